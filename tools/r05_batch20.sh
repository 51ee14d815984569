#!/bin/bash
# round 5, batch 20: modes d / i: the entries' keys with one load per read word; SQ counters of the seed-mode kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b20; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or mode" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for wl in C3-d C3-i; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms')"
done
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for wl in C3-i C3-d; do
timeout -k 5 300 rocprofv3 --kernel-trace --stats --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_$wl -- python3 bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 1 --warmup 1 > $O/pmc_$wl.txt 2> $O/pmc_$wl.err || echo "pmc failed"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/pmc_$wl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_seed") or k.startswith("k_rx"):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
print("$wl (sums over the run's 2 steps)")
for k, v in acc.items():
    busy = v["SQ_BUSY_CYCLES"] / 32.0
    print(f"{k[:34]:34s} n {n[k]:3d} VALU {v['SQ_INSTS_VALU']:.3g} valu_busy {v['SQ_INSTS_VALU']*4/1024/max(busy,1):.2f} lanes {v['SQ_THREAD_CYCLES_VALU']/64/max(v['SQ_INSTS_VALU'],1):.2f} wait {v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f} busy_ms {busy/2.4e6:.1f}")
PY
done

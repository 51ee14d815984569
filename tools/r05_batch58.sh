#!/bin/bash
# round 5, batch 58: modes d / i / e with the grouped heavy windows (units of 256 windows x 256 entries, 32 windows staged at a time): tests, bench lines with the CPU legs, PMC traffic, traces
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b58; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py "tests/test_gpu_fullsize.py::test_c3_full_size_seed_modes_bit_parity" -x -q -m gpu -k "seed or mode or golden or multi" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for wl in C3-d C3-i C3-e; do
  timeout -k 10 300 python bench.py --workload $wl --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', 'parity', (d.get('parity_sample') or {}).get('diff'), 'cpu', d['cpu_baseline']['value'])"
done
for wl in C3-d C3-i C3-e; do
  bash tools/pmc_groups.sh $O/pmc_$wl "FETCH_SIZE" "WRITE_SIZE" -- --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
  python3 tools/pmc_seed_traffic.py $O/pmc_$wl $O/$(echo $wl | tr 'A-Z' 'a-z')_traffic.json $wl
done
bash tools/r05_batch23.sh

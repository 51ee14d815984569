#!/bin/bash
# round 5, batch 41: SQ counters of the dual kernel at -M 3 and on the C5 shard (what are they bound by)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b41; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for wl in C3-M3 C5-shard; do
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_EA0_RDREQ"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${wl}_pass$i -- python3 bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 1 --warmup 1 > $O/${wl}_pass$i.txt 2> $O/${wl}_pass$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/${wl}_pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_copmem_match_dual" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v)/len(v) for k, v in acc.items()}
print("$wl", {k: f"{v:.4g}" for k, v in sorted(m.items())})
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("  ms %.1f  VALU issue %.2f of the SIMD cycles  lanes/instr %.2f  waiting for memory %.2f  waiting to issue %.2f  lines/s %.1f G" % (
    cyc / 2.4e6, m["SQ_INSTS_VALU"] * 4 / 1024 / cyc, m["SQ_THREAD_CYCLES_VALU"] / 64 / m["SQ_INSTS_VALU"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["TCC_EA0_RDREQ"] / (cyc / 2.4e9) / 1e9))
PY
done

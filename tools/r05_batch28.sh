#!/bin/bash
# round 5, batch 28: kernel traces of the long-text workloads (where does their index build's time go)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b28; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for wl in C5-shard P64; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --no-boundary --no-cpu-baseline --parity-sample-reads 0 --steps 2 --warmup 1 > $O/bench_$wl.json 2> $O/trace_$wl.err
find $O/trace_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$wl.csv
find $O/trace_$wl -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
print("$wl")
for r in csv.DictReader(open("$O/kernel_stats_$wl.csv")):
    t=int(r['TotalDurationNs'])/1e6
    if t>1: print(f"  {r['Name'][:72]:72s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:7.3f} total/3 {t/3:7.2f}")
PY
done

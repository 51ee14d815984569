#!/bin/bash
# round 5, batch 11: device times of the streamed blocks; kernel trace of mode d
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b11; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
PGRC_STREAM_TIMING=1 timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 2 > $O/boundary.json 2> $O/boundary_timing.txt; tail -30 $O/boundary_timing.txt
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in C3-d C3-i; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --no-boundary --no-cpu-baseline --parity-sample-reads 0 --steps 2 --warmup 1 > $O/bench_$wl.json 2> $O/trace_$wl.err
find $O/trace_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$wl.csv
find $O/trace_$wl -name "*kernel_trace.csv" -size +20M -delete
head -20 $O/kernel_stats_$wl.csv | cut -c1-160
done

#!/bin/bash
# round 5, batch 65: PMC traffic of the workloads that had none (C3-N, C3-PE, C2, P64) and their bench lines with it; kernel traces of C3-M3 / C3-N
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b65; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
for wl in C3-N C3-PE C2 P64; do
  w=$(echo $wl | tr 'A-Z' 'a-z')
  bash tools/pmc_groups.sh $O/pmc_$wl "FETCH_SIZE" "WRITE_SIZE" -- --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
  python3 tools/pmc_traffic.py $O/pmc_$wl k_copmem_match_ $O/${w}_traffic.json 1 dual | head -c 200; echo
  cp $O/${w}_traffic.json profiles/r05_${w}_traffic.json
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms; traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), 'parity', (d.get('parity_sample') or {}).get('diff'))"
done
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for wl in C3-M3 C3-N; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --no-boundary --no-cpu-baseline --parity-sample-reads 0 --steps 2 --warmup 1 > $O/tbench_$wl.json 2> $O/trace_$wl.err
find $O/trace_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$wl.csv
find $O/trace_$wl -name "*kernel_trace.csv" -size +20M -delete
head -4 $O/kernel_stats_$wl.csv | cut -c1-150
done

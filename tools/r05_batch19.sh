#!/bin/bash
# round 5, batch 19: the round's profile of the default bench command (kernel trace + PMC passes), then the other workloads' bench lines
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b19; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 bash tools/profile_c3.sh $O/profile_c3 > $O/profile_c3.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile_c3.log | cut -c1-200
for wl in C3-M3 C3-N C5-shard C2 P64 C3-PE; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms; dual', d['roofline'].get('kernel_ms'), 'index', d['roofline'].get('index',{}).get('ms'), 'parity diff', (d.get('parity_sample') or {}).get('diff'))"
done

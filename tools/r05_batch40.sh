#!/bin/bash
# round 5, batch 40: modes d / i / e at the round's last seedidx.hip / radix.hip: bench lines with the CPU legs, PMC traffic of d / i / e, kernel traces
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b40; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
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

#!/bin/bash
# round 5, batch 33: modes d / i / e with the segment sort: the three bench lines (with the CPU legs), mode d's PMC traffic
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b33; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
for wl in C3-d C3-i C3-e; do
  timeout -k 10 300 python bench.py --workload $wl --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', 'parity', (d.get('parity_sample') or {}).get('diff'), 'cpu', d['cpu_baseline']['value'])"
done
bash tools/pmc_groups.sh $O/pmc_d "FETCH_SIZE" "WRITE_SIZE" -- --workload C3-d --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
python3 tools/pmc_seed_traffic.py $O/pmc_d $O/c3-d_traffic.json C3-d

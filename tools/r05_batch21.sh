#!/bin/bash
# round 5, batch 21: k_seed_keys with LDS tables; how many heavy windows are there
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b21; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or mode" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for wl in C3-d C3-i; do
  PGRC_SEED_DEBUG=1 timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 1 --warmup 0 > $O/dbg_$wl.json 2> $O/dbg_$wl.err; grep "seed segment" $O/dbg_$wl.err | tail -4
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms')"
done

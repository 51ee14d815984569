#!/bin/bash
# round 5, batch 51: heavy windows grouped by key: tests, full-size parity, bench lines, kernel trace, soaks
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b51; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py -x -q -m gpu -k "seed or mode or golden or multi" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python -m pytest "tests/test_gpu_fullsize.py::test_c3_full_size_seed_modes_bit_parity" -x -q -m gpu > $O/pytest_full.log 2>&1; rc=$?; echo "pytest full rc=$rc"; tail -2 $O/pytest_full.log
[ $rc -eq 0 ] || exit 1
for wl in C3-d C3-i C3-e; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', 'parity', (d.get('parity_sample') or {}).get('diff'))"
done
bash tools/r05_batch23.sh
timeout -k 10 300 python tests/soak.py 150 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -1 $O/soak.log
timeout -k 10 300 python tests/soak_medium.py 150 > $O/soak_medium.log 2>&1; echo "soak_medium rc=$?"; tail -1 $O/soak_medium.log

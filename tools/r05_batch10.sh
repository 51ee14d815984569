#!/bin/bash
# round 5, batch 10: the sort-built table of modes d / i / e (tests, then the three benches); the boundary A/B with pinned host packing
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b10; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py -x -q -m gpu -k "seed or mode or golden or multi" > $O/pytest_seed.log 2>&1; rc=$?; echo "pytest seed rc=$rc"; tail -5 $O/pytest_seed.log
if [ $rc -eq 0 ]; then
  timeout -k 10 400 python -m pytest "tests/test_gpu_fullsize.py::test_c3_full_size_seed_modes_bit_parity" -x -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest full rc=$?"; tail -3 $O/pytest_full.log
  for wl in C3-d C3-i C3-e; do
    timeout -k 10 300 python bench.py --workload $wl --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', d['value'], d['ms_per_step'], d.get('roofline'), {k:d['config'].get(k) for k in ('index_ms','match_ms','hits')})"
  done
fi
for v in "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=0" "PGRC_STREAM_GRID=5 PGRC_HOST_PACK=0" "PGRC_STREAM_GRID=6 PGRC_HOST_PACK=0" "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=1" "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=1 PGRC_HOST_THREADS=4" "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=1 PGRC_HOST_THREADS=14" "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=0"; do
  echo "== $v"
  env $v timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 3 > $O/boundary_tmp.json 2> $O/boundary_tmp.err
  python3 - <<PY
import json
d=json.load(open("$O/boundary_tmp.json"))
print("   ", [(round(r["total_s"]*1e3,1), round(r["set_pg_s"]*1e3,1), round(r["reads_per_s_incl_pcie"]/1e6,1), r.get("equal_to_first_run")) for r in d["pipelined"]])
PY
done 2>&1 | tee $O/boundary_ab.txt
PGRC_HOST_PACK=1 PGRC_STREAM_TIMING=1 timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 2 > $O/boundary.json 2> $O/boundary_timing.txt; tail -22 $O/boundary_timing.txt

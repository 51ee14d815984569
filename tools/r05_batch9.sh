#!/bin/bash
# round 5, batch 9: host-side text packing + the streamed runs' grid: tests, then the boundary A/B; the fused flagged-partition list
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b9; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python -m pytest tests/test_gpu_stream.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_export.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=0" "PGRC_STREAM_GRID=8 PGRC_HOST_PACK=1" "PGRC_STREAM_GRID=6 PGRC_HOST_PACK=1" "PGRC_STREAM_GRID=5 PGRC_HOST_PACK=1" "PGRC_STREAM_GRID=4 PGRC_HOST_PACK=1" "PGRC_STREAM_GRID=5 PGRC_HOST_PACK=1 PGRC_HOST_THREADS=32"; do
  echo "== $v"
  env $v timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 3 > $O/boundary_tmp.json 2> $O/boundary_tmp.err
  python3 - <<PY
import json
d=json.load(open("$O/boundary_tmp.json"))
print("   ", [(round(r["total_s"]*1e3,1), round(r["set_pg_s"]*1e3,1), round(r["reads_per_s_incl_pcie"]/1e6,1), r.get("equal_to_first_run")) for r in d["pipelined"]])
PY
done 2>&1 | tee $O/boundary_ab.txt
PGRC_STREAM_TIMING=1 timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 2 > $O/boundary.json 2> $O/boundary_timing.txt; tail -22 $O/boundary_timing.txt
timeout -k 10 300 python tools/ab_match.py --workload C3 --rounds 3 PGRC_INDEX_CFG=1 PGRC_INDEX_CFG=0 > $O/ab_index_c3.txt 2>&1; cat $O/ab_index_c3.txt | tail -2

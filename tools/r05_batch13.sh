#!/bin/bash
# round 5, batch 13: a smaller first block in a streamed run (A/B, 5 reps each, alternating); mode d PMC passes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b13; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for v in "PGRC_STREAM_FIRST=0" "PGRC_STREAM_FIRST=1" "PGRC_STREAM_FIRST=0" "PGRC_STREAM_FIRST=1" "PGRC_STREAM_FIRST=1 PGRC_STREAM_TWO=1" "PGRC_STREAM_FIRST=1 PGRC_UPLOAD_CHUNK_MB=768"; do
  echo "== $v"
  env $v timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 5 > $O/boundary_tmp.json 2> $O/boundary_tmp.err
  python3 - <<PY
import json
d=json.load(open("$O/boundary_tmp.json"))
print("   ", [(round(r["total_s"]*1e3,1), round(r["set_pg_s"]*1e3,1), round(r["reads_per_s_incl_pcie"]/1e6,1), r.get("equal_to_first_run")) for r in d["pipelined"]])
PY
done 2>&1 | tee $O/boundary_ab.txt
PGRC_STREAM_TIMING=1 timeout -k 10 200 python tools/boundary_c3.py --legs pipelined --reps 2 > $O/boundary.json 2> $O/boundary_timing.txt; tail -34 $O/boundary_timing.txt
bash tools/pmc_groups.sh $O/pmc_d "FETCH_SIZE" "WRITE_SIZE" -- --workload C3-d --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
python3 tools/pmc_seed_traffic.py $O/pmc_d $O/c3-d_traffic.json C3-d

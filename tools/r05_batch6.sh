#!/bin/bash
# round 5, batch 6: the index build as kept (cb 13, XCD-aware order, round 4's finish); the full bench line with its new legs
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b6; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index or copmem_parity" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python tools/ab_match.py --workload C3 --rounds 3 PGRC_INDEX_CFG=1 PGRC_INDEX_CFG=0 > $O/ab_index_c3.txt 2>&1; echo "ab index C3 rc=$?"; cat $O/ab_index_c3.txt | tail -3
( time timeout -k 10 900 python bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err ) 2>&1 | grep real; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/bench_c3.json"))
print("value", d["value"], "ms", d["ms_per_step"], "phases", {k:(round(v,2) if isinstance(v,float) else v) for k,v in d["phases_ms"].items() if k in ("index_fwd","screen","other")})
print("roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_GBps"))
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","nproc","kind")}, "t16", (d["cpu_baseline"].get("t16") or {}).get("value"), "t1", (d["cpu_baseline"].get("t1") or {}).get("value"))
print("parity", d.get("parity_sample",{}).get("diff"), "boundary", {k:d.get("boundary",{}).get(k) for k in ("reads_per_s","s","link_GBps","frac_of_link_bound","error")})
PY

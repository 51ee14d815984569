#!/bin/bash
# round 5, batch 4: head-write micro-benchmark; pass 1 with reserved runs (no counting pre-pass): tests, then A/B and traces
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b4; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 120 ./tools/ubench/headwrite > $O/headwrite.txt 2>&1; cat $O/headwrite.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index or copmem_parity" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 300 python -m pytest tests/test_gpu_golden.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest golden rc=$?"; tail -3 $O/pytest2.log
timeout -k 10 300 python tools/ab_match.py --workload C3 --rounds 3 PGRC_INDEX_CFG=3 PGRC_INDEX_CFG=11 PGRC_INDEX_CFG=1 PGRC_INDEX_CFG=9 PGRC_INDEX_CFG=5 PGRC_INDEX_CFG=13 > $O/ab_index_c3.txt 2>&1; echo "ab index C3 rc=$?"; cat $O/ab_index_c3.txt | tail -7
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 3 1; do
  for turn in 1 0; do
    export PGRC_INDEX_CFG=$cfg PGRC_BUILD_STREAMS=$turn
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c${cfg}_t${turn} -- python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 4 --warmup 1 > $O/bench_c${cfg}_t${turn}.json 2> $O/bench_c${cfg}_t${turn}.err || echo "trace cfg $cfg turn $turn failed"
    find $O/trace_c${cfg}_t${turn} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats_c${cfg}_t${turn}.csv
    echo "cfg $cfg in_turn $turn:"; python3 - <<PY
import csv,json
try:
    d=json.load(open("$O/bench_c${cfg}_t${turn}.json")); print("  step", round(d["ms_per_step"],2), "index", round(d["phases_ms"]["index_fwd"],2), round(d["phases_ms"]["index_rc"],2), "dual", round(d["phases_ms"]["screen"],2))
except Exception as e: print("  no bench line", e)
for r in csv.DictReader(open("$O/stats_c${cfg}_t${turn}.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_os_","k_ps_")) and float(r["AverageNs"]) > 3000: print("   %-28s calls %3s avg %8.3f ms  min %8.3f max %8.3f" % (n.split("<")[0].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
PY
  done
done
rm -rf $O/trace_c*

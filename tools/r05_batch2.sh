#!/bin/bash
# round 5, batch 2: kernel traces of the index build variants (PGRC_INDEX_CFG 0..3), both builds at once and in turn; the
# mem / export / multi / stream tests with the library's own sorts and scans
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b2; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 0 1 2 3; do
  for turn in 0 1; do
    export PGRC_INDEX_CFG=$cfg PGRC_BUILD_STREAMS=$turn
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c${cfg}_t${turn} -- python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 4 --warmup 1 > $O/bench_c${cfg}_t${turn}.json 2> $O/bench_c${cfg}_t${turn}.err || echo "trace cfg $cfg turn $turn failed"
    find $O/trace_c${cfg}_t${turn} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stats_c${cfg}_t${turn}.csv
    echo "cfg $cfg in_turn $turn:"; python3 - <<PY
import csv,json
try:
    d=json.load(open("$O/bench_c${cfg}_t${turn}.json")); print("  step", round(d["ms_per_step"],2), "index", d["phases_ms"]["index_fwd"], d["phases_ms"]["index_rc"])
except Exception as e: print("  no bench line", e)
for r in csv.DictReader(open("$O/stats_c${cfg}_t${turn}.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_os_","k_ps_","k_psc_")): print("   %-28s calls %3s avg %8.3f ms  min %8.3f max %8.3f" % (n.split("<")[0].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6))
PY
  done
done
unset PGRC_INDEX_CFG PGRC_BUILD_STREAMS
rm -rf $O/trace_c*
timeout -k 10 700 python -m pytest tests/test_gpu_mem.py tests/test_gpu_export.py tests/test_gpu_stream.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log

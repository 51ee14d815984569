#!/bin/bash
# Where the time of the two scatter passes goes (idxsweep.hip, PGRC_OS_DBG): per-kernel ms at C3 with parts of the kernels
# switched off -- 1: blocks exit at once (launch cost), 2: no global stores, 4: no record loads / no hashing, 6: neither.
# The index is garbage under these knobs: timing only.  usage (on the GPU box): bash tools/os_dbg_sweep.sh <outdir> "<cfg> ..."
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/os_dbg}
CFGS=${2:-"0 1"}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for cfg in $CFGS; do
  for dbg in ${DBGS:-0 1 2 4 6}; do
    PGRC_OS_CFG=$cfg PGRC_OS_DBG=$dbg PGRC_BUILD_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof${cfg}_$dbg" -- python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 2 --warmup 1 > "$OUT/b${cfg}_$dbg.json" 2> "$OUT/b${cfg}_$dbg.err"
    python3 - "$OUT" "$cfg" "$dbg" <<'PY'
import csv, glob, sys
out, cfg, dbg = sys.argv[1:4]
f = glob.glob(f"{out}/prof{cfg}_{dbg}/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"]: r for r in csv.DictReader(open(f))}
def ms(prefix):
    return [round(float(r["AverageNs"]) / 1e6, 2) for n, r in rows.items() if prefix in n[:len(prefix) + 6]]
print("cfg", cfg, "dbg", dbg, "gen", ms("k_os_scatter_gen"), "bins", ms("k_os_scatter_bins"), "count1", ms("k_os_count_gen"), "count2", ms("k_os_count_bins"), "finish", ms("k_ps_finish_fast"), flush=True)
PY
  done
done

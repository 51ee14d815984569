#!/bin/bash
# Block shapes of the one-sweep index build (idxsweep.hip, PGRC_OS_CFG) at C3: per-kernel times with the two strands'
# builds in turn (PGRC_BUILD_STREAMS=1, rocprofv3 kernel stats), then the pair at once (bench phases).
# usage (on the GPU box): bash tools/os_cfg_sweep.sh <outdir> "<cfg> <cfg> ..."
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/os_cfg}
CFGS=${2:-"0 1 2 3 4 5 6 7 8"}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for cfg in $CFGS; do
  PGRC_OS_CFG=$cfg PGRC_BUILD_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$cfg" -- python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 2 --warmup 1 > "$OUT/turn$cfg.json" 2> "$OUT/turn$cfg.err"
  [ -s "$OUT/turn$cfg.json" ] || { echo "cfg $cfg: bench printed nothing"; tail -5 "$OUT/turn$cfg.err"; exit 1; }
  PGRC_OS_CFG=$cfg python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 4 --warmup 1 > "$OUT/pair$cfg.json" 2> "$OUT/pair$cfg.err"
  python3 - "$OUT" "$cfg" <<'PY'
import csv, glob, json, sys
out, cfg = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/prof{cfg}/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"]: r for r in csv.DictReader(open(f))}
def ms(prefix):
    return [round(float(r["AverageNs"]) / 1e6, 2) for n, r in rows.items() if n.startswith(prefix) or ("void " + prefix) in n[:len(prefix) + 5]]
turn = json.load(open(f"{out}/turn{cfg}.json")); pair = json.load(open(f"{out}/pair{cfg}.json"))
print("cfg", cfg, "count1", ms("k_os_count_gen"), "count2", ms("k_os_count_bins"), "scan", ms("k_psc_write"), "gen", ms("k_os_scatter_gen"), "bins", ms("k_os_scatter_bins"), "finish", ms("k_ps_finish_fast"), "general", ms("k_ps_finish<"),
      "| in turn: index", round(turn["phases_ms"]["index_fwd"] + turn["phases_ms"]["index_rc"], 2), "| pair:", round(pair["phases_ms"]["index_fwd"], 2), "step", round(pair["ms_per_step"], 1), flush=True)
PY
done

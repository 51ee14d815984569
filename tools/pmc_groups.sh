#!/bin/bash
# usage: tools/pmc_groups.sh <outdir> "<grp1>" "<grp2>" ... -- <bench args...>
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  # a counter group the hardware cannot collect makes rocprofv3 abort and then hang in its signal handler: bound it
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 bench.py "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
  echo "pass $i ($grp) done"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"

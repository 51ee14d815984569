#!/bin/bash
# per-context counters of identical match kernels (why do contexts differ by up to 13 %?)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/pmc_contexts}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
i=0
for grp in "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 tools/ab_libs.py --rounds 2 a=base b=base c=base d=base e=base f=base > "$OUT/pass$i.log" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; exit 1; }
  grep median "$OUT/pass$i.log"
done
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT/pass1/*/ | head

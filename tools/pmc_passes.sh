#!/bin/bash
# PMC passes (one counter group per rocprofv3 run, no tracing flags beside --kernel-trace).
# usage: tools/pmc_passes.sh <outdir> <bench args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 bench.py "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
  echo "pass $i ($grp) done"
done

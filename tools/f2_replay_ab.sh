#!/bin/bash
# row f2: the host replay with one thread (PGRC_MEM_REPLAY_THREADS=1 = the sequential scan) vs the parallel speculative
# replay, at the C3 pseudogenome size; the cases run twice in one process (the first call also grows the pinned mirrors).
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/f2_replay}
mkdir -p $OUT
for t in 1 16; do
  PGRC_MEM_REPLAY_THREADS=$t python tests/mem_scale.py --no-reference --cases fwd,fwd,lq,lq,long,long,rc --out $OUT/mem_scale_t$t.json > $OUT/mem_scale_t$t.log 2>&1
  python - <<PY
import json
for line in open("$OUT/mem_scale_t$t.log"):
    line=line.strip()
    if line.startswith("{") and "counters" in line:
        d=json.loads(line)
        for k,v in d.items():
            c=v["counters"]; print("threads $t", k, "matches", v["matches"], "events", c["events"], "call_s", round(v["gpu_s"],3), "ms_host", round(c["ms_host"],1), "ms_sort", round(c["ms_sort"],1), v["digest"])
PY
done

#!/bin/bash
# A/B of the index build variants (PGRC_INDEX_SORT = hybrid (default) | own | rocprim), then a rocprofv3 kernel trace of
# the default.  usage (on the GPU box): bash tools/ab_index_sort.sh <outdir>
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/ab_index_sort}
mkdir -p $OUT
for v in ${VARIANTS:-sweep hybrid own rocprim}; do
  PGRC_INDEX_SORT=$v python bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 5 --warmup 2 > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  [ -s $OUT/bench_$v.json ] || { echo "bench ($v) printed nothing: see $OUT/bench_$v.err"; exit 1; }
done
python - <<PY
import json
for f in "${VARIANTS:-sweep hybrid own rocprim}".split():
    d=json.load(open("$OUT/bench_%s.json"%f))
    print(f, round(d["value"]/1e6,1), "M reads/s", round(d["ms_per_step"],1), "ms", {k: round(v,2) for k,v in d["phases_ms"].items() if isinstance(v, float)})
PY
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 3 --warmup 1 > $OUT/prof_bench.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs head -16 | cut -c1-200

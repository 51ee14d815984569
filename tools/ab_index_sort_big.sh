#!/bin/bash
# index build variants on the larger tables (C5 shard: 2^30 buckets, 620 M samples; P64: 4.4 Gbp text, 880 M samples)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/ab_index_big}
mkdir -p $OUT
for w in C5-shard P64; do
  for v in hybrid own rocprim; do
    PGRC_INDEX_SORT=$v python bench.py --workload $w --no-cpu-baseline --parity-sample-reads 0 --steps 3 --warmup 1 > $OUT/bench_${w}_$v.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/bench_${w}_$v.json')); print('$w $v', round(d['value']/1e6,1), 'M reads/s', {k: round(v,2) for k,v in d['phases_ms'].items()})"
  done
done

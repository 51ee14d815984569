#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
for m in d e; do
OUT=$ROOT/gpurun_out/prof_mode_${m}_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/modes_c3.py $m > $OUT/run.jsonl 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo "mode $m"; head -14 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/trace
done

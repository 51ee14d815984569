#!/bin/bash
# Round profile of the default bench command: rocprofv3 kernel trace + stats, then separate PMC passes (never combined with
# other trace domains).  usage (on the GPU box): bash tools/profile_c3.sh <outdir>
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-gpurun_out/profile_c3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# (--no-boundary: the boundary leg launches the dual kernel on blocks of a streamed run -- other, shorter launches of the same kernel
#  that would enter its average)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-boundary > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
[ -s $OUT/bench_under_rocprof.json ] || { echo "bench printed nothing: see $OUT/trace.err"; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-180
bash tools/pmc_groups.sh $OUT/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" -- --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
python3 tools/pmc_traffic.py $OUT/pmc k_copmem_match_ $OUT/traffic.json 1 dual > /dev/null   # C3 takes the dual schedule: ONE match launch per step (round 4: no passes behind the dual kernel)
cat $OUT/pmc/summary.txt | head -40

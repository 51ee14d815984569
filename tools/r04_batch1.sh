#!/bin/bash
# round-4 measurement batch (run on the GPU box): tiny-N parity, guided-chunk and inline-N A/Bs, -M 3, N-kernel profile
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --workload tiny-N --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_tinyN.json 2> gpurun_out/r04_bench_tinyN.err
python -c "
import json;d=json.load(open('gpurun_out/r04_bench_tinyN.json'));print(d['parity_sample']);print(d['value'], d['config']['reads_with_N'])"
python tools/ab_match.py --workload C3-N --rounds 3 PGRC_NREAD_INLINE=0,PGRC_MATCH_GUIDED=0 PGRC_NREAD_INLINE=1,PGRC_MATCH_GUIDED=0 PGRC_NREAD_INLINE=1,PGRC_MATCH_GUIDED=1 > gpurun_out/r04_nread_inline_ab.txt 2>&1; cat gpurun_out/r04_nread_inline_ab.txt
python tools/ab_match.py --workload C3 --rounds 3 PGRC_MATCH_GUIDED=0 PGRC_MATCH_GUIDED=1 > gpurun_out/r04_guided_ab.txt 2>&1; cat gpurun_out/r04_guided_ab.txt
python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_HEAD_PAIR=0 PGRC_HEAD_PAIR=3 > gpurun_out/r04_m3_ab.txt 2>&1; cat gpurun_out/r04_m3_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PGRC_NREAD_INLINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_nkernel -- python3 tools/ab_match.py --workload C3-N --rounds 1 X=1 > gpurun_out/r04_prof_nkernel.log 2>&1
find gpurun_out/r04_prof_nkernel -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-160

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python -m pytest tests/test_gpu_divide.py tests/test_gpu_parity.py -x -q -m gpu -k "divide or own_hit" > gpurun_out/r04_batch12_tests.log 2>&1; tail -2 gpurun_out/r04_batch12_tests.log
for w in C2 C3-PE C5-shard P64; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err
  python -c "
import json;d=json.load(open('gpurun_out/r04_bench_$w.json'));print('$w', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],1), 'ms', d['phases_ms']['schedule'], 'parity diff', d['parity_sample']['diff'])"
done
python tools/ab_match.py --workload C2 --rounds 4 PGRC_DUAL=0 PGRC_DUAL=1 > gpurun_out/r04_c2_dual_ab.txt 2>&1; cat gpurun_out/r04_c2_dual_ab.txt

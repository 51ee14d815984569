#!/bin/bash
# round-4 batch 5: the profile of the default bench command (kernel stats + PMC passes), then the C3-N and C3-M3 lines
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or few_ns" > gpurun_out/r04_batch5_tests.log 2>&1; tail -2 gpurun_out/r04_batch5_tests.log
bash tools/profile_c3.sh gpurun_out/profile_c3_r04 > gpurun_out/profile_c3_r04.log 2>&1; tail -45 gpurun_out/profile_c3_r04.log
python bench.py --workload C3-N --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_c3n.json 2> gpurun_out/r04_bench_c3n.err; cut -c1-400 gpurun_out/r04_bench_c3n.json
python bench.py --workload C3-M3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_c3m3.json 2> gpurun_out/r04_bench_c3m3.err; cut -c1-400 gpurun_out/r04_bench_c3m3.json

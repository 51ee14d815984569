#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04_batch16_tests.log 2>&1; tail -3 gpurun_out/r04_batch16_tests.log
python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_min_own.jsonl 2> gpurun_out/r04_modes_c3_min_own.err; cut -c1-200 gpurun_out/r04_modes_c3_min_own.jsonl
PGRC_SEED_SORT=lib python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_min_lib.jsonl 2> gpurun_out/r04_modes_c3_min_lib.err; cut -c1-200 gpurun_out/r04_modes_c3_min_lib.jsonl

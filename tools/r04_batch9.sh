#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_multi.py -x -q -m gpu -k "seed or golden or mode or multi" > gpurun_out/r04_batch9_tests.log 2>&1; tail -3 gpurun_out/r04_batch9_tests.log
python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_own_sort.jsonl 2> gpurun_out/r04_modes_c3_own_sort.err; cat gpurun_out/r04_modes_c3_own_sort.jsonl
PGRC_SEED_SORT=lib python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_lib_sort.jsonl 2> gpurun_out/r04_modes_c3_lib_sort.err; cat gpurun_out/r04_modes_c3_lib_sort.jsonl

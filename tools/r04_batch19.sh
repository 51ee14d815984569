#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu -k "seed or golden or mode or exact or continuation or p64 or property or drop_in" > gpurun_out/r04_batch19_tests.log 2>&1; tail -3 gpurun_out/r04_batch19_tests.log
MODES_DIGEST=1 timeout -k 10 300 python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_min.jsonl 2> gpurun_out/r04_modes_c3_min.err; cut -c1-230 gpurun_out/r04_modes_c3_min.jsonl
MODES_DIGEST=1 PGRC_SEED_REDUCE=sort timeout -k 10 300 python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_sort.jsonl 2> gpurun_out/r04_modes_c3_sort.err; cut -c1-230 gpurun_out/r04_modes_c3_sort.jsonl
timeout -k 10 200 python tests/soak.py 120 4101 > gpurun_out/r04_soak_min.log 2>&1; tail -2 gpurun_out/r04_soak_min.log

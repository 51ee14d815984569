#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_export.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or golden or export or order or mode" > gpurun_out/r04_batch17_tests.log 2>&1; tail -3 gpurun_out/r04_batch17_tests.log
python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_min_own9.jsonl 2> gpurun_out/r04_modes_c3_min_own9.err; cut -c1-200 gpurun_out/r04_modes_c3_min_own9.jsonl
PGRC_RADIX_BITS=8 python tools/modes_c3.py d i > gpurun_out/r04_modes_c3_min_own8.jsonl 2> gpurun_out/r04_modes_c3_min_own8.err; cut -c1-200 gpurun_out/r04_modes_c3_min_own8.jsonl

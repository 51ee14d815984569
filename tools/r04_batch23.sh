#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or golden or mode or exact or continuation" > gpurun_out/r04_batch23_tests.log 2>&1; tail -3 gpurun_out/r04_batch23_tests.log
MODES_DIGEST=1 timeout -k 10 300 python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_canon.jsonl 2> gpurun_out/r04_modes_c3_canon.err; cut -c1-60,120-330 gpurun_out/r04_modes_c3_canon.jsonl

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r04_batch4_tests.log 2>&1; tail -4 gpurun_out/r04_batch4_tests.log
python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_SPEC_LIMIT=-1 PGRC_SPEC_LIMIT=0 PGRC_SPEC_LIMIT=1 PGRC_SPEC_LIMIT=2 PGRC_SPEC_LIMIT=3 PGRC_SPEC_LIMIT=5 > gpurun_out/r04_spec_ab.txt 2>&1; cat gpurun_out/r04_spec_ab.txt

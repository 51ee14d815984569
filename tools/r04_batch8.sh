#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r04_batch8_tests.log 2>&1; tail -3 gpurun_out/r04_batch8_tests.log
python tools/ab_match.py --workload C3 --rounds 3 PGRC_DUAL_AHEAD=0 PGRC_DUAL_AHEAD=1 > gpurun_out/r04_ahead_ab.txt 2>&1; cat gpurun_out/r04_ahead_ab.txt
python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_AHEAD=0 PGRC_DUAL_AHEAD=1 >> gpurun_out/r04_ahead_ab.txt 2>&1; tail -2 gpurun_out/r04_ahead_ab.txt
python tools/ab_match.py --workload C3-N --rounds 3 PGRC_DUAL_AHEAD=0 PGRC_DUAL_AHEAD=1 >> gpurun_out/r04_ahead_ab.txt 2>&1; tail -2 gpurun_out/r04_ahead_ab.txt

#!/bin/bash
# round-4 measurement batch 2 (on the GPU box)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_export.py tests/test_e2e_dropin.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_batch2_tests.log 2>&1; tail -5 gpurun_out/r04_batch2_tests.log
python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_REDO_ANY=1 PGRC_REDO_ANY=0 > gpurun_out/r04_m3_redo_ab.txt 2>&1; cat gpurun_out/r04_m3_redo_ab.txt
python tools/ab_match.py --workload C3 --rounds 3 PGRC_REDO_ANY=1 PGRC_REDO_ANY=0 PGRC_MATCH_GUIDED=0 PGRC_MATCH_GUIDED=1 PGRC_MATCH_GUIDED=40 PGRC_MATCH_GUIDED=1,PGRC_MATCH_CHUNK=4096 > gpurun_out/r04_guided_ab.txt 2>&1; cat gpurun_out/r04_guided_ab.txt
python tools/ab_match.py --workload C3-N --rounds 3 PGRC_MATCH_GUIDED=0 PGRC_MATCH_GUIDED=1 PGRC_MATCH_GUIDED=40 PGRC_MATCH_GUIDED=1,PGRC_MATCH_CHUNK=4096 > gpurun_out/r04_guided_n_ab.txt 2>&1; cat gpurun_out/r04_guided_n_ab.txt

#!/bin/bash
# round-4 batch 3 (on the GPU box): the whole GPU suite, then the from-end A/B
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite2.log 2>&1; tail -4 gpurun_out/r04_gpu_suite2.log
python tools/ab_match.py --workload C3-N --rounds 3 PGRC_MATCH_FROM_END=0 PGRC_MATCH_FROM_END=1 > gpurun_out/r04_from_end_ab.txt 2>&1; cat gpurun_out/r04_from_end_ab.txt
python tools/ab_match.py --workload C3 --rounds 3 PGRC_MATCH_FROM_END=0 PGRC_MATCH_FROM_END=1 >> gpurun_out/r04_from_end_ab.txt 2>&1; tail -2 gpurun_out/r04_from_end_ab.txt

#!/bin/bash
# round 5, batch 30: the dual kernel at seven waves per SIMD (72 registers, two verify-cache slots, 8 staged reads): in-context A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b30; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "copmem_parity" > $O/pytest0.log 2>&1; echo "pytest default rc=$?"; tail -2 $O/pytest0.log
PGRC_DUAL_VARIANT=7 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "copmem_parity" > $O/pytest7.log 2>&1; echo "pytest variant 7 rc=$?"; tail -2 $O/pytest7.log
timeout -k 10 400 python tools/ab_match.py --workload C3 --rounds 4 PGRC_DUAL_VARIANT=0 PGRC_DUAL_VARIANT=7 PGRC_DUAL_VARIANT=5 > $O/ab_c3.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_c3.txt
timeout -k 10 400 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_VARIANT=0 PGRC_DUAL_VARIANT=7 > $O/ab_c3m3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3m3.txt

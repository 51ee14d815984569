#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b63; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "copmem_parity or dual" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/ab_match.py --workload C3 --rounds 4 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3.txt
timeout -k 10 400 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c3m3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3m3.txt
timeout -k 10 300 python tools/ab_match.py --workload C2 --rounds 4 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c2.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c2.txt

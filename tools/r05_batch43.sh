#!/bin/bash
# round 5, batch 43: the dual kernel's last reads in small chunks: parity, then the in-context A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b43; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_stream.py -x -q -m gpu -k "not seedindex" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/ab_match.py --workload C3 --rounds 5 PGRC_DUAL_TAIL=0 PGRC_DUAL_TAIL=1 > $O/ab_c3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3.txt
timeout -k 10 400 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_TAIL=0 PGRC_DUAL_TAIL=1 > $O/ab_c3m3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3m3.txt
timeout -k 10 400 python tools/ab_match.py --workload C2 --rounds 5 PGRC_DUAL_TAIL=0 PGRC_DUAL_TAIL=1 > $O/ab_c2.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c2.txt

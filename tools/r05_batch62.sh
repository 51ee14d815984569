#!/bin/bash
# round 5, batch 62: the dual kernel with all of an iteration's loads in flight together (one register set, loads before any use) and the
# next reads staged by loads straight into LDS: parity, then the in-context A/B against the kernel before (PGRC_DUAL_VARIANT=6)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b62; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_stream.py tests/test_gpu_multi.py -x -q -m gpu -k "not seedindex" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/ab_match.py --workload C3 --rounds 4 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3.txt
timeout -k 10 400 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c3m3.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3m3.txt
timeout -k 10 400 python tools/ab_match.py --workload C3-N --rounds 3 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c3n.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c3n.txt
timeout -k 10 300 python tools/ab_match.py --workload C2 --rounds 4 PGRC_DUAL_VARIANT=6 PGRC_DUAL_VARIANT=0 > $O/ab_c2.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_c2.txt

#!/bin/bash
# round 5, batch 15: the dual kernel's gathers with the non-temporal hint (in-context A/B)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b15; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python tools/ab_match.py --workload C3 --rounds 4 PGRC_DUAL_NT=0 PGRC_DUAL_NT=1 PGRC_DUAL_NT=2 PGRC_DUAL_NT=4 PGRC_DUAL_NT=3 PGRC_DUAL_NT=5 PGRC_DUAL_NT=7 > $O/ab_nt_c3.txt 2>&1; echo "rc=$?"; tail -12 $O/ab_nt_c3.txt
timeout -k 10 500 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_NT=0 PGRC_DUAL_NT=1 PGRC_DUAL_NT=7 > $O/ab_nt_c3m3.txt 2>&1; echo "rc=$?"; tail -6 $O/ab_nt_c3m3.txt

#!/bin/bash
# round 5, batch 42: 250 bp reads (the C5 shard): the dual kernel at five waves per SIMD instead of four, one context
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b42; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python tools/ab_match.py --workload C5-shard --rounds 3 PGRC_DUAL_VARIANT=0 PGRC_DUAL_VARIANT=9 > $O/ab_c5.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_c5.txt

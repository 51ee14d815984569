#!/bin/bash
# round 5, batch 46: the 9-bit passes of the index build (texts whose hash table has 2^30 buckets) as 1024 x 5 blocks, two per CU; one context
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b46; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python tools/ab_match.py --workload C5-shard --rounds 3 PGRC_INDEX_CFG=1 PGRC_INDEX_CFG=2 > $O/ab_c5.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_c5.txt
timeout -k 10 500 python tools/ab_match.py --workload P64 --rounds 3 PGRC_INDEX_CFG=1 PGRC_INDEX_CFG=2 > $O/ab_p64.txt 2>&1; echo "rc=$?"; tail -3 $O/ab_p64.txt

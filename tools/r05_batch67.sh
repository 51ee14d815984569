#!/bin/bash
# round 5, batch 67: long soaks on the round's last build (other seeds than the short ones)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b67; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 560 python tests/soak.py 500 7 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -1 $O/soak.log
timeout -k 10 560 python tests/soak_medium.py 500 7 > $O/soak_medium.log 2>&1; echo "soak_medium rc=$?"; tail -1 $O/soak_medium.log

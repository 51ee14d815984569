#!/bin/bash
# round 5, batch 7: the whole -m gpu suite as the driver runs it
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b7; mkdir -p $O
(while sleep 50; do echo "... $(date +%T) $(tail -c 200 $O/pytest.log 2>/dev/null | tr '\n' ' ' | tail -c 120)"; done) &
HB=$!
trap "kill $HB" EXIT
( time timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1 ) 2>&1 | grep real; echo "pytest rc=$?"; tail -8 $O/pytest.log

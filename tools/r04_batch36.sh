#!/bin/bash
# modes d / i / e at the full C3 size: the GPU's results for a sample of the workload's reads, and the number of (window, part) pairs with
# equal keys per strand, against the oracle's serial scans of the whole 1.875 Gbp text
cd ${GRAFT_REPO_ROOT:-.}
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
for w in C3-d C3-i C3-e; do
timeout -k 10 500 python tests/fullscale_parity.py --workload $w --sample 200000 --checker port --out gpurun_out/r04_fullscale_parity_${w/-/}.json 2>&1 | tail -1
done
kill $HB

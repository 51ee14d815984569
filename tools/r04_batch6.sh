#!/bin/bash
# round-4 batch 6: the reference's whole encoder at 25 M x 150 bp, -t 16, CPU leg and GPU leg (stages 1 + 4 + 7), round trip
cd ${GRAFT_REPO_ROOT:-.}
W=/tmp/pgrc_e2e_big
rm -rf $W; mkdir -p $W
PGRC_HIP_TIMING=1 python tests/e2e_big.py $W --reads ${1:-25000000} --threads 16 > gpurun_out/r04_e2e_25m.log 2> gpurun_out/r04_e2e_25m.err
tail -1 gpurun_out/r04_e2e_25m.log > gpurun_out/r04_e2e_25m.json
cut -c1-1500 gpurun_out/r04_e2e_25m.json
grep -i "export\|sort\|stream\|phase" gpurun_out/r04_e2e_25m.err | tail -30
rm -rf $W

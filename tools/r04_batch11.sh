#!/bin/bash
# round-4 batch 11: soaks with the round's knobs, full-size parity samples against the real reference
cd ${GRAFT_REPO_ROOT:-.}
python tests/soak.py 240 41 > gpurun_out/r04_soak.log 2>&1; tail -2 gpurun_out/r04_soak.log
python tests/soak_medium.py 240 42 > gpurun_out/r04_soak_medium.log 2>&1; tail -2 gpurun_out/r04_soak_medium.log
for w in C3 C5-shard P64; do
  python tests/fullscale_parity.py --workload $w --out gpurun_out/r04_fullscale_parity_$w.json > gpurun_out/r04_fsp_$w.log 2>&1; tail -c 500 gpurun_out/r04_fullscale_parity_$w.json; echo
done

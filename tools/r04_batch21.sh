#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for d in 0; do
echo "dbg $d"; PGRC_SEED_DBG=$d timeout -k 10 200 python tools/modes_c3.py d 2>/dev/null | cut -c120-330
done > gpurun_out/r04_hmin_dbg.txt
cat gpurun_out/r04_hmin_dbg.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for d in 4 8; do echo "dbg $d"; PGRC_SEED_DBG=$d timeout -k 10 300 python tools/modes_c3.py d 2>/dev/null | cut -c1-60,120-330; done

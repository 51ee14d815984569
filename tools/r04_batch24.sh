#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for h in 32 16 64 8; do echo "heavy $h"; PGRC_SEED_HEAVY=$h timeout -k 10 300 python tools/modes_c3.py d e 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['mode'], 'best %.1f ms first %.1f' % (d['best_s'] * 1e3, d['first_s'] * 1e3))"; done

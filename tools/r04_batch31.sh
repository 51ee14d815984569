#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for s in 0 1; do echo "stagger $s"; PGRC_BUILD_STAGGER=$s python bench.py --no-cpu-baseline --parity-sample-reads 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.2f index %.2f dual %.2f' % (d['ms_per_step'], d['phases_ms']['index_fwd'], d['phases_ms']['screen']))"; done; done

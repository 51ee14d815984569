#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "copmem or hard or golden or screened or continuation or adapter" 2>&1 | tail -2
PGRC_VERIFY2=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "copmem or hard or golden or screened or continuation" 2>&1 | tail -2
for w in C3-M3 C3; do for rep in 1 2; do for s in 0 1; do echo "$w two-step $s"; PGRC_VERIFY2=$s python bench.py --workload $w --no-cpu-baseline --parity-sample-reads 200000 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.2f index %.2f dual %.2f verifies %d parity diff %s' % (d['ms_per_step'], d['phases_ms']['index_fwd'], d['phases_ms']['screen'], d['counters']['dual']['verifies'], d.get('parity_sample', {}).get('diff')))"; done; done; done

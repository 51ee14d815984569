#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or golden or mode or exact or palindromes" 2>&1 | tail -3
python tools/modes_vs_oracle.py 10000000 1000000 d i e 2>/dev/null
python tools/modes_vs_oracle.py 30000000 2000000 d 2>/dev/null

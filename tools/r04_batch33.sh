#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_gpu_suite_b33.log 2>&1; tail -3 gpurun_out/r04_full_gpu_suite_b33.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 200 python tests/soak.py 90 4511 2>&1 | tail -1
timeout -k 10 400 python tests/soak_medium.py 200 4512 2>&1 | tail -1

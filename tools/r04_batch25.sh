#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_gpu_suite_b25.log 2>&1; tail -4 gpurun_out/r04_full_gpu_suite_b25.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke_b25.log 2>&1; tail -2 gpurun_out/r04_smoke_b25.log

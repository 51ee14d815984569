#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "seed or golden or mode" > gpurun_out/r04_batch10_tests.log 2>&1; tail -3 gpurun_out/r04_batch10_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_mode_d -- python3 tools/modes_c3.py d > gpurun_out/r04_modes_c3_own_sort.jsonl 2> gpurun_out/r04_prof_mode_d.err
cat gpurun_out/r04_modes_c3_own_sort.jsonl
find gpurun_out/r04_prof_mode_d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_mode_d_kernel_stats.csv
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r04_mode_d_kernel_stats.csv')))[:14]:
    print(r['Name'][:60].replace('\n',' '), r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), round(float(r['AverageNs'])/1e6,2))
PY

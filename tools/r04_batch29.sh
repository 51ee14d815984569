#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_gpu_suite_b29.log 2>&1; tail -4 gpurun_out/r04_full_gpu_suite_b29.log
timeout -k 10 200 python tests/soak.py 100 4307 > gpurun_out/r04_soak_b29.log 2>&1; tail -2 gpurun_out/r04_soak_b29.log
MODES_DIGEST=1 timeout -k 10 300 python tools/modes_c3.py d i e > gpurun_out/r04_modes_c3_brev.jsonl 2> gpurun_out/r04_modes_c3_brev.err
python - <<'PY'
import json
for l in open('gpurun_out/r04_modes_c3_brev.jsonl'):
    d = json.loads(l); print(d["mode"], "best %.1f ms first %.1f ms" % (d["best_s"] * 1e3, d["first_s"] * 1e3), "%.0f M reads/s" % (d["reads_per_s"] / 1e6), d["digest"], d["candidates"])
PY

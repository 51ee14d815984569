#!/bin/bash
# round 5, batch 37: row f2's event-rich case under the kernel trace (why 36 -> 60 ms)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b37; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python tests/mem_scale.py --no-reference --cases fwd,rc,fwd --out $O/mem_scale.json > $O/mem_scale.log 2>&1; grep '^{"fwd"\|^{"rc"' $O/mem_scale.log | cut -c1-700
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tests/mem_scale.py --no-reference --cases fwd --out $O/mem_scale2.json > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    t=int(r['TotalDurationNs'])/1e6
    if t>0.3: print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:7.3f} total {t:7.2f}")
PY

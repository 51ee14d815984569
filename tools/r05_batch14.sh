#!/bin/bash
# round 5, batch 14: micro-benchmarks: cache policies / allocation kinds of a random gather; an occupancy filter in the memory-side cache
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b14; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 200 tools/ubench/filter_mall > $O/filter_mall.txt 2>&1; echo "filter_mall rc=$?"; cat $O/filter_mall.txt
timeout -k 10 300 tools/ubench/gather_modes 3 > $O/gather_modes.txt 2>&1; echo "gather_modes rc=$?"; cat $O/gather_modes.txt
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RD_UNCACHED_32B FETCH_SIZE --output-format csv -d $O/pmc_modes -- tools/ubench/gather_modes 1 > $O/pmc_modes.txt 2>&1; echo "pmc rc=$?"
python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("$O/pmc_modes/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0]); acc.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), v in sorted(acc.items(), key=lambda kv: int(kv[0][0])):
    if k.startswith("k_"): print(d, k, {c: int(x) for c, x in v.items()})
PY

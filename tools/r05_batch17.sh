#!/bin/bash
# round 5, batch 17: SQ counters of the two dual kernels in one process (is the kernel bound by VALU issue?)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b17; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pass$i -- python3 tools/ab_match.py --workload C3 --rounds 1 PGRC_DUAL_VARIANT=5 PGRC_DUAL_VARIANT=0 > $O/pass$i.txt 2> $O/pass$i.err || echo "pass $i failed"
  echo "pass $i done"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_copmem_match_dual" in r["Kernel_Name"]:
            k = "r05a" if "r05a" in r["Kernel_Name"] else "new"
            acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
for c in sorted(acc):
    print(f"{c:28s}", {k: f"{sum(v)/len(v):.4g} (n={len(v)})" for k, v in acc[c].items()})
PY

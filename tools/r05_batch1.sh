#!/bin/bash
# round 5, batch 1: (a) the slimmed dual kernel + the options refactor + the index variants through the parity and golden suites,
# (b) PMC calibration of random gathers, (c) in-context A/B: dual kernel r04 / slim at 5 waves / slim at 6 waves; index variants
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b1; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
set -o pipefail
timeout -k 10 560 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3-avail list > $O/avail.txt 2>&1 || true)
i=0; mkdir -p $O/pmc_gather
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_IO TCC_REQ_sum" "TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_gather/pass$i -- ./tools/ubench/gather_pmc > $O/pmc_gather/pass$i.out 2> $O/pmc_gather/pass$i.err || echo "gather pmc pass $i ($grp) failed"
done
python3 tools/pmc_summary.py $O/pmc_gather > $O/pmc_gather_summary.txt 2>&1; cat $O/pmc_gather_summary.txt | head -60
timeout -k 10 300 python tools/ab_match.py --workload C3 --rounds 3 PGRC_DUAL_VARIANT=4 PGRC_DUAL_VARIANT=5 PGRC_DUAL_VARIANT=6 > $O/ab_dual_c3.txt 2>&1; echo "ab dual C3 rc=$?"; cat $O/ab_dual_c3.txt | tail -5
timeout -k 10 300 python tools/ab_match.py --workload C3-M3 --rounds 3 PGRC_DUAL_VARIANT=4 PGRC_DUAL_VARIANT=5 PGRC_DUAL_VARIANT=6 > $O/ab_dual_c3m3.txt 2>&1; echo "ab dual C3-M3 rc=$?"; cat $O/ab_dual_c3m3.txt | tail -5
timeout -k 10 300 python tools/ab_match.py --workload C3 --rounds 3 PGRC_INDEX_CFG=2 PGRC_INDEX_CFG=3 PGRC_INDEX_CFG=0 PGRC_INDEX_CFG=1 > $O/ab_index_c3.txt 2>&1; echo "ab index C3 rc=$?"; cat $O/ab_index_c3.txt | tail -6

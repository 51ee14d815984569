#!/bin/bash
# round 5, batch 18: product build: the non-temporal hint in modes d / i (A/B in one context), then the whole -m gpu suite, then the bench line
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b18; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python tools/ab_match.py --workload C3-d --rounds 3 PGRC_SEED_NT=0 PGRC_SEED_NT=1 PGRC_SEED_NT=2 PGRC_SEED_NT=3 > $O/ab_seed_nt_d.txt 2>&1; echo "rc=$?"; tail -4 $O/ab_seed_nt_d.txt
timeout -k 10 300 python tools/ab_match.py --workload C3-i --rounds 2 PGRC_SEED_NT=0 PGRC_SEED_NT=3 > $O/ab_seed_nt_i.txt 2>&1; echo "rc=$?"; tail -2 $O/ab_seed_nt_i.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; grep '^{' $O/bench_c3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'), d.get('boundary'), d['cpu_baseline']['value'])"

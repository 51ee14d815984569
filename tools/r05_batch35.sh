#!/bin/bash
# round 5, batch 35: the whole -m gpu suite and the default bench line at the round's last commit
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b35; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; grep '^{' $O/bench_c3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'), d['roofline']['index']['ms'], d.get('boundary',{}).get('reads_per_s'), d['cpu_baseline']['value'], d['parity_sample']['diff'])"

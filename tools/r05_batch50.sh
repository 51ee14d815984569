#!/bin/bash
# round 5, batch 50: at the round's last commit: the whole -m gpu suite, the default bench line, the three seed-mode bench lines, the long-text bench lines
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b50; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_all.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; grep '^{' $O/bench_c3.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'), d['roofline']['index']['ms'], d.get('boundary',{}).get('reads_per_s'), d['cpu_baseline']['value'], d['parity_sample']['diff'])"
for wl in C3-d C3-i C3-e; do
  timeout -k 10 300 python bench.py --workload $wl --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', 'parity', (d.get('parity_sample') or {}).get('diff'), 'cpu', d['cpu_baseline']['value'])"
done
for wl in C5-shard P64; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; grep '^{' $O/bench_$wl.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  ', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms; dual', d['roofline'].get('kernel_ms'), 'index', d['roofline'].get('index',{}).get('ms'), 'parity diff', (d.get('parity_sample') or {}).get('diff'))"
done

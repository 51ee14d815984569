#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b31; mkdir -p $O
for wl in C3-d C3-i C3-e; do
  PGRC_SEED_DEBUG=1 timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 1 --warmup 0 > $O/dbg_$wl.json 2> $O/dbg_$wl.err; echo $wl; grep "seed build" $O/dbg_$wl.err | head -4
done

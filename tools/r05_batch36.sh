#!/bin/bash
# round 5, batch 36: the widening rows re-measured with the round's library (f2 with its own sorts, f1, f3); PMC traffic of C3-M3, C5-shard, C3-i, C3-e at the final kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b36; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python tests/mem_scale.py --no-reference --cases rc,fwd,lq --out $O/mem_scale.json > $O/mem_scale.log 2>&1; echo "mem_scale rc=$?"; tail -6 $O/mem_scale.log | cut -c1-300
timeout -k 10 300 python tools/divide_rate.py > $O/divide_rate.json 2> $O/divide_rate.err; echo "divide rc=$?"; tail -c 900 $O/divide_rate.json; echo
timeout -k 10 400 python tools/stage4_c3.py > $O/stage4_c3.json 2> $O/stage4_c3.err; echo "stage4 rc=$?"; tail -c 900 $O/stage4_c3.json; echo
for wl in C3-M3 C5-shard; do
  bash tools/pmc_groups.sh $O/pmc_$wl "FETCH_SIZE" "WRITE_SIZE" -- --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
  python3 tools/pmc_traffic.py $O/pmc_$wl k_copmem_match_ $O/$(echo $wl | tr 'A-Z' 'a-z')_traffic.json 1 dual | head -c 300; echo
done
for wl in C3-i C3-e; do
  bash tools/pmc_groups.sh $O/pmc_$wl "FETCH_SIZE" "WRITE_SIZE" -- --workload $wl --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1
  python3 tools/pmc_seed_traffic.py $O/pmc_$wl $O/$(echo $wl | tr 'A-Z' 'a-z')_traffic.json $wl
done

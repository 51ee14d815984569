#!/bin/bash
# round 5, batch 8: rocprofv3 kernel stats + PMC traffic of the default bench command (C3), PMC traffic of C3-M3 and the C5 shard,
# stream timing of the boundary, the N = 2 rehearsal (gloo, two ranks on one GPU)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b8; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
bash tools/profile_c3.sh $O/profile_c3 > $O/profile_c3.log 2>&1; echo "profile rc=$?"; tail -25 $O/profile_c3.log
for w in C3-M3 C5-shard; do
  bash tools/pmc_groups.sh $O/pmc_$w "FETCH_SIZE" "WRITE_SIZE" -- --workload $w --no-cpu-baseline --no-boundary --parity-sample-reads 0 --steps 2 --warmup 1 > $O/pmc_$w.log 2>&1
  python3 tools/pmc_traffic.py $O/pmc_$w k_copmem_match_ $O/traffic_$w.json 1 dual > /dev/null; echo "$w traffic rc=$?"
done
PGRC_STREAM_TIMING=1 timeout -k 10 300 python tools/boundary_c3.py --legs pipelined --reps 2 > $O/boundary.json 2> $O/boundary_timing.txt; echo "boundary rc=$?"; tail -40 $O/boundary_timing.txt
PGRC_BENCH_FORCE_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong --workload C3-PE --dist-backend gloo > $O/bench_n2_strong_gloo.json 2> $O/bench_n2.err; echo "n2 rc=$?"; head -c 600 $O/bench_n2_strong_gloo.json
rm -rf $O/profile_c3/trace $O/profile_c3/pmc/pass*/ $O/pmc_*/pass*/

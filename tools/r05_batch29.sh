#!/bin/bash
# round 5, batch 29: smoke(), the N = 2 rehearsals of bench.py (gloo, two ranks on the one GPU) at the round's last kernels, the soaks
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05_b29; mkdir -p $O
(while sleep 50; do echo "... $(date +%T)"; done) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
PGRC_BENCH_FORCE_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong --workload C3-PE --dist-backend gloo > $O/bench_n2_strong_gloo.json 2> $O/bench_n2.err; echo "n2 strong rc=$?"; grep '^{' $O/bench_n2_strong_gloo.json | head -c 500; echo
PGRC_BENCH_FORCE_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --workload C2 --dist-backend gloo > $O/bench_n2_weak_gloo.json 2> $O/bench_n2w.err; echo "n2 weak rc=$?"; grep '^{' $O/bench_n2_weak_gloo.json | head -c 500; echo
timeout -k 10 500 python tests/soak.py 240 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -3 $O/soak.log
timeout -k 10 400 python tests/soak_medium.py 200 > $O/soak_medium.log 2>&1; echo "soak_medium rc=$?"; tail -3 $O/soak_medium.log

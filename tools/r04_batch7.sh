#!/bin/bash
# round-4 batch 7: the -t 1 identity run at 5 M x 150 bp, partitioned-probing micro-benchmark, stage 4 at C3 size with the
# position order made on the device / by the reference's host sort
cd ${GRAFT_REPO_ROOT:-.}
(while sleep 60; do date >> gpurun_out/r04_batch7_heartbeat.txt; done) &
HB=$!
python tools/ubench_partjoin.py > gpurun_out/r04_ubench_partjoin.txt 2>&1; cat gpurun_out/r04_ubench_partjoin.txt
python tools/stage4_c3.py --cpu-reads 0 > gpurun_out/r04_stage4_c3_device_sort.json 2> gpurun_out/r04_stage4_c3_device_sort.err; cut -c1-1200 gpurun_out/r04_stage4_c3_device_sort.json
PGRC_DEVICE_SORT=0 python tools/stage4_c3.py --cpu-reads 0 > gpurun_out/r04_stage4_c3_host_sort.json 2> gpurun_out/r04_stage4_c3_host_sort.err; cut -c1-600 gpurun_out/r04_stage4_c3_host_sort.json
W=/tmp/pgrc_e2e_id; rm -rf $W; mkdir -p $W
python tests/e2e_big.py $W --reads 5000000 --identity > gpurun_out/r04_e2e_identity_5m.log 2> gpurun_out/r04_e2e_identity_5m.err
tail -1 gpurun_out/r04_e2e_identity_5m.log > gpurun_out/r04_e2e_identity_5m.json; cut -c1-1300 gpurun_out/r04_e2e_identity_5m.json
rm -rf $W
kill $HB

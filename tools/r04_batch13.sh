#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
timeout -k 5 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r04_pmc_mode_d -- python3 tools/modes_c3.py d > gpurun_out/r04_pmc_mode_d.log 2> gpurun_out/r04_pmc_mode_d.err
python3 tools/pmc_summary.py gpurun_out/r04_pmc_mode_d > gpurun_out/r04_pmc_mode_d_summary.txt
grep -A5 "k_seed_scan\|k_seed_hamming\|k_seed_insert\|k_seed_replay$" gpurun_out/r04_pmc_mode_d_summary.txt | head -40
python bench.py --workload C2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_C2.json 2> gpurun_out/r04_bench_C2.err
python -c "
import json;d=json.load(open('gpurun_out/r04_bench_C2.json'));print('C2', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],2), 'ms', d['phases_ms']['schedule'], 'parity diff', d['parity_sample']['diff'])"

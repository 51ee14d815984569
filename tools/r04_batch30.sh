#!/bin/bash
# round-4 profiles at the last commit: C3 bench under rocprofv3 (kernel stats) + the PMC passes, modes d / i / e kernel stats and mode d counters
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
bash tools/profile_c3.sh gpurun_out/profile_c3_final > gpurun_out/profile_c3_final.log 2>&1; tail -5 gpurun_out/profile_c3_final.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for m in d i e; do
  OUT=$ROOT/gpurun_out/prof_mode_${m}_final
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/modes_c3.py $m > $OUT/run.jsonl 2> $OUT/trace.err
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rm -rf $OUT/trace
done
OUT=$ROOT/gpurun_out/pmc_mode_d_final
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 tools/modes_c3.py d > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
find gpurun_out/profile_c3_final gpurun_out/pmc_mode_d_final -name "*.csv" -size +8M -delete
ls gpurun_out/profile_c3_final

#!/bin/bash
# final evidence of round 4: the default bench line at HEAD, kernel stats of modes d / i / e, counters of mode d
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
python bench.py > gpurun_out/r04_final_c3_bench.json 2> gpurun_out/r04_final_c3_bench.err; cut -c1-300 gpurun_out/r04_final_c3_bench.json
for m in d i e; do
  OUT=$ROOT/gpurun_out/prof_mode_${m}_final
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/modes_c3.py $m > $OUT/run.jsonl 2> $OUT/trace.err
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rm -rf $OUT/trace
  echo "mode $m"; head -8 $OUT/kernel_stats.csv | cut -c1-60,100-170
done
OUT=$ROOT/gpurun_out/pmc_mode_d_final
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 tools/modes_c3.py d > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || echo "pass $i failed"
  echo "pass $i ($grp) done"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"; grep -A5 "k_seed" $OUT/summary.txt | head -60
find $OUT -name "*.csv" -size +20M -delete

#!/bin/bash
# Builds compile-time variants of the match kernel next to the product library (pgrc_amd/variants/, git-ignored) and
# benches them back to back on ONE box (box-to-box variance is ~10 %): tools/variants.sh build | run
# (the first round of a fresh box runs a few percent slower: read the later rounds)
set -eu
cd "$(dirname "$0")/.."
V=pgrc_amd/variants
declare -A DEFS=( [base]="" [w7]="-DMATCH_WAVES_PER_EU=7" [w8]="-DMATCH_WAVES_PER_EU=8" [w5]="-DMATCH_WAVES_PER_EU=5" [vc8]="-DVC_BITS=3" [vc2]="-DVC_BITS=1" [noahead]="-DPROBE_AHEAD=0" [chunk256]="-DMATCH_CHUNK=256u" [d6]="-DDUAL_WAVES_PER_EU=6 -DMATCH_STAGE=16" [s16]="-DMATCH_STAGE=16" [vc2s16]="-DVC_BITS=1 -DMATCH_STAGE=16" [d6vc2]="-DDUAL_WAVES_PER_EU=6 -DMATCH_STAGE=16 -DVC_BITS=1" )
ORDER="${VARIANTS:-base w7 w8 vc8}"
if [ "${1:-build}" = build ]; then
  mkdir -p $V
  for v in "${!DEFS[@]}"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ipgrc_amd/csrc ${DEFS[$v]} -c pgrc_amd/csrc/copmem.hip -o $V/copmem_$v.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libpgrc_match_$v.so $V/copmem_$v.o $(ls pgrc_amd/csrc/build/*.o | grep -v copmem.o)
    echo built $v
  done
else
  for rep in 1 2 3; do
    for v in $ORDER; do
      PGRC_MATCH_LIB=$PWD/$V/libpgrc_match_$v.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$v', round(d['ms_per_step'],1), round(p['match_fwd'],1), round(p['match_rc'],1), d['counters']['verifies'])"
    done
  done
fi

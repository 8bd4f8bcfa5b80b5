#!/bin/bash
# Same-box A/B of library VARIANTS (signerf_amd/libsignerf_hip_<name>.so, built with signerf_amd.build.build(extra_flags=..., out_path=...))
# of the wide K1 shape against the product library in both shapes; per-launch HIP-event medians (tools/ab_bench.py), 2 interleaved rounds.
#   tools/ab_variants.sh <out_file> <name> [<name> ...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
one() {  # label, lib-or-empty, wide
  if [ -n "$2" ]; then export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$2.so; else unset SIGNERF_HIP_LIB; fi
  r=$(python tools/ab_bench.py SN_K1_WIDE $3 --rounds ${ROUNDS:-60} --config ${CONFIG:-bench} --precision fp16x2 ${SIZE:+--size $SIZE} 2>/dev/null | tail -1)
  echo "$1 $r" | tee -a "$OUT"
}
for rep in 1 2; do
  one "base(4-wave WG, 3/SIMD)" "" 0
  one "wide(product lib)" "" 1
  for v in "$@"; do one "wide($v)" "$v" 1; done
done

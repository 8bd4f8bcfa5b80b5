#!/bin/bash
# Same-box A/B of library VARIANTS (signerf_amd/libsignerf_hip_<name>.so, built with signerf_amd.build.build(extra_flags=..., out_path=...))
# against the product library: per-launch HIP-event medians (tools/ab_bench.py), REPS interleaved rounds.
#   [CONFIG=bench|proposal] [SIZE=..] [ROUNDS=60] [REPS=2] [PRECISION=fp16x2|fp32|fp16] [IMPL=tcnn] tools/ab_libs.sh <out_file> <name> [<name> ...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
one() {  # label, lib-or-empty
  if [ -n "$2" ]; then export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$2.so; else unset SIGNERF_HIP_LIB; fi
  r=$(python tools/ab_bench.py SN_AB_DUMMY 0 --rounds ${ROUNDS:-60} --config ${CONFIG:-bench} --precision ${PRECISION:-fp16x2} ${IMPL:+--implementation $IMPL} ${SIZE:+--size $SIZE} 2>/dev/null | tail -1)
  echo "$1 $r" | tee -a "$OUT"
}
for rep in $(seq 1 ${REPS:-2}); do
  one "product" ""
  for v in "$@"; do one "$v" "$v"; done
done

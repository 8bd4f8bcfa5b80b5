#!/bin/bash
# Same-box A/B of library VARIANTS (signerf_amd/libsignerf_hip_<name>.so) against the product library on the TRAINED scene
# (tools/trained_bench.py: ms per frame with the exact early termination on / off), REPS interleaved rounds.
#   [REPS=2] tools/ab_trained_libs.sh <out_file> <name> [<name> ...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
one() {  # label, lib-or-empty
  if [ -n "$2" ]; then export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$2.so; else unset SIGNERF_HIP_LIB; fi
  python tools/trained_bench.py --rounds ${ROUNDS:-6} --frames 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', ' | '.join('%s: off %.3f on %.3f ms' % (l['frame'].split(',')[0] + (' x64' if 'uniform' in l['frame'] else ''), l['ms_per_frame']['early_term_off'], l['ms_per_frame']['early_term_on']) for l in d['legs']))" | tee -a "$OUT"
}
for rep in $(seq 1 ${REPS:-2}); do
  one "product" ""
  for v in "$@"; do one "$v" "$v"; done
done

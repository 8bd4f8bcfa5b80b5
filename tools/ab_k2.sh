#!/bin/bash
# Same-box A/B of library variants on the 1080p nerfacto frame (K2 + K1 in bins mode), one stream and two frames in flight:
#   tools/ab_k2.sh <name> [<name> ...]     variants = signerf_amd/libsignerf_hip_<name>.so, "base" = the product library; interleaved, 3 rounds
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for rep in 1 2 3; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset SIGNERF_HIP_LIB; else export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$v.so; fi
    python bench.py --workload nerfacto1080 --steps 60 --warmup 6 --no-cpu-baseline --no-alt-precision --no-others --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'per launch median', round(d['kernel_ms']['median'],3), 'p05', round(d['kernel_ms']['p05'],3), '| frame period (2 in flight)', round(d['ms_per_step'],3), '| clock', round(d['roofline'].get('sustained_clock_ghz') or 0,3))"
  done
done

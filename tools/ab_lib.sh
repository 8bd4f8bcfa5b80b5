#!/bin/bash
# A/B of library variants built with signerf_amd.build.build(extra_flags=..., out_path=signerf_amd/libsignerf_hip_<name>.so):
#   tools/ab_lib.sh <name> [<name> ...]     (the product library is always included as "base"); interleaved, 2 rounds
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for rep in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset SIGNERF_HIP_LIB; else export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$v.so; fi
    python bench.py ${AB_ARGS:---steps 150 --warmup 10} --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'median', round(d['kernel_ms']['median'],4), 'p05', round(d['kernel_ms']['p05'],4), 'clock', d['roofline'].get('sustained_clock_ghz'))"
  done
done

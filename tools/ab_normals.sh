#!/bin/bash
# Same-box A/B of normals-kernel (K3) variants: tools/ab_normals.sh <out_file> <variant> ...   (variant = suffix of signerf_amd/libsignerf_hip_<variant>.so)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for rep in 1 2; do
  for v in product "$@"; do
    if [ "$v" = product ]; then unset SIGNERF_HIP_LIB; else export SIGNERF_HIP_LIB=$PWD/signerf_amd/libsignerf_hip_$v.so; fi
    echo "[$v] $(python tools/normals_bench.py --steps 20 --table-scale 1e-3 2>/dev/null | tail -1)" | tee -a "$OUT"
  done
done
unset SIGNERF_HIP_LIB
echo "[product] $(python tools/normals_bench.py --workload nerfacto1080 --steps 10 --table-scale 1e-3 2>/dev/null | tail -1)" | tee -a "$OUT"

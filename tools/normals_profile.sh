#!/bin/bash
# rocprofv3 evidence for the normals kernel K3: kernel stats + SQ / TCP PMC passes of `tools/normals_bench.py`
# (800x800x64, tables x 1e-3: the split-precision path a real checkpoint takes).   tools/normals_profile.sh <out_dir>
set -u
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/tools/normals_bench.py" --steps 20 --table-scale 1e-3 > "$OUT/stats.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$OUT/stats" > "$OUT/kernel_stats.txt" 2>&1
for pass in "sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "sq_b SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
            "tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/pmc/$name" -- python "$ROOT/tools/normals_bench.py" --steps 3 --table-scale 1e-3 > "$OUT/pmc_$name.log" 2>&1
  echo "pass $name rc=$?"
done
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc" "sn_normals_kernel<0, 0, 1" > "$OUT/pmc_summary.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
head -8 "$OUT/kernel_stats.txt"; tail -14 "$OUT/pmc_summary.txt"

#!/bin/bash
# r04 evidence for the 4-waves-per-SIMD K1 variant: rocprofv3 kernel stats, SQ PMC passes and the power / clock table,
# base shape vs SN_K1_WIDE=1, one box.   tools/k1_wide_evidence.sh <out_dir>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/r04_k1_wide}
mkdir -p "$OUT"
cd "$ROOT"
python tools/power_ab.py --rounds 2 --only base,wide4 > "$OUT/power_ab.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  export SN_K1_WIDE=$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_w$w" -- python "$ROOT/bench.py" --frames-in-flight 1 --steps 60 --no-cpu-baseline --no-alt-precision --no-others --no-traffic > "$OUT/stats_w$w.log" 2>&1
  python "$ROOT/tools/rocprof_summary.py" "$OUT/stats_w$w" > "$OUT/kernel_stats_w$w.txt" 2>&1
  for pass in "sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "sq_b SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
              "sq_c SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/pmc_w$w/$name" -- python "$ROOT/bench.py" --frames-in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --no-others --no-traffic > "$OUT/pmc_w${w}_$name.log" 2>&1
    echo "w=$w pass $name rc=$?"
  done
  python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_w$w" sn_render_main > "$OUT/pmc_summary_w$w.txt" 2>&1
done
unset SN_K1_WIDE
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
find "$OUT" -name "*.db" -delete 2>/dev/null
cat "$OUT/power_ab.txt" | tail -5; tail -12 "$OUT/pmc_summary_w0.txt"; tail -12 "$OUT/pmc_summary_w1.txt"; head -6 "$OUT/kernel_stats_w0.txt" "$OUT/kernel_stats_w1.txt"

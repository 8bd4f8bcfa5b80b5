#!/bin/bash
# Collect PMC counters for bench.py in SEPARATE rocprofv3 passes (counter slots: SQ 8, TCC 4, GRBM 2 per pass;
# never combined with --sys-trace etc.).  Usage (on the GPU box, from the repo root):
#   tools/pmc_passes.sh <out_dir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -- \
      python "$ROOT/bench.py" --frames-in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision --no-others --no-traffic ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run_pass tcc_fetch   FETCH_SIZE
run_pass tcc_write   WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pass tcc_ea      TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run_pass tcc_dram    TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum
run_pass sq_a        SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq_b        SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
run_pass tcp         TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE


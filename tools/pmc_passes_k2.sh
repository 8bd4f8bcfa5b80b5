#!/bin/bash
# PMC passes for the nerfacto1080 workload (proposal kernel K2 + main kernel K1<1>); see tools/pmc_passes.sh
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -- \
      python "$ROOT/bench.py" --frames-in-flight 1 --workload nerfacto1080 --width 960 --height 540 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision --no-others --no-traffic > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run_pass sq_a   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pass sq_b   SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS

run_pass tcp    TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE


#!/bin/bash
# Split-depth tail of K1 (sn_api.hip plan_tail) on / off over frame sizes, frames issued back to back on one stream (steady clocks; a frame
# timed on its own after an idle gap runs at whatever the power state happens to be):   tools/tail_split_sweep.sh [sizes...]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for s in "${@:-64 128 200 256 320 400 512 560 640 800 1024}"; do
  for sp in 1 0; do
    SN_TAIL_SPLIT=$sp python bench.py --width $s --height $s --frames-in-flight 1 --steps 200 --warmup 20 --no-cpu-baseline --no-alt-precision 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('size $s split $sp: frame period', round(d['ms_per_step'],4), 'ms; per-launch median', round(d['kernel_ms']['median'],4), 'p05', round(d['kernel_ms']['p05'],4))"
  done
done

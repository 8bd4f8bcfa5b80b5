#!/bin/bash
# Are the eight XCDs' clocks under K1 persistently different (then a static 1/8 split of every frame runs at the SLOWEST XCD's pace), or sampling noise?
# 12 amd-smi samples, 0.4 s apart, of GFX_0..7 while the bench frame renders back to back.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
(python tools/power_ab.py --one base --seconds 10 > /tmp/load.json 2>&1) &
LOAD=$!
sleep 4
for i in $(seq 1 12); do
  /opt/rocm/bin/amd-smi metric -g 0 --clock 2>/dev/null | awk '/GFX_[0-7]:/{g=$1} /^ +CLK:/{if(g!=""){printf "%s %s  ", g, $2; g=""}} END{print ""}'
  sleep 0.4
done
wait $LOAD
tail -1 /tmp/load.json

#!/bin/bash
# which limiter holds the clock under K1?  samples amd-smi / rocm-smi while the bench frame renders back to back
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
(python tools/power_ab.py --one base --seconds 8 > /tmp/load.json 2>&1) &
LOAD=$!
sleep 5
echo "== amd-smi metric (under load)"; timeout 20 /opt/rocm/bin/amd-smi metric -g 0 --power --clock --temperature --throttle 2>&1 | head -120
echo "== rocm-smi (under load)"; timeout 20 /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showperflevel --showprofile --showvoltage 2>&1 | head -60
wait $LOAD
cat /tmp/load.json | tail -1

#!/bin/bash
# Experiment (r04): does it pay to split K1's tiles between the XCDs in proportion to the clock each die sustains under load, instead of 1/8 each?
# Needs the kernel-side knob: `patch -p0 < tools/patches/xcd_weights_experiment.patch && python -m signerf_amd.build --force` (r04: measured, no gain, not kept in the library).
#   1. per-XCD clocks under the back-to-back bench frame (amd-smi, mean of 8 samples)   2. bench.py with SN_XCD_WEIGHTS = those clocks, = equal, = the
#   inverse (sanity: must be slower), interleaved.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
(python tools/power_ab.py --one base --seconds 9 > /tmp/load.json 2>&1) &
LOAD=$!
sleep 4
: > /tmp/clk.txt
for i in $(seq 1 8); do
  /opt/rocm/bin/amd-smi metric -g 0 --clock 2>/dev/null | awk '/GFX_[0-7]:/{g=$1} /^ +CLK:/{if(g!=""){printf "%s ", $2; g=""}} END{print ""}' >> /tmp/clk.txt
  sleep 0.4
done
wait $LOAD
W=$(awk '{for(i=1;i<=8;i++)s[i]+=$i; n++} END{for(i=1;i<=8;i++)printf "%d%s", s[i]/n, (i<8?",":"")}' /tmp/clk.txt)
INV=$(awk '{for(i=1;i<=8;i++)s[i]+=$i; n++} END{for(i=1;i<=8;i++)printf "%d%s", 3800-s[i]/n, (i<8?",":"")}' /tmp/clk.txt)
echo "per-XCD clocks under load (MHz, mean of 8 samples): $W"
one() {  # label, weights-or-empty
  if [ -n "$2" ]; then export SN_XCD_WEIGHTS=$2; else unset SN_XCD_WEIGHTS; fi
  python bench.py --steps 200 --warmup 10 --no-others --no-traffic --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'frame period (2 in flight) %.4f ms' % d['ms_per_step'], '| per launch median %.4f' % d['kernel_ms']['median'], '| clock %.3f' % d['roofline']['sustained_clock_ghz'])"
}
for rep in 1 2 3; do
  one "equal   " ""
  one "weighted" "$W"
  one "inverse " "$INV"
done

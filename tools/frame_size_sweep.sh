#!/bin/bash
# K1's per-launch time against the frame size (one stream): a launch is a whole number of 64-ray waves on 3072 wave slots, so the cost per
# sample is a staircase in the number of rounds ("Frames in flight").   bash tools/frame_size_sweep.sh
cd $GRAFT_REPO_ROOT
for wh in "768 768" "800 800" "832 800" "864 800" "896 800" "1024 768" "1024 832"; do set -- $wh
python bench.py --frames-in-flight 1 --width $1 --height $2 --steps 100 --warmup 5 --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w,h=$1,$2; tiles=((w+7)//8)*((h+7)//8); k=d['kernel_ms']['median']
print('%dx%d tiles %d rounds %.3f  kernel %.4f ms  ps/sample %.3f' % (w,h,tiles,tiles/3072, k, k*1e9/(w*h*64)))"
done

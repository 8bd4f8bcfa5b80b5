cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_wide.py -m gpu -q > gpurun_out/r04b/wide_test.log 2>&1; tail -8 gpurun_out/r04b/wide_test.log
python tools/ab_bench.py SN_K1_WIDE 0 1 --rounds 40 --config bench --precision fp16x2 > gpurun_out/r04b/ab_wide_800.txt 2>&1; cat gpurun_out/r04b/ab_wide_800.txt
python tools/ab_bench.py SN_K1_WIDE 0 1 --rounds 20 --config proposal --precision fp16x2 --size 1080 > gpurun_out/r04b/ab_wide_prop.txt 2>&1; cat gpurun_out/r04b/ab_wide_prop.txt
for w in 0 1; do SN_K1_WIDE=$w python bench.py --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide=$w', 'ms_per_step', round(d['ms_per_step'],4), 'kernel median', round(d['kernel_ms']['median'],4), 'clock', d['roofline'].get('sustained_clock_ghz'))"; done | tee gpurun_out/r04b/bench_wide.txt

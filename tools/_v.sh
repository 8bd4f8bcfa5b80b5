timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/ab_bench.py X 0 --precision fp16x2 --config proposal --size 1080 --rounds 8
python tools/determinism_probe.py --n 6 --config proposal | tail -1

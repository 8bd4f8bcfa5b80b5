#!/bin/bash
# End-of-round evidence run on the GPU box (from the repo root): bench lines, rocprofv3 kernel stats of the same commands, PMC passes
# of K1 and K2, concurrency / determinism probes.  Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-refresh}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python bench.py > "$OUT/bench_sheet64.json" 2> "$OUT/bench_sheet64.err"
timeout 600 python bench.py --workload nerfacto1080 --steps 60 --warmup 5 --no-others --no-traffic > "$OUT/bench_nerfacto1080.json" 2> "$OUT/bench_nerfacto1080.err"
timeout 300 python tools/kernel_counts.py > "$OUT/kernel_counts.txt" 2>&1
timeout 300 python tools/kernel_counts.py sn_render_main_kernelILi1ELi1ELi0ELi11E >> "$OUT/kernel_counts.txt" 2>&1
timeout 300 python -c "import sys; sys.path.insert(0, 'tools'); import kernel_counts as k; print('K2 marching loops (net 0, net 1):', [{x: c.get(x, 0) for x in ('valu', 'mfma', 'gather', 'packed_f32')} for c in k.mfma_loops('sn_proposal_kernelILi0ELi5ELi4E')])" >> "$OUT/kernel_counts.txt" 2>&1
timeout 600 python bench.py --scaling strong --steps 40 --no-cpu-baseline --no-others --no-traffic > "$OUT/bench_strong_n1.json" 2> "$OUT/bench_strong_n1.err"
timeout 300 python tools/normals_bench.py > "$OUT/normals_bench.txt" 2>&1
timeout 300 python tools/normals_bench.py --table-scale 1e-3 >> "$OUT/normals_bench.txt" 2>&1
timeout 300 python tools/normals_bench.py --workload nerfacto1080 --table-scale 1e-3 >> "$OUT/normals_bench.txt" 2>&1
timeout 600 python tools/config5_bench.py --size 800 > "$OUT/config5_800.txt" 2>&1
timeout 600 python tools/config5_bench.py --size 512 > "$OUT/config5_512.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_sheet64" -- python "$ROOT/bench.py" --frames-in-flight 1 --no-cpu-baseline --no-others --no-traffic > "$OUT/prof_sheet64.log" 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_nerfacto1080" -- python "$ROOT/bench.py" --frames-in-flight 1 --workload nerfacto1080 --steps 30 --warmup 3 --no-cpu-baseline --no-others --no-traffic > "$OUT/prof_nerfacto1080.log" 2>&1)
python tools/rocprof_summary.py "$OUT/prof_sheet64" > "$OUT/kernel_stats_sheet64.txt" 2>&1
python tools/rocprof_summary.py "$OUT/prof_nerfacto1080" > "$OUT/kernel_stats_nerfacto1080.txt" 2>&1
bash tools/pmc_passes.sh "$OUT/pmc_k1" > "$OUT/pmc_k1.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc_k1" "sn_render_main_kernel<0, 1" --json fp16x2 > "$OUT/pmc_k1_summary.txt" 2>&1
cp profiles/traffic.json "$OUT/traffic.json"
bash tools/normals_profile.sh "$OUT/normals" > "$OUT/normals_profile.log" 2>&1
timeout 600 python tools/fp16_mode_bench.py > "$OUT/fp16_mode_bench.txt" 2>&1
bash tools/pmc_passes_k2.sh "$OUT/pmc_k2" > "$OUT/pmc_k2.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc_k2" sn_proposal_kernel > "$OUT/pmc_k2_summary.txt" 2>&1
for m in shared two-models vs-uniform; do
  echo "== concurrency_probe --mode $m (renders unordered: the default)" >> "$OUT/probes.txt"
  timeout 300 python tools/concurrency_probe.py --mode $m 2>&1 | tail -3 >> "$OUT/probes.txt"
done
echo "== determinism_probe" >> "$OUT/probes.txt"
timeout 300 python tools/determinism_probe.py 2>&1 | tail -3 >> "$OUT/probes.txt"
# r05: the trained scene (fitted once, cached in /tmp for the legs below), the T = 2^21 PMC passes, the copy-budget curve, the early-termination A/B
timeout 600 python tools/trained_bench.py > "$OUT/trained_bench.json" 2> "$OUT/trained_bench.err"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trained" -- python "$ROOT/tools/trained_bench.py" --rounds 2 --frames 4 > "$OUT/prof_trained.log" 2>&1)
python tools/rocprof_summary.py "$OUT/prof_trained" > "$OUT/kernel_stats_trained.txt" 2>&1
BENCH_ARGS="--log2-hashmap-size 21" bash tools/pmc_passes.sh "$OUT/pmc_t21" > "$OUT/pmc_t21.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc_t21" "sn_render_main_kernel<0, 1" > "$OUT/pmc_t21_summary.txt" 2>&1
timeout 300 python tools/dense_sweep.py > "$OUT/dense_curve.txt" 2>&1
timeout 300 python tools/early_term_ab.py > "$OUT/early_term_ab.txt" 2>&1
timeout 900 python tools/full_frame_parity.py --only trained800 --crop 400 --out "$OUT/trained_parity_400.jsonl" > "$OUT/trained_parity_400.txt" 2>&1
timeout 1500 python tools/soak_random_parity.py --trained --n 600 > "$OUT/soak_trained.txt" 2>&1
# drop the bulky raw traces, keep the stats
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
ls "$OUT"

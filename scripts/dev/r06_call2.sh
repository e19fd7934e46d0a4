#!/bin/bash
# round 6, second GPU visit: the new tests (C3-size exchange, composition golden, tight harness pins), the gated issue ubench,
# SQ counters, and the dense scene's own PMC / SQ passes.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
./scripts/ubench/issue_clock.bin > gpurun_out/r06_issue_clock.txt 2>&1; cat gpurun_out/r06_issue_clock.txt
python -m pytest tests/test_render_gpu.py tests/test_blend_variants_gpu.py tests/test_harness_pin_gpu.py tests/test_abi_gpu.py -m gpu -x -q --tb=short \
  -k "golden or harness_scene or trajectory or abi" 2>&1 | tail -15 > gpurun_out/r06b_pytest_new.log; tail -5 gpurun_out/r06b_pytest_new.log
( time python -m pytest tests/test_dist_gpu.py -m gpu -x -q --tb=short --durations=12 ) 2>&1 | tail -40 > gpurun_out/r06b_pytest_dist.log; tail -25 gpurun_out/r06b_pytest_dist.log
bash scripts/gpu_sq.sh C2 sq > gpurun_out/sq.log2 2>&1
PMC_ARGS="--scene dense" PMC_SUFFIX=dense bash scripts/gpu_pmc.sh > gpurun_out/pmc_dense.log 2>&1
PMC_ARGS="--scene dense" bash scripts/gpu_sq.sh C2 sq_dense > gpurun_out/sq_dense.log2 2>&1
grep -i "blend" gpurun_out/pmc_WRITE_SIZE_summary_dense.csv gpurun_out/pmc_FETCH_SIZE_summary_dense.csv gpurun_out/sq_dense_summary.txt | cut -c1-300
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-harness 2>gpurun_out/r06b_bench_stderr.log | tail -1 > gpurun_out/r06b_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06b_bench.json"))
print("ms/step", d["ms_per_step"], "flavours", d["config"].get("blend_waves_per_tile"))
r = d["roofline"]; print({k: r.get(k) for k in ("frac", "valu_frac", "valu_frac_of_achievable", "traffic_error")})
print("dense", d["dense_scene"]["ms_per_step"], {k: (d["dense_scene"].get("roofline") or {}).get(k) for k in ("frac", "valu_frac", "traffic_error")}, d["dense_scene"].get("roofline_error"))
PY

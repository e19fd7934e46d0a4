#!/bin/bash
# Quick GPU-box visit while iterating on the blend kernels: raster/render parity tests + a short bench with per-kernel times.
#   gpurun -- 'bash scripts/gpu_quick.sh [tag] [pytest-selection...]'
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-quick}; shift || true
sel=${*:-tests/test_raster_gpu.py tests/test_render_gpu.py tests/test_fast_step_gpu.py}
python -m pytest $sel -m gpu -x -q --tb=short 2>&1 | tail -40 > gpurun_out/${tag}_pytest.log
python bench.py --steps 60 --warmup 10 --profile-all --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench.json
python - "$tag" <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open("gpurun_out/%s_bench.json" % t))
    k = d["kernels_ms"]
    print("ms/step %.4f  it/s %.1f  R %d" % (d["ms_per_step"], d["value"], d["config"]["num_rendered"]))
    print(" ".join("%s=%.1f" % (n, 1e3 * v["avg_ms"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["avg_ms"])))
    if d.get("tracking_step"): print("tracking ms/iter %.4f" % d["tracking_step"]["ms_per_iter"])
except Exception as e:
    print("bench failed:", e, open("gpurun_out/%s_bench.json" % t).read()[-2000:])
PY
cat gpurun_out/${tag}_pytest.log | tail -4

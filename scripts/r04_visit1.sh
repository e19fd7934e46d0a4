#!/bin/bash
# round 4, first GPU visit: new tests, the bench line with the harness / drop-in extras, the 8-rank one-GPU smoke,
# the lane-utilisation counters (diagnostics flavour built on the box)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_fast_step_gpu.py tests/test_staging_gpu.py tests/test_harness_gpu.py tests/test_harness_pin_gpu.py -x -q 2>&1 | tail -15
echo "== bench (driver flags)"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_v1_bench.json 2> gpurun_out/r04_v1_bench.err; echo rc $?; tail -3 gpurun_out/r04_v1_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_v1_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "blocks", d["timed_blocks"]["blocks"], d["timed_blocks"]["ms_per_step_min"], d["timed_blocks"]["ms_per_step_max"])
    print("roofline", {k: d["roofline"].get(k) for k in ("frac", "valu_frac", "valu_busy", "avg_kernel_ms")})
    print("harness", json.dumps(d.get("harness"), indent=1)[:6000])
    print("drop_in", json.dumps(d.get("drop_in_step"), indent=1))
    print("also", d["config"]["also_measured"])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== 8 ranks on one GPU (gloo)"; FSGS_DIST_ONE_GPU=1 FSGS_BENCH_EXTRAS_DEADLINE=400 timeout 1200 python bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/r04_bench_one_gpu_8ranks.json 2> gpurun_out/r04_bench_one_gpu_8ranks.err; echo rc $?; tail -5 gpurun_out/r04_bench_one_gpu_8ranks.err; head -c 3000 gpurun_out/r04_bench_one_gpu_8ranks.json
echo "== lane utilisation"; FSGS_DIAG=1 python free-surgs_amd/build.py && FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so timeout 900 python scripts/lane_utilisation.py 2>&1 | tail -12

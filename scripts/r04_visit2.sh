#!/bin/bash
# round 4, second GPU visit: the new tests, the harness numbers with the one-launch frame begin, the achieved-error records of
# the full-size parity tests, 60 runs of the pinned harness (distribution of the post-densification deviation)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r04_full_size_parity.jsonl gpurun_out/r04_outlier_statistics.jsonl
echo "== tests"; timeout 900 python -m pytest tests/test_fast_step_gpu.py tests/test_staging_gpu.py tests/test_harness_gpu.py tests/test_harness_pin_gpu.py tests/test_abi_gpu.py -x -q 2>&1 | grep -v Warning | tail -12
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_v2_bench.json 2> gpurun_out/r04_v2_bench.err; echo rc $?
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_v2_bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    h = d["harness"]["progressive"]
    print("progressive", {k: h[k] for k in ("tracking_ms_per_frame", "tracking_iterations_ms_per_frame", "per_frame_setup_ms", "mapping_ms_per_frame", "ms_per_frame", "first_tracked_frame_ms")})
    print("global", {k: d["harness"]["global"][k] for k in ("ms_per_iter", "over_bare_step")})
    print("drop_in", json.dumps(d.get("drop_in_step"), indent=1))
except Exception as e:
    print("bench parse failed", e)
PY
echo "== full-size parity (achieved errors)"; timeout 1500 python -m pytest tests/test_full_size_oracle_gpu.py tests/test_raster_gpu.py -x -q -k "full_size or c1_init or witnessed" 2>&1 | grep -v Warning | tail -5
echo "== pin deviation x60"; timeout 900 python scripts/dev/pin_deviation.py 60 > gpurun_out/r04_pin_deviation.txt 2>&1; tail -4 gpurun_out/r04_pin_deviation.txt

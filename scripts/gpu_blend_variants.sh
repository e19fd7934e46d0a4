#!/bin/bash
# GPU-box visit for the two flavours of the blend kernels: the direct comparison tests, then bench.py at C1 / C2 (/ C4) with
# each flavour forced (FSGS_BLEND_VARIANT, read by fsgs_amd/rasterizer.py) -> gpurun_out/<tag>_*.json + a summary table.
#   gpurun -- 'bash scripts/gpu_blend_variants.sh [tag] [configs...]'
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-variants}; shift || true
cfgs=${*:-C1 C2}
timeout -k 5 900 python -m pytest tests/test_blend_variants_gpu.py -m gpu -x -q --tb=short --durations=8 2>&1 | tail -40 > gpurun_out/${tag}_pytest.log
tail -15 gpurun_out/${tag}_pytest.log
for c in $cfgs; do
  for v in one quad auto; do
    FSGS_BLEND_VARIANT=$v timeout -k 5 300 python bench.py --config $c --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-harness 2>&1 | tail -1 > gpurun_out/${tag}_${c}_${v}.json
  done
done
python - "$tag" $cfgs <<'PY'
import json, sys
t = sys.argv[1]
for c in sys.argv[2:]:
    for v in ("one", "quad", "auto"):
        try:
            d = json.load(open("gpurun_out/%s_%s_%s.json" % (t, c, v)))
            k = d["kernels_ms"]
            trk = d.get("tracking_step") or {}
            print("%s %-4s ms/step %.4f  tracking %.4f  R %d | %s" % (c, v, d["ms_per_step"], trk.get("ms_per_iter", float("nan")), d["config"]["num_rendered"],
                  " ".join("%s=%.1f" % (n, 1e3 * x["avg_ms"]) for n, x in sorted(k.items(), key=lambda kv: -kv[1]["avg_ms"])[:7])))
        except Exception as e:
            print(c, v, "failed:", e, open("gpurun_out/%s_%s_%s.json" % (t, c, v)).read()[-1500:])
PY

#!/bin/bash
# A/B: bin_scatter with XCD-owned sub-lists whose cursor atomics run in the XCD's own L2 (FSGS_BIN_SUB_XCC=2, workgroup scope)
# against the product (sub-list by Gaussian index, agent-scope atomics).   gpurun -- 'bash scripts/dev/ab_bin_l2.sh'
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
V=$PWD/free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.binl2.so
out=gpurun_out/ab_bin_l2.txt; : > $out
FSGS_LIB_PATH=$V python -m pytest tests/test_raster_gpu.py tests/test_render_gpu.py tests/test_fast_step_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -5 | tee -a $out
for rep in 1 2 3; do
  for L in product variant; do
    for cfg in C2 C1 C4; do
      if [ $L = variant ]; then export FSGS_LIB_PATH=$V; else unset FSGS_LIB_PATH; fi
      python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-harness --no-tracking --profile-all 2>/dev/null | tail -1 > /tmp/line.json
      python - $L $cfg $rep <<'PY' | tee -a $out
import json, sys
d = json.load(open("/tmp/line.json")); k = d["kernels_ms"]
print("%-8s %s rep %s: ms/step %.4f  " % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"]) +
      " ".join("%s=%.1f" % (n, 1e3 * k[n]["avg_ms"]) for n in ("sort_depth", "sort_tile", "render_pre_fwd", "blend_fwd") if n in k))
PY
    done
  done
done
unset FSGS_LIB_PATH
for L in product variant; do
  for cfg in C2; do
    if [ $L = variant ]; then export FSGS_LIB_PATH=$V; else unset FSGS_LIB_PATH; fi
    python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-harness --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L $cfg plain bench ms/step %.4f' % d['ms_per_step'])" | tee -a $out
  done
done

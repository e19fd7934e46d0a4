#!/bin/bash
# round 6: the product with the runtime priority step (vs a build with it off), parity of the backward, forward priority A/B
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_blend_variants_gpu.py tests/test_raster_gpu.py tests/test_fast_step_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -3
LIBS="diag/libfsgs_hip.prio0.so libfsgs_hip.so diag/libfsgs_hip.fprio32.so diag/libfsgs_hip.fprio64.so"
{
echo "# product (runtime step = mean list length / 6 when every tile is resident) vs the same build with the priorities off (prio0), and the forward's"
echo "# own priority by remaining entries on top of the product (fprio32 / fprio64); alternating runs of whole libraries on one box"
bash scripts/dev/ab_libs.sh "$LIBS" "C2 C4 C1" 3
for r in 1 2; do for L in $LIBS; do
  for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_fwd %.1f us  blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
  done
done; done
} > gpurun_out/r06_ab_prio_product.txt 2>&1
cat gpurun_out/r06_ab_prio_product.txt

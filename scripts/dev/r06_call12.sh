#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
V=diag/libfsgs_hip.fbf.so
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$V python -m pytest tests/test_blend_variants_gpu.py tests/test_raster_gpu.py tests/test_render_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -3
{
echo "# A/B of whole libraries, alternating runs on one box: product vs -DFSGS_FWD_BRANCHFREE=1 (four-waves forward: every lane runs the whole body, w = 0 where it does not blend; one round of record reads)"
bash scripts/dev/ab_libs.sh "libfsgs_hip.so $V" "C2 C1 C4" 3
for r in 1 2; do for L in libfsgs_hip.so $V; do for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_fwd %.1f us  blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
done; done; done
} > gpurun_out/r06_ab_fwd_branchfree.txt 2>&1
cat gpurun_out/r06_ab_fwd_branchfree.txt

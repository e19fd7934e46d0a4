#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "# product (branch-free forward body) vs its lean form (flean: the three lane-mask combinations replaced by one unsigned compare of test_T's bits and w > 0)"
bash scripts/dev/ab_libs.sh "libfsgs_hip.so diag/libfsgs_hip.flean.so" "C2 C1" 3
for r in 1 2; do for L in libfsgs_hip.so diag/libfsgs_hip.flean.so; do for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_fwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms']))"
done; done; done
} > gpurun_out/r06_ab_fwd_lean.txt 2>&1
cat gpurun_out/r06_ab_fwd_lean.txt
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.flean.so python -m pytest tests/test_blend_variants_gpu.py tests/test_raster_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -3
{
echo "# scripts/soak_raster.py 2000 600 100 under each forced flavour of the blend kernels, round-6 tree (branch-free forward body, backward issue priorities), one MI355X"
for v in quad one; do echo "== FSGS_BLEND_VARIANT=$v"; FSGS_BLEND_VARIANT=$v python scripts/soak_raster.py 2000 600 100 2>&1 | grep -v Warning | tail -12; done
} > gpurun_out/r06_soak.txt 2>&1
tail -30 gpurun_out/r06_soak.txt

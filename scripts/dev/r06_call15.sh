#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_fast_step_gpu.py tests/test_abi_gpu.py tests/test_blend_variants_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -5
{
echo "# (1) the backward's dispatch order refined from the forward's body counts (FSGS_REFINE_ORDER=1, the product) vs left as the sort launch made it (=0); same library"
for r in 1 2 3; do for v in 0 1; do
  FSGS_REFINE_ORDER=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 refine=$v ms/step %.4f tracking %.4f' % (d['ms_per_step'], d['tracking_step']['ms_per_iter']))"
done; done
for r in 1 2; do for v in 0 1; do for sc in default dense; do
  FSGS_REFINE_ORDER=$v python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel refine=$v', 'ms/step %.4f  blend_fwd %.1f us  blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
done; done; done
echo "# (2) product (branch-free forward body) vs two records per trip on top of it (fbfpair); fbranchy = round 5's branchy body"
bash scripts/dev/ab_libs.sh "diag/libfsgs_hip.fbranchy.so libfsgs_hip.so diag/libfsgs_hip.fbfpair.so" "C2" 3
for r in 1 2; do for L in diag/libfsgs_hip.fbranchy.so libfsgs_hip.so diag/libfsgs_hip.fbfpair.so; do for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_fwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms']))"
done; done; done
} > gpurun_out/r06_ab_refine_order_and_fwd_pair.txt 2>&1
cat gpurun_out/r06_ab_refine_order_and_fwd_pair.txt

#!/bin/bash
# round 6: row-streams prototype with its three accumulation variants; A/B of the backward whose first body assigns its slots
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
./scripts/ubench/row_streams_bwd.bin > gpurun_out/r06_row_streams_bwd.txt 2>&1; cat gpurun_out/r06_row_streams_bwd.txt
V=diag/libfsgs_hip.bwdinit.so
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$V python -m pytest tests/test_blend_variants_gpu.py tests/test_raster_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -3
{
echo "# A/B of whole libraries, alternating runs on one box: product vs -DFSGS_BWD_INIT_BODY=1 (the first quadrant body of a Gaussian assigns the reduction slots)"
bash scripts/dev/ab_libs.sh "libfsgs_hip.so $V" "C2 C4" 3
for L in libfsgs_hip.so $V libfsgs_hip.so $V; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 events around every kernel', '$L', 'blend_fwd %.1f us  blend_bwd %.1f us' % (1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
done
} > gpurun_out/r06_ab_bwd_init_body.txt 2>&1
cat gpurun_out/r06_ab_bwd_init_body.txt

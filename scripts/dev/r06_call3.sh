#!/bin/bash
# round 6, third GPU visit: issue ubench (kernel-time accounting), the row-streams prototype, A/B of the two-records-per-trip forward
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
./scripts/ubench/issue_clock.bin > gpurun_out/r06_issue_clock.txt 2>&1; cat gpurun_out/r06_issue_clock.txt
./scripts/ubench/row_streams_bwd.bin > gpurun_out/r06_row_streams_bwd.txt 2>&1; cat gpurun_out/r06_row_streams_bwd.txt
PAIR=diag/libfsgs_hip.fwdpair.so
FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$PAIR python -m pytest tests/test_blend_variants_gpu.py tests/test_raster_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -4
{
echo "# A/B of whole libraries, alternating runs on one box: product (one record per trip of the four-waves forward's set-bit walk) vs -DFSGS_FWD_PAIR=1"
bash scripts/dev/ab_libs.sh "libfsgs_hip.so $PAIR" "C2 C1 C4" 3
for L in libfsgs_hip.so $PAIR libfsgs_hip.so $PAIR; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 events around every kernel', '$L', 'blend_fwd %.1f us  blend_bwd %.1f us' % (1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
done
for L in libfsgs_hip.so $PAIR; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene dense --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 dense', '$L', 'ms/step %.4f blend_fwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms']))"
done
} > gpurun_out/r06_ab_fwd_pair.txt 2>&1
cat gpurun_out/r06_ab_fwd_pair.txt

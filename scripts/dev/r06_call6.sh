#!/bin/bash
# round 6: A/B of issue priority by remaining work in the one-wave backward blend (s_setprio per 64-record batch)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LIBS="libfsgs_hip.so diag/libfsgs_hip.prio48.so diag/libfsgs_hip.prio64.so diag/libfsgs_hip.prio96.so"
{
echo "# A/B of whole libraries, alternating runs on one box: product vs -DFSGS_BWD_PRIO_STEP=48/64/96 (s_setprio(min(3, remaining entries / STEP)) per batch of the one-wave backward)"
bash scripts/dev/ab_libs.sh "$LIBS" "C2 C4" 2
for r in 1 2; do for L in $LIBS; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 events around every kernel', '$L', 'blend_fwd %.1f us  blend_bwd %.1f us' % (1e3*k['blend_fwd']['avg_ms'], 1e3*k['blend_bwd']['avg_ms']))"
done; done
for L in libfsgs_hip.so diag/libfsgs_hip.prio64.so; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene dense --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 dense', '$L', 'ms/step %.4f blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_bwd']['avg_ms']))"
done
} > gpurun_out/r06_ab_bwd_prio.txt 2>&1
cat gpurun_out/r06_ab_bwd_prio.txt

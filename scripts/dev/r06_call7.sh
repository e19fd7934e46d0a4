#!/bin/bash
# round 6: finer sweep of the backward's issue-priority step
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LIBS="libfsgs_hip.so diag/libfsgs_hip.prio16.so diag/libfsgs_hip.prio24.so diag/libfsgs_hip.prio32.so diag/libfsgs_hip.prio48.so"
{
echo "# second sweep: -DFSGS_BWD_PRIO_STEP=16/24/32/48, alternating runs of whole libraries on one box"
bash scripts/dev/ab_libs.sh "$LIBS" "C2 C4 X3" 3
for r in 1 2; do for L in $LIBS; do
  for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_bwd']['avg_ms']))"
  done
done; done
} > gpurun_out/r06_ab_bwd_prio2.txt 2>&1
cat gpurun_out/r06_ab_bwd_prio2.txt

#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LIBS="libfsgs_hip.so diag/libfsgs_hip.pfine.so diag/libfsgs_hip.pfine24.so diag/libfsgs_hip.pgeo96.so diag/libfsgs_hip.pgeo160.so"
{
echo "# product (priority level = remaining / (mean list length / 6), set once per 64 records) vs: pfine = the same set every 16 records; pfine24 = step 24, every 16;"
echo "# pgeo96 / pgeo160 = thresholds step/4, step/2, step with step 96 / 160, every 16 records.  Alternating whole-library runs on one box."
bash scripts/dev/ab_libs.sh "$LIBS" "C2" 3
for r in 1 2; do for L in $LIBS; do
  for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_bwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_bwd']['avg_ms']))"
  done
done; done
} > gpurun_out/r06_ab_bwd_prio3.txt 2>&1
cat gpurun_out/r06_ab_bwd_prio3.txt

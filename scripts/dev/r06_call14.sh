#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
V=diag/libfsgs_hip.fbfpair.so
{
echo "# product (branch-free forward body) vs the same with two records per trip of the set-bit walk (fbfpair); fbranchy = round 5's branchy body"
bash scripts/dev/ab_libs.sh "diag/libfsgs_hip.fbranchy.so libfsgs_hip.so $V" "C2 C4" 3
for r in 1 2; do for L in diag/libfsgs_hip.fbranchy.so libfsgs_hip.so $V; do for sc in default dense; do
  FSGS_LIB_PATH=$PWD/free-surgs_amd/fsgs_amd/lib/$L python bench.py --scene $sc --steps 100 --warmup 10 --profile-all --no-cpu-baseline --no-extras --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('C2 $sc events around every kernel', '$L', 'ms/step %.4f  blend_fwd %.1f us' % (d['ms_per_step'], 1e3*k['blend_fwd']['avg_ms']))"
done; done; done
} > gpurun_out/r06_ab_fwd_branchfree_pair.txt 2>&1
cat gpurun_out/r06_ab_fwd_branchfree_pair.txt

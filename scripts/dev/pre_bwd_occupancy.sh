#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
FSGS_DIAG=1 python free-surgs_amd/build.py >/dev/null || exit 1
L=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so
for lds in 0 8192 20000 36000 70000; do
  FSGS_DBG_LDS_PRE_BWD=$lds FSGS_LIB_PATH=$L python bench.py --steps 150 --warmup 20 --no-extras --no-cpu-baseline --no-tracking --profile-all 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extra dynamic LDS $lds B: render_pre_bwd %.1f us, step %.4f ms' % (1e3*d['kernels_ms']['render_pre_bwd']['avg_ms'], d['ms_per_step']))"
done

#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
DIAG=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so
FSGS_LIB_PATH=$DIAG timeout 300 python scripts/dev/diag_tile_times.py > gpurun_out/r06_tile_times_bwd_prio.txt 2>&1
grep -v "^{" gpurun_out/r06_tile_times_bwd_prio.txt | head -24
{
echo "# C1 / X1 / X2 (four-waves backward): product (priority per 64-record sub-batch of the quad backward) vs priorities off"
bash scripts/dev/ab_libs.sh "diag/libfsgs_hip.prio0.so libfsgs_hip.so" "C1 X1 X2" 4
} > gpurun_out/r06_ab_prio_quad.txt 2>&1
cat gpurun_out/r06_ab_prio_quad.txt

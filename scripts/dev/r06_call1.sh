#!/bin/bash
# round 6, first GPU visit: parity subset + bench on the XCD-owned binning sub-lists, the two new micro-benchmarks, PMC traffic,
# the row-stream balance counters and the shader clock inside the blend kernels (diagnostics flavour).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash scripts/gpu_quick.sh r06a
./scripts/ubench/issue_clock.bin > gpurun_out/r06_issue_clock.txt 2>&1
./scripts/ubench/mfma_valu_overlap.bin > gpurun_out/r06_mfma_valu_overlap.txt 2>&1
cat gpurun_out/r06_issue_clock.txt gpurun_out/r06_mfma_valu_overlap.txt
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1
grep -i "bin_scatter\|sort_tiles\|photometric\|blend" gpurun_out/pmc_WRITE_SIZE_summary.csv gpurun_out/pmc_FETCH_SIZE_summary.csv
DIAG=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so
FSGS_LIB_PATH=$DIAG timeout 600 python scripts/lane_utilisation.py > gpurun_out/r06_lane_utilisation.log 2>&1
tail -c 1500 gpurun_out/r06_lane_utilisation.log | grep -o '"row_streams": {[^}]*}'
FSGS_LIB_PATH=$DIAG timeout 300 python scripts/dev/diag_tile_times.py > gpurun_out/r06_tile_times_bwd.txt 2>&1
FSGS_LIB_PATH=$DIAG timeout 300 python scripts/dev/diag_tile_times.py --fwd > gpurun_out/r06_tile_times_fwd.txt 2>&1
grep "shader clock" gpurun_out/r06_tile_times_bwd.txt gpurun_out/r06_tile_times_fwd.txt

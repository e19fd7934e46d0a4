#!/bin/bash
# round 6, the evidence set in one visit: scripts/gpu_profile_round.sh (bench lines, rocprofv3 stats, PMC, SQ, timelines) + the dense
# scene's own PMC / SQ passes + the shader clock inside the blend kernels (diagnostics flavour) + the issue micro-benchmark.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash scripts/gpu_profile_round.sh > gpurun_out/r06_profile_round.log 2>&1
PMC_ARGS="--scene dense" PMC_SUFFIX=dense bash scripts/gpu_pmc.sh > gpurun_out/pmc_dense.log 2>&1
PMC_ARGS="--scene dense" bash scripts/gpu_sq.sh C2 sq_dense > gpurun_out/sq_dense.log2 2>&1
DIAG=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so
FSGS_LIB_PATH=$DIAG timeout 300 python scripts/dev/diag_tile_times.py > gpurun_out/r06_tile_times_bwd.txt 2>&1
FSGS_LIB_PATH=$DIAG timeout 300 python scripts/dev/diag_tile_times.py --fwd > gpurun_out/r06_tile_times_fwd.txt 2>&1
./scripts/ubench/issue_clock.bin > gpurun_out/r06_issue_clock.txt 2>&1
FSGS_LIB_PATH=$DIAG timeout 600 python scripts/lane_utilisation.py > gpurun_out/r06_lane_utilisation.log 2>&1
tail -3 gpurun_out/bench.json | cut -c1-600
ls gpurun_out | wc -l

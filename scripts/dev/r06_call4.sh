#!/bin/bash
# round 6: the whole GPU suite (parity logs land in gpurun_out/r06_*.jsonl) + the row-streams prototype with its VALU floor
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r06_full_size_parity.jsonl gpurun_out/r06_outlier_statistics.jsonl gpurun_out/r06_harness_c1.jsonl
./scripts/ubench/row_streams_bwd.bin > gpurun_out/r06_row_streams_bwd.txt 2>&1; cat gpurun_out/r06_row_streams_bwd.txt
( time python -m pytest tests -m gpu -x -q --tb=short --durations=15 ) > gpurun_out/r06_pytest_full.log 2>&1; tail -30 gpurun_out/r06_pytest_full.log
wc -l gpurun_out/r06_*.jsonl

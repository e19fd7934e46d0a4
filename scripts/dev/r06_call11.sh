#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
bash scripts/dev/r06_final.sh
rm -f gpurun_out/r06_full_size_parity.jsonl gpurun_out/r06_outlier_statistics.jsonl gpurun_out/r06_harness_c1.jsonl
( time python -m pytest tests -m gpu -x -q --tb=short --durations=10 ) > gpurun_out/r06_pytest_full.log 2>&1; tail -16 gpurun_out/r06_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

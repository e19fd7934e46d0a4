#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof summary.  Run through gpurun from the repo root.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_gpu_full.log 2>&1; tail -8 gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
python bench.py --steps 10 --warmup 3 --profile-all --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_profile_all.json
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.json
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-tracking > gpurun_out/rocprof.log 2>&1
find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/ \;
ls -la gpurun_out

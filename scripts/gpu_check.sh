#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof summary.  Run through gpurun from the repo root.
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
python bench.py --steps 10 --warmup 3 --profile-all 2>&1 | tail -3 | tee gpurun_out/bench_profile_all.log
python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rocprof.log 2>&1
ls -R gpurun_out/prof | head -30

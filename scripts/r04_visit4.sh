#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r04_full_size_parity.jsonl gpurun_out/r04_outlier_statistics.jsonl
echo "== raster tests with IMAGE_TOL"; timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_full_size_oracle_gpu.py tests/test_render_golden_gpu.py -x -q 2>&1 | grep -v Warning | tail -8
echo "== soak (raster sweep, 1000 further seeds)"; timeout 1200 python scripts/soak_raster.py 1000 0 0 2>&1 | grep -v Warning | tail -12

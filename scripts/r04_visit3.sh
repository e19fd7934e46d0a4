#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r04_full_size_parity.jsonl gpurun_out/r04_outlier_statistics.jsonl
echo "== tests (kernel restructure)"; timeout 1200 python -m pytest tests/test_fast_step_gpu.py tests/test_render_gpu.py tests/test_optim_gpu.py tests/test_render_golden_gpu.py tests/test_harness_pin_gpu.py -x -q 2>&1 | grep -v Warning | tail -6
echo "== full-size parity (achieved errors)"; timeout 1500 python -m pytest tests/test_full_size_oracle_gpu.py tests/test_raster_gpu.py -x -q -k "full_size or c1_init or witnessed" 2>&1 | grep -v Warning | tail -3
echo "== pin deviation x200"; timeout 900 python scripts/dev/pin_deviation.py 200 > gpurun_out/r04_pin_deviation.txt 2>&1; tail -2 gpurun_out/r04_pin_deviation.txt

#!/bin/bash
# One visit, two A/Bs against the product library (alternating, HIP events around every kernel):
#   subs4 / subs16 / subs32 : FSGS_BIN_SUBS sub-lists per tile in the binning (product: 8)
#   lossv2                  : FSGS_LOSS_V2 -- photometric kernels with 32-bit plane offsets (scalar base + one offset register per
#                             load), interior strips of the backward without the zero-padding selects, the forward's 80 second-round
#                             row tasks rotated over the waves from tile to tile
#   gpurun -- 'bash scripts/dev/ab_round6b.sh'
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
D=$PWD/free-surgs_amd/fsgs_amd/lib/diag
out=gpurun_out/ab_round6b.txt; : > $out
for t in subs16 subs32; do FSGS_LIB_PATH=$D/libfsgs_hip.$t.so python -m pytest tests/test_raster_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -1 | sed "s/^/$t: /" | tee -a $out; done
FSGS_LIB_PATH=$D/libfsgs_hip.lossv2.so python -m pytest tests/test_loss_gpu.py tests/test_fast_step_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -1 | sed "s/^/lossv2: /" | tee -a $out
for rep in 1 2; do
  for L in product subs4 subs16 subs32 lossv2; do
    for cfg in C2 C4 C1; do
      if [ $L != product ]; then export FSGS_LIB_PATH=$D/libfsgs_hip.$L.so; else unset FSGS_LIB_PATH; fi
      python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-harness --no-tracking --profile-all 2>/dev/null | tail -1 > /tmp/line.json
      python - $L $cfg $rep <<'PY' | tee -a $out
import json, sys
d = json.load(open("/tmp/line.json")); k = d["kernels_ms"]
print("%-8s %s rep %s: ms/step %.4f  " % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"]) +
      " ".join("%s=%.1f" % (n, 1e3 * k[n]["avg_ms"]) for n in ("sort_depth", "sort_tile", "loss_rgb_fwd", "loss_rgb_bwd") if n in k))
PY
    done
  done
done
unset FSGS_LIB_PATH
for L in product subs16 lossv2 product subs16 lossv2; do
  if [ $L != product ]; then export FSGS_LIB_PATH=$D/libfsgs_hip.$L.so; else unset FSGS_LIB_PATH; fi
  python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-harness 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L C2 plain bench ms/step %.4f tracking %.4f' % (d['ms_per_step'], d['tracking_step']['ms_per_iter']))" | tee -a $out
done

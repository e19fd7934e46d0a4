#!/bin/bash
# A/B: render_pre_bwd<ADAM> keeping the first 4 / 6 / 8 staged float4 of a thread's SH-rest rows in registers for its Adam pass
# (FSGS_PRE_BWD_KEEP) instead of reading them from memory a second time.   gpurun -- 'bash scripts/dev/ab_pre_bwd_keep.sh'
set -u  # (needs the FSGS_PRE_BWD_KEEP patch of profiles/r06_ab_pre_bwd_keep.txt re-applied: not in the tree)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
D=$PWD/free-surgs_amd/fsgs_amd/lib/diag
out=gpurun_out/ab_pre_bwd_keep.txt; : > $out
python -m pytest tests/test_optim_gpu.py tests/test_fast_step_gpu.py tests/test_harness_pin_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -1 | sed "s/^/product (unrolled Adam trips): /" | tee -a $out
for t in keep6 keep8; do FSGS_LIB_PATH=$D/libfsgs_hip.$t.so python -m pytest tests/test_optim_gpu.py tests/test_fast_step_gpu.py tests/test_harness_pin_gpu.py -m gpu -x -q --tb=short 2>&1 | tail -1 | sed "s/^/$t: /" | tee -a $out; done
for rep in 1 2 3; do
  for L in product keep4 keep6 keep8; do
    for cfg in C2 C4; do
      if [ $L != product ]; then export FSGS_LIB_PATH=$D/libfsgs_hip.$L.so; else unset FSGS_LIB_PATH; fi
      python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-harness --no-tracking --profile-all 2>/dev/null | tail -1 > /tmp/line.json
      python - $L $cfg $rep <<'PY' | tee -a $out
import json, sys
d = json.load(open("/tmp/line.json")); k = d["kernels_ms"]
print("%-8s %s rep %s: ms/step %.4f  " % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"]) +
      " ".join("%s=%.1f" % (n, 1e3 * k[n]["avg_ms"]) for n in ("render_pre_bwd", "render_pre_fwd", "blend_bwd") if n in k))
PY
    done
  done
done
unset FSGS_LIB_PATH
for L in product keep6 keep8 product keep6 keep8; do
  if [ $L != product ]; then export FSGS_LIB_PATH=$D/libfsgs_hip.$L.so; else unset FSGS_LIB_PATH; fi
  python bench.py --config C2 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-harness --no-tracking 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L C2 plain bench ms/step %.4f' % d['ms_per_step'])" | tee -a $out
done

#!/bin/bash
# same-box A/B of render_pre_bwd<OUT_ADAM>'s alternating phase order (round 4): two diagnostics-flavour libraries, with and
# without the swap, bench.py --profile-all twice each in alternation
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
FSGS_DIAG=1 python free-surgs_amd/build.py >/dev/null && FSGS_DIAG=1 FSGS_CFLAGS=-DFSGS_EXP_NO_PHASE_SWAP FSGS_LIB_TAG=noswap python free-surgs_amd/build.py >/dev/null || exit 1
D=free-surgs_amd/fsgs_amd/lib/diag
for rep in 1 2; do
  for tag in diag noswap; do
    FSGS_LIB_PATH=$D/libfsgs_hip.$tag.so python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --profile-all > gpurun_out/ab_$tag.$rep.json 2>/dev/null
    python - "$tag" "$rep" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
k = d["kernels_ms"]
print("%-7s rep %s: step %.4f ms  render_pre_bwd %.2f us  blend_bwd %.1f us  (%s)" % (sys.argv[1], sys.argv[2], d["ms_per_step"],
      1e3 * k["render_pre_bwd"]["avg_ms"], 1e3 * k["blend_bwd"]["avg_ms"], ", ".join("%s %.0f" % (n, 1e3 * v["avg_ms"]) for n, v in sorted(k.items()))))
PY
  done
done

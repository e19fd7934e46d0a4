#!/bin/bash
# SQ issue/stall counters of the blend kernels (separate --pmc passes, kernel-trace only).
#   gpurun -- 'bash scripts/gpu_sq.sh [config] [tag]'   (FSGS_BLEND_VARIANT is honoured)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
cfg=${1:-C2}; tag=${2:-sq}
: > gpurun_out/${tag}_summary.txt
pass=0
for counters in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
                "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  pass=$((pass + 1))
  rm -rf /tmp/sq && mkdir -p /tmp/sq
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d /tmp/sq -o sq -- python bench.py --config $cfg ${PMC_ARGS:-} --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-extras > gpurun_out/${tag}_pass${pass}.log 2>&1
  f=$(find /tmp/sq -name "*counter_collection.csv" | head -1)
  python - "$f" "gpurun_out/${tag}_summary.txt" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name); name = (m.group(1) if m else name)[:50]
    if not any(k in name for k in ("blend", "render_pre", "photometric", "adam", "bin_scatter", "sort_tiles", "pearson")): continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
out = open(sys.argv[2], "a")
for k, d in agg.items():
    n = max(max(cnt[k].values()), 1)
    line = "%-44s launches %d " % (k, n) + " ".join("%s=%.3g" % (c, v / n) for c, v in sorted(d.items()))
    print(line); out.write(line + "\n")
PY
  tail -2 gpurun_out/${tag}_pass${pass}.log
done

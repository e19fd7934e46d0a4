#!/bin/bash
# SQ issue/stall counters of the blend kernels (one --pmc pass, kernel-trace only).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/sq && mkdir -p /tmp/sq
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/sq -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-extras > gpurun_out/sq.log 2>&1
f=$(find /tmp/sq -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name); name = (m.group(1) if m else name)[:50]
    if not any(k in name for k in ("blend", "render_pre", "photometric", "adam")): continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": cnt[name] += 1
out = open("gpurun_out/sq_summary.txt", "w")
for k, d in agg.items():
    n = max(cnt[k], 1)
    line = "%-34s launches %d " % (k, n) + " ".join("%s=%.3g" % (c, v / n) for c, v in sorted(d.items()))
    print(line); out.write(line + "\n")
PY
tail -3 gpurun_out/sq.log

#!/bin/bash
# HBM traffic counters for the bench kernels (separate --pmc passes; MI355X_MICROARCH.md HBM section).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
sfx=${PMC_SUFFIX:+_$PMC_SUFFIX}   # PMC_ARGS="--scene dense" PMC_SUFFIX=dense: the dense scene's own pass (round 6)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && mkdir -p /tmp/pmc_$c
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py ${PMC_ARGS:-} --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-extras > gpurun_out/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" "$sfx" <<'PY'
import csv, sys, collections
f, c, sfx = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != c: continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    import re
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name)
    name = (m.group(1) if m else name)[:70]
    agg[name][0] += float(r["Counter_Value"]); agg[name][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]
with open("gpurun_out/pmc_%s_summary%s.csv" % (c, sfx), "w") as o:
    o.write("kernel,launches,%s_total,%s_per_launch\n" % (c, c))
    for k, (v, n) in rows:
        o.write("%s,%d,%.1f,%.1f\n" % (k.replace(",", ";"), n, v, v / n))
print(open("gpurun_out/pmc_%s_summary%s.csv" % (c, sfx)).read())
PY
done

#!/bin/bash
# kernel + memory-copy timeline of two-view mapping iterations on a staged sequence (capacity 4): does the copy of the next
# keyframe run beside the kernels?   gpurun -- 'bash scripts/dev/trace_staging.sh'  -> gpurun_out/trace_staging.txt
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/trs && mkdir -p /tmp/trs
STAGING_ONLY=4 timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trs -o tr -- python scripts/dev/staging_steps.py > gpurun_out/trace_staging.log 2>&1
k=$(find /tmp/trs -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/trs -name "*memory_copy_trace.csv" | head -1)
python - "$k" "$m" <<'PY'
import csv, sys, re
K = list(csv.DictReader(open(sys.argv[1]))); M = list(csv.DictReader(open(sys.argv[2])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"], r.get("Queue_Id", "")) for r in K]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "%s %s bytes" % (r.get("Direction", ""), r.get("Size", "")), "copy") for r in M]
ev.sort()
idx = [i for i, e in enumerate(ev) if "adam_compact_kernel" in e[3]]
a, b = idx[-4], idx[-1]
seg = ev[a + 1:b + 1]
t0 = seg[0][0]
out = open("gpurun_out/trace_staging.txt", "w")
queues = {}
for s, e, kind, name, q in seg:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    mm = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name) if kind == "K" else None
    name = (mm.group(1) if mm else name)[-48:]
    qi = queues.setdefault(q, len(queues))
    line = "%8.1f .. %8.1f  dur %7.1f  q%d %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qi, "    " * qi, name)
    print(line); out.write(line + "\n")
PY

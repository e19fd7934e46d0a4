#!/bin/bash
# kernel timeline of ONE tracking iteration (render + masked rgb loss + flow loss + pose Adam)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/trk && mkdir -p /tmp/trk
timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk -o tr -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/trace_trk.log 2>&1
f=$(find /tmp/trk -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pose_adam_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
out = open("gpurun_out/trace_tracking.txt", "w")
def P(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
P("one tracking iteration (pose update to pose update): wall %.1f us, kernels %d, busy %.1f us" % ((t1 - t0) / 1e3, len(seg), busy / 1e3))
prev_end = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name); name = (m.group(1) if m else name)[-64:]
    P("%8.1f gap %6.1f dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    prev_end = max(prev_end, e)
PY

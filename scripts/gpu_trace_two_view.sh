#!/bin/bash
# Kernel timeline of ONE two-view mapping iteration (train.py:236-259) at C2: which kernels of view 1 run beside which
# of view 0 (queue column).   gpurun -- 'bash scripts/gpu_trace_two_view.sh'  -> gpurun_out/trace_two_view.txt
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/tr2 && mkdir -p /tmp/tr2
cat > /tmp/two_view_driver.py <<'PY'
import sys, time
sys.path[:0] = [".", "free-surgs_amd"]
import torch, bench
from fsgs_amd.fast_step import FastStepper
pc, poses, frames, cam, sc = bench.build_problem("C2", torch.device("cuda", 0), 0, 1)
fs = FastStepper(pc, poses, frames)
fs.overlap_views = "--serial" not in sys.argv
for it in range(6):
    fs.mapping_step([it % 8, (it + 3) % 8])
torch.cuda.synchronize()
t = time.perf_counter()
for it in range(40):
    fs.mapping_step([it % 8, (it + 3) % 8])
torch.cuda.synchronize()
print("two-view step: %.4f ms (%s)" % ((time.perf_counter() - t) / 40 * 1e3, "overlapped" if fs.overlap_views else "serial"))
PY
python /tmp/two_view_driver.py --serial | tail -1 > gpurun_out/trace_two_view.txt
python /tmp/two_view_driver.py | tail -1 >> gpurun_out/trace_two_view.txt
timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o tr -- python /tmp/two_view_driver.py > gpurun_out/trace_two_view.log 2>&1
f=$(find /tmp/tr2 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_compact_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
out = open("gpurun_out/trace_two_view.txt", "a")
def P(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
# union of busy intervals (kernels of the two views overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e in iv)
P("one two-view step (between two adam_compact launches): wall %.1f us, %d kernels, GPU busy (union) %.1f us, sum of kernel "
  "durations %.1f us, idle %.1f us" % ((t1 - t0) / 1e3, len(seg), busy / 1e3, total / 1e3, (t1 - t0 - busy) / 1e3))
qcol = "Queue_Id" if "Queue_Id" in seg[0] else ("Stream_Id" if "Stream_Id" in seg[0] else None)
queues = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9a-z, ]+>)?)", name); name = (m.group(1) if m else name)[-64:]
    q = queues.setdefault(r[qcol], len(queues)) if qcol else 0
    P("%8.1f .. %8.1f  dur %7.1f  q%d %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, "    " * q, name))
PY

#!/usr/bin/env python3
"""Instruction stream of one compiled kernel (no GPU needed): `bash scripts/kernel_regs.sh render` leaves the assembly in
/tmp/isa; this prints the function whose mangled name contains every given substring, and per basic block the counts of
VALU / SALU / LDS / VMEM instructions.   python scripts/dump_kernel_isa.py render blend_fwd_quad_kernelILi6 [--full]"""
import re
import sys

src, pats = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
full = "--full" in sys.argv
s = open("/tmp/isa/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % src).read()
for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
    if all(p in m.group(1) for p in pats):
        name, body = m.group(1), m.group(2)
        break
else:
    raise SystemExit("no such kernel")
print(name)
blocks, cur = [], ["entry", []]
for ln in body.split("\n"):
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\S+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], []]
        continue
    cur[1].append(t.split(";")[0].strip())
blocks.append(cur)
for label, ins in blocks:
    c = {"v": 0, "s": 0, "ds": 0, "mem": 0}
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_"): c["v"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["mem"] += 1
        elif op.startswith("s_"): c["s"] += 1
    print("%-12s valu %3d salu %3d lds %2d vmem %2d" % (label, c["v"], c["s"], c["ds"], c["mem"]))
    if full:
        for i in ins:
            print("      " + i)

"""gpurun_out/<tag>_issue_clock.txt (scripts/ubench/issue_clock.bin) + gpurun_out/<tag>_tile_times_{bwd,fwd}.txt
(scripts/dev/diag_tile_times.py on the diagnostics flavour) -> profiles/<tag>_issue_rates.json, read by bench.py:

    python scripts/make_issue_rates_json.py r06

  shader_clock_mhz   the clock the blend kernels' waves ran at, MEASURED inside the kernels (delta s_memtime / delta s_memrealtime
                     per tile, one-wave flavour of the diagnostics library; median over the stamped steps) -- replaces the 2.4 GHz
                     bench.py's valu_frac assumed until round 5 (VERDICT r5 #4 a)
  ubench             ns per wave64 VALU instruction and SIMD at W = 1..8 resident waves: independent FMAs (the ceiling) and the
                     backward blend body's instruction mix -- the denominator of roofline.valu_frac_of_achievable (#4 b)
  waves_per_simd     what the register allocation of each blend kernel allows (scripts/kernel_regs.sh)"""
import json
import os
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = {"_about": __doc__.strip(), "shader_clock_mhz": {}, "shader_clock_samples_mhz": {}, "ubench": {"fma": {}, "blend": {}},
       "waves_per_simd": {}}
for kind in ("bwd", "fwd"):
    f = os.path.join(root, "gpurun_out", "%s_tile_times_%s.txt" % (tag, kind))
    if os.path.exists(f):
        v = [json.loads(l)["shader_clock_mhz"] for l in open(f) if l.startswith('{"kernel"')]
        if v:
            out["shader_clock_samples_mhz"]["blend_" + kind] = v
            out["shader_clock_mhz"]["blend_" + kind] = sorted(v)[len(v) // 2]
pat = re.compile(r"(fma|blend)\s+W=(\d+) waves/SIMD: kernel\s+([\d.]+) ms \(mean wave loop\s+([\d.]+) ms; (.*?)\)\s+shader clock\s+(\d+) MHz\s+"
                 r"([\d.]+) ns/inst/SIMD =\s+([\d.]+) cycles/inst/SIMD")
f = os.path.join(root, "gpurun_out", "%s_issue_clock.txt" % tag)
for l in open(f):
    m = pat.search(l)
    if m:
        out["ubench"][m.group(1)][m.group(2)] = {"ns_per_inst_per_simd": float(m.group(7)), "cycles_per_inst_per_simd": float(m.group(8)),
                                                 "shader_clock_mhz": float(m.group(6)), "all_resident": m.group(5).startswith("all"),
                                                 "kernel_ms": float(m.group(3)), "wave_loop_ms": float(m.group(4))}
r = subprocess.run(["bash", os.path.join(root, "scripts", "kernel_regs.sh")], capture_output=True, text=True).stdout
for l in r.splitlines():
    m = re.match(r"(\S.*?)\s+vgpr\s+(\d+) \(waves/SIMD (\d+)\)", l)
    if m:
        out["waves_per_simd"][m.group(1).strip().replace(", ", "; ")] = {"vgpr": int(m.group(2)), "waves_per_simd": int(m.group(3))}
with open(os.path.join(root, "profiles", "%s_issue_rates.json" % tag), "w") as fo:
    json.dump(out, fo, indent=1)
with open(os.path.join(root, "profiles", "%s_issue_clock_ubench.txt" % tag), "w") as fo:
    fo.write(open(f).read())
print(json.dumps({k: out[k] for k in ("shader_clock_mhz",)}), len(out["ubench"]["blend"]), "ubench rows", len(out["waves_per_simd"]), "kernels")

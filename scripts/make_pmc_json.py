"""gpurun_out/pmc_{FETCH,WRITE}_SIZE_summary.csv (scripts/gpu_pmc.sh) + gpurun_out/sq_summary.txt (scripts/gpu_sq.sh)
-> profiles/<tag>_pmc_*.csv, profiles/<tag>_sq_counters.txt and profiles/<tag>_pmc_traffic.json (read by bench.py).

    python scripts/make_pmc_json.py r02 <R> [<P> <W> <H> [<suffix>]]

<suffix> (round 6): e.g. "dense" -- reads gpurun_out/pmc_*_summary_dense.csv / sq_dense_summary.txt (scripts/gpu_pmc.sh and
gpu_sq.sh run with PMC_ARGS="--scene dense") and writes profiles/<tag>_pmc_traffic_dense.json for the bench's dense-scene block.

Correction of FETCH_SIZE (MI355X_MICROARCH.md, HBM section, and the calibration in profiles/r02_fetch_size_calibration.txt,
scripts/ubench/gather_fetch.hip): wide COALESCED streaming reads are tallied at half their bytes (x2); a gather of whole
64-byte records -- the blend kernels' per-pair record fetch since round 2 -- is tallied at face value (x1: 275.0 MB counted
for 268.4 MB of records + 16.8 MB of coalesced indices at half).  So for the blend kernels
    traffic = FETCH_SIZE + (coalesced bytes the kernel is known to stream) / 2 + WRITE_SIZE
and for every streaming kernel  traffic = 2 FETCH_SIZE + WRITE_SIZE."""
import csv
import json
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280
H = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
sfx = ("_" + sys.argv[6]) if len(sys.argv) > 6 else ""
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy("gpurun_out/pmc_%s_summary%s.csv" % (c, sfx), "profiles/%s_pmc_%s_summary%s.csv" % (tag, c, sfx))


def load(f, col):
    return {r["kernel"]: float(r[col]) for r in csv.DictReader(open(f))}


fe = load("profiles/%s_pmc_FETCH_SIZE_summary%s.csv" % (tag, sfx), "FETCH_SIZE_per_launch")
wr = load("profiles/%s_pmc_WRITE_SIZE_summary%s.csv" % (tag, sfx), "WRITE_SIZE_per_launch")
sq = {}
try:
    sq_src = "gpurun_out/sq%s_summary.txt" % sfx
    shutil.copy(sq_src, "profiles/%s_sq_counters%s.txt" % (tag, sfx))
    for line in open(sq_src):
        m = re.match(r"(\S.*?)\s+launches \d+ (.*)", line)
        if m:
            # (one line per kernel and --pmc pass: merge the passes)
            sq.setdefault(m.group(1).strip(), {}).update({k: float(v) for k, v in (kv.split("=") for kv in m.group(2).split())})
except FileNotFoundError:
    pass
HW = W * H
# bytes each blend kernel streams COALESCED (tallied at half by FETCH_SIZE): the tile list, the per-pixel planes
coalesced = {
    "blend_fwd": 4 * R,                              # sorted Gaussian ids
    "blend_bwd": 4 * R + HW * (4 * 4 + 8),           # ids + dL/dpixel (4 planes) + final_T + n_contrib
}
out = {"_about": __doc__.strip().split("\n\n")[-1] + "  Workload: bench.py config C2 (%dx%d, %d Gaussians, R=%d, fused "
       "6-channel render).  Counter unit = KiB." % (W, H, P, R),
       "config": "C2", "scene": sfx[1:] or "default", "num_rendered": R,
       "source": "profiles/%s_pmc_*_summary%s.csv, profiles/%s_sq_counters%s.txt" % (tag, sfx, tag, sfx),
       "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    if k.startswith(("at::", "rocprim", "__amd")):
        continue
    f, w = fe.get(k, 0.0) * 1024, wr.get(k, 0.0) * 1024
    fam = "blend_fwd" if k.startswith("blend_fwd") else "blend_bwd" if k.startswith("blend_bwd") else None
    if fam:
        ent = {"fetch_raw_bytes": f, "write_bytes": w, "coalesced_bytes_streamed": coalesced[fam],
               "traffic_bytes": f + 0.5 * coalesced[fam] + w, "correction": "gather x1 + half of the coalesced bytes"}
    else:
        ent = {"fetch_raw_bytes": f, "write_bytes": w, "traffic_bytes": 2 * f + w, "correction": "streaming x2"}
    for name, c in sq.items():
        if name.replace(",", ";").replace("; ", "; ").startswith(k[:40]) or k.startswith(name.replace(", ", "; ")[:40]):
            iv = c.get("SQ_INSTS_VALU")
            if iv:
                ent["valu_wave_insts"] = iv
                ent["salu_wave_insts"] = c.get("SQ_INSTS_SALU")
                ent["active_valu_cycles_per_inst"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / iv
                ent["sq_wave_cycles"] = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0)
                ent["sq_wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)
                # round 6 (VERDICT r5 #4 d): the counters as collected (SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles per
                # the guide; SQ_BUSY_CYCLES summed over the shader engines) -- per launch
                ent["sq"] = {n: c[n] for n in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                               "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
                                               "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA") if n in c}
    out["kernels"][k] = ent
json.dump(out, open("profiles/%s_pmc_traffic%s.json" % (tag, sfx), "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-40s fetch_raw %7.1f MB  write %7.1f MB  traffic %7.1f MB  %s" % (
        k[:40], v["fetch_raw_bytes"] / 1e6, v["write_bytes"] / 1e6, v["traffic_bytes"] / 1e6,
        ("VALU %.1f M" % (v["valu_wave_insts"] / 1e6)) if "valu_wave_insts" in v else ""))

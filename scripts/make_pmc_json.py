"""gpurun_out/pmc_{FETCH,WRITE}_SIZE_summary.csv (from scripts/gpu_pmc.sh) -> profiles/<tag>_pmc_*.csv + profiles/r01_pmc_traffic.json"""
import csv, json, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = sys.argv[2] if len(sys.argv) > 2 else "?"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy("gpurun_out/pmc_%s_summary.csv" % c, "profiles/%s_pmc_%s_summary.csv" % (tag, c))


def load(f, col):
    return {r["kernel"]: float(r[col]) for r in csv.DictReader(open(f))}


fe = load("profiles/%s_pmc_FETCH_SIZE_summary.csv" % tag, "FETCH_SIZE_per_launch")
wr = load("profiles/%s_pmc_WRITE_SIZE_summary.csv" % tag, "WRITE_SIZE_per_launch")
out = {"_about": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/gpu_pmc.sh) "
       "on bench.py config C2 (1280x1024, 300k Gaussians, R=%s, fused 6-channel). Counter unit = KiB. Correction per "
       "MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly 1/2 of coalesced streaming reads on gfx950 -> x2; calibrated "
       "on adam_kernel (known 283.2 MB read / 212.4 MB written per launch) and pearson_bwd (10.49 MB read). For "
       "gather-dominated kernels (blend_*) the x2 is an upper bound." % R,
       "config": "C2", "source": "profiles/%s_pmc_*_summary.csv" % tag, "kernels": {}}
for k in sorted(set(fe) | set(wr)):
    if k.startswith(("at::", "rocprim", "__amd")):
        continue
    f, w = fe.get(k, 0.0) * 1024, wr.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_raw_bytes": f, "write_bytes": w, "traffic_bytes": 2 * f + w}
json.dump(out, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-28s fetch_raw %7.1f MB  write %7.1f MB  traffic(2f+w) %7.1f MB" % (
        k, v["fetch_raw_bytes"] / 1e6, v["write_bytes"] / 1e6, v["traffic_bytes"] / 1e6))

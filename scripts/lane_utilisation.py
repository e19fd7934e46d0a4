"""Lane utilisation of the blend kernels: of the 64 lanes of every executed 8x8 quadrant body, how many
blend (forward) / carry a non-zero alpha (backward), and what a body over 64 lanes chosen by 4x4-pixel blocks could save.
Needs the diagnostics flavour of the library (the counters are compiled out of the product build):

    gpurun -- 'FSGS_DIAG=1 python free-surgs_amd/build.py && \
               FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so python scripts/lane_utilisation.py'

Prints one JSON object per workload (C2 default scene, C2 dense scene, C4) -> profiles/r05_lane_utilisation.jsonl.

Round 5: the counters live in the ONE-WAVE flavour of the blend kernels (the diagnostics library instantiates
blend_*_kernel<..., DIAG = true>: the product's instruction stream plus counters behind `if constexpr`), so the script forces
FSGS_BLEND_VARIANT=one; and before it counts anything it times that flavour against the product library on the same workload
(bench.py, both forced to one wave per tile, counters off): the diagnostics kernels must be within 10 % of the product's
(VERDICT r4 #5), and every record states the flavour and both step times."""
import json
import os
import subprocess
import sys

os.environ["FSGS_BLEND_VARIANT"] = "one"

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402


def report(name, c):
    c = [int(v) for v in c]
    pairs, bodies, lanes = c[0], c[1], c[2]
    hist = c[3:12]
    out = {"kernel": name, "pairs": pairs, "bodies": bodies, "bodies_per_pair": bodies / max(pairs, 1),
           "contributing_lanes": lanes, "lane_utilisation": lanes / max(64 * bodies, 1),
           "bodies_by_contributing_lanes": dict(zip(["0", "1-8", "9-16", "17-24", "25-32", "33-40", "41-48", "49-56", "57-64"],
                                                    [h / max(bodies, 1) for h in hist])),
           # a body over 64 lanes chosen by 4x4 blocks: bodies a pair would need = max over the four lane sets of the
           # number of quadrants whose block is alive
           "packed_bodies_over_bodies": {"blocks_that_contributed (ceiling)": c[12] / max(bodies, 1),
                                         "blocks_by_footprint_test (decidable up front)": c[14] / max(bodies, 1)},
           "alive_4x4_blocks_per_body": {"contributed": c[13] / max(bodies, 1), "footprint_test": c[15] / max(bodies, 1)},
           "pairs_with_a_gain_footprint_test": c[16] / max(pairs, 1)}
    if len(c) > 18 and c[17]:
        # round 6: four 16-lane row streams per quadrant wave (one 4x4 block each, its own list): wave steps a 256-record
        # batch would take (max over the four rows) against the 8x8 bodies executed today, and how full the rows would be
        out["row_streams"] = {"wave_steps": c[17], "wave_steps_over_bodies": c[17] / max(bodies, 1),
                              "wave_steps_per_pair": c[17] / max(pairs, 1), "row_bodies": c[18],
                              "row_balance": c[18] / max(4 * c[17], 1),
                              "lane_utilisation_of_a_row_step": lanes / max(64 * c[17], 1)}
    return out


def flavour_times():
    """blend kernel times and step time of the product library and of the diagnostics library (counters off), one wave per
    tile forced on both, via bench.py in subprocesses"""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = {}
    for name, lib in (("product", os.path.join(root, "free-surgs_amd", "fsgs_amd", "lib", "libfsgs_hip.so")),
                      ("diag", os.environ.get("FSGS_LIB_PATH", ""))):
        env = dict(os.environ, FSGS_LIB_PATH=lib, FSGS_BLEND_VARIANT="one")
        for k in ("FSGS_DBG_LANES_FWD", "FSGS_DBG_LANES"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline",
                            "--no-extras", "--no-tracking"], env=env, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out[name] = {"step_ms": d["ms_per_step"], "blend_fwd_us": 1e3 * d["kernels_ms"]["blend_fwd"]["avg_ms"],
                     "blend_bwd_us": 1e3 * d["kernels_ms"]["blend_bwd"]["avg_ms"]}
    for k in ("blend_fwd_us", "blend_bwd_us"):
        ratio = out["diag"][k] / out["product"][k]
        out[k + "_diag_over_product"] = ratio
        assert ratio < 1.10, "the diagnostics flavour's %s is %.0f %% above the product's" % (k, 100 * (ratio - 1))
    return out


def main():
    flavour = flavour_times()
    print(json.dumps({"flavour_check": flavour}), flush=True)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bufs = {"fwd": torch.zeros((32,), dtype=torch.int64, device=dev), "bwd": torch.zeros((32,), dtype=torch.int64, device=dev)}
    os.environ["FSGS_DBG_LANES_FWD"] = str(bufs["fwd"].data_ptr())  # read once per process, at the first launch
    os.environ["FSGS_DBG_LANES"] = str(bufs["bwd"].data_ptr())
    from fsgs_amd import _lib
    from fsgs_amd.fast_step import FastStepper

    _lib.load()
    if "diag" not in _lib.LIB_PATH:
        sys.exit("needs the diagnostics flavour: FSGS_LIB_PATH=free-surgs_amd/fsgs_amd/lib/diag/libfsgs_hip.diag.so")
    os.makedirs("gpurun_out", exist_ok=True)
    lines = []
    for cfg, scene in (("C2", "default"), ("C2", "dense"), ("C4", "default")):
        torch.manual_seed(0)
        np.random.seed(0)
        pc, poses, frames, cam, sc = bench.build_problem(cfg, dev, 0, 1, scene=scene)
        st = FastStepper(pc, poses, frames)
        for it in range(8):
            st.mapping_step([it % len(frames.colors)])
        torch.cuda.synchronize()
        for b in bufs.values():
            b.zero_()
        st.pairs_total = st.forward_calls = 0
        n = 8
        for it in range(n):  # the eight cameras of the bench
            st.mapping_step([it % len(frames.colors)])
        torch.cuda.synchronize()
        R = st.pairs_total / max(st.forward_calls, 1)
        for k in ("fwd", "bwd"):
            rep = report("blend_" + k, bufs[k].cpu().numpy())
            if rep["bodies"] == 0:
                sys.exit("no counts: the loaded library has no diagnostics hooks")
            rep.update({"config": cfg, "scene": scene, "steps": n, "num_rendered": R, "kernel_flavour": "one wave per tile, DIAG = true",
                        "flavour_step_ms": flavour["diag"]["step_ms"], "product_one_wave_step_ms": flavour["product"]["step_ms"],
                        "flavour_blend_us_over_product": {"fwd": flavour["blend_fwd_us_diag_over_product"],
                                                          "bwd": flavour["blend_bwd_us_diag_over_product"]},
                        "pairs_walked_over_num_rendered": rep["pairs"] / n / R})
            lines.append(rep)
            print(json.dumps(rep), flush=True)
        del st, pc, poses, frames
        torch.cuda.empty_cache()
    with open("gpurun_out/lane_utilisation.jsonl", "w") as f:
        for l in lines:
            f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()

"""Time GaussianModel.densify_and_prune: the reference's torch sequence vs csrc/densify.hip, on a bench scene."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "free-surgs_amd"))
import numpy as np, torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
for mode in ("torch", "device"):
    ts = []
    for rep in range(3):
        torch.manual_seed(0)
        pc, poses, frames, cam, sc = bench.build_problem(cfg, "cuda", 0, 1)
        P = pc.num_points
        for name in pc.params:  # give Adam a state to carry along
            pc.params[name].grad = torch.zeros_like(pc.params[name])
        pc.optimizer.step()
        g = torch.Generator(device="cuda").manual_seed(1)
        pc.variables["xyz_gradient_accum"] = torch.rand((P, 1), device="cuda", generator=g) * 6e-4
        pc.variables["denom"] = torch.ones((P, 1), device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "torch":
            pc.densify_and_prune(2e-4, 0.05, 20)
        else:
            pc.densify_and_prune_device(2e-4, 0.05, 20)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        n = pc.num_points
        del pc
    print("%s %-6s P %d -> %d : %.2f ms (min of %s)" % (cfg, mode, P, n, min(ts), ["%.2f" % t for t in ts]))

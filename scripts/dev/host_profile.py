"""cProfile of the host side of FastStepper.mapping_step / tracking_step (where the ~110 us of issue work per step go).
   gpurun -- 'python scripts/dev/host_profile.py [C1|C2]'"""
import cProfile, pstats, sys, io
sys.path.insert(0, "free-surgs_amd"); sys.path.insert(0, ".")
import torch
import bench
from fsgs_amd.fast_step import FastStepper

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
pc, poses, frames, cam, sc = bench.build_problem(cfg, "cuda", 0, 1)
fs = FastStepper(pc, poses, frames)
for it in range(30): fs.mapping_step([it % 8])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(300): fs.mapping_step([it % 8])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print(s.getvalue()[:9000])

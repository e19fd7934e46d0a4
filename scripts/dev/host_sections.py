"""Host time of every C-ABI call and of the Python around them in FastStepper.mapping_step (one view, Adam fused), by wrapping
the library's functions:   gpurun -- 'python scripts/dev/host_sections.py [C1|C2]'"""
import sys, time, collections
sys.path.insert(0, "free-surgs_amd"); sys.path.insert(0, ".")
import torch
import bench
from fsgs_amd import _lib
from fsgs_amd.fast_step import FastStepper

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
pc, poses, frames, cam, sc = bench.build_problem(cfg, "cuda", 0, 1)
fs = FastStepper(pc, poses, frames)
acc = collections.defaultdict(float); cnt = collections.Counter()
class Wrap:
    def __init__(self, lib): self._lib = lib; self._cache = {}
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not callable(f): return f
        w = self._cache.get(name)
        if w is None:
            def w(*a, _f=f, _n=name):
                t = time.perf_counter(); r = _f(*a); acc[_n] += time.perf_counter() - t; cnt[_n] += 1; return r
            self._cache[name] = w
        return w
for it in range(30): fs.mapping_step([it % 8])
torch.cuda.synchronize()
fs.lib = Wrap(fs.lib)
N = 300
t0 = time.perf_counter()
for it in range(N): fs.mapping_step([it % 8])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("host issue per step %.1f us" % (t_issue / N * 1e6))
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-44s %6.1f us/step  (%d calls/step)" % (k, v / N * 1e6, cnt[k] // N)); tot += v
print("  inside C-ABI calls %.1f us, Python around them %.1f us" % (tot / N * 1e6, (t_issue - tot) / N * 1e6))

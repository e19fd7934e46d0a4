"""Where the host spends a mapping step at C2: time blocked inside the forward call (which returns once the pinned mailbox
has the pair count, i.e. when the GPU reaches the forward blend) vs. time issuing the rest of the step."""
import sys, time, torch
sys.path.insert(0, "free-surgs_amd"); sys.path.insert(0, ".")
import bench
from fsgs_amd.fast_step import FastStepper

pc, poses, frames, cam, sc = bench.build_problem(sys.argv[1] if len(sys.argv) > 1 else "C2", "cuda", 0, 1)
fs = FastStepper(pc, poses, frames)
orig = fs._render_forward
acc = {"fwd": 0.0, "n": 0}
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); acc["fwd"] += time.perf_counter() - t; acc["n"] += 1; return r
fs._render_forward = timed
for it in range(20): fs.mapping_step([it % 8])
torch.cuda.synchronize(); acc["fwd"] = 0.0; acc["n"] = 0
t0 = time.perf_counter()
N = 200
for it in range(N): fs.mapping_step([it % 8])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: wall %.1f us, host issue %.1f us, of which inside the forward call (enqueue + blocked on the mailbox) %.1f us"
      % (t_all / N * 1e6, t_issue / N * 1e6, acc["fwd"] / N * 1e6))
print("=> host work outside the forward call: %.1f us per step" % ((t_issue - acc["fwd"]) / N * 1e6))

"""Wall time per frame of Runner.progressive_run (50 tracking + 30 mapping iterations, train.py:322-345) at C2
resolution on a synthetic sequence, against the sum of the step times bench.py reports."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd.sequence import learner_from_first_frame, make_sequence
from fsgs_amd.trainer import PoseTrack, Runner

W, H, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 1024, 6)
P = int(0.1 * W * H)
torch.manual_seed(0)
frames, cam = make_sequence(W, H, n, P)
pc = learner_from_first_frame(frames, cam)
poses = PoseTrack(n, "cuda")
run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=False)
marks = []
orig_tracking, orig_mapping = run.tracking, run.mapping
def timed(fn, tag):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks.append((tag, (time.perf_counter() - t0) * 1e3)); return r
    return w
run.tracking = timed(orig_tracking, "tracking x50")
run.mapping = timed(orig_mapping, "mapping")
torch.cuda.synchronize(); t0 = time.perf_counter()
run.progressive_run()
torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
print("P", pc.num_points, "frames", n, "total %.1f ms" % total)
for tag, ms in marks: print("  %-14s %8.2f ms" % (tag, ms))
tr = [m for t, m in marks if t.startswith("tracking")]; mp_ = [m for t, m in marks if t == "mapping"][1:]
print("tracking/frame %.2f ms (%.3f ms/iter incl. per-frame setup), mapping/frame %.2f ms (%.3f ms/iter, 2 views), PSNR %.2f" % (
    np.mean(tr), np.mean(tr) / 50, np.mean(mp_), np.mean(mp_) / 30, run.validation() if len(run.frames.i_test) else float("nan")))
print("pose metrics", run.eval_pose())
if len(sys.argv) > 4:  # soak: global_run across densification boundaries
    n_glob = int(sys.argv[4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P0 = run.pc.num_points
    run.global_run(n_glob)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("global_run %d iterations: %.1f ms (%.3f ms/iter), P %d -> %d, iteration counter %d, PSNR %.2f" % (
        n_glob, dt * 1e3, dt * 1e3 / n_glob, P0, run.pc.num_points, run.iteration, run.validation()))
    assert all(torch.isfinite(v).all() for v in run.pc.params.values())

"""wall time of the progressive phase per frame at BASELINE C2's image size, resident vs staged (capacity 4): the copies of
a frame (31 MB: 3 + 1 + 2 planes of 1280 x 1024 floats) must hide behind the ~60 ms of steps the frame takes"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd.sequence import learner_from_first_frame, make_sequence
from fsgs_amd.staging import StagedFrames
from fsgs_amd.trainer import PoseTrack, Runner

W, H, n = 1280, 1024, 10
for staged in (False, True, False, True):
    torch.manual_seed(0)
    frames, cam = make_sequence(W, H, n, P=200000, seed=1)
    res = frames
    if staged:
        frames = StagedFrames([c.cpu() for c in res.colors], [m.cpu() for m in res.monodeps], flows_fw=[f.cpu() for f in res.flows_fw],
                              K=res.K, gt_w2c=res.gt_w2c, device="cuda", capacity=4)
    pc = learner_from_first_frame(res, cam, ratio=0.1)
    run = Runner(pc, PoseTrack(n, "cuda"), frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.progressive_run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-8s %d frames %dx%d P %d: %.1f ms (%.2f ms per frame after the first)  rpe/ate %s  %s" % (
        "staged" if staged else "resident", n, W, H, pc.num_points, dt * 1e3, dt * 1e3 / (n - 1), np.round(run.eval_pose(), 5),
        frames.stats() if staged else ""))

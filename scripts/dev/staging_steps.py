"""two-view mapping iterations with a random keyframe each (train.py:239) at C2's image size: resident frames vs staged ones
with 4 / 16 device buffers per lane (10 frames: 16 = no copies after the first pass)"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd.sequence import learner_from_first_frame, make_sequence
from fsgs_amd.staging import StagedFrames
from fsgs_amd.trainer import PoseTrack, Runner

W, H, n = 1280, 1024, 10
torch.manual_seed(0)
res, cam = make_sequence(W, H, n, P=200000, seed=1)
caps = (None, 4, 16, None, 4, 16) if not os.environ.get("STAGING_ONLY") else (int(os.environ["STAGING_ONLY"]),)
for cap in caps:
    frames = res if cap is None else StagedFrames([c.cpu() for c in res.colors], [m.cpu() for m in res.monodeps],
                                                  flows_fw=[f.cpu() for f in res.flows_fw], K=res.K, gt_w2c=res.gt_w2c, device="cuda", capacity=cap)
    pc = learner_from_first_frame(res, cam, ratio=0.1)
    poses = PoseTrack(n, "cuda")
    for i in range(n):
        poses.set_pose(i, [1, 0, 0, 0], [0, 0, 0])
    run = Runner(pc, poses, frames, densify=False)
    run.keyframes = list(range(n - 1))
    pc.initialize_optimizer()
    run.mapping(n - 1, 30, progressive=True)
    torch.cuda.synchronize()
    spent = [0.0, 0]
    if cap is not None:
        inner = frames.prefetch
        def timed(t, flows=True):
            a = time.perf_counter(); inner(t, flows=flows); spent[0] += time.perf_counter() - a; spent[1] += 1
        frames.prefetch = timed
    t0 = time.perf_counter()
    run.mapping(n - 1, 300, progressive=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %.3f ms per two-view iteration %s" % ("resident" if cap is None else "staged/%d" % cap, dt / 300 * 1e3,
                                                      "" if cap is None else (frames.stats()["colors"], "prefetch calls: %.1f us of CPU each" % (spent[0] / max(spent[1], 1) * 1e6))))

"""run-to-run spread of the short synthetic harness run (resident twice, staged once): max |t| difference"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "free-surgs_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from fsgs_amd.sequence import learner_from_first_frame, make_sequence
from fsgs_amd.trainer import PoseTrack, Runner
from fsgs_amd.staging import StagedFrames

def go(staged):
    torch.manual_seed(0)
    W, H, n = 320, 256, 7
    frames, cam = make_sequence(W, H, n, P=40000, seed=1)
    res = frames
    if staged:
        frames = StagedFrames([c.cpu() for c in res.colors], [m.cpu() for m in res.monodeps], flows_fw=[f.cpu() for f in res.flows_fw], K=res.K, gt_w2c=res.gt_w2c, device="cuda", capacity=4)
    pc = learner_from_first_frame(res, cam, ratio=0.25)
    poses = PoseTrack(n, "cuda")
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=100, row0_depth_quirk=False)
    run.progressive_run()
    t = poses.t.detach().clone()
    run.global_run(20, eval_every=10)
    torch.cuda.synchronize()
    return t, np.array(run.eval_pose()), pc.num_points, [round(m["psnr"], 2) for _, m in run.eval_log]
runs = [go(False), go(False), go(False), go(True), go(True), go(True)]
a = runs[0]
for k, r in enumerate(runs):
    print("resident" if k < 3 else "staged  ", "max |t - t_run0| %.2e" % (a[0] - r[0]).abs().max().item(), "rpe/ate", r[1], "P", r[2], "psnr", r[3])

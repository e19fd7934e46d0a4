import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from fsgs_amd.sequence import learner_from_first_frame, make_sequence
from fsgs_amd.trainer import PoseTrack, Runner
torch.manual_seed(0)
W, H, n = 320, 256, 7
frames, cam = make_sequence(W, H, n, P=40000, seed=1)
pc = learner_from_first_frame(frames, cam, ratio=0.25)
poses = PoseTrack(n, "cuda")
run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=(len(sys.argv) < 2))
run.progressive_run()
gt = np.stack(frames.gt_w2c)
with torch.no_grad():
    pred = np.stack([poses.get_pose(i).cpu().numpy() for i in range(n)])
np.set_printoptions(precision=5, suppress=True)
print("gt t:\n", gt[:, :3, 3]); print("pred t:\n", pred[:, :3, 3])
print("gt R01 R02 R12:", gt[:, 0, 1], gt[:, 0, 2], gt[:, 1, 2]); print("pred:", pred[:, 0, 1], pred[:, 0, 2], pred[:, 1, 2])
for l in run.log: print(l)
print("metrics", run.eval_pose(), "psnr", run.validation())
d = frames.gt_depths[0]; print("gt depth range", d.min().item(), d.max().item(), "mono", frames.monodeps[0].min().item(), frames.monodeps[0].max().item())

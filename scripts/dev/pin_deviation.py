import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "free-surgs_amd")]
import numpy as np, torch
from tests import ref_harness
import tests.test_harness_pin_gpu as T
fx = dict(np.load(T.FX))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    run = T._run_gpu(fx)
    maps = [e for e in run.trace if e[0] == "map"]; tracks = [e for e in run.trace if e[0] == "track"]
    got_map = np.array([e[3] for e in maps]); n_pre = int((fx["map_iter"] < fx["densify"][0, 0]).sum())
    rel = np.abs(got_map - fx["map_loss"]) / np.abs(fx["map_loss"])
    got_trk = np.array([[e[3], e[4], e[5]] for e in tracks])
    relt = np.abs(got_trk - fx["track_loss"]) / (np.abs(fx["track_loss"]) + 1e-30)
    sgn = np.sign(got_map - fx["map_loss"])[n_pre:]
    print("P %d  map pre %.2e post max %.2e (signs %s)  track max %.2e  pose_t err %.2e" % (
        run.pc.num_points, rel[:n_pre].max(), rel[n_pre:].max(), "".join("+" if s_ > 0 else "-" for s_ in sgn), relt.max(),
        np.abs(run.poses.t.detach().cpu().numpy() - fx["pose_t"]).max()), flush=True)

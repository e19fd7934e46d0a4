import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "free-surgs_amd")]
import numpy as np, torch, json
from tests import ref_harness
from oracle.fsgs_oracle import Oracle, usable_cores
from fsgs_amd.render import render
from fsgs_amd.trainer import Runner
o = Oracle(np.float32); o.set_threads(usable_cores())
C1 = ref_harness.C1
inputs = ref_harness.make_c1_inputs(o)
print("stats", ref_harness.c1_input_stats(inputs)[:12].tolist())
for rep in range(3):
    pc, poses, frames = ref_harness.load_inputs(inputs, "cuda")
    pc.training_setup()
    run = Runner(pc, poses, frames, tracking_iter=C1["tracking_iter"], mapping_iter=C1["mapping_iter"], first_mapping_iter=C1["first_mapping_iter"],
                 densify_interval=C1["densify_interval"], seed=C1["seed"], trace=True)
    torch.manual_seed(0)
    with ref_harness.deterministic_rng(C1["rng_seed"]):
        run.progressive_run()
    torch.cuda.synchronize()
    def render_fn(t):
        with torch.no_grad():
            return render(run.poses, t, run.pc, gs_grad=False, cam_grad=False)["render"]
    got = ref_harness.c1_outcome(run.trace, run.pc, run.poses, frames, render_fn)
    print(json.dumps({k: np.asarray(v).tolist() for k, v in got.items() if k in ("densify", "final_P", "pose_metrics", "psnr_test", "track_last", "map_mean")}))

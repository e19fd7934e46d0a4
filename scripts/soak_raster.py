"""Soak of the randomised parity sweeps beyond the seeds in tests/ (same test bodies, same attribution rule: every element
beyond 1e-4 of its tensor's inf-norm must be covered by a near-tie the oracle itself witnesses):
rasteriser vs oracle (600 more scenes), fused render vs the two-pass statement + oracle (240 more), step driver vs the
autograd route at random image sizes (60 more).   gpurun -- 'python scripts/soak_raster.py'
Last run (round 2, final kernels): see profiles/r02_soak.txt."""
import sys
import time

import numpy as np

sys.path.insert(0, "free-surgs_amd")
sys.path.insert(0, ".")
import tests.test_fast_step_gpu as TF  # noqa: E402
import tests.test_raster_gpu as TR  # noqa: E402
import tests.test_render_gpu as TG  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402

o = Oracle(np.float32)
o.set_threads(8)
for name, fn, seeds, with_oracle in (
        ("rasteriser vs oracle", TR.test_randomised_small_scenes_match_oracle, range(40, 640), True),
        ("fused render vs two-pass + oracle", TG.test_randomised_fused_render_equals_two_pass, range(12, 252), True),
        ("step driver vs autograd route", TF.test_randomised_image_sizes_step_driver_equals_autograd, range(6, 66), False)):
    bad, t0 = [], time.time()
    for seed in seeds:
        try:
            fn(o, seed) if with_oracle else fn(seed)
        except Exception as e:  # noqa: BLE001 -- report, keep sweeping
            bad.append((seed, type(e).__name__, str(e)[:300]))
    print("%-36s %4d seeds, %d failures, %.0f s" % (name, len(seeds), len(bad), time.time() - t0), flush=True)
    for b in bad[:10]:
        print("   ", b, flush=True)

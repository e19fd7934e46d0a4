"""Soak of the randomised parity sweeps beyond the seeds in tests/ (same test bodies, same attribution rule: every element
beyond 1e-4 of its tensor's inf-norm must be covered by a near-tie the oracle itself witnesses):
rasteriser vs oracle (600 more scenes), fused render vs the two-pass statement + oracle (240 more), step driver vs the
autograd route at random image sizes (60 more).   gpurun -- 'python scripts/soak_raster.py'
Last runs: profiles/r02_soak.txt, profiles/r03_soak.txt (final kernels of each round).
   python scripts/soak_raster.py [n_raster n_render n_step]"""
import sys
import time

import numpy as np

sys.path.insert(0, "free-surgs_amd")
sys.path.insert(0, ".")
import tests.test_fast_step_gpu as TF  # noqa: E402
import tests.test_raster_gpu as TR  # noqa: E402
import tests.test_render_gpu as TG  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402

from tests.util import ATTRIBUTION_LOG, sign_balance  # noqa: E402

n_r, n_g, n_s = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (600, 240, 60)
o = Oracle(np.float32)
o.set_threads(8)
for name, fn, seeds, with_oracle in (
        ("rasteriser vs oracle", TR.test_randomised_small_scenes_match_oracle, range(40, 40 + n_r), True),
        ("fused render vs two-pass + oracle", TG.test_randomised_fused_render_equals_two_pass, range(12, 12 + n_g), True),
        ("step driver vs autograd route", TF.test_randomised_image_sizes_step_driver_equals_autograd, range(6, 6 + n_s), False)):
    bad, t0 = [], time.time()
    del ATTRIBUTION_LOG[:]
    for seed in seeds:
        try:
            fn(o, seed) if with_oracle else fn(seed)
        except Exception as e:  # noqa: BLE001 -- report, keep sweeping
            bad.append((seed, type(e).__name__, str(e)[:300]))
    print("%-36s %4d seeds, %d failures, %.0f s" % (name, len(seeds), len(bad), time.time() - t0), flush=True)
    for b in bad[:10]:
        print("   ", b, flush=True)
    if ATTRIBUTION_LOG:  # the witnessed outliers of the sweep: how many of how many elements, and on which side of the oracle
        pos, neg, z = sign_balance(ATTRIBUTION_LOG)
        print("    witnessed outliers: %d of %d compared elements (%.1e), %d above / %d below the reference (z = %.2f)" % (
            sum(r["outliers"] for r in ATTRIBUTION_LOG), sum(r["size"] for r in ATTRIBUTION_LOG),
            sum(r["outliers"] for r in ATTRIBUTION_LOG) / max(1, sum(r["size"] for r in ATTRIBUTION_LOG)), pos, neg, z), flush=True)

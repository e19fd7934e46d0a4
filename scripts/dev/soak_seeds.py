"""Re-run single seeds of the randomised rasteriser sweep (tests/test_raster_gpu.py) and say what happened.
   python scripts/dev/soak_seeds.py 757 1459 2066      (FSGS_LIB_PATH selects an experiment build)"""
import sys
sys.path[:0] = ["free-surgs_amd", "."]
import numpy as np
import tests.test_raster_gpu as TR
from oracle.fsgs_oracle import Oracle
o = Oracle(np.float32)
o.set_threads(8)
for seed in [int(a) for a in sys.argv[1:]]:
    try:
        TR.test_randomised_small_scenes_match_oracle(o, seed)
        print(seed, "ok")
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:400])

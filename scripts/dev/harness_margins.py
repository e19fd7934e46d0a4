"""How close the C1 harness test (tests/test_harness_c1_gpu.py, product path) comes to each of its bounds: N runs of the test body
with _within recording |got - ref| / bound instead of asserting.   gpurun -- 'python scripts/dev/harness_margins.py 150'"""
import collections
import json
import sys

import numpy as np

sys.path.insert(0, "free-surgs_amd")
sys.path.insert(0, ".")
import tests.test_harness_c1_gpu as T  # noqa: E402
from oracle.fsgs_oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ratios = collections.defaultdict(list)
cur = {}


def within(got, ref, alt, mult, floor, what):
    got, ref, alt = (np.asarray(a, np.float64) for a in (got, ref, alt))
    bound = mult * np.abs(alt - ref) + floor
    cur[what] = max(cur.get(what, 0.0), float(np.max(np.abs(got - ref) / bound)))


T._within = within
T._record = lambda *a, **k: None
o = Oracle(np.float32)
fx = dict(np.load(T.FX, allow_pickle=True))
inputs = T._c1_inputs(o)
for i in range(n):
    cur.clear()
    got, frames, _t, _f = T._run_c1(inputs)
    T._check_outcome(got, fx, frames, tight=False, record=False)
    for k, v in cur.items():
        ratios[k].append(v)
out = {}
for k, v in ratios.items():
    v = np.array(v)
    out[k] = dict(median=float(np.median(v)), p90=float(np.quantile(v, .9)), p99=float(np.quantile(v, .99)), max=float(v.max()),
                  over_0p8=int((v > 0.8).sum()), over_1=int((v > 1).sum()))
    print("%-45s median %.3f p90 %.3f p99 %.3f max %.3f  (> 0.8: %d, > 1: %d of %d)" % (k, out[k]["median"], out[k]["p90"], out[k]["p99"],
          out[k]["max"], out[k]["over_0p8"], out[k]["over_1"], len(v)))
json.dump(dict(runs=n, ratios=out), open("gpurun_out/harness_margins.json", "w"), indent=1)

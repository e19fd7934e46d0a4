import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "free-surgs_amd"))
import numpy as np, torch
from oracle.fsgs_oracle import Oracle
from fsgs_amd import synth
import tests.test_raster_gpu as T
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
W, H = int(rng.integers(5, 150)), int(rng.integers(5, 120))
P = int(rng.choice([1, 2, 7, 64, 65, 300, 1500]))
w2c = None
if seed % 3 == 1:
    w2c = synth.pose_matrix(np.array([1.0, 0, 0, 0]) + 0.05 * rng.standard_normal(4), 0.05 * rng.standard_normal(3))
cam = synth.make_camera(W, H, w2c=w2c)
lo, hi = [(0.05, 0.8), (1.5, 6.0), (4.0, 40.0), (0.3, 120.0)][seed % 4]
ch = (3, 6, 1)[seed % 3 if seed % 5 else 2]
xyz, col, op, s, r = synth.random_small_scene(P, cam, seed=seed, zmin=0.25, zmax=2.0, scale_px=(lo, hi), channels=ch)
if w2c is not None:
    xyz = (np.linalg.inv(w2c) @ np.concatenate([xyz, np.ones((P, 1))], 1).T).T[:, :3]
s[rng.random(P) < 0.2, 0] *= 12.0
op[rng.random(P) < 0.1] = 0.003
op[rng.random(P) < 0.1] = 1.0
xyz[rng.random(P) < 0.05, 2] = 0.2
f = lambda a: np.ascontiguousarray(a, np.float32)
xyz, col, op, s, r = f(xyz), f(col), f(op), f(s), f(r)
print("W H P ch", W, H, P, ch)
dL = (np.random.default_rng(seed).uniform(-1, 1, (ch, H, W)) / (ch * H * W)).astype(np.float32)
img, dep, radii, g = T._run_hip(cam, xyz, col, op, s, r, dL)
o32, o64 = Oracle(np.float32), Oracle(np.float64)
res = {}
for name, o in (("f32", o32), ("f64", o64)):
    oi, od, orad, st = o.raster_forward(cam, xyz, col, op, s, r)
    res[name] = o.raster_backward(st, dL)
for k in ("means3D", "scales", "rotations", "means2D", "opacities", "colors"):
    a, b, c = g[k].reshape(P, -1), res["f32"][k].reshape(P, -1), res["f64"][k].reshape(P, -1)
    sc = np.abs(c).max() + 1e-30
    print(k, "scale", sc, "hip-f64", np.abs(a - c).max() / sc, "f32-f64", np.abs(b - c).max() / sc, "hip-f32", np.abs(a - b).max() / sc)
i = np.unravel_index(np.argmax(np.abs(g["means3D"] - res["f32"]["means3D"])), g["means3D"].shape)
print("worst means3D", i, g["means3D"][i], res["f32"]["means3D"][i], res["f64"]["means3D"][i], "radii", radii[i[0]], "xyz", xyz[i[0]], "scale", s[i[0]], "op", op[i[0]])
if len(sys.argv) > 2:  # python diag_seed.py <seed> <tensor> <row>: one Gaussian's row in the three evaluations
    k, row = sys.argv[2], int(sys.argv[3])
    print(k, "row", row, "hip", g[k].reshape(P, -1)[row], "\n   f32", res["f32"][k].reshape(P, -1)[row], "\n   f64", res["f64"][k].reshape(P, -1)[row])
    print("   radius", radii[row], "xyz", xyz[row], "scale", s[row], "rot", r[row], "op", op[row])
    for kk in ("means3D", "scales", "opacities", "means2D"):
        print("   ", kk, "hip", g[kk].reshape(P, -1)[row], "f32", res["f32"][kk].reshape(P, -1)[row], "f64", res["f64"][kk].reshape(P, -1)[row])

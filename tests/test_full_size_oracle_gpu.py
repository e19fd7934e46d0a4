"""BASELINE.json's own sizes in front of the CPU oracle: the operator-boundary
rasteriser AND the fused 6-channel render at full C2 (1280x1024, 300 k Gaussians -- the bench's trained-like scene,
identity + the perturbed pose) and full C4 (1920x1080, 1 M), forward and every gradient, with flip attribution
(tests/util.py:assert_close_attributed) and the witnessed-outlier fraction / sign balance bounded.

The oracle runs on the cores the cgroup really grants (oracle.fsgs_oracle.usable_cores: 16 on the GPU boxes; the 256
logical CPUs they show make OpenMP and torch thrash).  One nominal pass costs ~1.2 s at C2 there, a run with the
allowances (3 threshold settings x (1 forward + 4 pixel-class backwards) + an fp64 pass) ~15 s; C4 is ~3.5x that.
Match: SURVEY.md s8d "Parity thresholds"; gaussian_renderer/__init__.py:56-92."""
import json

import numpy as np
import pytest
import torch

from fsgs_amd import synth
from fsgs_amd.model import GaussianCloud
from fsgs_amd.render import render, render_two_pass
from fsgs_amd.trainer import PoseTrack, settings_from_cam
from tests.test_raster_gpu import IMAGE_TOL, _compare
from tests.test_render_gpu import _check_against_reference
from tests.util import ATTRIBUTION_LOG, assert_sign_balanced, dump_attribution_log, sh0_colors, to_camera_frame

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIZES = {"C2": (1280, 1024, 300_000), "C4": (1920, 1080, 1_000_000)}
_scenes = {}


@pytest.fixture(scope="module")
def oracle_all_cores(oracle32):
    from oracle.fsgs_oracle import usable_cores

    n = usable_cores()  # (not min(., max_threads()): the session fixture set 1 thread, which is what max_threads reports)
    old = torch.get_num_threads()
    oracle32.set_threads(n)
    torch.set_num_threads(n)
    yield oracle32
    oracle32.set_threads(1)
    torch.set_num_threads(old)


def _scene(cfg):
    """the bench's scene for that configuration (bench.build_problem: trained-like cloud, scales from the HIP KNN)."""
    if cfg not in _scenes:
        from simple_knn._C import distCUDA2

        W, H, P = SIZES[cfg]
        knn = lambda pts: distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
        _scenes.clear()  # one at a time: a C4 cloud is 236 MB on the host
        _scenes[cfg] = synth.trained_like_scene(W, H, P, seed=0, knn_fn=knn)
    return _scenes[cfg]


def _report(name, rec):
    line = json.dumps(dict(test=name, **rec))
    print(line)
    dump_attribution_log("r06_full_size_parity", json.loads(line))


def _assert_mostly_plain(stats, what):
    """the allowances must stay the exception: at full size at most 6 % of a tensor's elements may have one that reaches
    the plain tolerance (measured: images 0.5-1 %, gradients 1-4 %) -- everything else is held to 1e-4 with no excuse"""
    for k, v in stats.items():
        if v.size >= 100_000:
            assert v.fragile <= 0.06 * v.size, "%s %s: %d of %d elements carry an allowance >= the tolerance" % (
                what, k, v.fragile, v.size)


def _summary(stats):
    # max_err / p9999: the achieved error over ALL elements (witnessed outliers included) and its 99.99th percentile,
    # max_err_plain: over the elements whose allowance does not reach the tolerance -- all relative to `scale`
    return {k: dict(outliers=v.outliers, fragile=v.fragile, size=v.size, pos=v.pos, neg=v.neg,
                    witnessed_fraction=v.outliers / max(v.size, 1), max_err=v.max_err, p9999=v.p9999,
                    max_err_plain=v.max_err_plain, scale=v.scale, max_err_zero_amp=v.max_err_zero_amp,
                    zero_amp_fraction=v.zero_amp_fraction) for k, v in stats.items()}


@pytest.mark.parametrize("cfg,pose", [("C2", "identity"), ("C2", "perturbed"), ("C4", "perturbed")])
def test_rasteriser_at_the_operator_boundary_matches_the_oracle_at_full_size(oracle_all_cores, cfg, pose):
    """`GaussianRasterizer(...)` as the reference calls it (camera-frame means, identity raster view matrix,
    gaussian_renderer/__init__.py:56-69) against oracle/raster_oracle.c: image, depth, radii, visibility and all six
    gradients, at BASELINE.json's configs[1] and configs[3] sizes."""
    W, H, P = SIZES[cfg]
    sc = _scene(cfg)
    cam = synth.make_camera(W, H)
    w2c = synth.pose_matrix() if pose == "identity" else synth.pose_matrix(**synth.PERTURBED_POSE)
    xyz = to_camera_frame(sc["_xyz"], w2c)
    s, r, o = synth.activate(sc)
    n_log = len(ATTRIBUTION_LOG)
    R, stats = _compare(oracle_all_cores, cam, xyz, sh0_colors(sc), o.reshape(-1), s, r, seed=7, tag="%s/%s" % (cfg, pose),
                        image_tol=IMAGE_TOL)
    assert R > P
    _assert_mostly_plain(stats, "raster op %s/%s" % (cfg, pose))
    # one near-tie moves an image element up or down with the colour behind it: no systematic sign over a full frame
    pos, neg, z, share = assert_sign_balanced(ATTRIBUTION_LOG[n_log:], "raster op %s/%s" % (cfg, pose))
    _report("raster_op", dict(cfg=cfg, pose=pose, P=P, num_rendered_upstream=R, tensors=_summary(stats),
                              sign_balance=dict(pos=pos, neg=neg, z=z, positive_share=share)))


@pytest.mark.parametrize("cfg,gs_grad,cam_grad", [("C2", True, True), ("C2", False, True), ("C4", True, True)])
def test_fused_render_matches_the_cpu_reference_render_at_full_size(oracle_all_cores, cfg, gs_grad, cam_grad):
    """`render(...)` -- the fused 6-channel op AND the reference's two-pass sequence on the drop-in -- against the CPU
    reference render (tests/ref_cpu.py: reference-pinned torch glue around the oracle), SH degree 3, the perturbed
    pose: the three image planes, uncertainty, radii / visibility / presence, every parameter gradient, the
    densification statistic's viewspace gradient and the pose gradient."""
    W, H, P = SIZES[cfg]
    sc = _scene(cfg)
    cam = synth.make_camera(W, H)
    pc = GaussianCloud(dict(sc), sh_degree=3, device=DEV)
    pc.cam = settings_from_cam(cam, DEV)
    pc.active_sh_degree = 3
    poses = PoseTrack(3, DEV)
    poses.set_pose(1, q=synth.PERTURBED_POSE["q"], t=synth.PERTURBED_POSE["t"])
    g = torch.Generator(device="cpu").manual_seed(11)
    wi = (torch.rand(3, H, W, generator=g) - 0.5).to(DEV) / (H * W)
    wd = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    ws = (torch.rand(H, W, generator=g) - 0.5).to(DEV) / (H * W)
    n_log = len(ATTRIBUTION_LOG)
    stats = _check_against_reference(oracle_all_cores, pc, poses, gs_grad, cam_grad, wi, wd, ws,
                                     fns=(render, render_two_pass), ctx="%s gs=%d cam=%d" % (cfg, gs_grad, cam_grad))
    _assert_mostly_plain(stats, "render %s" % cfg)
    pos, neg, z, share = assert_sign_balanced(ATTRIBUTION_LOG[n_log:], "render %s" % cfg)
    _report("fused_render", dict(cfg=cfg, gs_grad=gs_grad, cam_grad=cam_grad, P=P, tensors=_summary(stats),
                                 sign_balance=dict(pos=pos, neg=neg, z=z, positive_share=share)))

"""GPU parity of the fused loss kernels (csrc/loss.hip) against (a) the golden vectors captured from
the reference's Python and (b) the plain PyTorch fp32 statement of the same op at C1/C2 sizes."""
import os

import numpy as np
import pytest
import torch

from fsgs_amd import losses

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"
T = lambda a: torch.tensor(np.asarray(a), device=DEV)


def test_rgb_loss_matches_reference_golden():
    g = np.load(os.path.join(G, "rgb_loss.npz"))
    gt = T(g["gt"])
    for tag, m in (("nomask", None), ("mask", T(g["mask"]))):
        x = T(g["img"]).requires_grad_(True)
        l = losses.rgb_loss_func(x, gt, mask=m)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"loss_{tag}"], rtol=2e-5)
        # north_star: 1e-4 relative fp32 -- of the gradient's inf-norm (the L1 part is +-0.8 / N almost everywhere)
        want = g[f"grad_{tag}"]
        assert np.abs(x.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()


def test_pearson_matches_reference_golden():
    g = np.load(os.path.join(G, "pearson.npz"))
    src = T(g["src"])
    x = T(g["tgt"]).requires_grad_(True)
    l = losses.pearson_depth_loss(src, x)
    l.backward()
    np.testing.assert_allclose(l.item(), g["pearson"], rtol=1e-4)
    scale = np.abs(g["pearson_grad_tgt"]).max()
    assert np.abs(x.grad.cpu().numpy() - g["pearson_grad_tgt"]).max() <= 1e-4 * scale
    x = T(g["src"]).requires_grad_(True)
    losses.pearson_depth_loss(x, T(g["tgt"])).backward()
    scale = np.abs(g["pearson_grad_src"]).max()
    assert np.abs(x.grad.cpu().numpy() - g["pearson_grad_src"]).max() <= 1e-4 * scale
    x = T(g["tgt"]).requires_grad_(True)
    l = losses.local_pearson_loss(src, x, int(g["lp_box"]), float(g["lp_p"]), (T(g["lp_x0"]), T(g["lp_y0"])))
    l.backward()
    np.testing.assert_allclose(l.item(), g["lp_loss"], rtol=1e-4)
    scale = np.abs(g["lp_grad_tgt"]).max()
    assert np.abs(x.grad.cpu().numpy() - g["lp_grad_tgt"]).max() <= 1e-4 * scale


@pytest.mark.parametrize("shape", [(3, 512, 640), (3, 1024, 1280), (3, 77, 45)])
def test_rgb_loss_matches_torch_statement(shape):
    torch.manual_seed(0)
    gt = torch.rand(shape, device=DEV)
    img = (gt + 0.1 * torch.randn(shape, device=DEV)).clamp(0, 1)
    mask = (torch.rand((1,) + shape[1:], device=DEV) > 0.2)
    for m in (None, mask):
        a = img.clone().requires_grad_(True)
        b = img.clone().requires_grad_(True)
        (3.0 * losses.rgb_loss_func(a, gt, mask=m)).backward()
        (3.0 * losses.rgb_loss_torch(b, gt, mask=m)).backward()
        la, lb = losses.rgb_loss_func(a, gt, mask=m).item(), losses.rgb_loss_torch(b, gt, mask=m).item()
        assert abs(la - lb) <= 2e-5 * abs(lb)
        scale = b.grad.abs().max().item()
        assert (a.grad - b.grad).abs().max().item() <= 2e-4 * scale


@pytest.mark.parametrize("hw", [(512, 640), (1024, 1280)])
def test_pearson_losses_match_torch_statement(hw):
    torch.manual_seed(1)
    H, W = hw
    mono = torch.rand(H, W, device=DEV) + 0.5
    dep = (mono * 0.8 + 0.3 * torch.rand(H, W, device=DEV)).contiguous()
    corners = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    assert len(corners[0]) == int(0.5 * (H // 128) * (W // 128))  # 10 at C1, 40 at C2
    a = dep.clone().requires_grad_(True)
    b = dep.clone().requires_grad_(True)
    la = 0.05 * losses.pearson_depth_loss(mono, a) + 0.15 * losses.local_pearson_loss(mono, a, 128, 0.5, corners)
    lb = 0.05 * losses.pearson_torch(mono, b) + 0.15 * losses.local_pearson_torch(mono, b, 128, 0.5, corners)
    la.backward()
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-4 * abs(lb.item())
    scale = b.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 2e-4 * scale


def test_local_pearson_consumes_rng_like_the_reference():
    """two randint draws on the device, in this order (utils/loss_utils.py:120-121)."""
    H, W = 512, 640
    torch.manual_seed(7)
    x0 = torch.randint(0, H - 128, size=(10,), device=DEV)
    y0 = torch.randint(0, W - 128, size=(10,), device=DEV)
    torch.manual_seed(7)
    c = losses.draw_patch_corners(H, W, 128, 0.5, DEV)
    assert torch.equal(c[0], x0) and torch.equal(c[1], y0)


def test_full_size_properties():
    """size-independent properties at C2: loss(x, x) = 0 for rgb; Pearson is invariant to positive affine
    maps of either argument; gradient of the Pearson loss is orthogonal to constants and to tgt-mean shifts."""
    torch.manual_seed(3)
    H, W = 1024, 1280
    img = torch.rand(3, H, W, device=DEV)
    assert abs(losses.rgb_loss_func(img, img.clone()).item()) < 1e-6
    mono = torch.rand(H, W, device=DEV) + 0.5
    dep = (mono + 0.2 * torch.rand(H, W, device=DEV)).requires_grad_(True)
    l1 = losses.pearson_depth_loss(mono, dep)
    l2 = losses.pearson_depth_loss(3.0 * mono + 2.0, 0.5 * dep.detach() + 7.0)
    assert abs(l1.item() - l2.item()) < 2e-5
    l1.backward()
    assert abs(dep.grad.sum().item()) < 1e-6 * dep.grad.abs().sum().item()


@pytest.mark.parametrize("seed", range(16))
def test_randomised_shapes_rgb_loss(seed):
    """image shapes around every edge of the tiling: narrower / shorter than the 11-tap window, one row, one column,
    one pixel, sizes straddling the 32-pixel tile and the 64-wide streaming segment; with and without a mask; the
    kernels' scratch is reused from call to call."""
    rng = np.random.default_rng(50 + seed)
    H = int(rng.choice([1, 2, 5, 10, 11, 12, 31, 32, 33, 63, 64, 65, 97, 130]))
    W = int(rng.choice([1, 3, 6, 11, 16, 31, 32, 33, 63, 64, 65, 127, 129, 200]))
    g = torch.Generator(device=DEV).manual_seed(seed)
    gt = torch.rand((3, H, W), device=DEV, generator=g)
    img = (gt + 0.2 * torch.randn((3, H, W), device=DEV, generator=g)).clamp(0, 1)
    mask = torch.rand((1, H, W), device=DEV, generator=g) > 0.3
    for m in (None, mask):
        a = img.clone().requires_grad_(True)
        b = img.clone().requires_grad_(True)
        la = losses.rgb_loss_func(a, gt, mask=m)
        lb = losses.rgb_loss_torch(b, gt, mask=m)
        la.backward()
        lb.backward()
        assert abs(la.item() - lb.item()) <= 2e-5 * abs(lb.item()) + 1e-7, (H, W, m is not None)
        scale = b.grad.abs().max().item()
        assert (a.grad - b.grad).abs().max().item() <= 2e-4 * scale + 1e-12, (H, W, m is not None)


@pytest.mark.parametrize("seed", range(10))
def test_randomised_shapes_pearson(seed):
    """global + local Pearson at random sizes (patches clipped to the image, 1..12 patches incl. duplicates and
    patches flush with the border), constant-offset / scaled inputs."""
    rng = np.random.default_rng(80 + seed)
    box = int(rng.choice([8, 32, 128]))
    H, W = int(rng.integers(box + 1, box + 200)), int(rng.integers(box + 1, box + 260))
    n = int(rng.integers(1, 13))
    g = torch.Generator(device=DEV).manual_seed(seed)
    mono = torch.rand((H, W), device=DEV, generator=g) * float(rng.uniform(0.1, 5.0)) + float(rng.uniform(-2, 2))
    dep = (mono * 0.7 + torch.rand((H, W), device=DEV, generator=g)).contiguous()
    x0 = torch.tensor(rng.integers(0, H - box, n), device=DEV)
    y0 = torch.tensor(rng.integers(0, W - box, n), device=DEV)
    if n > 2:
        x0[1], y0[1] = x0[0], y0[0]              # the same patch twice
        x0[2], y0[2] = H - box - 1, W - box - 1  # the last corner randint can return
    a = dep.clone().requires_grad_(True)
    b = dep.clone().requires_grad_(True)
    la = 0.05 * losses.pearson_depth_loss(mono, a) + 0.15 * losses.local_pearson_loss(mono, a, box, 0.5, (x0, y0))
    lb = 0.05 * losses.pearson_torch(mono, b) + 0.15 * losses.local_pearson_torch(mono, b, box, 0.5, (x0, y0))
    la.backward()
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-4 * abs(lb.item()) + 1e-7, (H, W, box, n)
    scale = b.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 2e-4 * scale, (H, W, box, n)


@pytest.mark.parametrize("shape,masked", [((3, 64, 96), False), ((3, 250, 333), True), ((3, 1024, 1280), True)])
def test_fused_forward_backward_entry_equals_the_two_calls(shape, masked):
    """fsgs_photometric_loss_forward_backward (what the step driver calls: forward + backward launches, the loss value
    finished by an extra workgroup of the backward) against fsgs_photometric_loss_forward + _backward, through the raw
    C ABI: same kernels, so loss terms and gradient must be bit-identical; with and without mask / presence plane."""
    import ctypes as C

    from fsgs_amd import _lib

    lib = _lib.load()
    Cc, H, W = shape
    torch.manual_seed(3)
    gt = torch.rand(shape, device=DEV)
    img = (gt + 0.1 * torch.randn(shape, device=DEV)).clamp(0, 1).contiguous()
    mask = (torch.rand((H, W), device=DEV) > 0.2).float().contiguous() if masked else None
    presence = (torch.rand((H, W), device=DEV) - 0.1).contiguous() if masked else None
    up = torch.tensor([0.7], device=DEV)
    nb = int(lib.fsgs_photometric_scratch_bytes(Cc, H, W))
    stream = _lib.current_stream()

    def run(fused):
        maps = torch.empty((3, Cc, H, W), device=DEV)
        sums = torch.zeros((nb,), dtype=torch.uint8, device=DEV)
        out3 = torch.zeros((3,), device=DEV)
        dimg = torch.zeros(shape, device=DEV)
        if fused:
            _lib.check(lib.fsgs_photometric_loss_forward_backward(
                Cc, H, W, _lib.ptr(img), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(presence), 0.2, _lib.ptr(maps), _lib.ptr(sums),
                _lib.ptr(out3), _lib.ptr(up), _lib.ptr(dimg), stream), "fsgs_photometric_loss_forward_backward")
        else:
            _lib.check(lib.fsgs_photometric_loss_forward(
                Cc, H, W, _lib.ptr(img), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(presence), 0.2, _lib.ptr(maps), _lib.ptr(sums),
                _lib.ptr(out3), stream), "fsgs_photometric_loss_forward")
            _lib.check(lib.fsgs_photometric_loss_backward(
                Cc, H, W, _lib.ptr(img), _lib.ptr(gt), _lib.ptr(mask), _lib.ptr(presence), _lib.ptr(maps), _lib.ptr(up), 0.2,
                _lib.ptr(dimg), stream), "fsgs_photometric_loss_backward")
        torch.cuda.synchronize()
        return out3.cpu(), dimg.cpu()

    a3, ad = run(False)
    b3, bd = run(True)
    # the finish runs as one wave (fused) or one 256-thread block (two calls): both sum the per-workgroup partials in
    # double precision, in a different order -- the float results may differ in the last place
    assert torch.allclose(a3, b3, rtol=2e-7, atol=0), (a3, b3)
    assert torch.equal(ad, bd)
    assert float(b3[0]) > 0 and bool(torch.isfinite(bd).all()) and float(bd.abs().max()) > 0
    # and against the torch statement of the op (same check as the two-call route gets above)
    x = img.clone().requires_grad_(True)
    m = None if not masked else (mask * (presence > 0).float())[None]
    ref = losses.rgb_loss_torch(x, gt, mask=m)
    (0.7 * ref).backward()
    assert abs(float(b3[0]) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach()))
    scale = float(x.grad.abs().max())
    assert float((bd.to(DEV) - x.grad).abs().max()) <= 1e-4 * scale
    # argument checks of the new entry
    assert lib.fsgs_photometric_loss_forward_backward(0, H, W, _lib.ptr(img), _lib.ptr(gt), None, None, 0.2, None, None, None,
                                                      None, None, stream) == _lib.FSGS_ERR_INVALID


@pytest.mark.parametrize("shape,n_patches,box", [((3, 64, 96), 0, 0), ((3, 250, 333), 5, 100), ((3, 512, 640), 12, 128),
                                                 ((3, 1024, 1280), 40, 128), ((3, 1080, 1920), 63, 128)])
def test_one_stream_view_loss_stage_equals_the_three_calls(shape, n_patches, box):
    """fsgs_view_losses_forward_backward (round 6: one view's loss stage in two launches on one stream -- the Pearson statistics
    as extra workgroups of the photometric forward, the Pearson gradient, whose waves finish the regions' sums themselves, as
    extra workgroups of the photometric backward) against fsgs_photometric_loss_forward_backward + fsgs_pearson_forward +
    fsgs_pearson_backward through the raw C ABI: every output bit-identical (utils/loss_utils.py:41-127, train.py:250-258)."""
    from fsgs_amd import _lib

    lib = _lib.load()
    Cc, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(11 + n_patches)
    gt = torch.rand(shape, generator=g).to(DEV)
    img = (gt + 0.1 * torch.randn(shape, generator=g).to(DEV)).clamp(0, 1).contiguous()
    mono = torch.rand((H, W), generator=g).to(DEV)
    depth = (0.6 * mono + 0.4 * torch.rand((H, W), generator=g).to(DEV) + 0.5).contiguous()
    rows = torch.randint(0, max(H - box, 1), (max(n_patches, 1),), generator=g).to(DEV)
    cols = torch.randint(0, max(W - box, 1), (max(n_patches, 1),), generator=g).to(DEV)
    if n_patches >= 2:  # one patch in the corner, one overlapping it: the ragged cases of the patch list
        rows[0], cols[0] = H - box - 1, W - box - 1
        rows[1], cols[1] = H - box - 1 - box // 2, W - box - 1 - box // 3
    up = torch.tensor([5.0], device=DEV)
    wreg = torch.full((n_patches + 1,), 0.15 / max(n_patches, 1), device=DEV)
    wreg[0] = 0.05
    nbp = int(lib.fsgs_photometric_scratch_bytes(Cc, H, W))
    nbs = int(lib.fsgs_pearson_scratch_bytes(H, W, n_patches, box))
    stream = _lib.current_stream()

    def run(fused):
        maps = torch.empty((3, Cc, H, W), device=DEV)
        sums = torch.zeros((nbp,), dtype=torch.uint8, device=DEV)
        stats = torch.zeros((nbs,), dtype=torch.uint8, device=DEV)
        out3, out2 = torch.zeros((3,), device=DEV), torch.zeros((2,), device=DEV)
        coef = torch.zeros((8 * (n_patches + 1),), device=DEV)
        dimg, ddep = torch.full(shape, 7.0, device=DEV), torch.full((H, W), 7.0, device=DEV)
        if fused:
            _lib.check(lib.fsgs_view_losses_forward_backward(
                Cc, H, W, _lib.ptr(img), _lib.ptr(gt), None, None, 0.2, _lib.ptr(maps), _lib.ptr(sums), _lib.ptr(out3),
                _lib.ptr(up), _lib.ptr(dimg), n_patches, box, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(mono), _lib.ptr(depth),
                _lib.ptr(stats), _lib.ptr(coef), _lib.ptr(out2), _lib.ptr(wreg), _lib.ptr(ddep), stream),
                "fsgs_view_losses_forward_backward")
        else:
            _lib.check(lib.fsgs_photometric_loss_forward_backward(
                Cc, H, W, _lib.ptr(img), _lib.ptr(gt), None, None, 0.2, _lib.ptr(maps), _lib.ptr(sums), _lib.ptr(out3),
                _lib.ptr(up), _lib.ptr(dimg), stream), "fsgs_photometric_loss_forward_backward")
            _lib.check(lib.fsgs_pearson_forward(H, W, n_patches, box, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(mono),
                                                _lib.ptr(depth), _lib.ptr(stats), _lib.ptr(coef), _lib.ptr(out2), stream),
                       "fsgs_pearson_forward")
            _lib.check(lib.fsgs_pearson_backward(H, W, n_patches, box, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(mono),
                                                 _lib.ptr(depth), _lib.ptr(coef), _lib.ptr(wreg), 0, _lib.ptr(ddep), stream),
                       "fsgs_pearson_backward")
        torch.cuda.synchronize()
        return [t.cpu() for t in (out3, out2, coef.view(-1, 8)[:, :6], dimg, ddep)]

    want, got = run(False), run(True)
    for name, a, b in zip(("rgb terms", "pearson terms", "coefficient rows", "dL/dimage", "dL/ddepth"), want, got):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert bool(torch.isfinite(got[4]).all()) and float(got[4].abs().max()) > 0 and float(got[1][0]) > 0
    if n_patches:
        assert float(got[1][1]) > 0
    # a second call over the same buffers (scratch not re-zeroed) gives the same bits
    again = run(True)
    assert all(torch.equal(a, b) for a, b in zip(got, again))
    # argument checks: more than 63 patches / boxes beyond 128 belong to the three-call route
    bad = lib.fsgs_view_losses_forward_backward(
        Cc, H, W, _lib.ptr(img), _lib.ptr(gt), None, None, 0.2, _lib.ptr(img), _lib.ptr(img), _lib.ptr(img), None, _lib.ptr(img),
        64, box, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(mono), _lib.ptr(depth), _lib.ptr(img), _lib.ptr(img), _lib.ptr(img),
        _lib.ptr(wreg), _lib.ptr(img), stream)
    assert bad == _lib.FSGS_ERR_INVALID
    if H > 200 and W > 200:
        bad = lib.fsgs_view_losses_forward_backward(
            Cc, H, W, _lib.ptr(img), _lib.ptr(gt), None, None, 0.2, _lib.ptr(img), _lib.ptr(img), _lib.ptr(img), None,
            _lib.ptr(img), 1, 130, _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(mono), _lib.ptr(depth), _lib.ptr(img), _lib.ptr(img),
            _lib.ptr(img), _lib.ptr(wreg), _lib.ptr(img), stream)
        assert bad == _lib.FSGS_ERR_INVALID

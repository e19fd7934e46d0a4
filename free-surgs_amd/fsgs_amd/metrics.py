"""Evaluation metrics that define "PSNR/ATE parity" (BASELINE.md s2): PSNR as
utils/general_utils.py:24-30, trajectory metrics as utils/geometry_utils.py:18-29 +
utils/utils_poses/{align_traj,comp_ate}.py (Sim(3) Umeyama alignment of the translation parts, ATE
RMSE, mean relative translation / rotation error).  Host-side numpy; pinned by golden vectors."""
import numpy as np


def psnr(gts, preds):
    """mean over images of -10 log10(per-image MSE); inputs [N,3,H,W] clamped to [0,1]."""
    g = np.clip(np.asarray(gts, np.float32), 0, 1)
    p = np.clip(np.asarray(preds, np.float32), 0, 1)
    mse = ((g - p) ** 2).reshape(g.shape[0], -1).mean(1)
    return float((-10.0 * np.log10(mse)).mean())


def ssim(gts, preds, win_size=7, K1=0.01, K2=0.03, data_range=1.0):
    """The SSIM `rgb_evaluation` reports (utils/general_utils.py:36-48): skimage.metrics.structural_similarity with its
    defaults for a channel-last colour image -- 7x7 UNIFORM window, sample covariance (factor N/(N-1), N = 49),
    C1 = (K1 R)^2, C2 = (K2 R)^2, the window-radius border cropped before averaging, mean over pixels and channels --
    averaged over the images.  scikit-image (0.2x, unpinned in the reference's requirements) is not in this image: the
    definition is restated from its documentation and pinned by closed-form cases in tests/test_golden_host.py
    (identical images -> 1, a constant offset d on a flat image -> (2 m (m+d) + C1) / (m^2 + (m+d)^2 + C1)).
    NOT the Gaussian-window SSIM of the training loss (losses.ssim_torch).  inputs [N,3,H,W] in [0,1]."""
    from scipy.ndimage import uniform_filter

    g = np.asarray(gts, np.float64)
    p = np.asarray(preds, np.float64)
    npx = float(win_size * win_size)
    cov_norm = npx / (npx - 1.0)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    pad = (win_size - 1) // 2
    vals = []
    for n in range(g.shape[0]):
        per_channel = []
        for c in range(g.shape[1]):
            x, y = g[n, c], p[n, c]
            ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
            uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
            vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
            S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
            per_channel.append(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())
        vals.append(np.mean(per_channel))
    return float(np.mean(vals))


def umeyama_sim3(model, data):
    """s, R, t minimising || model - (s R data + t) ||  (Umeyama 1991)."""
    model = np.asarray(model, np.float64)
    data = np.asarray(data, np.float64)
    mu_m, mu_d = model.mean(0), data.mean(0)
    mc, dc = model - mu_m, data - mu_d
    n = model.shape[0]
    cov = mc.T @ dc / n
    var_d = (dc * dc).sum() / n
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / var_d)
    t = mu_m - s * R @ mu_d
    return s, R, t


def align_trajectory(pred, gt):
    """Sim(3)-align `pred` [N,4,4] onto `gt` using the translation columns (align_ate_c2b_use_a2b)."""
    pred = np.asarray(pred, np.float32)
    gt = np.asarray(gt, np.float32)
    s, R, t = umeyama_sim3(gt[:, :3, 3], pred[:, :3, 3])
    R32, t32 = R.astype(np.float32), t.astype(np.float32)
    out = np.tile(np.eye(4, dtype=np.float32), (pred.shape[0], 1, 1))
    out[:, :3, :3] = R32[None] @ pred[:, :3, :3]
    out[:, :3, 3] = np.float32(s) * (pred[:, :3, 3] @ R32.T) + t32
    return out


def compute_ate(gt, pred):
    e = np.linalg.norm(np.asarray(gt)[:, :3, 3] - np.asarray(pred)[:, :3, 3], axis=1)
    return float(np.sqrt(np.mean(e ** 2)))


def compute_rpe(gt, pred):
    te, re = [], []
    for i in range(len(gt) - 1):
        g_rel = np.linalg.inv(gt[i]) @ gt[i + 1]
        p_rel = np.linalg.inv(pred[i]) @ pred[i + 1]
        err = np.linalg.inv(g_rel) @ p_rel
        te.append(np.linalg.norm(err[:3, 3]))
        d = 0.5 * (np.trace(err[:3, :3]) - 1.0)
        re.append(np.arccos(max(min(d, 1.0), -1.0)))
    return float(np.mean(te)), float(np.mean(re))


def pose_metrics(pred, gt):
    """-> aligned trajectory, [rpe_trans, rpe_rot (deg), ate]   (align_pose)."""
    aligned = align_trajectory(pred, gt)
    gt = np.asarray(gt, np.float32)
    ate = compute_ate(gt, aligned)
    rt, rr = compute_rpe(gt, aligned)
    return aligned, [rt, rr * 180.0 / np.pi, ate]

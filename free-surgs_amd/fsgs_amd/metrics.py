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

"""`render(viewpoint_camera, index, pc, gs_grad, cam_grad)` -- the operator boundary L3 of the hot
path (gaussian_renderer/__init__.py:49-92), returning the same 10-key dict.

Two implementations with identical results:
  * render_two_pass: the reference's own sequence of small torch ops around TWO calls of the
    rasteriser drop-in (what an unchanged gaussian_renderer/__init__.py does on this library);
  * render (default): ONE fused HIP op (csrc/render.hip) -- transform_to_frame + activations +
    eval_sh + both passes share one preprocess, one binning and one 6-channel blend.
"""
import torch

from .pose import transform_to_frame
from .rasterizer import GaussianRasterizer
from .sh import eval_sh


def rendervars(pc, transformed, means2D, cam_center):
    """transformed_params2rendervar + transformed_params2depthplussilhouette
    (scene/gaussian_model.py:260-333): colours from SH in torch, (z, 1, z^2) pseudo-colours."""
    scales, rot, opac = pc.get_scaling, pc.get_rotation, pc.get_opacity
    feats = pc.get_features
    K = (pc.max_sh_degree + 1) ** 2
    shs_view = feats.transpose(1, 2).reshape(-1, 3, K)
    d = pc.get_xyz - cam_center.reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, d) + 0.5, 0.0)
    # get_depth_and_silhouette multiplies by cam.viewmatrix[0] AS STORED (transposed storage used as a
    # matrix, scene/gaussian_model.py:266-267); identity for Free-SurGS (first-frame pose), kept verbatim.
    V = pc.cam.viewmatrix.reshape(4, 4)
    z = (transformed @ V[2, :3].reshape(3, 1) + V[2, 3]).reshape(-1, 1)
    depth_sil = torch.cat([z, torch.ones_like(z), z * z], dim=1)
    rv = dict(means3D=transformed, colors_precomp=rgb, rotations=rot, opacities=opac, scales=scales, means2D=means2D)
    dv = dict(means3D=transformed, colors_precomp=depth_sil, rotations=rot, opacities=opac, scales=scales,
              means2D=torch.zeros_like(pc.get_xyz, requires_grad=True) + 0)
    return rv, dv


def _package(pc, im, radius, depth_sil, w2c, means2D):
    depth = depth_sil[0]
    silhouette = depth_sil[1]
    depth_sq = depth_sil[2].unsqueeze(0)
    uncertainty = (depth_sq - depth ** 2).detach()
    seen = radius > 0
    pc.variables["means2D"] = means2D
    # == max_radii2D[seen] = max(radius[seen], max_radii2D[seen]) (radius is 0 where unseen, maxima are >= 0)
    pc.variables["max_radii2D"] = torch.maximum(pc.variables["max_radii2D"], radius.to(torch.float32))
    pc.variables["seen"] = seen
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(uncertainty))
    return {"render": im, "render_dep": depth, "render_w2c": w2c, "render_opacity": silhouette,
            "nan_mask": nan_mask, "presence_mask": silhouette > 0.3, "uncertainty": uncertainty,
            "viewspace_points": means2D, "visibility_filter": radius > 0, "radii": radius}


def render_two_pass(poses, index, pc, gs_grad=True, cam_grad=True):
    means2D = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    if gs_grad:
        means2D.retain_grad()
    w2c = poses.get_pose(index)
    transformed = transform_to_frame(pc.get_xyz, w2c, gs_grad, cam_grad)
    rv, dv = rendervars(pc, transformed, means2D, poses.cam_center)
    im, radius, _ = GaussianRasterizer(raster_settings=pc.cam)(**rv)
    depth_sil, _, _ = GaussianRasterizer(raster_settings=pc.cam)(**dv)
    return _package(pc, im, radius, depth_sil, w2c, means2D)


def render(poses, index, pc, gs_grad=True, cam_grad=True):
    from . import render_ops  # fused HIP op; raises if the library is missing

    means2D = torch.zeros_like(pc.get_xyz, requires_grad=True) + 0
    if gs_grad:
        means2D.retain_grad()
    w2c = poses.get_pose(index)
    im, depth_sil, radius = render_ops.fused_render(pc, w2c, means2D, poses.cam_center, gs_grad, cam_grad)
    return _package(pc, im, radius, depth_sil, w2c, means2D)

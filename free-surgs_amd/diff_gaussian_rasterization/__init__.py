"""Drop-in for the un-vendored `diff_gaussian_rasterization` package (requirements.txt:26):
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:15, scene/gaussian_model.py:18, scene/pose_optimizer.py:5).
Put `free-surgs_amd/` on PYTHONPATH; the implementation is the gfx950 HIP library."""
from fsgs_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)

# FSGS_AUTOBIND=1: also rebind render / the three losses / Adam of an unchanged checkout to the fused path (fsgs_amd/autobind.py)
from fsgs_amd import autobind as _autobind  # noqa: E402

_autobind.install_from_env()

"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:23) on the gfx950 HIP library."""
from fsgs_amd.knn import distCUDA2  # noqa: F401

"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:23) on the gfx950 HIP library."""
from fsgs_amd.knn import distCUDA2  # noqa: F401
from fsgs_amd import autobind as _autobind  # noqa: E402  (FSGS_AUTOBIND=1, see fsgs_amd/autobind.py)

_autobind.install_from_env()

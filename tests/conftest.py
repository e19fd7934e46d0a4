import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "free-surgs_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    from oracle.fsgs_oracle import Oracle

    o = Oracle(np.float32)
    o.set_threads(1)  # deterministic accumulation order
    return o


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    from oracle.fsgs_oracle import Oracle

    o = Oracle(np.float64)
    o.set_threads(1)
    return o


# ---- flavours of the blend kernels (include/fsgs.h FSGS_FLAG_BLEND_ONE_WAVE / _QUAD_WAVES) ---------------------------
# Left alone the library blends forward with four waves per tile and picks the backward's flavour by the size of the tile
# grid, so a small test scene would only ever see the four-waves-per-tile kernels and nothing the one-wave forward.  The parity modules of the rasteriser and of the fused
# render therefore run every test under BOTH forced flavours (tests/test_blend_variants_gpu.py compares the two
# directly, at BASELINE.json's sizes too).  Not doubled: tests that never reach a blend kernel, and the largest sweeps.
BLEND_VARIANT_MODULES = ("test_raster_gpu", "test_render_gpu", "test_render_golden_gpu")
BLEND_VARIANT_SINGLE = {
    "test_wave_transposing_reduction_selftest", "test_alpha_evaluation_is_unbiased_around_the_skip_threshold",
    "test_unsupported_channel_count_is_an_error_not_a_wrong_image", "test_malformed_inputs_raise_before_any_kernel_runs",
    "test_witnessed_outliers_of_c1_and_the_sweep_are_few_and_unsigned",  # reads the log the other tests leave
    "test_binning_is_upstream_order_minus_unreachable_pairs", "test_pair_capacity_overflow_is_reported_and_retried",
    "test_hip_preprocess_covariance_equals_the_reference_golden",
    "test_fused_sh_colours_and_gradients_equal_the_reference_golden",
}


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in BLEND_VARIANT_MODULES and metafunc.function.__name__ not in BLEND_VARIANT_SINGLE:
        metafunc.parametrize("blend_variant", ["one", "quad"], indirect=True)


@pytest.fixture(autouse=True)
def blend_variant(request):
    name = getattr(request, "param", None)
    if name is None:
        yield "auto"
        return
    from fsgs_amd import rasterizer

    prev = rasterizer.set_blend_variant(name)
    try:
        yield name
    finally:
        rasterizer.set_blend_variant(prev)

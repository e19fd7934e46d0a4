"""rasterizer._host_floats: host copies of the tiny camera tensors (no device sync per render).  Oldest-first eviction --
a sweep over more cameras than the cache holds must not throw the whole set away."""
import torch


def test_host_cache_evicts_oldest_first_and_follows_versions(monkeypatch):
    from fsgs_amd import rasterizer as R

    monkeypatch.setattr(R, "_HOST_CACHE_ENTRIES", 8)
    R._host_cache.clear()
    cams = [torch.full((4, 4), float(i)) for i in range(12)]
    for c in cams:
        assert R._host_floats(c, 16)[0] == float(c[0, 0])
    assert len(R._host_cache) == 8
    assert all(id(c) not in R._host_cache for c in cams[:4])  # the four oldest fell out ...
    assert all(id(c) in R._host_cache for c in cams[4:])      # ... and nothing else
    kept = R._host_cache[id(cams[5])][2]
    assert R._host_floats(cams[5], 16) is kept  # a hit returns the cached list (no device read)
    cams[5].add_(1.0)  # in-place update bumps the version: the entry is refreshed, not served stale
    assert R._host_floats(cams[5], 16)[0] == 6.0
    assert len(R._host_cache) == 8
    try:
        R._host_floats(cams[6], 17)
    except ValueError:
        pass
    else:
        raise AssertionError("a camera tensor with too few elements must be refused")
    R._host_cache.clear()

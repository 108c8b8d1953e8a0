

def test_release_version_arithmetic():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                        "tool", "release", "release.py")
    spec = importlib.util.spec_from_file_location("release_tool", path)
    rel = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rel)
    assert rel.parse("4.2.3-rc1") == (4, 2, 3, 1)
    assert rel.render(rel.bump(rel.parse("0.1.9"), "patch")) == "0.1.10"
    assert rel.render(rel.bump(rel.parse("0.1.9"), "minor")) == "0.2.0"
    assert rel.render(rel.bump(rel.parse("v1.4.2"), "major")) == "2.0.0"
    assert rel.render(rel.bump(rel.parse("1.0.0"), "rc")) == "1.0.0-rc1"
    assert rel.render(rel.bump(rel.parse("1.0.0-rc1"), "rc")) == "1.0.0-rc2"
    assert rel.render(rel.bump(rel.parse("1.0.0-rc2"), "stable")) == "1.0.0"
    import pytest
    with pytest.raises(ValueError):
        rel.bump(rel.parse("1.0.0-rc2"), "minor")
    with pytest.raises(ValueError):
        rel.bump(rel.parse("1.0.0"), "stable")
    assert rel.main(["patch", "--dry-run"]) is not None


def test_winograd_route_bookkeeping_on_the_host():
    """Slice counts of the Winograd weight gradient (divisors of the tile
    count, enough workgroups, slices of at least 128 tiles) and the scratch
    bound that keeps huge layers on the direct kernels."""
    import torch
    from singa_amd.core import math_hip
    hm = math_hip.HipMath
    # 56 x 56 x 64 at batch 32, F(4x4): 36 planes x 1 tile of 64 x 64
    T = 32 * 14 * 14
    s = hm._winograd_slices(36, 64, 64, T)
    assert s > 1 and T % s == 0 and T // s >= 128
    # 14 x 14 x 256: 36 x 16 output tiles fill the chip already
    assert hm._winograd_slices(36, 256, 256, 32 * 4 * 4) == 1
    # tile counts without a usable divisor stay whole
    assert hm._winograd_slices(16, 8, 8, 127) == 1
    x = torch.empty(2, 8, 12, 12)
    w = torch.empty(8, 8, 3, 3)
    y = torch.empty(2, 8, 12, 12)
    assert hm._winograd_applies(x, w, y, (1, 1), (1, 1), 1)
    assert not hm._winograd_applies(x, w, y, (2, 2), (1, 1), 1)
    assert not hm._winograd_applies(x, w, y, (1, 1), (0, 0), 1)
    assert not hm._winograd_applies(x, torch.empty(8, 8, 5, 5), y, (1, 1),
                                    (1, 1), 1)
    assert not hm._winograd_applies(x, w, y, (1, 1), (1, 1), 2)
    big = torch.empty((4096, 64, 224, 224), device="meta")   # shapes only
    assert not hm._winograd_applies(big, torch.empty(64, 64, 3, 3,
                                                     device="meta"),
                                    big, (1, 1), (1, 1), 1)

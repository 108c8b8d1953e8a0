

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

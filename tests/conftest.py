import os
import sys

import numpy as np
import pytest

try:  # load torch's HIP runtime before libart so that the process holds exactly one libamdhip64
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; PARITY UNPINNED, see oracle/art_oracle.h)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def scenes():
    from araytracingjourney_amd import scenes as _s
    return _s


_SCENE_CACHE = {}


@pytest.fixture(scope="session")
def get_scene(scenes):
    def _get(name, detail=1.0):
        key = (name, detail)
        if key not in _SCENE_CACHE:
            _SCENE_CACHE[key] = scenes.get_scene(name, detail)
        return _SCENE_CACHE[key]
    return _get


def assert_radiance_close(got, want, rel=1e-4, floor=1e-6, what="radiance"):
    """per-pixel fp32 radiance within 1e-4 relative (abs floor 1e-6)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    err = np.abs(got - want)
    tol = rel * np.abs(want) + floor
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} values off; worst rel err {float((err / (np.abs(want) + floor)).max()):.3e}"


def radiance_margin(got, want, rel=1e-4, floor=1e-6):
    """how much of the 1e-4 tolerance a frame uses: worst |err| / (rel * |want| + floor) (must stay <= 1), the worst plain relative
    error over values above 1e-3 (where the floor plays no part), and the mean relative error there"""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    big = np.abs(want) > 1e-3
    rel_err = err[big] / np.abs(want[big])
    return dict(worst_tolerance_fraction=float((err / (rel * np.abs(want) + floor)).max()), worst_rel_err=float(rel_err.max()) if rel_err.size else 0.0,
                mean_rel_err=float(rel_err.mean()) if rel_err.size else 0.0, values_compared=int(err.size), values_off_by_more_than_1e5_rel=int((rel_err > 1e-5).sum()))


def record_margin(tag, margin):
    """measured on the GPU box: kept under gpurun_out/margins/ (merged back by gpurun); the committed copy lives in tests/golden/<tag>.stats.json"""
    import json
    d = os.path.join(ROOT, "gpurun_out", "margins")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(margin, open(os.path.join(d, tag + ".json"), "w"), indent=1)
    except OSError:
        pass
    print(f"\n[margin] {tag}: {json.dumps(margin)}")

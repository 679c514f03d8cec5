import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PY = os.path.join(ROOT, "pointcloud-raster_amd", "python")
for p in (ROOT, PKG_PY, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A run that contains GPU tests must never pass on the host engine: a GPU pipeline that cannot get its device fails at
    # create instead of falling back as the reference would (Pipeline::create, PCR_REQUIRE_GPU_ENGINE).
    if "not gpu" not in (config.getoption("-m") or "") and any(item.get_closest_marker("gpu") for item in items):
        os.environ.setdefault("PCR_REQUIRE_GPU_ENGINE", "1")


def _denan(v):
    if isinstance(v, list):
        return [_denan(a) for a in v]
    return float("nan") if v == "nan" else v


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def denan():
    return _denan


def grid_from_json(oracle_py, gj):
    g = oracle_py.make_grid(tuple(float(b) for b in gj["bounds"]), cell=tuple(gj["cell"]),
                            tile=tuple(gj["tile"]), dims=tuple(gj["dims"]) if gj["dims"] else None)
    return g


def assert_band_close(got, want, rtol=0.0, atol=0.0, what=""):
    """NaN mask must match exactly; finite values within tolerance (0,0 -> exact equality)."""
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), f"{what}: NaN mask differs at {np.argwhere(gn != wn)[:5].tolist()}"
    g, w = got[~gn], want[~wn]
    if rtol == 0.0 and atol == 0.0:
        bad = g != w
    else:
        bad = np.abs(g - w) > atol + rtol * np.abs(w)
    assert not bad.any(), (f"{what}: {int(bad.sum())} cells differ, max abs err "
                           f"{float(np.max(np.abs(g - w)))}")


def load_cabi():
    """The ctypes view of the C-ABI (pcr/_cabi.py), loaded without importing the pybind module."""
    import importlib.util
    name = "pcr_cabi_standalone"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_PY, "pcr", "_cabi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod

"""Shared pytest configuration: the ``gpu`` marker and golden-vector loading."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def nonfinite_inputs():
    """The seeded inputs of tests/golden/g15_nonfinite.npz (make_golden.g15_inputs)."""
    rng = np.random.default_rng(1515)
    x = rng.standard_normal((3, 600000))
    x[1, 250123] = np.nan
    x[2, 590000:] = np.nan
    xi = x.copy()
    xi[0, 100] = np.inf
    return x, xi


def nonfinite_runs(arr):
    """(channel, start, stop) runs of non-finite samples along the last axis."""
    bad = ~np.isfinite(arr)
    runs = []
    for c in range(arr.shape[0]):
        edges = np.flatnonzero(np.diff(np.concatenate(([0], bad[c].astype(np.int8), [0]))))
        runs += [(c, a, b) for a, b in zip(edges[::2], edges[1::2])]
    return np.array(runs, dtype=np.int64).reshape(-1, 3)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """A fresh checkout has no built artefacts (they are git-ignored): compile
    libosz_hip.so (hipcc cross-compiles gfx950 without a GPU) and the oracle's C
    part once per session if they are missing."""
    import shutil
    from openseize_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and shutil.which("hipcc") or (
            not os.path.exists(_lib.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc")):
        _lib.build()
    from oracle import oracle as orc
    if not os.path.exists(os.path.join(ROOT, "oracle", "libosz_oracle.so")):
        orc.build()
    yield

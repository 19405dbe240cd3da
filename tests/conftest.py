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

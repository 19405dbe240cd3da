"""Welch PSD (mean mode) at lengths only Bluestein's chirp transform keeps on chip, 256 ch x 2^20
resident, beside the rocFFT staging route of the same handle type (OSZ_SPEC_MIX=0).
    PYTHONPATH=. python benchmarks/blue_rates.py"""
import json
import os
import time

import scipy.signal as sps
import torch

from openseize_amd import _device as dev
from openseize_amd import _lib

CH, N = 256, 1 << 20
x = dev.synth_normal(CH, N, seed=0)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for nfft in (347, 1001, 2049, 3001, 4095):
    w = sps.get_window("hann", nfft)
    sc = 1.0 / (nfft * float((w ** 2).sum())) ** 0.5
    row = {"nfft": nfft}
    for route in ("chirp transform", "rocFFT staging"):
        if route.startswith("rocFFT"):
            os.environ["OSZ_SPEC_MIX"] = "0"
        spm = dev.SpecStream(nfft, nfft, nfft - nfft // 2, w, sc, "constant", _lib.SPEC_PSD_MEAN, CH)
        os.environ.pop("OSZ_SPEC_MIX", None)
        dt = timed(lambda: spm.push(x))
        row[route + " ms"] = dt * 1e3
        spm.close()
    row["algorithmic_TBps"] = 8 * CH * N / (row["chirp transform ms"] * 1e-3) / 1e12
    print(json.dumps(row), flush=True)

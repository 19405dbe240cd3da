"""Polyphase resampler alone, 256 ch x 2^20 resident: /5 (cfg-5's first stage), /10, /25, 3/2, 2/3.
    PYTHONPATH=. python benchmarks/poly_rates.py"""
import json
import time

import torch

from openseize_amd import _device as dev
from openseize_amd.filtering.fir import Kaiser

CH, N = 256, 1 << 20
x = dev.synth_normal(CH, N, seed=0)


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for L, M, fs in ((1, 5, 20480), (1, 10, 5000), (1, 25, 5000), (3, 2, 5000), (2, 3, 5000), (1, 2, 5000)):
    cut = fs / (2 * max(L, M))
    h = Kaiser(cut - cut / 10, cut + cut / 10, fs, gpass=0.1, gstop=40).coeffs
    xin = x if L == 1 else x[:, : N // 2].contiguous()
    poly = dev.PolyStream(h, L, M, CH)
    dt = timed(lambda: poly.push(xin, final=False))
    n = xin.shape[1]
    print(json.dumps({"L": L, "M": M, "taps": len(h), "ms": dt * 1e3, "G_input_samples_s": CH * n / dt / 1e9,
                      "algorithmic_TBps": (8 + 8 * L / M) * CH * n / dt / 1e12}), flush=True)
    poly.close()

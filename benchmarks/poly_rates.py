"""Polyphase resampler alone, 256 ch x 2^20 resident: /5 (cfg-5's first stage), /10, /25, 3/2, 2/3 --
the window as the generator hands it over (SciPy's zero padding around it, numerical._resample_padded)
and bare.
    python benchmarks/poly_rates.py"""
import json
import sys
import time

sys.path.insert(0, ".")

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
    from openseize_amd.core import numerical as nm
    n = xin.shape[1]
    row = {"L": L, "M": M, "taps": len(h)}
    for name in ("padded", "bare", "padded", "bare"):
        if name == "padded":
            taps, centre = nm._resample_padded(h, L, M, 96 * n)
            poly = dev.PolyStream(taps, L, M, CH, centre=centre)
            row["taps_padded"] = len(taps)
        else:
            poly = dev.PolyStream(h, L, M, CH)
        dt = timed(lambda: poly.push(xin, final=False))
        row.setdefault(f"ms_{name}", []).append(round(dt * 1e3, 4))
        if name == "padded":
            row.update(G_input_samples_s=round(CH * n / dt / 1e9, 1), algorithmic_TBps=round((8 + 8 * L / M) * CH * n / dt / 1e12, 3))
        poly.close()
    print(json.dumps(row), flush=True)

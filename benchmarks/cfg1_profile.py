"""cfg-1 (16 ch x 1e6, 256-tap FIR, chunksize 30 000) through the public API, host-fed and
resident: wall time and where the host's time goes (cProfile, top entries by own time).

    PYTHONPATH=. python benchmarks/cfg1_profile.py
"""
import cProfile
import io
import pstats
import sys
import time

import numpy as np
import scipy.signal as sps
import torch

from openseize_amd import producer
from openseize_amd.core import numerical as nm

xh = np.random.default_rng(0).standard_normal((16, 1_000_000))
xd = torch.from_numpy(xh).cuda()
hh = sps.firwin(256, 0.2)
sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")


def drain(gen):
    n = 0
    for c in gen:
        n += c.shape[-1]
    torch.cuda.synchronize()
    return n


cases = {
    "oaconvolve host-fed": lambda: drain(nm.oaconvolve(producer(xh, 30000, -1), hh, -1, "same")),
    "oaconvolve resident": lambda: drain(nm.oaconvolve(producer(xd, 30000, -1), hh, -1, "same")),
    "sosfiltfilt host-fed": lambda: drain(nm.sosfiltfilt(producer(xh, 30000, -1), sos, -1)),
    "sosfiltfilt resident": lambda: drain(nm.sosfiltfilt(producer(xd, 30000, -1), sos, -1)),
}
for name, fn in cases.items():
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: {min(ts) * 1e3:.2f} ms for 34 chunks = {min(ts) / 34 * 1e6:.0f} us per chunk", flush=True)
    if len(sys.argv) > 1:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            fn()
        pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(14)
        print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:3500], flush=True)

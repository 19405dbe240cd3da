"""cfg-1 geometry on a device-resident signal: per-chunk cost of small chunks (launches + Python)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, scipy.signal as sps, torch
from openseize_amd import producer
from openseize_amd.core import numerical as nm
x = torch.from_numpy(np.random.default_rng(0).standard_normal((16, 1_000_000))).cuda()
h = sps.firwin(256, 0.2)
sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
def run(fn, cs):
    n = 0
    for c in fn(producer(x, cs, -1)):
        n += c.shape[-1]
    torch.cuda.synchronize()
    return n
for name, fn in (("oaconvolve", lambda p: nm.oaconvolve(p, h, -1, "same")), ("sosfilt", lambda p: nm.sosfilt(p, sos, -1)), ("sosfiltfilt", lambda p: nm.sosfiltfilt(p, sos, -1))):
    for cs in (30000, 300000):
        run(fn, cs)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(fn, cs); ts.append(time.perf_counter() - t0)
        nchunks = -(-1_000_000 // cs)
        print(f"{name} resident 16 x 1e6, cs {cs}: {min(ts)*1e3:.2f} ms = {16e6/min(ts)/1e6:.0f} Msamples/s, {min(ts)/nchunks*1e6:.0f} us per chunk")

import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.signal as sps, torch
from openseize_amd import _device as dev, _lib
_lib.load()
import test_gpu_zp as T
h = sps.firwin(1024, 0.2)
for name, sos in (("bp6", T.BP), ("hp", sps.ellip(4, 0.5, 50, 0.25, "highpass", output="sos")), ("lp5", sps.butter(5, 0.3, output="sos"))):
    for dc in (0.0, 1e2, 1e4, 1e6):
        C, lens = 4, [150000, 150000, 150000, 90001]
        total = sum(lens)
        x = dev.synth_normal(C, total, seed=5)
        drift = torch.linspace(0, 1, total, dtype=torch.float64, device="cuda")[None] * (0.1 * dc)
        x = x + dc + drift
        got, lag = T.run_stream(dev, x, h, sos, lens, False)
        ref = T.whole_stream_reference(x.cpu().numpy(), h, sos)
        g = got.cpu().numpy()
        hi = total - lag - 6000
        lo = 20000      # away from the stream's start (the reference's own start transient)
        err = np.max(np.abs(g[:, lag + lo:lag + hi] - ref[:, lo:hi]))
        print(name, "dc", dc, "abs err", err, "rel to output", err / np.max(np.abs(ref[:, lo:hi])), "out scale", np.max(np.abs(ref[:, lo:hi])), flush=True)

#!/usr/bin/env python3
"""cfg-3 through the PUBLIC API on a device-resident signal: Kaiser-free
1024-tap FIR ('same') -> 6-section Butterworth band-pass, zero phase,
chunksize 2^20, 256 channels x 8 chunks.  Compares with bench.py's
kernel-level steady state (same kernels, no producer glue)."""
import os
import sys
import time
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev, producer
    from openseize_amd.core import numerical as nm
    C, cs, nchunks = 256, 1 << 20, 8
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    x = torch.cat([dev.synth_normal(C, cs, seed=0, n0=k * cs) for k in range(nchunks)], 1)

    def chain():
        fir = producer(partial(nm.oaconvolve, producer(x, cs, -1), h, -1, "same"), cs, -1,
                       shape=tuple(x.shape))
        n = 0
        for out in nm.sosfiltfilt(fir, sos, -1):
            n += out.shape[-1]
        return n

    chain()
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    n = chain()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st1 = torch.cuda.memory_stats()
    print("host-side issue time %.1f ms of %.1f ms; device allocations during the run: %d, frees: %d"
          % (t_host * 1e3, dt * 1e3, st1["num_device_alloc"] - st0["num_device_alloc"],
             st1["num_device_free"] - st0["num_device_free"]))
    print("API chain: %.2f ms per 256 x 2^20 chunk (%.1f Gsamples/s) over %d samples per channel"
          % (dt / nchunks * 1e3, C * n / dt / 1e9, n))


if __name__ == "__main__":
    main()

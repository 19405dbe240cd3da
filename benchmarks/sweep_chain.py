#!/usr/bin/env python3
"""The cfg-3 chain (1024-tap FIR -> 6-section sosfiltfilt, steady state, inputs
resident) over channel counts and chunk sizes: how the kernels hold up away
from the headline shape (the 8-GPU split of cfg-4/5 leaves 32 / 128 channels
per GPU).  One JSON line per shape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(C, CHUNK, steps=12):
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    fo = torch.empty((C, CHUNK), dtype=torch.float64, device="cuda")
    fwd = [torch.empty_like(fo) for _ in range(3)]
    y = torch.empty_like(fo)
    iir.set_state_scaled(ring[0], 0)

    def step(k):
        fir.push(ring[k % 3], 0, out=fo)
        if k < 2:
            iir.forward(fo, out=fwd[k % 3])
        else:
            iir.step(fo, fwd[(k - 2) % 3], fwd[(k - 1) % 3], f_out=fwd[k % 3], y_out=y)

    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4, 4 + steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fir.close()
    iir.close()
    return {"channels": C, "chunksize": CHUNK, "ms_per_chunk": dt * 1e3,
            "Gsamples_s": C * CHUNK / dt / 1e9, "chain_TBps": 48 * C * CHUNK / dt / 1e12}


if __name__ == "__main__":
    for C, CHUNK in ((16, 1 << 20), (32, 1 << 20), (64, 1 << 20), (128, 1 << 20), (256, 1 << 20),
                     (512, 1 << 20), (1024, 1 << 19), (256, 1 << 18), (256, 1 << 16), (256, 1 << 22)):
        print(json.dumps(run(C, CHUNK)), flush=True)

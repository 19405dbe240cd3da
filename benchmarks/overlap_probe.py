#!/usr/bin/env python3
"""Probe: does running the FIR of chunk k+1 beside the sosfiltfilt step of
chunk k (two HIP streams) raise the chain's throughput?  Same workload as
bench.py (cfg-3).  Prints ms per chunk for the serial and the overlapped
schedule."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    C, CHUNK, STEPS = 256, 1 << 20, 20
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    fo = [torch.empty((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(3)]
    fwd = [torch.empty_like(fo[0]) for _ in range(3)]
    y = torch.empty_like(fo[0])
    iir.set_state_scaled(ring[0], 0)

    def sos_step(k):
        if k < 2:
            iir.forward(fo[k % 3], out=fwd[k % 3])
        else:
            iir.step(fo[k % 3], fwd[(k - 2) % 3], fwd[(k - 1) % 3], f_out=fwd[k % 3], y_out=y)

    # serial
    for k in range(3):
        fir.push(ring[k % 3], 0, out=fo[k % 3]); sos_step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(3, 3 + STEPS):
        fir.push(ring[k % 3], 0, out=fo[k % 3]); sos_step(k)
    torch.cuda.synchronize()
    print("serial     %.3f ms per chunk" % ((time.perf_counter() - t0) / STEPS * 1e3))

    # overlapped: FIR(k+1) on stream A beside SOS(k) on stream B
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fir_done = [torch.cuda.Event() for _ in range(3)]
    sos_done = [torch.cuda.Event() for _ in range(3)]
    torch.cuda.synchronize()
    k0 = 3 + STEPS
    with torch.cuda.stream(sa):
        fir.push(ring[k0 % 3], 0, out=fo[k0 % 3]); fir_done[k0 % 3].record(sa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(k0, k0 + STEPS):
        with torch.cuda.stream(sa):   # FIR of the next chunk (its buffer was read by SOS(k-2))
            sa.wait_event(sos_done[(k + 1) % 3])
            fir.push(ring[(k + 1) % 3], 0, out=fo[(k + 1) % 3]); fir_done[(k + 1) % 3].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(fir_done[k % 3])
            sos_step(k); sos_done[k % 3].record(sb)
    torch.cuda.synchronize()
    print("overlapped %.3f ms per chunk" % ((time.perf_counter() - t0) / STEPS * 1e3))


if __name__ == "__main__":
    main()

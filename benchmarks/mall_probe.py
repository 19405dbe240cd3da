#!/usr/bin/env python3
"""Probe: does cutting the chunk into sub-chunks that fit the 256 MB Infinity
Cache let the FIR -> forward-SOS intermediate stay on chip?  Times FIR push +
forward SOS over one 256 x 2^20 chunk, processed in sub-chunks of 2^k samples
(same total work; the intermediate buffer is reused by every sub-chunk)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    C, CHUNK = 256, 1 << 20
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    x = dev.synth_normal(C, CHUNK, seed=0)
    out = torch.empty_like(x)
    for k in (20, 18, 17, 16, 15, 14):
        n = 1 << k
        fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
        tmp = torch.empty((C, n), dtype=torch.float64, device="cuda")

        def run():
            for s in range(0, CHUNK, n):
                fir.push(x[:, s:s + n], 0, out=tmp)
                iir.forward(tmp, out=out[:, s:s + n])
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("sub-chunk 2^%d (%4d MB intermediate): %.3f ms per 256 x 2^20" % (k, C * n * 8 >> 20, dt * 1e3))


if __name__ == "__main__":
    main()

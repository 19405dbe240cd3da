#!/usr/bin/env python3
"""cfg-5 through the PUBLIC API on a device-resident signal: downsample by 5
(fs 20480 -> 4096, default Kaiser anti-alias filter) then STFT (nfft 4096,
50 % overlap) as a producer of segments; 256 channels x 8 chunks of 2^20."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from openseize_amd import _device as dev, producer
    from openseize_amd.resampling.resampling import downsample
    from openseize_amd.spectra.estimators import stft
    C, cs, nchunks = 256, 1 << 20, 8
    x = torch.cat([dev.synth_normal(C, cs, seed=0, n0=k * cs) for k in range(nchunks)], 1)

    def chain():
        y = downsample(producer(x, cs, -1), M=5, fs=20480, chunksize=cs, axis=-1)
        f, t, pro = stft(y, fs=4096, axis=-1, resolution=1.0, asarray=False)
        return sum(1 for _ in pro)

    chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nseg = chain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("API cfg-5: %.2f ms per 256 x 2^20 input chunk (%.1f G input samples/s), %d segments"
          % (dt / nchunks * 1e3, C * cs * nchunks / dt / 1e9, nseg))


if __name__ == "__main__":
    main()

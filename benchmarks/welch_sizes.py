"""Welch PSD (PSD_MEAN mode) timing per on-chip transform size, 256 ch x 2^20, 50 % overlap:
the cube kernel at 4096, the fft8 kernels (OSZ_SPEC_V8=2 forces them at 4096 too)."""
import json
import os
import sys
import time

import numpy as np
import scipy.signal as sps

sys.path.insert(0, ".")


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    CH, N = 256, 1 << 20
    x = dev.synth_normal(CH, N, seed=3)
    for nf, v8 in ((4096, None), (4096, "2"), (512, None), (1024, None), (2048, None), (8192, None),
                   (1000, None), (10000, None)):
        if v8:
            os.environ["OSZ_SPEC_V8"] = v8
        wn = sps.get_window("hann", nf)
        sc = float(np.sqrt(1 / (float(nf) * np.sum(wn ** 2))))
        sp = dev.SpecStream(nf, nf, nf // 2, wn, sc, "constant", _lib.SPEC_PSD_MEAN, CH)
        os.environ.pop("OSZ_SPEC_V8", None)
        dt = timed(lambda: sp.push(x), 9)
        print(json.dumps({"nfft": nf, "forced_fft8": bool(v8), "ms_per_chunk": round(dt * 1e3, 4),
                          "Gsamples_s": round(CH * N / dt / 1e9, 1),
                          "algorithmic_TBps": round(8 * CH * N / dt / 1e12, 3)}), flush=True)
        sp.close()


if __name__ == "__main__":
    main()

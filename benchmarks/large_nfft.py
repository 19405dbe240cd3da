"""Welch PSD (mean mode) and STFT segments for transform lengths whose half is beyond the LDS:
the pairs-of-sub-transforms kernel (csrc/specsplit.h) beside the staging route (spec_prep ->
rocFFT -> spec_post, OSZ_SPEC_SPLIT=0), 256 ch x 2^20 (PSD) / 2^18 (STFT), 50 % overlap.
One JSON line per length; lengths left to rocFFT (a large prime in the way) are marked."""
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
    sizes = [int(v) for v in sys.argv[1:]] or [20412, 20480, 30000, 32768, 40000, 44100, 50000, 60000, 65536,
                                               88200, 100000, 20014]
    for nf in sizes:
        wn = sps.get_window("hann", nf)
        sc = float(np.sqrt(1 / (float(nf) * np.sum(wn ** 2))))
        row = {"nfft": nf}
        for mode, name, xx in ((_lib.SPEC_PSD_MEAN, "psd", x), (_lib.SPEC_DFT_SEGMENTS, "stft", x[:, : 1 << 18])):
            for route, env in (("on_chip", None), ("staging", "0")):
                if env:
                    os.environ["OSZ_SPEC_SPLIT"] = os.environ["OSZ_SPEC_MIX"] = env
                sp = dev.SpecStream(nf, nf, nf // 2, wn, sc, "constant", mode, CH)
                os.environ.pop("OSZ_SPEC_SPLIT", None)
                os.environ.pop("OSZ_SPEC_MIX", None)
                row[f"{name}_{route}_ms"] = round(timed(lambda: sp.push(xx), 5) * 1e3, 3)
                sp.close()
        row["psd_speedup"] = round(row["psd_staging_ms"] / row["psd_on_chip_ms"], 2)
        row["psd_algorithmic_TBps"] = round(8 * CH * N / row["psd_on_chip_ms"] / 1e9, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

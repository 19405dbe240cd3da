#!/usr/bin/env python3
"""The only numbers the reference publishes for this path are
tutorial wall times on its 4-channel x 18 875 000-sample demo recording (5 kHz, about an
hour): filtering.ipynb:1683-3106, resampling.ipynb:650.  The same calls on synthetic data of
that shape, through the CLASS API, host-fed (ndarray in, every chunk back on the host) and
device-resident; four channels is the regime where only parallelism in TIME fills the chip.
One JSON line per call; `reference_s` is the tutorial's wall time (hardware unstated, EDF
decode from disk included).

    PYTHONPATH=. python benchmarks/tutorial_shape.py > profiles/rNN_tutorial_shape.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

C, N, FS = 4, 18_875_000, 5000


def cases():
    from openseize_amd.filtering import fir, iir
    from openseize_amd.resampling.resampling import downsample
    kaiser = fir.Kaiser(fpass=200, fstop=400, gpass=0.5, gstop=40, fs=FS)
    cheb1 = iir.Cheby1(fpass=200, fstop=400, gpass=0.5, gstop=40, fs=FS, fmt="sos")
    notch = iir.Notch(60, width=6, fs=FS)
    return [
        ("Kaiser 57-tap low-pass, mode='same', chunksize 10e6", 3.39,
         lambda pro: kaiser(pro, chunksize=10e6, axis=-1, mode="same")),
        ("Cheby1 3-section low-pass, dephase=True (sosfiltfilt), chunksize 10e6", 4.06,
         lambda pro: cheb1(pro, chunksize=10e6, axis=-1, dephase=True)),
        ("Cheby1 3-section low-pass, dephase=False (sosfilt), chunksize 10e6", 1.45,
         lambda pro: cheb1(pro, chunksize=10e6, axis=-1, dephase=False)),
        ("Notch 60 Hz (ba), dephase=True (filtfilt), chunksize 5e6", 3.54,
         lambda pro: notch(pro, chunksize=5e6, axis=-1, dephase=True)),
        ("downsample(M=25, fs=5000, chunksize=5e6) polyphase", 3.74,
         lambda pro: downsample(pro, M=25, fs=FS, chunksize=5e6)),
    ]


def drain(result):
    n = 0
    for chunk in result:
        n += chunk.shape[-1]
    return n


def main():
    import torch
    from openseize_amd import producer
    from openseize_amd import _device as dev
    xd = dev.synth_normal(C, N, seed=0)
    xh = xd.cpu().numpy()
    for name, ref_s, call in cases():
        row = {"call": name, "shape": [C, N], "reference_s": ref_s,
               "reference_Msamples_s": C * N / ref_s / 1e6}
        for kind, data in (("host_fed", xh), ("resident", xd)):
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = drain(call(producer(data, chunksize=1e6, axis=-1)))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            row[kind + "_s"] = best
            row[kind + "_Msamples_s"] = C * N / best / 1e6
            row["samples_out"] = got
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Measured parity: the largest error of the HIP path, relative to the output
scale, against the golden vectors the reference produced (tests/golden/).  The
tests assert 1e-9; this prints how far below that the path actually sits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300)


def main():
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    from openseize_amd.spectra.estimators import psd, stft
    g = lambda name: np.load(os.path.join(ROOT, "tests", "golden", name))
    rows = []
    g2 = g("g2_fir.npz")
    for taps in (76, 256, 1024):
        e = max(rel(np.concatenate(list(nm.oaconvolve(producer(g2["x"], 4096, -1), g2[f"h{taps}"], -1, m)), -1),
                    g2[f"y_t{taps}_{m}"]) for m in ("full", "same", "valid"))
        rows.append((f"oaconvolve, {taps} taps, 3 modes", e))
    g3 = g("g3_sosfilt.npz")
    keys = [k for k in g3.files if k.startswith("y_")]
    sos_names = sorted({k.split("_cs")[0][2:] for k in keys})
    for name in sos_names:
        ks = [k for k in keys if k.startswith(f"y_{name}_cs") and "zi" not in k]
        if not ks or f"sos_{name}" not in g3.files:
            continue
        e = 0.0
        for k in ks:
            cs = int(k.split("_cs")[1].split("_")[0])
            y = np.concatenate(list(nm.sosfilt(producer(g3["x"], cs, -1), g3[f"sos_{name}"], -1)), -1)
            e = max(e, rel(y, g3[k]))
        rows.append((f"sosfilt, {name}", e))
    g4 = g("g4_sosfiltfilt.npz")
    keys = [k for k in g4.files if k.startswith("y_")]
    for name in sorted({k.split("_cs")[0][2:] for k in keys}):
        if f"sos_{name}" not in g4.files:
            continue
        e = 0.0
        for k in [k for k in keys if k.startswith(f"y_{name}_cs")]:
            cs = int(k.split("_cs")[1])
            y = np.concatenate(list(nm.sosfiltfilt(producer(g4["x"], cs, -1), g4[f"sos_{name}"], -1)), -1)
            e = max(e, rel(y, g4[k]))
        rows.append((f"sosfiltfilt, {name}", e))
    from openseize_amd.resampling import resampling as rs
    g5 = g("g5_resample.npz")
    for L, M in ((1, 5), (3, 1), (3, 2), (2, 7), (3, 11)):
        e = max(rel(rs.resample(g5["x"], L, M, 5000, cs, -1), g5[f"y_L{L}_M{M}_cs{cs}"])
                for cs in (3000, 7001))
        rows.append((f"resample {L}/{M}", e))
    g7 = g("g7_welch.npz")
    for ov in (0.0, 0.5, 0.6):
        cnt, f, p = psd(g7["x"], 1024, axis=-1, resolution=1.0, overlap=ov)
        rows.append((f"psd nfft 1024 (rocFFT path), overlap {ov}", rel(p, g7[f"psd_ov{ov}"])))
    import scipy.signal as sps
    xx = np.random.default_rng(3).standard_normal((4, 200000))
    for ov in (0.5, 0.25):
        cnt, f, p = psd(xx, 4096, axis=-1, resolution=1.0, overlap=ov)
        _, pr = sps.welch(xx, 4096, window="hann", nperseg=4096, noverlap=int(4096 * ov), axis=-1)
        rows.append((f"psd nfft 4096 (on-chip path) vs scipy.signal.welch, overlap {ov}", rel(p, pr)))
    g8 = g("g8_stft.npz")
    e = 0.0
    for b in (1, 0):
        for pd_ in (1, 0):
            f, t, X = stft(g8["x"], 256, axis=-1, resolution=1.0, boundary=bool(b), padded=bool(pd_))
            e = max(e, rel(X, g8[f"X_b{b}_p{pd_}_density"]))
    rows.append(("stft nfft 256 (rocFFT path), 4 boundary/padding cases", e))
    for label, e in rows:
        print(f"{label:44s} {e:9.2e}")


if __name__ == "__main__":
    main()

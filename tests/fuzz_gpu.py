#!/usr/bin/env python3
"""Randomised stress run of the public API against whole-array SciPy / NumPy
(not part of the test suite: minutes, not seconds).  Hunts for geometry bugs:
lengths around block / tile / chunk boundaries, odd chunk sizes, long filters,
ragged last chunks, every mode and axis position.

    python tests/fuzz_gpu.py [iterations] [seed]
"""
import os
import sys
import time

import numpy as np
import scipy.signal as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300) if a.size else 0.0


def interesting_length(rng, lo, hi):
    """Lengths clustered around multiples of the kernel geometries."""
    if rng.random() < 0.6:
        base = int(rng.choice([256, 512, 2048, 3072, 3073, 4096, 6144, 8192, 16384, 30000, 32768, 65536]))
        n = base * int(rng.integers(1, 5)) + int(rng.integers(-3, 4))
    else:
        n = int(rng.integers(lo, hi))
    return int(min(max(n, lo), hi))


def case_array(rng, n):
    ndim = int(rng.integers(1, 4))
    axis = int(rng.integers(0, ndim))
    shape = [int(rng.integers(1, 4)) for _ in range(ndim)]
    shape[axis] = n
    return rng.standard_normal(shape), axis


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import torch
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    from openseize_amd.resampling import resampling as rs
    from openseize_amd.spectra.estimators import psd
    from openseize_amd.filtering.fir import Kaiser
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    designs = [sps.butter(2, 0.3, output="sos"), sps.butter(6, [0.05, 0.3], "bandpass", output="sos"),
               sps.cheby1(5, 1, 0.3, output="sos"), sps.ellip(4, 0.5, 40, [0.1, 0.4], "bandpass", output="sos"),
               sps.butter(8, 0.02, output="sos"), sps.butter(3, 0.6, "highpass", output="sos")]
    bad, t0 = 0, time.time()
    for it in range(iters):
        kind = it % 21
        try:
            if kind == 0:      # FIR
                taps = int(rng.choice([2, 3, 17, 76, 255, 256, 257, 511, 1023, 1024, 1025, 2049, 2050, 3000, 4097]))
                n = interesting_length(rng, taps + 1, 70000)
                x, axis = case_array(rng, n)
                h = rng.standard_normal(taps) / np.sqrt(taps)
                mode = ("full", "same", "valid")[int(rng.integers(0, 3))]
                cs = int(rng.integers(max(taps // 8, 1), n + 100))
                dev_in = rng.random() < 0.5
                src = torch.from_numpy(x).cuda() if dev_in else x
                out = list(nm.oaconvolve(producer(src, cs, axis), h, axis, mode))
                y = np.concatenate([o.cpu().numpy() if dev_in else o for o in out], axis)
                ref = np.apply_along_axis(lambda r: sps.fftconvolve(r, h, mode=mode), axis, x)
                e, what = rel(y, ref), f"fir taps={taps} n={n} shape={x.shape} axis={axis} mode={mode} cs={cs} dev={dev_in}"
            elif kind == 1:    # sosfilt
                n = interesting_length(rng, 50, 80000)
                x, axis = case_array(rng, n)
                sos = designs[int(rng.integers(0, len(designs)))]
                cs = int(rng.integers(10, n + 100))
                y = np.concatenate(list(nm.sosfilt(producer(x, cs, axis), sos, axis)), axis)
                e, what = rel(y, sps.sosfilt(sos, x, axis=axis)), f"sosfilt n={n} shape={x.shape} axis={axis} cs={cs}"
            elif kind == 2:    # sosfiltfilt (chunk-local oracle)
                n = interesting_length(rng, 200, 80000)
                x, axis = case_array(rng, n)
                sos = designs[int(rng.integers(0, len(designs)))]
                cs = int(rng.integers(100, n + 100))
                y = np.concatenate(list(nm.sosfiltfilt(producer(x, cs, axis), sos, axis)), axis)
                x2 = np.moveaxis(x, axis, -1)
                ref = orc.sosfiltfilt(x2.reshape(-1, n), sos, cs).reshape(x2.shape)
                e, what = rel(np.moveaxis(y, axis, -1), ref), f"sosfiltfilt n={n} shape={x.shape} axis={axis} cs={cs}"
            elif kind == 3:    # resample
                L, M = [(1, 2), (1, 5), (1, 10), (1, 20), (2, 1), (3, 1), (3, 2), (2, 7), (5, 3), (4, 25), (7, 5)][int(rng.integers(0, 11))]
                n = interesting_length(rng, 12000, 90000)
                x, axis = case_array(rng, n)
                cs = int(rng.integers(3000, 40000))
                y = rs.resample(x, L, M, 5000, cs, axis)
                fc = 5000 / (2 * max(L, M))
                h = Kaiser(fc - fc / 10, fc + fc / 10, 5000, gpass=0.1, gstop=40).coeffs
                e, what = rel(y, sps.resample_poly(x, L, M, axis=axis, window=h)), f"resample {L}/{M} n={n} shape={x.shape} axis={axis} cs={cs}"
            elif kind == 5:    # stft against the oracle's restatement
                from openseize_amd.spectra.estimators import stft
                fs = float(rng.choice([250, 500, 1000, 173.61, 700, 1111]))
                res = float(rng.choice([0.5, 1.0, 2.0]))
                if rng.random() < 0.1:
                    fs, res = float(rng.choice([10240, 25000, 32768, 22000])), float(rng.choice([0.5, 1.0]))
                nfft = int(fs / res)
                n = interesting_length(rng, 3 * nfft, max(40000, 4 * nfft))
                x = rng.standard_normal((int(rng.integers(1, 4)), n))
                ov = float(rng.choice([0.25, 0.5, 0.75]))
                b, pd_ = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                det = ("constant", "linear")[int(rng.integers(0, 2))]
                f, t, X = stft(x, fs, axis=-1, resolution=res, overlap=ov, detrend=det, boundary=b, padded=pd_)
                rf, rt, rX = orc.stft(x, fs, resolution=res, overlap=ov, detrend=det, boundary=b, padded=pd_)
                e = max(rel(X, rX), 0.0 if np.allclose(t, rt, rtol=0, atol=1e-12) else float("inf"))
                what = f"stft fs={fs} res={res} n={n} ov={ov} boundary={b} padded={pd_} {det}"
            elif kind == 6:    # masked producer -> sosfilt, device tensors
                n = interesting_length(rng, 500, 60000)
                x = rng.standard_normal((int(rng.integers(1, 5)), n))
                mask = rng.random(n) > rng.random() * 0.8
                if mask.sum() < 50:
                    mask[:50] = True
                sos = designs[int(rng.integers(0, len(designs)))]
                cs = int(rng.integers(100, n + 10))
                dev_in = rng.random() < 0.5
                src = torch.from_numpy(x).cuda() if dev_in else x
                out = list(nm.sosfilt(producer(src, cs, -1, mask=mask), sos, -1))
                y = np.concatenate([o.cpu().numpy() if dev_in else o for o in out], -1)
                e, what = rel(y, sps.sosfilt(sos, x[:, mask], axis=-1)), f"masked sosfilt n={n} kept={int(mask.sum())} cs={cs} dev={dev_in}"
            elif kind == 7:    # transfer-function (ba) filters: lfilter and filtfilt
                n = interesting_length(rng, 500, 60000)
                x = rng.standard_normal((int(rng.integers(1, 4)), n))
                order = int(rng.choice([1, 2, 3, 4, 6]))
                b_, a_ = sps.butter(order, float(rng.uniform(0.05, 0.6)))
                cs = int(rng.integers(200, n + 10))
                y = np.concatenate(list(nm.lfilter(producer(x, cs, -1), (b_, a_), -1)), -1)
                e1 = rel(y, sps.lfilter(b_, a_, x, axis=-1))
                z = np.concatenate(list(nm.filtfilt(producer(x, cs, -1), (b_, a_), -1)), -1)
                e2 = rel(z, orc.filtfilt(x, (b_, a_), cs))
                e, what = max(e1 / 10, e2 / 100), f"ba order={order} n={n} cs={cs} (lfilter {e1:.1e}, filtfilt {e2:.1e})"
            elif kind == 19:   # non-finite samples through the transfer-function filters: as far as in the reference
                n = interesting_length(rng, 1000, 60000)
                C = int(rng.integers(1, 4))
                x = rng.standard_normal((C, n))
                for _ in range(int(rng.integers(1, 4))):
                    x[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan if rng.random() < 0.7 else np.inf
                order = int(rng.choice([1, 2, 4]))
                b_, a_ = sps.butter(order, float(rng.uniform(0.05, 0.6)))
                cs = int(rng.integers(300, n + 10))
                dev_in = rng.random() < 0.5
                src = torch.from_numpy(x).cuda() if dev_in else x
                y = np.concatenate([o.cpu().numpy() if dev_in else o for o in nm.lfilter(producer(src, cs, -1), (b_, a_), -1)], -1)
                z = np.concatenate([o.cpu().numpy() if dev_in else o for o in nm.filtfilt(producer(src, cs, -1), (b_, a_), -1)], -1)
                with np.errstate(invalid="ignore"):
                    ry, rz = orc.lfilter(x, (b_, a_), cs)[0], orc.filtfilt(x, (b_, a_), cs)
                e = 0.0
                for got_, ref_ in ((y, ry), (z, rz)):
                    ok = np.isfinite(ref_)
                    if got_.shape != ref_.shape or not np.array_equal(ok, np.isfinite(got_)):
                        e = float("inf")
                    elif ok.any():
                        e = max(e, float(np.max(np.abs(got_[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) / 100)
                what = f"ba nonfinite order={order} n={n} C={C} cs={cs} dev={dev_in}"
            elif kind == 20:   # non-finite samples through the per-segment Welch producer
                fs = float(rng.choice([250, 500, 1000, 173.61, 4096, 347]))
                nfft = int(fs / float(rng.choice([0.5, 1.0, 2.0])))
                n = interesting_length(rng, 3 * nfft, max(30000, 5 * nfft))
                C = int(rng.integers(1, 4))
                x = rng.standard_normal((C, n))
                for _ in range(int(rng.integers(1, 4))):
                    x[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan if rng.random() < 0.7 else np.inf
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                cs = int(rng.integers(nfft, 4 * nfft))
                dev_in = rng.random() < 0.5
                src = torch.from_numpy(x).cuda() if dev_in else x
                freqs, pro_ = nm.welch(producer(src, cs, -1), fs, nfft, "hann", ov, -1, "constant", "density")
                got_ = np.stack([o.cpu().numpy() if dev_in else o for o in pro_], 0)
                with np.errstate(invalid="ignore"):
                    ref_ = np.stack(orc.welch_segments(x, fs, nfft, "hann", ov, "constant", "density")[1], 0)
                ok = np.isfinite(ref_)
                if got_.shape != ref_.shape or not np.array_equal(ok, np.isfinite(got_)):
                    e = float("inf")
                else:
                    e = float(np.max(np.abs(got_[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) if ok.any() else 0.0
                what = f"welch segments nonfinite fs={fs} nfft={nfft} n={n} C={C} ov={ov} cs={cs} dev={dev_in}"
            elif kind == 8:    # sosfilt with a user zi, chunked == whole
                n = interesting_length(rng, 300, 50000)
                x = rng.standard_normal((int(rng.integers(1, 4)), n))
                sos = designs[int(rng.integers(0, len(designs)))]
                zi = rng.standard_normal((sos.shape[0], x.shape[0], 2))
                cs = int(rng.integers(50, n + 10))
                y = np.concatenate(list(nm.sosfilt(producer(x, cs, -1), sos, -1, zi=zi)), -1)
                ref, _ = sps.sosfilt(sos, x, axis=-1, zi=zi)
                e, what = rel(y, ref), f"sosfilt zi n={n} cs={cs}"
            elif kind == 9:    # fused FIR -> forward SOS against the separate kernels
                from openseize_amd import _device as dev
                taps = int(rng.choice([300, 513, 777, 1024, 1300, 1793, 2049]))
                C = int(rng.integers(1, 10))
                step = min((4097 - taps) // 256, 15) * 256
                n = int(rng.integers(8, 40)) * 2 * step + int(rng.integers(0, 2 * step))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                sos = designs[int(rng.integers(0, len(designs)))]
                xs = [torch.from_numpy(rng.standard_normal((C, n))).cuda() for _ in range(2)]
                fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
                ref = [iir.forward(fir.push(x, 0)) for x in xs]
                fir.close(); iir.close()
                fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
                got = [dev.chain_forward(fir, iir, x) for x in xs]
                fir.close(); iir.close()
                e = max(rel(g_.cpu().numpy(), r_.cpu().numpy()) for g_, r_ in zip(got, ref))
                what = f"chain taps={taps} C={C} n={n}"
            elif kind == 10:   # FIR producer -> sosfiltfilt through the API: fused step vs the two generators
                import os
                from functools import partial
                taps = int(rng.choice([2, 64, 301, 512, 1024, 1025, 2049]))
                C = int(rng.integers(1, 6))
                cs = int(rng.integers(65536, 140000))
                nch_ = int(rng.integers(5, 9)) if rng.random() < 0.5 else int(rng.integers(9, 26))   # (long: grouped steps)
                total = cs * (nch_ - 1) + int(rng.integers(1, cs + 1))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                sos = designs[int(rng.integers(0, len(designs)))]
                xdev = torch.from_numpy(rng.standard_normal((C, total))).cuda()

                def through():
                    src = producer(xdev, cs, -1)
                    fir_ = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
                    return torch.cat(list(nm.sosfiltfilt(fir_, sos, -1)), -1).cpu().numpy()
                got_ = through()
                os.environ["OSZ_CHAIN_API"] = "0"
                try:
                    ref_ = through()
                finally:
                    del os.environ["OSZ_CHAIN_API"]
                e, what = rel(got_, ref_) * 100, f"api chain taps={taps} C={C} cs={cs} total={total}"
            elif kind == 14:   # non-finite samples through FIR -> sosfiltfilt on the one-kernel route: the reference's reach
                from functools import partial
                taps = int(rng.choice([64, 256, 513, 1024]))
                C = int(rng.integers(1, 5))
                cs = int(rng.integers(65536, 140000))
                nch_ = int(rng.integers(6, 13))
                total = cs * (nch_ - 1) + int(rng.integers(1, cs + 1))
                h = sps.firwin(taps, 0.3)
                sos = designs[1]
                xh = rng.standard_normal((C, total))
                for _ in range(int(rng.integers(1, 4))):
                    c, at = int(rng.integers(0, C)), int(rng.integers(0, total))
                    r = rng.random()
                    if r < 0.25:
                        xh[c, at:] = np.nan
                    else:
                        xh[c, at] = np.nan if r < 0.8 else np.inf
                data = torch.from_numpy(xh).cuda() if rng.random() < 0.6 else xh
                src = producer(data, cs, -1)
                fir_ = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
                got_ = np.concatenate([o.cpu().numpy() if torch.is_tensor(o) else o for o in nm.sosfiltfilt(fir_, sos, -1)], -1)
                with np.errstate(invalid="ignore"):
                    ref_ = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(xh, h, "same"), -1), sos, cs)
                ok = np.isfinite(ref_)
                if not np.array_equal(ok, np.isfinite(got_)):
                    e = float("inf")
                else:
                    e = float(np.max(np.abs(got_[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) if ok.any() else 0.0
                what = f"chain nonfinite taps={taps} C={C} cs={cs} total={total} resident={torch.is_tensor(data)}"
            elif kind == 15:   # non-finite samples through oaconvolve alone: the reference's segments
                taps = int(rng.choice([3, 17, 64, 65, 255, 256, 1024, 2049, 2050, 3000]))
                total = int(rng.integers(max(4 * taps, 3000), 400000))
                C = int(rng.choice([1, 2, 3, 7]))
                cs = int(rng.integers(max(taps, 500), total + 100))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                mode = ("full", "same", "valid")[int(rng.integers(0, 3))]
                xh = rng.standard_normal((C, total))
                for _ in range(int(rng.integers(1, 5))):
                    c, at = int(rng.integers(0, C)), int(rng.integers(0, total))
                    r = rng.random()
                    if r < 0.2:
                        xh[c, at:] = np.nan
                    elif r < 0.3:
                        xh[c, :at] = np.nan
                    else:
                        xh[c, at] = np.nan if r < 0.8 else np.inf
                if (orc.oa_plan(total, taps, 32)[0] & 1) == 0:       # (an odd fallback nfft: the reference raises)
                    data = torch.from_numpy(xh).cuda() if rng.random() < 0.6 else xh
                    got_ = np.concatenate([o.cpu().numpy() if torch.is_tensor(o) else o
                                           for o in nm.oaconvolve(producer(data, cs, -1), h, -1, mode)], -1)
                    with np.errstate(invalid="ignore"):
                        ref_ = np.concatenate(orc.oaconvolve(xh, h, mode), -1)
                    ok = np.isfinite(ref_)
                    if got_.shape != ref_.shape or not np.array_equal(ok, np.isfinite(got_)):
                        e = float("inf")
                    else:
                        e = float(np.max(np.abs(got_[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) if ok.any() else 0.0
                else:
                    e = 0.0
                what = f"fir nonfinite taps={taps} C={C} cs={cs} total={total} mode={mode}"
            elif kind == 16:   # non-finite samples through FIR -> sosfilt on the fused launch: lost from the segment's start
                taps = int(rng.choice([17, 65, 255, 256, 1024, 2049]))
                cs = int(rng.integers(65536, 90000))
                nch_ = int(rng.integers(3, 8))
                total = cs * (nch_ - 1) + int(rng.integers(max(taps, 600), cs + 1))
                C = int(rng.choice([1, 2, 5]))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                sos = designs[int(rng.integers(0, 4))]
                xh = rng.standard_normal((C, total))
                for _ in range(int(rng.integers(0, 4))):
                    c, at = int(rng.integers(0, C)), int(rng.integers(0, total))
                    r = rng.random()
                    if r < 0.2:
                        xh[c, at:] = np.nan
                    else:
                        xh[c, at] = np.nan if r < 0.8 else np.inf
                data = torch.from_numpy(xh).cuda() if rng.random() < 0.6 else xh
                src = producer(data, cs, -1)
                fir_ = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
                got_ = np.concatenate([o.cpu().numpy() if torch.is_tensor(o) else o for o in nm.sosfilt(fir_, sos, -1)], -1)
                with np.errstate(invalid="ignore"):
                    ref_ = orc.sosfilt(np.concatenate(orc.oaconvolve(xh, h, "same"), -1), sos, cs)[0]
                ok = np.isfinite(ref_)
                if got_.shape != ref_.shape or not np.array_equal(ok, np.isfinite(got_)):
                    e = float("inf")
                else:
                    e = float(np.max(np.abs(got_[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) if ok.any() else 0.0
                what = f"fir->sosfilt nonfinite taps={taps} C={C} cs={cs} total={total} resident={torch.is_tensor(data)}"
            elif kind == 17:   # non-finite samples through the resampler: the outputs SciPy's padded window touches
                L, M = [(1, 2), (1, 5), (1, 10), (1, 20), (2, 1), (3, 1), (3, 2), (2, 7), (5, 3), (4, 25), (7, 5)][int(rng.integers(0, 11))]
                n = interesting_length(rng, 12000, 90000)
                C = int(rng.choice([1, 2, 5]))
                xh = rng.standard_normal((C, n))
                for _ in range(int(rng.integers(1, 5))):
                    c, at = int(rng.integers(0, C)), int(rng.integers(0, n))
                    r = rng.random()
                    if r < 0.2:
                        xh[c, at:] = np.nan
                    else:
                        xh[c, at] = np.nan if r < 0.8 else np.inf
                cs = int(rng.integers(3000, 40000))
                data = torch.from_numpy(xh).cuda() if rng.random() < 0.5 else xh
                y = rs.resample(data, L, M, 5000, cs, -1)
                y = y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y)
                fc = 5000 / (2 * max(L, M))
                h = Kaiser(fc - fc / 10, fc + fc / 10, 5000, gpass=0.1, gstop=40).coeffs
                with np.errstate(invalid="ignore"):
                    ref_ = sps.resample_poly(xh, L, M, axis=-1, window=h)
                ok = np.isfinite(ref_)
                if y.shape != ref_.shape or not np.array_equal(ok, np.isfinite(y)):
                    e = float("inf")
                else:
                    e = float(np.max(np.abs(y[ok] - ref_[ok])) / np.max(np.abs(ref_[ok]))) if ok.any() else 0.0
                what = f"resample nonfinite {L}/{M} n={n} C={C} cs={cs} resident={torch.is_tensor(data)}"
            elif kind == 18:   # non-finite samples through the STFT: the segments that hold them, on every route
                from openseize_amd.spectra.estimators import stft
                fs = float(rng.choice([250, 500, 1000, 173.61, 700, 1111, 4096, 2048, 347]))
                res = float(rng.choice([0.5, 1.0, 2.0]))
                nfft = int(fs / res)
                n = interesting_length(rng, 3 * nfft, max(40000, 4 * nfft))
                C = int(rng.integers(1, 4))
                xh = rng.standard_normal((C, n))
                for _ in range(int(rng.integers(1, 4))):
                    xh[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan if rng.random() < 0.7 else np.inf
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                b, pd_ = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                det = ("constant", "linear")[int(rng.integers(0, 2))]
                with np.errstate(invalid="ignore"):
                    rf, rt, rX = orc.stft(xh, fs, resolution=res, overlap=ov, detrend="constant", boundary=b, padded=pd_)
                if det == "linear":
                    # (SciPy's least-squares trend refuses non-finite data -- the reference raises -- as soon
                    # as a segment holds such a sample: iff the constant-trend result has a lost segment)
                    try:
                        stft(xh, fs, axis=-1, resolution=res, overlap=ov, detrend=det, boundary=b, padded=pd_)
                        e = 0.0 if np.isfinite(rX).all() else float("inf")
                    except ValueError as exc:
                        e = 0.0 if "infs or NaNs" in str(exc) and not np.isfinite(rX).all() else float("inf")
                else:
                    f, t, X = stft(xh, fs, axis=-1, resolution=res, overlap=ov, detrend=det, boundary=b, padded=pd_)
                    ok = np.isfinite(rX)
                    if X.shape != rX.shape or not np.array_equal(ok, np.isfinite(X)):
                        e = float("inf")
                    else:
                        e = float(np.max(np.abs(X[ok] - rX[ok])) / np.max(np.abs(rX[ok]))) if ok.any() else 0.0
                what = f"stft nonfinite fs={fs} res={res} n={n} C={C} ov={ov} boundary={b} padded={pd_} {det}"
            elif kind == 13:   # plain sosfiltfilt of a long resident stream, any layout: grouped zero-phase steps
                ndim = int(rng.integers(1, 4))
                axis = int(rng.integers(0, ndim))
                cs = int(rng.integers(65536, 90000))
                nch_ = int(rng.integers(6, 22))
                total = cs * (nch_ - 1) + int(rng.integers(1, cs + 1))
                shape = [int(rng.integers(1, 3)) for _ in range(ndim)]
                shape[axis] = total
                sos = designs[int(rng.integers(0, len(designs)))]
                xh = rng.standard_normal(shape)
                xdev = torch.from_numpy(xh).cuda()
                got_ = torch.cat(list(nm.sosfiltfilt(producer(xdev, cs, axis), sos, axis)), axis).cpu().numpy()
                x2 = np.moveaxis(xh, axis, -1)
                ref_ = orc.sosfiltfilt(x2.reshape(-1, total), sos, cs).reshape(x2.shape)
                e, what = rel(np.moveaxis(got_, axis, -1), ref_), f"long sosfiltfilt shape={tuple(shape)} axis={axis} cs={cs}"
            elif kind == 12:   # FIR producer -> sosfilt through the API: one launch per chunk vs SciPy
                from functools import partial
                taps = int(rng.choice([2, 64, 301, 512, 1024, 1025, 2049]))
                C = int(rng.integers(1, 6))
                cs = int(rng.integers(65536, 140000))
                nch_ = int(rng.integers(3, 7))
                total = cs * (nch_ - 1) + int(rng.integers(1, cs + 1))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                sos = designs[int(rng.integers(0, len(designs)))]
                xh = rng.standard_normal((C, total))
                src = producer(torch.from_numpy(xh).cuda() if it % 2 else xh, cs, -1)
                fir_ = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
                pieces = list(nm.sosfilt(fir_, sos, -1))
                got_ = np.concatenate([p_.cpu().numpy() if torch.is_tensor(p_) else p_ for p_ in pieces], -1)
                lens_ = [p_.shape[-1] for p_ in pieces]
                u_ = sps.oaconvolve(xh, h[None], axes=-1)[:, (taps - 1) // 2:(taps - 1) // 2 + total]
                e, what = rel(got_, sps.sosfilt(sos, u_, axis=-1)), f"api causal chain taps={taps} C={C} cs={cs} total={total}"
                if lens_ != [cs] * (nch_ - 1) + [total - cs * (nch_ - 1)]:
                    e = float("inf")
            elif kind == 4:    # psd
                fs = float(rng.choice([250, 500, 1000, 4096, 173.61, 700, 1111, 3001]))
                res = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
                if rng.random() < 0.15:      # halves beyond the LDS (specsplit.h), some left to rocFFT
                    fs = float(rng.choice([10240, 15000, 22050, 25000, 32768, 22000, 10007]))
                    res = float(rng.choice([0.5, 1.0]))
                nfft = int(fs / res)
                n = interesting_length(rng, 3 * nfft, max(120000, 5 * nfft))
                x, axis = case_array(rng, n)
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                det = ("constant", "linear")[int(rng.integers(0, 2))]
                sc = ("density", "spectrum")[int(rng.integers(0, 2))]
                win = ("hann", "hamming", "boxcar")[int(rng.integers(0, 3))]
                cnt, f, p = psd(x, fs, axis=axis, resolution=res, window=win, overlap=ov, detrend=det, scaling=sc)
                _, pr = sps.welch(x, fs, window=win, nperseg=nfft, noverlap=int(nfft * ov), detrend=det,
                                  scaling=sc, axis=axis)
                e, what = rel(p, pr), f"psd fs={fs} res={res} n={n} shape={x.shape} axis={axis} ov={ov} {det} {sc} {win}"
            elif kind == 11:   # non-finite samples: as far as in the serial recurrence (time segments!)
                n = int(rng.integers(70000, 700000))
                C = int(rng.choice([1, 2, 3, 5, 40, 130]))
                if C > 5:
                    n = min(n, 200000)
                x = rng.standard_normal((C, n))
                for _ in range(int(rng.integers(1, 4))):
                    c, at = int(rng.integers(0, C)), int(rng.integers(0, n))
                    if rng.random() < 0.4:
                        x[c, at:] = np.nan                     # EDF-style padding to the end
                    else:
                        x[c, at] = np.nan if rng.random() < 0.7 else np.inf
                sos = designs[int(rng.integers(0, 4))]
                cs = int(rng.choice([8192 * int(rng.integers(2, 30)), int(rng.integers(20000, n + 100))]))
                dev_in = rng.random() < 0.5
                src = torch.from_numpy(x).cuda() if dev_in else x
                back = (lambda o: o.cpu().numpy()) if dev_in else (lambda o: o)
                y = np.concatenate([back(o) for o in nm.sosfilt(producer(src, cs, -1), sos, -1)], -1)
                z = np.concatenate([back(o) for o in nm.sosfiltfilt(producer(src, cs, -1), sos, -1)], -1)
                ry, rz = orc.sosfilt(x, sos, cs)[0], orc.sosfiltfilt(x, sos, cs)
                e = 0.0
                for got, want in ((y, ry), (z, rz)):
                    ok = np.isfinite(want)
                    if not np.array_equal(ok, np.isfinite(got)):
                        e = float("inf")
                    elif ok.any():
                        e = max(e, float(np.max(np.abs(got[ok] - want[ok])) / np.max(np.abs(want[ok]))))
                what = f"nonfinite C={C} n={n} cs={cs} dev={dev_in}"
                if e == float("inf"):     # where the masks differ, for a replay
                    for nm_, got, want in (("sosfilt", y, ry), ("sosfiltfilt", z, rz)):
                        diff = np.isfinite(got) != np.isfinite(want)
                        for c_ in np.flatnonzero(diff.any(-1)):
                            idx = np.flatnonzero(diff[c_])
                            badx = np.flatnonzero(~np.isfinite(x[c_]))
                            what += (f" | {nm_} ch {c_}: {len(idx)} samples {idx[0]}..{idx[-1]} differ, got finite there: "
                                     f"{bool(np.isfinite(got[c_, idx[0]]))}; input non-finite at {badx[:4].tolist()} ({len(badx)})")
        except Exception as exc:   # noqa: BLE001 - report and continue
            e, what = float("inf"), f"kind {kind} raised {type(exc).__name__}: {exc}"
        if not e < (1e-8 if kind in (4, 5) else TOL):
            bad += 1
            print(f"FAIL it={it} err={e:.3e} {what}", flush=True)
        if (it + 1) % 50 == 0:
            print(f"{it + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {iters} cases, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

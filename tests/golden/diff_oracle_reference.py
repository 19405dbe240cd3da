"""Differential run of the CPU oracle against the REFERENCE itself on random geometries, with and
without non-finite samples (build container only: the reference does not travel; nothing is written).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src:/root/repo \\
        python /root/repo/tests/golden/diff_oracle_reference.py [cases] [seed]

Prints one line per mismatch (operator, geometry, where) and a summary.  The golden vectors pin the
oracle at fixed geometries; this looks for geometries they do not hold."""
import sys
import warnings
from functools import partial

import numpy as np
import scipy.signal as sps

warnings.filterwarnings("ignore")


def same(got, want, tol=1e-9):
    if got.shape != want.shape:
        return f"shape {got.shape} vs {want.shape}"
    ok = np.isfinite(want)
    if not np.array_equal(ok, np.isfinite(got)):
        d = np.argwhere(ok != np.isfinite(got))
        return f"masks differ at {len(d)} entries, first {d[0].tolist()}, last {d[-1].tolist()}"
    if ok.any():
        e = np.max(np.abs(got[ok] - want[ok])) / max(np.max(np.abs(want[ok])), 1e-300)
        if e > tol:
            return f"values differ by {e:.2e}"
    return None


def poison(rng, x, p=0.6):
    if rng.random() > p:
        return x
    for _ in range(int(rng.integers(1, 4))):
        c, at = int(rng.integers(0, x.shape[0])), int(rng.integers(0, x.shape[1]))
        r = rng.random()
        if r < 0.2:
            x[c, at:] = np.nan
        else:
            x[c, at] = np.nan if r < 0.8 else np.inf
    return x


def main():
    from openseize import producer
    from openseize.core import numerical as ref
    from openseize.resampling import resampling as ref_rs
    from openseize.spectra import estimators as ref_est
    from oracle import oracle as orc
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    designs = [sps.butter(2, 0.3, output="sos"), sps.butter(6, [0.05, 0.3], "bandpass", output="sos"),
               sps.cheby1(5, 1, 0.3, output="sos"), sps.butter(3, 0.6, "highpass", output="sos")]
    bad = 0
    for it in range(cases):
        kind = it % 10
        C = int(rng.integers(1, 4))
        msg = None
        try:
            if kind == 0:
                taps = int(rng.choice([3, 17, 64, 65, 256, 1024]))
                n = int(rng.integers(max(4 * taps, 600), 150000))
                x = poison(rng, rng.standard_normal((C, n)))
                h = rng.standard_normal(taps) / np.sqrt(taps)
                mode = ("full", "same", "valid")[int(rng.integers(0, 3))]
                cs = int(rng.integers(max(taps, 300), n + 50))
                if orc.oa_plan(n, taps, 32)[0] & 1:
                    continue
                want = np.concatenate(list(ref.oaconvolve(producer(x, cs, -1), h, -1, mode)), -1)
                what = f"oaconvolve taps={taps} n={n} cs={cs} {mode}"
                msg = same(np.concatenate(orc.oaconvolve(x, h, mode), -1), want)
            elif kind == 1:
                n = int(rng.integers(300, 60000))
                x = poison(rng, rng.standard_normal((C, n)))
                sos = designs[int(rng.integers(0, 4))]
                cs = int(rng.integers(50, n + 50))
                want = np.concatenate(list(ref.sosfilt(producer(x, cs, -1), sos, -1)), -1)
                what = f"sosfilt n={n} cs={cs}"
                msg = same(orc.sosfilt(x, sos, cs)[0], want)
            elif kind == 2:
                n = int(rng.integers(600, 60000))
                x = poison(rng, rng.standard_normal((C, n)))
                sos = designs[int(rng.integers(0, 4))]
                cs = int(rng.integers(200, n + 50))
                want = np.concatenate(list(ref.sosfiltfilt(producer(x, cs, -1), sos, -1)), -1)
                what = f"sosfiltfilt n={n} cs={cs}"
                msg = same(orc.sosfiltfilt(x, sos, cs), want)
            elif kind == 3:
                L, M = [(1, 2), (1, 5), (1, 10), (2, 1), (3, 2), (2, 7), (5, 3), (4, 25)][int(rng.integers(0, 8))]
                n = int(rng.integers(6000, 60000))
                x = poison(rng, rng.standard_normal((C, n)))
                cs = int(rng.integers(1500, 20000))
                want = ref_rs.resample(x, L, M, 5000, chunksize=cs, axis=-1)
                what = f"resample {L}/{M} n={n} cs={cs}"
                msg = same(orc.polyphase_resample(x, L, M, orc.resample_filter(L, M, 5000)), want)
            elif kind == 4:
                fs = float(rng.choice([250, 500, 173.61, 1000]))
                res = float(rng.choice([0.5, 1.0, 2.0]))
                nfft = int(fs / res)
                n = int(rng.integers(3 * nfft, 12 * nfft))
                x = poison(rng, rng.standard_normal((C, n)))
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                cnt, f, p = ref_est.psd(x, fs, axis=-1, resolution=res, overlap=ov)
                what = f"psd fs={fs} res={res} n={n} ov={ov}"
                oc, of, op = orc.psd(x, fs, resolution=res, overlap=ov)
                msg = same(op, p) if oc == cnt else f"count {oc} vs {cnt}"
            elif kind == 6:     # transfer-function filters (b, a), forward
                n = int(rng.integers(300, 40000))
                x = poison(rng, rng.standard_normal((C, n)))
                coeffs = [sps.butter(2, 0.3), sps.iirnotch(0.24, 8.0), sps.butter(4, [0.1, 0.4], "bandpass"),
                          sps.cheby1(3, 1, 0.25)][int(rng.integers(0, 4))]
                cs = int(rng.integers(50, n + 50))
                want = np.concatenate(list(ref.lfilter(producer(x, cs, -1), coeffs, -1)), -1)
                what = f"lfilter order={len(coeffs[1]) - 1} n={n} cs={cs}"
                msg = same(orc.lfilter(x, coeffs, cs)[0], want, 1e-8)
            elif kind == 7:     # ... forward-backward
                n = int(rng.integers(600, 40000))
                x = poison(rng, rng.standard_normal((C, n)))
                coeffs = [sps.butter(2, 0.3), sps.iirnotch(0.24, 8.0), sps.butter(4, [0.1, 0.4], "bandpass")][int(rng.integers(0, 3))]
                cs = int(rng.integers(200, n + 50))
                want = np.concatenate(list(ref.filtfilt(producer(x, cs, -1), coeffs, -1)), -1)
                what = f"filtfilt order={len(coeffs[1]) - 1} n={n} cs={cs}"
                msg = same(orc.filtfilt(x, coeffs, cs), want, 1e-8)
            elif kind == 8:     # periodogram: padding and cropping, both trends on finite data, windows, scalings
                n = int(rng.integers(50, 3000))
                x = rng.standard_normal((C, n)) + rng.standard_normal()
                det = ("constant", "linear")[int(rng.integers(0, 2))]
                if det == "constant":
                    x = poison(rng, x, 0.4)
                nfft = [None, n, n + int(rng.integers(1, 500)), max(n - int(rng.integers(1, 40)), 8)][int(rng.integers(0, 4))]
                win = ("hann", "hamming", "boxcar", "blackman")[int(rng.integers(0, 4))]
                sc = ("density", "spectrum")[int(rng.integers(0, 2))]
                f, p = ref.periodogram(x, 500.0, nfft=nfft, window=win, axis=-1, detrend=det, scaling=sc)
                what = f"periodogram n={n} nfft={nfft} {win} {det} {sc}"
                of, op = orc.periodogram(x, 500.0, nfft=nfft, window=win, detrend=det, scaling=sc)
                msg = same(op, p) or (None if np.allclose(of, f) else "freqs differ")
            elif kind == 9:     # the per-segment Welch producer
                fs = float(rng.choice([250, 500, 173.61]))
                nfft = int(fs / float(rng.choice([0.5, 1.0, 2.0])))
                n = int(rng.integers(3 * nfft, 10 * nfft))
                x = poison(rng, rng.standard_normal((C, n)), 0.5)
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                win = ("hann", "hamming", "boxcar")[int(rng.integers(0, 3))]
                sc = ("density", "spectrum")[int(rng.integers(0, 2))]
                cs = int(rng.integers(nfft, 4 * nfft))
                freqs, pro = ref.welch(producer(x, cs, -1), fs, nfft, win, ov, -1, "constant", sc)
                want = np.stack(list(pro), -1)
                what = f"welch segments fs={fs} nfft={nfft} n={n} ov={ov} {win} {sc} cs={cs}"
                got = orc.welch_segments(x, fs, nfft, win, ov, "constant", sc)
                got = got[1] if isinstance(got, tuple) else got
                msg = same(np.asarray(got), want) if np.asarray(got).shape == want.shape else same(np.moveaxis(np.asarray(got), 0, -1), want)
            elif kind == 5:
                fs = float(rng.choice([250, 500, 173.61, 1000]))
                res = float(rng.choice([0.5, 1.0, 2.0]))
                nfft = int(fs / res)
                n = int(rng.integers(3 * nfft, 12 * nfft))
                x = poison(rng, rng.standard_normal((C, n)))
                ov = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
                b, pd_ = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                f, t, X = ref_est.stft(x, fs, axis=-1, resolution=res, overlap=ov, boundary=b, padded=pd_)
                what = f"stft fs={fs} res={res} n={n} ov={ov} boundary={b} padded={pd_}"
                of, ot, oX = orc.stft(x, fs, resolution=res, overlap=ov, boundary=b, padded=pd_)
                msg = same(oX, X) or (None if np.allclose(ot, t) else "times differ")
        except Exception as exc:      # noqa: BLE001 - a differential run reports, it does not stop
            what = locals().get("what", f"kind {kind}")
            msg = f"{type(exc).__name__}: {exc}"
        if msg:
            bad += 1
            print(f"MISMATCH it={it} {what}: {msg}", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")


if __name__ == "__main__":
    main()

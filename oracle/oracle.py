"""CPU parity oracle: a NumPy/C restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``openseize_amd/`` imports this
module; it is used by ``tests/``, by ``bench.py``'s ``cpu_baseline`` leg and by
``__graft_entry__.smoke()`` -- always as the checker, never as the product.

Parity status: PINNED.  ``tests/test_oracle.py`` checks every function below
against the golden vectors in ``tests/golden/`` which were produced by running
the reference (``tests/golden/make_golden.py``).

All citations are ``path:line`` under ``/root/reference/src/openseize``.
The oracle deliberately restates the reference's *algorithm*, including its
chunk-dependent behaviour (``sosfiltfilt``) and its quirks (``oaconvolve``
strict ``>`` loop, odd-nfft ``irfft``), so that it can stand in for the
reference on the GPU box where the reference does not exist.

Everything works on arrays whose sample axis is LAST; callers move axes.
"""

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (gcc) next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libosz_oracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libosz_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.osz_ref_sosfilt.argtypes = [dp, ctypes.c_int, dp, ctypes.c_ssize_t,
                                        dp, ctypes.c_ssize_t, ctypes.c_long, dp]
        lib.osz_ref_sosfilt.restype = None
        lib.osz_ref_convolve_full.argtypes = [dp, ctypes.c_long, dp,
                                              ctypes.c_long, dp]
        lib.osz_ref_convolve_full.restype = None
        lib.osz_ref_resample.argtypes = [dp, ctypes.c_long, dp, ctypes.c_long,
                                         ctypes.c_int, ctypes.c_int, dp,
                                         ctypes.c_long]
        lib.osz_ref_resample.restype = None
        lib.osz_ref_lfilter.argtypes = [dp, dp, ctypes.c_int, dp, ctypes.c_ssize_t,
                                        dp, ctypes.c_ssize_t, ctypes.c_long, dp]
        lib.osz_ref_lfilter.restype = None
        _LIB = lib
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# ---------------------------------------------------------------------------
# a1-a4: producer chunking (core/producer.py:54-444, core/queues.py:9-70)
# ---------------------------------------------------------------------------
def array_chunk_lengths(n, chunksize):
    """ArrayProducer.__iter__ (core/producer.py:289-295): slices of
    ``chunksize`` along the axis, last one short."""
    cs = int(chunksize)
    return [min(cs, n - s) for s in range(0, n, cs)]


def rechunk_lengths(total, chunksize):
    """GenProducer.__iter__ (core/producer.py:331-376) and the MaskedProducer
    collector (:426-444): whatever the piece sizes coming in, the stream is
    re-cut at multiples of chunksize and a non-empty remainder is yielded."""
    cs = int(chunksize)
    out = [cs] * (total // cs)
    if total % cs:
        out.append(total % cs)
    return out


def masked_stream(x, mask, chunksize):
    """MaskedProducer.__iter__ (core/producer.py:423-444) on an array: data and
    mask are cut with the same chunksize and zipped, so iteration ends with the
    shorter of the two; chunks whose mask is all False are skipped (:429-430);
    survivors are gathered with np.take(flatnonzero) (:432)."""
    cs = int(chunksize)
    n = x.shape[-1]
    pieces = []
    for s in range(0, min(n, len(mask)), cs):
        m = mask[s:s + cs]
        arr = x[..., s:s + cs]
        if not np.any(m):
            continue
        pieces.append(np.take(arr, np.flatnonzero(m), axis=-1))
    if not pieces:
        return np.zeros(x.shape[:-1] + (0,), dtype=x.dtype)
    return np.concatenate(pieces, axis=-1)


# ---------------------------------------------------------------------------
# a5: overlap-add FIR (core/numerical.py:19-298)
# ---------------------------------------------------------------------------
def optimal_nffts(wlen):
    """core/numerical.py:19-38."""
    return int(8 * 2 ** math.ceil(math.log2(wlen)))


def oa_plan(n, wlen, nfft_factor=32):
    """nfft and step as chosen at core/numerical.py:202-217."""
    nfft = optimal_nffts(wlen) * nfft_factor
    if nfft - wlen + 1 > n:
        nfft = min(optimal_nffts(wlen), n)
    return nfft, nfft - wlen + 1


def oa_boundary_cuts(wlen, mode):
    """Samples removed from the left of the first piece and the right of the
    last piece (core/numerical.py:143-150)."""
    if mode == "full":
        return 0, 0
    if mode == "same":
        return (wlen - 1) // 2, int(math.ceil((wlen - 1) / 2))
    if mode == "valid":
        return wlen - 1, wlen - 1
    raise ValueError(mode)


def oaconvolve(x, h, mode, nfft_factor=32):
    """List of pieces yielded by the reference generator
    (core/numerical.py:158-298) for whole-array input x (..., n).

    The pieces do not depend on the producer's chunksize: the FIFO re-cuts the
    stream into ``step`` samples, and the strict ``>`` at :258 always leaves the
    last (non-empty, <= step) segment to the tail branch :285-298.
    """
    n = x.shape[-1]
    wlen = len(h)
    nfft, step = oa_plan(n, wlen, nfft_factor)
    H = np.fft.rfft(h, nfft)
    overlap = np.zeros(x.shape[:-1] + (wlen - 1,))
    lcut, rcut = oa_boundary_cuts(wlen, mode)

    def cconv(seg):
        # :229-241 (pad by wlen-1, rfft to nfft, multiply, irfft default len)
        seg = np.concatenate(
            [seg, np.zeros(seg.shape[:-1] + (wlen - 1,))], axis=-1)
        return np.fft.irfft(np.fft.rfft(seg, nfft, axis=-1) * H, axis=-1).real

    nfull = -(-n // step) - 1          # segments taken by the while loop
    pieces = []
    for k in range(nfull):
        z = cconv(x[..., k * step:(k + 1) * step])
        y, new_overlap = z[..., :step].copy(), z[..., step:]
        y[..., :wlen - 1] += overlap        # raises like the reference if odd
        overlap = new_overlap
        if k == 0:
            y = y[..., lcut:]
        pieces.append(y)
    rem = x[..., nfull * step:]
    if rem.shape[-1] > 0:
        z = cconv(rem)
        y = z[..., :rem.shape[-1] + wlen - 1].copy()
        y[..., :wlen - 1] += overlap
        ns = y.shape[-1]
        pieces.append(y[..., :ns - rcut])
    return pieces


def convolve_direct(x, h, mode):
    """np.convolve semantics per channel by direct summation (C loop); the
    exact-arithmetic meaning of oaconvolve, independent of segmentation."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    n, m = x.shape[-1], len(h)
    flat = x.reshape(-1, n)
    full = np.empty((flat.shape[0], n + m - 1))
    for c in range(flat.shape[0]):
        _lib().osz_ref_convolve_full(_dp(flat[c]), n, _dp(h), m, _dp(full[c]))
    lcut, rcut = oa_boundary_cuts(m, mode)
    out = full[:, lcut:full.shape[1] - rcut]
    return out.reshape(x.shape[:-1] + (out.shape[1],))


# ---------------------------------------------------------------------------
# a6 / a7: SOS IIR (core/numerical.py:301-411)
# ---------------------------------------------------------------------------
def _sosfilt_block(sos, x, z, reverse=False):
    """One scipy.signal.sosfilt call (core/numerical.py:334) on (C, n) data
    with state z (nsec, C, 2), in place on z.  reverse=True filters the block
    back to front and returns it in natural order, which is what
    flip -> sosfilt -> flip does at :397-403."""
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    C, n = x.shape
    y = np.empty_like(x)
    lib = _lib()
    for c in range(C):
        zc = np.ascontiguousarray(z[:, c, :])
        if reverse:
            lib.osz_ref_sosfilt(_dp(sos), sos.shape[0],
                                _dp(x[c, n - 1:]) if n else _dp(x[c]), -1,
                                _dp(y[c, n - 1:]) if n else _dp(y[c]), -1, n,
                                _dp(zc))
        else:
            lib.osz_ref_sosfilt(_dp(sos), sos.shape[0], _dp(x[c]), 1,
                                _dp(y[c]), 1, n, _dp(zc))
        z[:, c, :] = zc
    return y


def sosfilt(x, sos, chunksize, zi=None):
    """core/numerical.py:301-335: forward cascade with the state carried from
    chunk to chunk.  x is (C, n); zi is (nsec, C, 2) or None.  Returns (y, zf).
    (The result does not depend on chunksize; it is kept in the signature so
    the chunk boundaries are exercised like the reference's.)"""
    x = np.asarray(x, dtype=np.float64)
    C, n = x.shape
    z = np.zeros((len(sos), C, 2)) if zi is None else np.array(zi, dtype=np.float64)
    out = [_sosfilt_block(sos, x[:, s:s + int(chunksize)], z)
           for s in range(0, n, int(chunksize))]
    return np.concatenate(out, axis=-1), z


def sosfilt_zi(sos):
    """Steady-state unit-step state of each section, what
    scipy.signal.sosfilt_zi returns at core/numerical.py:378.  For a DF2T
    biquad with a0 = 1 the fixed point of the state update under constant input
    1 and constant section output g (its DC gain) is
        z1 = b2 - a2*g,  z0 = g - b0   (from y = b0*x + z0)
    and the next section sees input scaled by g."""
    sos = np.asarray(sos, dtype=np.float64)
    zi = np.empty((sos.shape[0], 2))
    scale = 1.0
    for s, (b0, b1, b2, a0, a1, a2) in enumerate(sos):
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        g = (b0 + b1 + b2) / (1.0 + a1 + a2)
        zi[s, 0] = scale * (g - b0)
        zi[s, 1] = scale * (b2 - a2 * g)
        scale *= g
    return zi


def sosfiltfilt(x, sos, chunksize):
    """core/numerical.py:338-411.  Forward pass from zi*x[0] (:374-386); the
    backward pass of chunk i starts from the state left by back-filtering the
    forward output of chunk i+1 ONLY, itself started at zi*(its last sample)
    (:397-403); the last chunk starts from zi*(its own last sample)
    (:408-411).  n = ceil(N / chunksize) (:389)."""
    x = np.asarray(x, dtype=np.float64)
    C, N = x.shape
    cs = int(chunksize)
    zi = sosfilt_zi(sos)[:, None, :]                      # (nsec, 1, 2)
    z = zi * x[:, :1][None, :, :]                          # zi * x0
    z = np.ascontiguousarray(np.broadcast_to(z, (len(sos), C, 2))).copy()
    fwd = [_sosfilt_block(sos, x[:, s:s + cs], z) for s in range(0, N, cs)]
    n = int(math.ceil(N / cs))
    out = []
    for idx, a in enumerate(fwd, 1):
        if idx < n:
            b = fwd[idx]
            zb = np.ascontiguousarray(zi * b[:, -1:][None, :, :]).copy()
            _sosfilt_block(sos, b, zb, reverse=True)        # warm-up -> zf
            out.append(_sosfilt_block(sos, a, zb, reverse=True))
        else:
            za = np.ascontiguousarray(zi * a[:, -1:][None, :, :]).copy()
            out.append(_sosfilt_block(sos, a, za, reverse=True))
    return np.concatenate(out, axis=-1)


# ---------------------------------------------------------------------------
# 8f rank 1: transfer-function filters (core/numerical.py:414-520)
# ---------------------------------------------------------------------------
def _norm_ba(b, a):
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    K = max(len(b), len(a)) - 1
    bb, aa = np.zeros(K + 1), np.zeros(K + 1)
    bb[:len(b)], aa[:len(a)] = b / a[0], a / a[0]
    return bb, aa, K


def _lfilter_block(b, a, K, x, z, reverse=False):
    """One scipy.signal.lfilter call on (C, n) data, state z (C, K) in place."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    C, n = x.shape
    y = np.empty_like(x)
    for c in range(C):
        zc = np.ascontiguousarray(z[c])
        if reverse:
            _lib().osz_ref_lfilter(_dp(b), _dp(a), K, _dp(x[c, n - 1:]), -1,
                                   _dp(y[c, n - 1:]), -1, n, _dp(zc))
        else:
            _lib().osz_ref_lfilter(_dp(b), _dp(a), K, _dp(x[c]), 1, _dp(y[c]), 1,
                                   n, _dp(zc))
        z[c] = zc
    return y


def lfilter_zi(b, a):
    """Steady-state unit-step state of the direct form (what
    scipy.signal.lfilter_zi returns at core/numerical.py:487): with DC gain
    g = sum(b)/sum(a), z[k] = sum_{i>k} (b[i] - a[i] g)."""
    b, a, K = _norm_ba(b, a)
    g = b.sum() / a.sum()
    return np.array([np.sum(b[k + 1:] - a[k + 1:] * g) for k in range(K)])


def lfilter(x, coeffs, chunksize, zi=None):
    """core/numerical.py:414-446 on (C, n); zi (C, K) or None."""
    b, a, K = _norm_ba(*coeffs)
    x = np.asarray(x, dtype=np.float64)
    z = np.zeros((x.shape[0], K)) if zi is None else np.array(zi, dtype=np.float64)
    out = [_lfilter_block(b, a, K, x[:, s:s + int(chunksize)], z)
           for s in range(0, x.shape[1], int(chunksize))]
    return np.concatenate(out, axis=-1), z


def filtfilt(x, coeffs, chunksize):
    """core/numerical.py:449-520: as sosfiltfilt, with lfilter and lfilter_zi."""
    b, a, K = _norm_ba(*coeffs)
    x = np.asarray(x, dtype=np.float64)
    C, N = x.shape
    cs = int(chunksize)
    zi = lfilter_zi(b, a)[None, :]
    z = np.ascontiguousarray(zi * x[:, :1])
    fwd = [_lfilter_block(b, a, K, x[:, s:s + cs], z) for s in range(0, N, cs)]
    n = int(math.ceil(N / cs))
    out = []
    for idx, xa in enumerate(fwd, 1):
        if idx < n:
            yb = fwd[idx]
            zb = np.ascontiguousarray(zi * yb[:, -1:])
            _lfilter_block(b, a, K, yb, zb, reverse=True)
            out.append(_lfilter_block(b, a, K, xa, zb, reverse=True))
        else:
            za = np.ascontiguousarray(zi * xa[:, -1:])
            out.append(_lfilter_block(b, a, K, xa, za, reverse=True))
    return np.concatenate(out, axis=-1)


# ---------------------------------------------------------------------------
# a8: polyphase resampling (core/numerical.py:523-632, resampling.py:72-311)
# ---------------------------------------------------------------------------
def kaiser_lowpass(fpass, fstop, fs, gpass, gstop):
    """The Kaiser design the resampler asks for at core/numerical.py:579-583
    (filtering/fir.py:52-137 + filtering/bases.py:321-361): odd tap count from
    kaiserord on max(pass attenuation, gstop), firwin at the band midpoint."""
    import scipy.signal as sps
    ripple = max(-20 * np.log10(1 - 10 ** (-gpass / 20)), gstop)
    width = abs(fstop - fpass)
    ntaps, _ = sps.kaiserord(ripple, width / (fs / 2))
    ntaps = ntaps + 1 if ntaps % 2 == 0 else ntaps
    cutoff = min(fpass, fstop) + width / 2
    return sps.firwin(ntaps, cutoff=cutoff, width=None,
                      window=("kaiser", sps.kaiser_beta(ripple)),
                      pass_zero="lowpass" if fpass < fstop else "highpass",
                      scale=True, fs=fs)


def resample_filter(L, M, fs, **kwargs):
    """Default anti-alias / interpolation filter, core/numerical.py:579-583."""
    cutoff = fs / (2 * max(L, M))
    fstop = kwargs.pop("fstop", cutoff + cutoff / 10)
    fpass = kwargs.pop("fpass", cutoff - cutoff / 10)
    gpass, gstop = kwargs.pop("gpass", 0.1), kwargs.pop("gstop", 40)
    return kaiser_lowpass(fpass, fstop, fs, gpass, gstop)


def polyphase_resample(x, L, M, h):
    """Whole-stream result of core/numerical.py:523-632 (the chunk/overhang
    machinery reproduces the global resample_poly definition), length
    ceil(N*L/M) (resampling/resampling.py:91)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    n = x.shape[-1]
    if M >= n:
        raise ValueError("Decimation factor must be < samples")  # :569-571
    nout = int(np.ceil(n * L / M))
    flat = x.reshape(-1, n)
    y = np.empty((flat.shape[0], nout))
    for c in range(flat.shape[0]):
        _lib().osz_ref_resample(_dp(flat[c]), n, _dp(h), len(h), L, M,
                                _dp(y[c]), nout)
    bad = ~np.isfinite(flat)
    if bad.any():
        y[resample_lost(bad, L, M, len(h))] = np.nan
    return y.reshape(x.shape[:-1] + (nout,))


def resample_lost(bad, L, M, ntaps):
    """Outputs a non-finite input sample costs scipy.signal.resample_poly -- and so the reference,
    whatever its chunking (core/numerical.py:590-632; pinned by tests/golden/g20): SciPy pads the
    window with zeros in front (n_pre_pad = M - half % M) and behind (n_post_pad, until the output
    is long enough) and upfirdn pads every phase to the same count of taps, K = ceil(padded / L);
    0 x NaN is NaN there, so output j is lost iff one of the K samples
    x[((j + n_pre_remove) M) // L - k], k < K, is non-finite.  bad: (..., n) bool -> (..., nout)."""
    n = bad.shape[-1]
    nout = -(-n * L // M)
    half = (ntaps - 1) // 2
    pre = M - half % M
    remove = (half + pre) // M

    def upfirdn_len(len_h):                    # scipy.signal._upfirdn._output_len
        return ((n - 1) * L + len_h - 1) // M + 1

    post = 0
    while upfirdn_len(ntaps + pre + post) < nout + remove:
        post += 1
    K = -(-(ntaps + pre + post) // L)
    count = np.concatenate([np.zeros(bad.shape[:-1] + (1,), dtype=np.int64), np.cumsum(bad, axis=-1)], axis=-1)
    top = ((np.arange(nout) + remove) * M) // L                       # the newest sample an output touches
    lo, hi = np.clip(top - K + 1, 0, n), np.clip(top + 1, 0, n)
    return (count[..., hi] - count[..., lo]) > 0


# ---------------------------------------------------------------------------
# a9-a12: windowed DFT, periodogram, Welch, STFT
# (core/numerical.py:635-1087, spectra/estimators.py:59-284)
# ---------------------------------------------------------------------------
def _detrend(x, kind):
    """scipy.signal.detrend along the last axis (core/numerical.py:691)."""
    if kind == "constant":
        return x - x.mean(axis=-1, keepdims=True)
    if kind == "linear":
        if not np.isfinite(x).all():
            # scipy.signal.detrend(type='linear') fits by scipy.linalg.lstsq, which refuses
            # non-finite data: the reference raises here (core/numerical.py:691)
            raise ValueError("array must not contain infs or NaNs")
        n = x.shape[-1]
        t = np.arange(1, n + 1, dtype=np.float64) / n
        A = np.stack([t, np.ones(n)], axis=1)
        coef, *_ = np.linalg.lstsq(A, x.reshape(-1, n).T, rcond=None)
        return x - (A @ coef).T.reshape(x.shape)
    raise ValueError(kind)


def get_window(window, n):
    import scipy.signal as sps
    return sps.get_window(window, n)


def modified_dft(x, fs, nfft, window, detrend, scaling):
    """core/numerical.py:635-718 on (..., nsamples)."""
    if nfft < x.shape[-1]:
        x = x[..., :nfft]
    x = _detrend(x, detrend)
    w = get_window(window, x.shape[-1])
    X = np.fft.rfft(x * w, nfft, axis=-1)
    if scaling == "spectrum":
        norm = 1 / np.sum(w) ** 2
    elif scaling == "density":
        norm = 1 / (fs * np.sum(w ** 2))
    else:
        raise ValueError("Unknown scaling: {}".format(scaling))
    return np.fft.rfftfreq(nfft, d=1 / fs), X * np.sqrt(norm)


def periodogram(x, fs, nfft=None, window="hann", detrend="constant",
                scaling="density"):
    """core/numerical.py:721-796."""
    nfft = x.shape[-1] if not nfft else int(nfft)
    f, X = modified_dft(x, fs, nfft, window, detrend, scaling)
    P = X.real ** 2 + X.imag ** 2
    if nfft % 2:
        P[..., 1:] *= 2
    else:
        P[..., 1:-1] *= 2
    return f, P


def segment_starts(n, nfft, overlap):
    """_spectra_estimatives (core/numerical.py:799-849): one estimate per
    ``stride`` while at least nfft samples are buffered; a trailing partial
    segment is dropped."""
    noverlap = int(nfft * overlap)
    stride = nfft - noverlap
    return list(range(0, n - nfft + 1, stride)), stride


def welch_segments(x, fs, nfft, window, overlap, detrend, scaling):
    starts, _ = segment_starts(x.shape[-1], nfft, overlap)
    f = np.fft.rfftfreq(nfft, 1 / fs)
    return f, [periodogram(x[..., s:s + nfft], fs, nfft, window, detrend,
                           scaling)[1] for s in starts]


def welch_reported_nsegs(n, nfft, overlap):
    """The ``shape[axis]`` the reference's welch/stft producers report
    (core/numerical.py:941, :1071) -- float arithmetic kept as is."""
    return int((n - nfft) // (nfft * (1 - overlap)) + 1)


def psd(x, fs, resolution=0.5, window="hann", overlap=0.5, detrend="constant",
        scaling="density"):
    """spectra/estimators.py:59-156: nfft = int(fs/resolution) (:144); running
    mean over the segment PSDs (:149-152); returns (cnt, freqs, mean)."""
    nfft = int(fs / resolution)
    f, segs = welch_segments(x, fs, nfft, window, overlap, detrend, scaling)
    result = 0
    cnt = 0
    for cnt, arr in enumerate(segs, 1):
        result = result + 1 / cnt * (arr - result)
    return cnt, f, result


def stft(x, fs, resolution=0.5, window="hann", overlap=0.5, detrend="constant",
         scaling="density", boundary=True, padded=True):
    """core/numerical.py:950-1087 through spectra/estimators.py:160-284:
    zero-extend nfft//2 both sides if boundary (:1041-1044), then a whole
    stride of zeros if padded and N % stride (:1046-1051); segment times
    :1076-1083.  Returns (freqs, time, X[..., nfreq, nseg])."""
    nfft = int(fs / resolution)
    noverlap = int(nfft * overlap)
    stride = nfft - noverlap
    n0 = x.shape[-1]
    data = x
    if boundary:
        z = np.zeros(x.shape[:-1] + (nfft // 2,))
        data = np.concatenate([z, data, z], axis=-1)
    if padded:
        amt = stride if n0 % stride else 0
        data = np.concatenate(
            [data, np.zeros(x.shape[:-1] + (amt,))], axis=-1)
    npad = data.shape[-1]
    starts, _ = segment_starts(npad, nfft, overlap)
    segs = [modified_dft(data[..., s:s + nfft], fs, nfft, window, detrend,
                         scaling)[1] for s in starts]
    if boundary:
        time = 1 / fs * np.arange(0, npad - nfft + 1, stride)
    else:
        time = 1 / fs * np.arange(nfft // 2, npad + 1 - nfft // 2, stride)
    f = np.fft.rfftfreq(nfft, 1 / fs)
    return f, time, np.stack(segs, axis=-1)

"""Randomised GPU parity, in the manner of the reference's own tests
(tests/test_oaconvolve.py:30-83, test_iir.py:77-158, test_resampling.py:39-139,
test_spectra.py:166-330): data of 1 to 4 dimensions with the sample axis in a
random position, random chunksizes, checked against whole-array NumPy / SciPy
(present in the image; not the reference).  Seeds are fixed."""

import numpy as np
import pytest
import scipy.signal as sps

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300)


@pytest.fixture(scope="module")
def nm():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from openseize_amd import _lib
    _lib.load()
    from openseize_amd.core import numerical
    return numerical


def producer(*a, **k):
    from openseize_amd import producer as p
    return p(*a, **k)


def random_case(rng, nmin, nmax):
    """(array, axis): 1-4 dims, the sample axis anywhere."""
    ndim = int(rng.integers(1, 5))
    axis = int(rng.integers(0, ndim))
    shape = [int(rng.integers(1, 4)) for _ in range(ndim)]
    shape[axis] = int(rng.integers(nmin, nmax))
    return rng.standard_normal(shape), axis


def test_random_oaconvolve(nm):
    rng = np.random.default_rng(9001)
    for _ in range(14):
        x, axis = random_case(rng, 3000, 26000)
        taps = int(rng.integers(8, 700))
        h = rng.standard_normal(taps) / np.sqrt(taps)
        mode = ("full", "same", "valid")[int(rng.integers(0, 3))]
        cs = int(rng.integers(taps + 1, x.shape[axis] + 500))
        y = np.concatenate(list(nm.oaconvolve(producer(x, cs, axis), h, axis, mode)), axis)
        ref = np.apply_along_axis(np.convolve, axis, x, h, mode=mode)
        assert rel_err(y, ref) < RTOL, (x.shape, axis, taps, mode, cs)


def test_random_sosfilt_and_sosfiltfilt(nm):
    from oracle import oracle as orc
    rng = np.random.default_rng(9002)
    designs = [sps.butter(4, 0.2, output="sos"), sps.butter(6, [0.05, 0.3], "bandpass", output="sos"),
               sps.cheby1(5, 1, 0.3, output="sos"), sps.ellip(4, 0.5, 40, [0.1, 0.4], "bandpass", output="sos"),
               sps.butter(3, 0.6, "highpass", output="sos")]
    for it in range(12):
        x, axis = random_case(rng, 2500, 30000)
        sos = designs[it % len(designs)]
        cs = int(rng.integers(300, x.shape[axis] + 500))
        y = np.concatenate(list(nm.sosfilt(producer(x, cs, axis), sos, axis)), axis)
        assert rel_err(y, sps.sosfilt(sos, x, axis=axis)) < RTOL, (x.shape, axis, cs)
        # zero phase: the chunk-local definition of the reference (oracle, 2-D, last axis)
        x2 = np.moveaxis(x, axis, -1)
        flat = x2.reshape(-1, x2.shape[-1])
        ref = orc.sosfiltfilt(flat, sos, cs).reshape(x2.shape)
        z = np.concatenate(list(nm.sosfiltfilt(producer(x, cs, axis), sos, axis)), axis)
        assert rel_err(np.moveaxis(z, axis, -1), ref) < RTOL, (x.shape, axis, cs)


def test_random_resample(nm):
    """resample / downsample / upsample against SciPy's whole-array
    resample_poly with the anti-aliasing filter the reference designs
    (core/numerical.py:579-583); M = 10 and 25 use the smaller kernel tiles."""
    from openseize_amd.filtering.fir import Kaiser
    from openseize_amd.resampling import resampling as rs
    rng = np.random.default_rng(9003)
    for L, M in ((1, 5), (1, 10), (1, 3), (3, 1), (3, 2), (2, 7), (5, 3), (4, 25)):
        x, axis = random_case(rng, 20000, 60000)
        fs = 5000
        cs = int(rng.integers(9000, 30000))
        y = rs.resample(x, L, M, fs, cs, axis)
        z = np.concatenate(list(rs.resample(producer(x, cs, axis), L, M, fs, cs, axis)), axis)
        assert np.array_equal(y, z)                     # ndarray in -> ndarray out, same numbers
        n = x.shape[axis]
        assert y.shape[axis] == -(-n * L // M)          # ceil(n L / M), resampling.py:91
        fc = fs / (2 * max(L, M))
        h = Kaiser(fc - fc / 10, fc + fc / 10, fs, gpass=0.1, gstop=40).coeffs
        ref = sps.resample_poly(x, L, M, axis=axis, window=h)
        assert rel_err(y, ref) < RTOL, (x.shape, axis, L, M, cs)


def test_random_welch_psd(nm):
    from openseize_amd.spectra import estimators as est
    rng = np.random.default_rng(9004)
    for it in range(8):
        x, axis = random_case(rng, 30000, 90000)
        fs = float((500, 1000, 4096)[it % 3])
        res = (0.5, 1.0, 2.0)[int(rng.integers(0, 3))]
        window = ("hann", "hamming", "boxcar")[int(rng.integers(0, 3))]
        overlap = (0.5, 0.25, 0.0)[int(rng.integers(0, 3))]
        scaling = ("density", "spectrum")[int(rng.integers(0, 2))]
        detrend = ("constant", "linear")[int(rng.integers(0, 2))]
        cnt, freqs, p = est.psd(x, fs, axis=axis, resolution=res, window=window, overlap=overlap,
                                detrend=detrend, scaling=scaling)
        nfft = int(fs / res)
        f_ref, p_ref = sps.welch(x, fs, window=window, nperseg=nfft, noverlap=int(nfft * overlap),
                                 detrend=detrend, scaling=scaling, axis=axis)
        assert cnt == (x.shape[axis] - nfft) // (nfft - int(nfft * overlap)) + 1
        assert np.allclose(freqs, f_ref)
        assert rel_err(p, p_ref) < 1e-8, (x.shape, axis, fs, res, window, overlap, scaling, detrend)

"""GPU parity of the generic on-chip spectra path (csrc/fft8.h + spec8_kernel):
power-of-two nfft from 512 to 8192, windows shorter than nfft (zero padding),
any overlap, both detrends, PSD mean / PSD segments / STFT segments -- against
the CPU oracle and whole-array SciPy.  nfft = int(fs / resolution) in the
reference (spectra/estimators.py:144), so these sizes are what fs = 512 ... 8192
Hz at 1 Hz resolution, or 256 ... 4096 Hz at the default 0.5 Hz, produce.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9
SIZES = (512, 1024, 2048, 4096, 8192)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300)


def producer(*a, **k):
    from openseize_amd import producer as p
    return p(*a, **k)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from openseize_amd import _lib
    _lib.load()


@pytest.fixture(params=["auto", "fft8_for_4096"])
def path(request):
    """Second pass: nfft = 4096 also goes through the fft8 kernel (it defaults to
    the 256-thread cube kernel there)."""
    old = os.environ.get("OSZ_SPEC_V8")
    if request.param == "fft8_for_4096":
        os.environ["OSZ_SPEC_V8"] = "2"
    yield request.param
    if old is None:
        os.environ.pop("OSZ_SPEC_V8", None)
    else:
        os.environ["OSZ_SPEC_V8"] = old


@pytest.mark.parametrize("nfft", SIZES)
def test_psd_all_sizes_vs_oracle(path, nfft):
    """psd() with fs = nfft, resolution 1: host-fed in small ragged chunks (the
    carry of every push feeds the head segments) and device-resident."""
    import torch
    from oracle import oracle as orc
    from openseize_amd.spectra.estimators import psd
    if path == "fft8_for_4096" and nfft != 4096:
        pytest.skip("only 4096 changes path")
    rng = np.random.default_rng(nfft)
    x = rng.standard_normal((5, 9 * nfft + 1234)) + 0.3
    for overlap, detrend in ((0.5, "constant"), (0.6, "linear"), (0.0, "constant")):
        rc, rf, rp = orc.psd(x, nfft, resolution=1.0, overlap=overlap, detrend=detrend)
        cnt, f, p = psd(x, fs=nfft, axis=-1, resolution=1.0, overlap=overlap, detrend=detrend)
        assert cnt == rc and np.array_equal(f, rf)
        assert rel_err(p, rp) < RTOL
        cnt, f, p = psd(torch.from_numpy(x).cuda(), fs=nfft, axis=-1, resolution=1.0,
                        overlap=overlap, detrend=detrend)
        assert cnt == rc and rel_err(p.cpu().numpy(), rp) < RTOL
    # ragged pushes straight on the handle: chunk cuts that leave every carry length
    import scipy.signal as sps
    from openseize_amd import _device as dev, _lib
    w = sps.get_window("hann", nfft)
    scale = float(np.sqrt(1 / (float(nfft) * np.sum(w ** 2))))
    spec = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant", _lib.SPEC_PSD_MEAN, 5)
    xd = torch.from_numpy(x).cuda()
    cuts = [0, 100, nfft - 1, nfft + 7, 3 * nfft + 11, 3 * nfft + 12, 7 * nfft, x.shape[1]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        spec.push(xd[:, a:b].contiguous())
    cnt, mean = spec.mean()
    spec.close()
    rc, _, rp = orc.psd(x, nfft, resolution=1.0)
    assert cnt == rc and rel_err(mean, rp) < RTOL


@pytest.mark.parametrize("nfft", SIZES)
def test_stft_and_welch_segments_all_sizes(path, nfft):
    import scipy.signal as sps
    from oracle import oracle as orc
    from openseize_amd.core import numerical as nm
    from openseize_amd.spectra.estimators import stft
    if path == "fft8_for_4096" and nfft != 4096:
        pytest.skip("only 4096 changes path")
    rng = np.random.default_rng(7 + nfft)
    x = rng.standard_normal((3, 6 * nfft + 321))
    f, t, X = stft(x, fs=nfft, axis=-1, resolution=1.0, overlap=0.5, boundary=True, padded=True)
    rf, rt, rX = orc.stft(x, nfft, resolution=1.0)
    assert np.array_equal(f, rf) and np.allclose(t, rt, rtol=0, atol=1e-12)
    assert X.shape == rX.shape and rel_err(X, rX) < RTOL
    # per-segment periodograms of welch() against scipy.signal.spectrogram-free welch
    freqs, pro = nm.welch(producer(x, 3000, -1), nfft, nfft, "hann", 0.5, -1, "constant", "density")
    segs = np.stack(list(pro), -1)
    fw, pw = sps.welch(x, fs=nfft, window="hann", nperseg=nfft, noverlap=nfft // 2,
                       detrend="constant", scaling="density", axis=-1)
    assert rel_err(segs.mean(-1), pw) < RTOL


def test_short_window_zero_padded():
    """periodogram / modified_dft with nfft above the sample count: the window
    covers the samples only, the transform is zero padded (reference
    core/numerical.py:697-699)."""
    import scipy.signal as sps
    from openseize_amd.core import numerical as nm
    rng = np.random.default_rng(5)
    for n, nfft in ((700, 1024), (300, 512), (5000, 8192), (2049, 4096), (1, 512)):
        x = rng.standard_normal((4, n)) + 1.0
        for detrend in ("constant", "linear"):
            if n == 1 and detrend == "linear":
                continue
            for scaling in ("density", "spectrum"):
                f, p = nm.periodogram(x, fs=500.0, nfft=nfft, window="hamming", axis=-1,
                                      detrend=detrend, scaling=scaling)
                rf, rp = sps.periodogram(x, fs=500.0, nfft=nfft, window="hamming", axis=-1,
                                         detrend=detrend, scaling=scaling)
                assert np.allclose(f, rf) and p.shape == rp.shape
                assert np.max(np.abs(p - rp)) < RTOL * max(np.max(np.abs(rp)), 1e-300)


def test_fullsize_welch_every_size():
    """256 channels x 2^20 samples through every on-chip size: the one-sided
    density integrates to the variance (Parseval) and the count is exact."""
    import scipy.signal as sps
    from openseize_amd import _device as dev, _lib
    C, n = 256, 1 << 20
    x = dev.synth_normal(C, n, seed=17)
    for nfft in SIZES:
        w = sps.get_window("hann", nfft)
        fs = float(nfft)
        scale = float(np.sqrt(1 / (fs * np.sum(w ** 2))))
        spec = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant", _lib.SPEC_PSD_MEAN, C)
        spec.push(x)
        cnt, p = spec.mean()
        spec.close()
        assert cnt == (n - nfft) // (nfft // 2) + 1
        var = p.sum(axis=1) * (fs / nfft)
        assert np.all(np.abs(var - 1.0) < 0.03), (nfft, float(np.max(np.abs(var - 1.0))))

"""GPU parity of the mixed-radix on-chip spectra path (csrc/specmix.h): even
nfft whose half is a product of 2, 3, 5 and 7 -- what nfft = int(fs / resolution)
(reference spectra/estimators.py:144) is for fs = 250, 350, 500, 700, 1000, 2500, 5000,
10 000 Hz at the default 0.5 Hz resolution -- against the CPU oracle,
whole-array SciPy and the rocFFT route of this library (OSZ_SPEC_MIX=0), for
PSD mean / PSD segments / STFT segments, both detrends, short windows, any
overlap, chunked pushes.  Lengths the plan cannot factor take the chirp transform up to 4096
(BLUE_SIZES); even lengths whose half is beyond the LDS run as pairs of sub-transforms
(csrc/specsplit.h, SPLIT_SIZES); what is left (odd above 4096, a large prime in the way) stays on
rocFFT and is checked by value.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9
SIZES = (96, 200, 500, 600, 1000, 1500, 2000, 5000, 10000, 20000, 98, 700, 1400, 4900)
# lengths the mixed-radix plan cannot factor, up to 4096: Bluestein's chirp transform on the fft8
# transforms (spec_blue_kernel) -- odd, prime, a factor 11, one below a power of two, tiny
BLUE_SIZES = (347, 1001, 694, 2200, 3001, 4095, 2049, 129, 63, 22)
# nfft / 2 = R0 S0 beyond the LDS (specsplit_kernel): R0 = 4 (512 threads), 3, 4, 11 (a prime first
# pass), 5, 8, 9, 10
SPLIT_SIZES = (20480, 30000, 32768, 44000, 50000, 65536, 88200, 100000)


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


class rocfft_route:
    """Handles created inside take the staging-row rocFFT route."""

    def __enter__(self):
        self.old = os.environ.get("OSZ_SPEC_MIX")
        os.environ["OSZ_SPEC_MIX"] = "0"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("OSZ_SPEC_MIX", None)
        else:
            os.environ["OSZ_SPEC_MIX"] = self.old


@pytest.mark.parametrize("nfft", SIZES + BLUE_SIZES + SPLIT_SIZES)
def test_psd_all_sizes_vs_oracle(nfft):
    """psd() with fs = nfft, resolution 1: host-fed in ragged chunks (the carry
    of every push feeds the head segments) and device-resident; the rocFFT
    route of the same call agrees to 1e-12."""
    import torch
    from oracle import oracle as orc
    from openseize_amd.spectra.estimators import psd
    rng = np.random.default_rng(nfft)
    x = rng.standard_normal((5, 9 * nfft + 1234)) + 0.3
    for overlap, detrend in ((0.5, "constant"), (0.25, "linear"), (0.8, "constant")):
        rc, rf, rp = orc.psd(x, nfft, resolution=1.0, overlap=overlap, detrend=detrend)
        cnt, f, p = psd(x, fs=nfft, axis=-1, resolution=1.0, overlap=overlap, detrend=detrend)
        assert cnt == rc and np.allclose(f, rf)
        assert rel_err(p, rp) < RTOL, (nfft, overlap, detrend)
        cnt, f, p = psd(torch.from_numpy(x).cuda(), fs=nfft, axis=-1, resolution=1.0,
                        overlap=overlap, detrend=detrend)
        assert cnt == rc and rel_err(p.cpu().numpy(), rp) < RTOL
    with rocfft_route():
        cnt0, _, p0 = psd(x, fs=nfft, axis=-1, resolution=1.0)
    cnt1, _, p1 = psd(x, fs=nfft, axis=-1, resolution=1.0)
    assert cnt0 == cnt1 and rel_err(p1, p0) < 1e-12
    # pushes cut anywhere, including inside a segment and of length < nfft
    import scipy.signal as sps
    from openseize_amd import _device as dev, _lib
    w = sps.get_window("hann", nfft)
    scale = float(np.sqrt(1 / (float(nfft) * np.sum(w ** 2))))
    spec = dev.SpecStream(nfft, nfft, nfft - int(0.5 * nfft), w, scale, "constant", _lib.SPEC_PSD_MEAN, 5)   # stride = nfft - noverlap
    xd = torch.from_numpy(x).cuda()
    cuts = [0, nfft // 3, nfft - 1, nfft + 7, 3 * nfft + 11, 3 * nfft + 12, 7 * nfft, x.shape[1]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        spec.push(xd[:, a:b].contiguous())
    cnt, p = spec.mean()
    spec.close()
    rc, _, rp = orc.psd(x, nfft, resolution=1.0)
    assert cnt == rc and rel_err(p, rp) < RTOL


@pytest.mark.parametrize("nfft", SIZES + BLUE_SIZES + SPLIT_SIZES)
def test_stft_and_welch_segments_all_sizes(nfft):
    """STFT (complex segments) and the per-segment Welch producer."""
    import scipy.signal as sps
    from oracle import oracle as orc
    from openseize_amd.core import numerical as nm
    from openseize_amd.spectra.estimators import stft
    rng = np.random.default_rng(7 + nfft)
    x = rng.standard_normal((3, 6 * nfft + 321))
    f, t, X = stft(x, fs=nfft, axis=-1, resolution=1.0, overlap=0.5, boundary=True, padded=True)
    rf, rt, rX = orc.stft(x, nfft, resolution=1.0)
    assert X.shape == rX.shape and np.allclose(t, rt) and np.allclose(f, rf)
    assert np.max(np.abs(X - rX)) < RTOL * np.max(np.abs(rX))
    chunk = max(3000, nfft + 17)
    freqs, pro = nm.welch(producer(x, chunk, -1), nfft, nfft, "hann", 0.5, -1, "constant", "density")
    segs = np.stack(list(pro), -1)
    fw, pw = sps.welch(x, fs=nfft, window="hann", nperseg=nfft, noverlap=nfft // 2,
                       detrend="constant", scaling="density", axis=-1)
    assert rel_err(segs.mean(-1), pw) < RTOL


def test_short_window_padded_to_nfft():
    """periodogram / modified_dft / welch with the window shorter than nfft: the
    padding is zeros inside the kernel (reference numerical.py:805-812)."""
    import scipy.signal as sps
    from openseize_amd.core import numerical as nm
    rng = np.random.default_rng(11)
    for n, nfft in ((700, 1000), (300, 500), (5000, 10000), (2049, 6000), (1, 96), (300, 347), (2500, 3001), (7, 1001),
                    (30000, 50000), (12345, 65536)):
        x = rng.standard_normal((4, n)) + 1.0
        for detrend in ("constant", "linear"):
            for scaling in ("density", "spectrum"):
                if n == 1 and detrend == "linear":
                    continue
                f, p = nm.periodogram(x, fs=500.0, nfft=nfft, window="hamming", axis=-1,
                                      detrend=detrend, scaling=scaling)
                rf, rp = sps.periodogram(x, fs=500.0, nfft=nfft, window="hamming", axis=-1,
                                         detrend=detrend, scaling=scaling)
                assert np.allclose(f, rf)
                assert rel_err(p, rp) < RTOL, (n, nfft, detrend, scaling)


def test_lengths_the_plan_leaves_to_rocfft():
    """Above 4096 and not a product of 2, 3, 5, 7 (a prime, an odd length, twice a prime, below and
    beyond the LDS): the staging route answers, and it matches SciPy."""
    import scipy.signal as sps
    from openseize_amd.core import numerical as nm
    rng = np.random.default_rng(3)
    for nfft in (4099, 5001, 8198, 20014, 70001):
        x = rng.standard_normal((2, 4 * nfft + 50))
        freqs, pro = nm.welch(producer(x, 2 * nfft + 5, -1), float(nfft), nfft, "hann", 0.5, -1,
                              "constant", "density")
        segs = np.stack(list(pro), -1)
        fw, pw = sps.welch(x, fs=float(nfft), window="hann", nperseg=nfft, noverlap=nfft // 2,
                           detrend="constant", scaling="density", axis=-1)
        assert rel_err(segs.mean(-1), pw) < RTOL, nfft


def test_fullsize_256ch_default_resolution_5khz():
    """BASELINE chunk shape (256 ch x 2^20) at fs = 5 kHz, the reference's demo
    rate: nfft = 10 000.  Parseval on white noise and three channels against
    the oracle."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev, _lib
    C, n, nfft, fs = 256, 1 << 20, 10000, 5000.0
    x = dev.synth_normal(C, n, seed=21)
    w = sps.get_window("hann", nfft)
    scale = float(np.sqrt(1 / (fs * np.sum(w ** 2))))
    spec = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant", _lib.SPEC_PSD_MEAN, C)
    spec.push(x)
    cnt, p = spec.mean()
    spec.close()
    assert cnt == (n - nfft) // (nfft // 2) + 1
    var = p.sum(axis=1) * (fs / nfft)
    assert np.all(np.abs(var - 1.0) < 0.03), float(np.max(np.abs(var - 1.0)))
    pick = [0, 131, 255]
    rc, _, rp = orc.psd(x[pick].cpu().numpy(), fs, resolution=0.5)
    assert rc == cnt and rel_err(p[pick], rp) < RTOL


def test_which_route_a_length_takes():
    """The on-chip kernels run where the plan says so (kernel timers of the library: every
    on-chip spectra kernel reports as `spec_fused`, the staging route as `spec_rocfft`):
    a product of 2, 3, 5, 7 and a Bluestein length stay on chip, a prime above 4096 and
    any length under OSZ_SPEC_MIX=0 go through rocFFT."""
    import ctypes
    import torch
    from openseize_amd import _lib
    from openseize_amd.spectra.estimators import psd
    lib = _lib.load()

    def launches(nfft):
        x = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 6 * nfft + 5))).cuda()
        _lib.check(lib.osz_profile_reset())
        _lib.check(lib.osz_profile_enable(1))
        try:
            psd(x, fs=nfft, axis=-1, resolution=1.0)
            torch.cuda.synchronize()
        finally:
            _lib.check(lib.osz_profile_enable(0))
        out = {}
        for name in ("spec_fused", "spec_rocfft"):
            n, ms = ctypes.c_int64(), ctypes.c_double()
            _lib.check(lib.osz_profile_query(name.encode(), ctypes.byref(n), ctypes.byref(ms)))
            out[name] = n.value
        return out

    # (20 412: the longest whose half the LDS takes; 20 480, 50 000: pairs of sub-transforms)
    for nfft in (1400, 347, 3001, 4095, 20412, 20480, 50000):
        got = launches(nfft)
        assert got["spec_fused"] > 0 and got["spec_rocfft"] == 0, (nfft, got)
    got = launches(4099)
    assert got["spec_fused"] == 0 and got["spec_rocfft"] > 0, got
    with rocfft_route():
        got = launches(347)
    assert got["spec_fused"] == 0 and got["spec_rocfft"] > 0, got

"""GPU parity at the BASELINE.json configurations themselves (cfg-1 ... cfg-5):
the exact channel counts, tap counts and chunk sizes, including the ragged last
chunk of the 1e8-sample streams (1e8 = 95 * 2^20 + 385 280), checked against
the CPU oracle on the same seeded inputs.  Streams are shortened to a few whole
chunks plus that ragged one: the launch geometry of every kernel depends on
(channels, chunk length), not on how many chunks follow.

Tolerance 1e-9 of the output scale (north_star: 1e-6); chunk lengths are
compared exactly (reference core/producer.py:289-295, :331-376).
"""

from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9
RAGGED = 385280            # 1e8 - 95 * 2^20: the last chunk of cfg-2 ... cfg-5
CS = 1 << 20


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


def _synth(C, n, seed):
    """(C, n) device-resident N(0,1) keyed by (seed, channel, sample)."""
    import torch
    from openseize_amd import _device as dev
    parts = [dev.synth_normal(C, min(CS, n - lo), seed=seed, n0=lo) for lo in range(0, n, CS)]
    return parts[0] if len(parts) == 1 else torch.cat(parts, 1)


# ------------------------------------------------------------------- cfg-1
def test_cfg1_exact_host_and_resident(nm):
    """cfg-1 as BASELINE.json states it: 16 ch x 1e6 samples, firwin(256, 0.2),
    oaconvolve(producer(x, 30000, -1), h, -1, 'same'); host-fed (ndarray in,
    ndarray out) and device-resident (CUDA tensor in and out) against the direct
    convolution; the re-chunked stream has 33 x 30000 + 10000 samples."""
    import scipy.signal as sps
    import torch
    rng = np.random.default_rng(0)
    x = rng.standard_normal((16, 1_000_000))
    h = sps.firwin(256, 0.2)
    ref = np.stack([np.convolve(row, h, mode="same") for row in x])
    want_lengths = [30000] * 33 + [10000]
    assert [a.shape[-1] for a in producer(x, 30000, -1)] == want_lengths
    # host-fed
    gen = producer(partial(nm.oaconvolve, producer(x, 30000, -1), h, -1, "same"),
                   30000, -1, shape=x.shape)
    pieces = list(gen)
    assert all(isinstance(p, np.ndarray) for p in pieces)
    assert [p.shape[-1] for p in pieces] == want_lengths
    assert rel_err(np.concatenate(pieces, -1), ref) < RTOL
    # raw generator: concatenation is the whole convolution
    raw = np.concatenate(list(nm.oaconvolve(producer(x, 30000, -1), h, -1, "same")), -1)
    assert rel_err(raw, ref) < RTOL
    # device resident
    xd = torch.from_numpy(x).cuda()
    gen = producer(partial(nm.oaconvolve, producer(xd, 30000, -1), h, -1, "same"),
                   30000, -1, shape=x.shape)
    pieces = list(gen)
    assert all(p.is_cuda for p in pieces)
    assert [p.shape[-1] for p in pieces] == want_lengths
    assert rel_err(torch.cat(pieces, -1).cpu().numpy(), ref) < RTOL


# ------------------------------------------------------------------- cfg-2
def test_cfg2_128ch_1024taps_ragged(nm):
    """cfg-2 geometry: 128 channels, 1024-tap FIR, chunksize 2^20, three whole
    chunks and the ragged 385 280-sample last chunk; three channels against the
    oracle (FFT convolution: the direct form would take minutes)."""
    import scipy.signal as sps
    import torch
    C, n = 128, 3 * CS + RAGGED
    h = sps.firwin(1024, 0.2)
    x = _synth(C, n, seed=21)
    src = producer(x, CS, -1)
    assert [a.shape[-1] for a in src] == [CS] * 3 + [RAGGED]
    gen = producer(partial(nm.oaconvolve, src, h, -1, "same"), CS, -1, shape=tuple(x.shape))
    pick = [0, 63, 127]
    got, lengths = [], []
    for out in gen:
        assert out.is_cuda
        lengths.append(out.shape[-1])
        got.append(out[pick].cpu().numpy())
    assert lengths == [CS] * 3 + [RAGGED]
    xh = x[pick].cpu().numpy()
    ref = np.stack([sps.fftconvolve(row, h, mode="same") for row in xh])
    assert rel_err(np.concatenate(got, -1), ref) < RTOL
    # 'full' and 'valid' lengths at this geometry
    for mode, total in (("full", n + 1023), ("valid", n - 1023)):
        m = sum(a.shape[-1] for a in nm.oaconvolve(producer(x, CS, -1), h, -1, mode))
        assert m == total
    del x
    torch.cuda.empty_cache()


# ------------------------------------------------------------------- cfg-3
def test_cfg3_256ch_chain_ragged(nm):
    """cfg-3: 256 channels, FIR(1024) -> 6-section Butterworth band-pass
    sosfiltfilt at chunksize 2^20 over six whole chunks plus the ragged
    385 280-sample last chunk (its backward pass starts from sosfilt_zi * last
    sample, reference core/numerical.py:408-411, and the chunk before it warms up
    over a SHORT next chunk, :397-399): seven chunks, so the route under test is the
    headline's -- osz_chain_zp_step for chunks 0 .. 4, the separate kernels for the
    last two -- and the test says so."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev
    C, n = 256, 6 * CS + RAGGED
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    x = _synth(C, n, seed=22)
    fir = producer(partial(nm.oaconvolve, producer(x, CS, -1), h, -1, "same"), CS, -1,
                   shape=tuple(x.shape))
    pick = [0, 128, 255]
    got, lengths = [], []
    steps, plain = [], dev.chain_zp_step
    dev.chain_zp_step = lambda *a, **k: (steps.extend([1] * (a[2].shape[1] // CS)), plain(*a, **k))[1]
    try:
        for out in nm.sosfiltfilt(fir, sos, -1):
            lengths.append(out.shape[-1])
            got.append(out[pick].cpu().numpy())
    finally:
        dev.chain_zp_step = plain
    assert lengths == [CS] * 6 + [RAGGED]
    assert len(steps) == 5, len(steps)          # the zero-phase kernel took five chunks (one launch each at 256 channels)
    xh = x[pick].cpu().numpy()
    del x
    torch.cuda.empty_cache()
    fir_ref = np.stack([sps.fftconvolve(row, h, mode="same") for row in xh])
    ref = orc.sosfiltfilt(fir_ref, sos, CS)
    assert rel_err(np.concatenate(got, -1), ref) < RTOL


# ------------------------------------------------------------------- cfg-4
def test_cfg4_shard_32ch_ragged(nm):
    """cfg-4's per-GPU shard (256 channels over 8 GPUs = 32 channels) with a
    ragged stream end: psd(fs=4096, resolution=1.0, hann, 50 %)."""
    from oracle import oracle as orc
    from openseize_amd.spectra.estimators import psd
    C, n = 32, 2 * CS + RAGGED
    x = _synth(C, n, seed=23)
    cnt, freqs, p = psd(x, fs=4096, axis=-1, resolution=1.0, window="hann", overlap=0.5,
                        detrend="constant", scaling="density")
    p = p.cpu().numpy() if hasattr(p, "cpu") else np.asarray(p)
    pick = [0, 17, 31]
    rc, rf, rp = orc.psd(x[pick].cpu().numpy(), 4096, resolution=1.0)
    assert cnt == rc == (n - 4096) // 2048 + 1
    assert np.array_equal(freqs, rf)
    assert rel_err(p[pick], rp) < RTOL


# ------------------------------------------------------------------- cfg-5
def test_cfg5_shard_128ch_ragged(nm):
    """cfg-5's per-GPU shard (1024 channels over 8 GPUs = 128 channels):
    downsample(M=5, fs=20480, chunksize=2^20) -> stft(fs=4096, resolution=1.0,
    boundary, padded) over two whole chunks and the ragged last chunk."""
    from oracle import oracle as orc
    from openseize_amd.resampling.resampling import downsample
    from openseize_amd.spectra.estimators import stft
    C, n = 128, 2 * CS + RAGGED
    x = _synth(C, n, seed=24)
    pick = [0, 64, 127]
    xh = x[pick].cpu().numpy()
    y = downsample(producer(x, CS, -1), M=5, fs=20480, chunksize=CS, axis=-1)
    assert y.shape[-1] == -(-n // 5)
    f, t, pro = stft(y, fs=4096, axis=-1, resolution=1.0, window="hann", overlap=0.5,
                     detrend="constant", scaling="density", boundary=True, padded=True,
                     asarray=False)
    yh = orc.polyphase_resample(xh, 1, 5, orc.resample_filter(1, 5, 20480))
    rf, rt, rX = orc.stft(yh, 4096, resolution=1.0)
    assert np.array_equal(f, rf) and np.allclose(t, rt, rtol=0, atol=1e-12)
    scale = np.max(np.abs(rX))
    nseg = 0
    for k, seg in enumerate(pro):
        seg = seg[pick].cpu().numpy() if hasattr(seg, "cpu") else np.asarray(seg)[pick]
        assert np.max(np.abs(seg - rX[..., k])) < RTOL * scale
        nseg += 1
    assert nseg == rX.shape[-1]


# --------------------------------------------------- in-place / state safety
def test_sos_forward_in_place_and_state_pingpong(nm):
    """osz_sos_forward with y aliasing x (allowed by include/osz_hip.h) must not
    cut the pass into time segments (their pre-roll would read samples another
    workgroup is writing); and the carried state after whole-tile chunks (the
    time-split launch) equals the oracle's at every chunk boundary."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    C, n = 16, 1 << 18          # few channels: the launch is cut into segments
    x = dev.synth_normal(C, n, seed=31)
    xh = x.cpu().numpy()
    ref, zref = orc.sosfilt(xh, sos, n)
    s = dev.SosStream(sos, C)
    out = s.forward(x)                       # out of place (split launch)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    zf_split = s.get_state()
    assert np.max(np.abs(zf_split - zref)) < 1e-9 * max(np.max(np.abs(zref)), 1.0)
    s.set_state(None)
    buf = x.clone()
    s.forward(buf, out=buf)                  # in place
    assert rel_err(buf.cpu().numpy(), ref) < RTOL
    assert np.max(np.abs(s.get_state() - zf_split)) < 1e-9 * max(np.max(np.abs(zf_split)), 1.0)
    # state carried across four whole-tile chunks == one pass
    s.set_state(None)
    parts = [s.forward(x[:, k * (n // 4):(k + 1) * (n // 4)]) for k in range(4)]
    assert rel_err(torch.cat(parts, 1).cpu().numpy(), ref) < RTOL
    s.close()

"""The geometry of the reference's own tutorials (the only numbers it
publishes): 4 channels x 18 875 000 samples at 5 kHz in chunks of 10e6 / 5e6 -- few channels,
very long chunks, where only parallelism in time fills the chip.  The class API on
device-resident and on host data against the CPU oracle with the same chunking
(docs/tutorials/filtering.ipynb:1683-3106, resampling.ipynb:650 of the reference)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9
C, N, FS = 4, 18_875_000, 5000


@pytest.fixture(scope="module")
def data():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from openseize_amd import _lib
    _lib.load()
    from openseize_amd import _device as dev
    xd = dev.synth_normal(C, N, seed=21)
    return xd, xd.cpu().numpy()


def collect(result):
    import torch
    parts = [c for c in result]
    if torch.is_tensor(parts[0]):
        return torch.cat(parts, -1).cpu().numpy()
    return np.concatenate(parts, -1)


def rel(got, want):
    return float(np.max(np.abs(got - want)) / np.max(np.abs(want)))


def test_kaiser_fir_same(data):
    from oracle import oracle as orc
    from openseize_amd import producer
    from openseize_amd.filtering import fir
    xd, xh = data
    kaiser = fir.Kaiser(fpass=200, fstop=400, gpass=0.5, gstop=40, fs=FS)
    assert len(kaiser.coeffs) == 57                      # "this will make a 57 tap filter"
    want = orc.convolve_direct(xh, kaiser.coeffs, "same")
    for src in (xd, xh):
        pro = kaiser(producer(src, chunksize=1e6, axis=-1), chunksize=10e6, axis=-1, mode="same")
        lens = [c.shape[-1] for c in pro]
        assert lens == [10_000_000, 8_875_000]
        assert rel(collect(pro), want) < RTOL


@pytest.mark.parametrize("dephase", [True, False])
def test_cheby1_sos(data, dephase):
    from oracle import oracle as orc
    from openseize_amd import producer
    from openseize_amd.filtering import iir
    xd, xh = data
    cheb1 = iir.Cheby1(fpass=200, fstop=400, gpass=0.5, gstop=40, fs=FS, fmt="sos")
    assert cheb1.coeffs.shape == (3, 6)
    cs = 10_000_000
    want = orc.sosfiltfilt(xh, cheb1.coeffs, cs) if dephase else orc.sosfilt(xh, cheb1.coeffs, cs)[0]
    for src in (xd, xh):
        pro = cheb1(producer(src, chunksize=1e6, axis=-1), chunksize=10e6, axis=-1, dephase=dephase)
        assert rel(collect(pro), want) < RTOL


def test_notch_filtfilt(data):
    from oracle import oracle as orc
    from openseize_amd import producer
    from openseize_amd.filtering import iir
    xd, xh = data
    notch = iir.Notch(60, width=6, fs=FS)
    want = orc.filtfilt(xh, notch.coeffs, 5_000_000)
    for src in (xd, xh):
        pro = notch(producer(src, chunksize=1e6, axis=-1), chunksize=5e6, axis=-1, dephase=True)
        assert rel(collect(pro), want) < 1e-7            # a ba filter runs as a biquad cascade here


def test_downsample_25(data):
    from oracle import oracle as orc
    from openseize_amd import producer
    from openseize_amd.resampling.resampling import downsample
    xd, xh = data
    h = orc.resample_filter(1, 25, FS)
    want = orc.polyphase_resample(xh, 1, 25, h)
    assert want.shape == (C, N // 25)
    for src in (xd, xh):
        pro = downsample(producer(src, chunksize=1e6, axis=-1), M=25, fs=FS, chunksize=5e6)
        assert rel(collect(pro), want) < RTOL

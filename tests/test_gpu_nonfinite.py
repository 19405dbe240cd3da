"""GPU parity on non-finite input: a NaN (or Inf) never leaves the state of an IIR
section, so the reference (scipy.signal.sosfilt behind core/numerical.py:334, :399-410)
is NaN from that sample to the end of the STREAM in a forward pass and, in
sosfiltfilt, over the whole chunk that holds it and the chunk before it (whose
backward warm-up runs over it).  The kernels cut a pass into time segments that start
from zero states; `sos_fwd_seal` / `probe` (csrc/sos_tile.h) make the segments agree
with the serial recurrence.  Pinned by tests/golden/g15_nonfinite.npz (reference
outputs) and, for other geometries, by the oracle (itself pinned by g15)."""

from functools import partial

import numpy as np
import pytest

from conftest import nonfinite_inputs, nonfinite_runs

pytestmark = pytest.mark.gpu

RTOL = 1e-9


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


def same_where_finite(got, want, tol=RTOL):
    ok = np.isfinite(want)
    assert np.array_equal(ok, np.isfinite(got))
    assert np.array_equal(np.isnan(want), np.isnan(got))
    if ok.any():
        assert np.max(np.abs(got[ok] - want[ok])) < tol * np.max(np.abs(want[ok]))


@pytest.mark.parametrize("tag", ["nan", "inf"])
def test_golden_nonfinite_reach(nm, golden, tag):
    """Three channels (every pass cut into time segments, main + ragged remainder
    launches), chunks of 200 000: the reference's own masks and values."""
    from oracle import oracle as orc
    g = golden("g15_nonfinite.npz")
    sos, cs = g["sos"], int(g["chunksize"])
    data = nonfinite_inputs()[0 if tag == "nan" else 1]
    y = np.concatenate(list(nm.sosfilt(producer(data, cs, -1), sos, -1)), -1)
    yy = np.concatenate(list(nm.sosfiltfilt(producer(data, cs, -1), sos, -1)), -1)
    for name, arr in (("sosfilt", y), ("sosfiltfilt", yy)):
        assert np.array_equal(nonfinite_runs(arr), g[f"{name}_{tag}_runs"]), name
        dec, want = arr[:, ::97], g[f"{name}_{tag}_dec"]
        ok = np.isfinite(want)
        assert np.array_equal(ok, np.isfinite(dec))
        assert np.max(np.abs(dec[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))
    if tag == "nan":
        same_where_finite(y, orc.sosfilt(data, sos, cs)[0])
        same_where_finite(yy, orc.sosfiltfilt(data, sos, cs))


def test_nonfinite_reach_many_channels_whole_tiles(nm):
    """128 channels in chunks of 2^17 (whole tiles: sosfiltfilt takes the dual launch,
    forward of chunk k and backward of chunk k - 2 in time segments), device resident:
    a NaN in the FIRST tile of a chunk (behind every later segment's pre-roll), one in
    the last chunk only, a NaN tail, and the carried state across chunks."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    cs, nchunks, C = 1 << 17, 5, 128
    rng = np.random.default_rng(77)
    x = rng.standard_normal((C, cs * nchunks))
    x[5, cs + 17] = np.nan                      # first tile of chunk 1
    x[40, 4 * cs + 3] = np.nan                  # last chunk only
    x[77, 3 * cs + 99999:] = np.nan             # tail
    x[100, 2 * cs - 1] = np.nan                 # last sample of chunk 1
    pick = [0, 5, 40, 77, 100, 127]
    xd = torch.from_numpy(x).cuda()
    y = torch.cat(list(nm.sosfilt(producer(xd, cs, -1), sos, -1)), -1)[pick].cpu().numpy()
    same_where_finite(y, orc.sosfilt(x[pick], sos, cs)[0])
    yy = torch.cat(list(nm.sosfiltfilt(producer(xd, cs, -1), sos, -1)), -1)[pick].cpu().numpy()
    want = orc.sosfiltfilt(x[pick], sos, cs)
    same_where_finite(yy, want)
    # what the masks are, spelled out: channel 5 -> chunks 0.. all NaN; channel 40 ->
    # chunks 3, 4; channel 77 -> chunks 2, 3, 4; channel 100 -> everything
    nanchunks = [[bool(np.isnan(yy[i, k * cs:(k + 1) * cs]).all()) for k in range(nchunks)]
                 for i in range(len(pick))]
    assert nanchunks == [[False] * 5, [True] * 5, [False, False, False, True, True],
                         [False, False, True, True, True], [True] * 5, [False] * 5]


@pytest.mark.parametrize("zero_phase", [True, False])
def test_nonfinite_reach_fused_chain(nm, zero_phase):
    """FIR -> sosfiltfilt on the zero-phase kernel (osz_chain_zp_step + osz_chain_zp_seal)
    and, with OSZ_CHAIN_ZP=0, on the two-kernel step (several runs per channel that
    start from nothing; backward pass beside it): a NaN in the input reaches the
    end of the stream in the forward half, so from the chunk before the NaN on every
    output chunk of that channel is NaN; the other channels equal the oracle.  How far
    a NaN spreads inside the FIR: on the zero-phase route as in the reference -- its
    overlap-add loses the sample's whole segment of 261 121 input samples at 1024 taps,
    i.e. the chain from up to two of these chunks before the sample (round 5:
    osz_chain_zp_reach; the oracle runs the reference's segments) --, on the two-kernel
    step by this library's own 4096-point blocks (an FFT block is NaN as a whole: a
    documented divergence), which is why there the chunks BEFORE are only required to
    be finite up to one chunk ahead of the NaN."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    taps = sps.firwin(1024, 0.2)
    C, cs, nchunks = 64, 6144 * 24, 7
    x = dev.synth_normal(C, cs * nchunks, seed=5)
    x[9, 3 * cs + 8000] = float("nan")         # inside the first block pair of chunk 3
    x[30, 5 * cs + cs // 2:] = float("nan")
    # 400 samples into input chunk 4: the FIR output turns NaN 112 samples before the END of
    # output chunk 3 ('same' cuts 511) -- inside the part of that chunk a fused step has not
    # got yet when it probes the chunk for the backward pass of chunk 2
    x[50, 4 * cs + 400] = float("nan")
    src = producer(x, cs, -1)
    fir = producer(partial(nm.oaconvolve, src, taps, -1, "same"), cs, -1, shape=src.shape)
    import os
    steps, plain_step, plain_zp = [], dev.chain_step, dev.chain_zp_step
    dev.chain_step = lambda *a, **k: (steps.append("step"), plain_step(*a, **k))[1]
    dev.chain_zp_step = lambda *a, **k: (steps.append("zp"), plain_zp(*a, **k))[1]
    if not zero_phase:
        os.environ["OSZ_CHAIN_ZP"] = "0"
    try:
        got = torch.cat(list(nm.sosfiltfilt(fir, sos, -1)), -1).cpu().numpy()
    finally:
        dev.chain_step, dev.chain_zp_step = plain_step, plain_zp
        os.environ.pop("OSZ_CHAIN_ZP", None)
    assert steps and set(steps) == {"zp" if zero_phase else "step"}, "the fused path did not run"
    nan_chunk = np.array([[bool(np.isnan(got[c, k * cs:(k + 1) * cs]).all()) for k in range(nchunks)]
                          for c in range(C)])
    some_nan = np.array([[bool(np.isnan(got[c, k * cs:(k + 1) * cs]).any()) for k in range(nchunks)]
                         for c in range(C)])
    assert np.array_equal(nan_chunk, some_nan)            # a chunk is NaN as a whole or not at all
    if zero_phase:
        xb = x[[9, 30, 50]].cpu().numpy()
        with np.errstate(invalid="ignore"):
            wb = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(xb, taps, "same"), -1), sos, cs)
        ref_lost = [[bool(np.isnan(wb[i, k * cs:(k + 1) * cs]).all()) for k in range(nchunks)] for i in range(3)]
        assert nan_chunk[[9, 30, 50]].tolist() == ref_lost
        assert ref_lost[0] == [True] * 7                  # (the segment of sample 3 cs + 8000 starts in chunk 1)
        ok = np.isfinite(wb)
        assert np.array_equal(ok, np.isfinite(got[[9, 30, 50]]))
        assert np.max(np.abs(got[[9, 30, 50]][ok] - wb[ok])) < RTOL * np.max(np.abs(wb[ok]))
    else:
        assert nan_chunk[9].tolist() == [False, False, True, True, True, True, True]
        assert nan_chunk[30].tolist() == [False, False, False, False, True, True, True]
        assert nan_chunk[50].tolist() == [False, False, True, True, True, True, True]
    clean = [c for c in range(C) if c not in (9, 30, 50)]
    assert not nan_chunk[clean].any()
    pick = [0, 31, 63]
    xh = x[pick].cpu().numpy()
    want = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(xh, taps, "same"), -1), sos, cs)
    assert np.max(np.abs(got[pick] - want)) < RTOL * np.max(np.abs(want))
    # and (two-kernel step) the finite chunks of the two NaN channels equal the oracle on their finite prefix
    for c, upto in (() if zero_phase else ((9, 2), (30, 4))):
        # chunks < upto see the forward output up to chunk `upto`, i.e. the input up to
        # 511 samples into the next chunk: no NaN there
        xc = x[c:c + 1, :(upto + 1) * cs + 600].cpu().numpy()
        assert np.isfinite(xc).all()
        w = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(xc, taps, "same"), -1), sos, cs)
        assert np.max(np.abs(got[c, :upto * cs] - w[0, :upto * cs])) < RTOL * np.max(np.abs(w))


def test_nonfinite_reach_at_the_headline_geometry(nm):
    """cfg-3's own geometry (256 channels x 2^20-sample chunks through osz_chain_step -- the
    two-kernel step, asked for with OSZ_CHAIN_ZP=0 --, two fused runs and three backward segments
    per channel): one NaN in one channel of chunk 2
    leaves every other channel bit-identical to the clean run, and that channel NaN from
    chunk 1 on (the chunk before the NaN: its backward warm-up runs over it)."""
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    taps = sps.firwin(1024, 0.2)
    C, cs, nchunks = 256, 1 << 20, 5
    ring = [dev.synth_normal(C, cs, seed=8, n0=k * cs) for k in range(nchunks)]

    def run(poison):
        if poison:
            saved = float(ring[2][77, 400000])
            ring[2][77, 400000] = float("nan")
        def chunks():
            yield from ring

        src = producer(chunks, cs, -1, shape=(C, cs * nchunks))
        fir = producer(partial(nm.oaconvolve, src, taps, -1, "same"), cs, -1, shape=src.shape)
        sums, nanmask = [], []
        plain_step = dev.chain_step
        dev.chain_step = lambda *a, **k: (steps.append(1), plain_step(*a, **k))[1]
        import os
        os.environ["OSZ_CHAIN_ZP"] = "0"          # (the two-kernel step is what this test is about)
        try:
            outs = list(nm.sosfiltfilt(fir, sos, -1))
        finally:
            del os.environ["OSZ_CHAIN_ZP"]
        for out in outs:
            bad = torch.isnan(out).any(dim=1)
            allbad = torch.isnan(out).all(dim=1)
            assert torch.equal(bad, allbad)            # a channel's chunk is NaN as a whole or not at all
            nanmask.append(bad.cpu().numpy())
            sums.append(torch.where(bad[:, None], torch.zeros_like(out), out).view(torch.int64).sum(dim=1).cpu().numpy())
            del out
        dev.chain_step = plain_step
        if poison:
            ring[2][77, 400000] = saved
        return np.array(nanmask), np.array(sums)

    steps = []
    clean_mask, clean_sums = run(False)
    assert len(steps) == nchunks - 2, "the fused step did not run"
    assert not clean_mask.any()
    mask, sums = run(True)
    want = np.zeros_like(mask)
    want[1:, 77] = True
    assert np.array_equal(mask, want)
    keep = np.ones(C, dtype=bool)
    keep[77] = False
    assert np.array_equal(sums[:, keep], clean_sums[:, keep])      # bit patterns, channel by channel
    assert np.array_equal(sums[0], clean_sums[0])


def test_nan_reach_is_decided_by_the_sample_not_by_the_block(nm):
    """sosfiltfilt of a long resident stream of few channels runs several chunks per zero-phase
    launch (round 5), so the kernel's blocks straddle chunk boundaries; the reference's NaN reach
    is per CHUNK (core/numerical.py:397-411: a chunk is NaN as a whole when the forward stream is
    NaN in it or in the chunk after it).  A NaN tail that begins 2579 samples into chunk 3 lies in
    a block that begins in chunk 2: the kernel records the SAMPLE (zp_exact_nanpos, chain_zp.h),
    chunk 1 stays finite as in the reference (found by tests/fuzz_gpu.py, seed 505)."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    C, cs, n = 5, 90112, 689048
    rng = np.random.default_rng(505)
    x = rng.standard_normal((C, n))
    x[3, 3 * cs + 2579:] = np.nan
    x[1, 5 * cs + 7] = np.inf
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    xd = torch.from_numpy(x).cuda()
    for axis, data in ((-1, xd), (0, xd.t().contiguous())):
        got = torch.cat(list(nm.sosfiltfilt(producer(data, cs, axis), sos, axis)), axis).cpu().numpy()
        got = got if axis == -1 else got.T
        want = orc.sosfiltfilt(x, sos, cs)
        assert np.array_equal(np.isfinite(got), np.isfinite(want))
        ok = np.isfinite(want)
        assert ok[3, :2 * cs].all() and not ok[3, 2 * cs:].any()
        assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))


@pytest.mark.parametrize("fed", ["resident", "host"])
@pytest.mark.parametrize("taps_n, cs", [(256, 65664), (513, 65536), (1024, 131072)])
def test_fir_chain_nan_reach_is_the_references(nm, fed, taps_n, cs):
    """FIR -> sosfiltfilt on the one-kernel route: the reference's FIR turns a non-finite input
    sample into a whole non-finite SEGMENT of its overlap-add (nfft - wlen + 1 input samples,
    core/numerical.py:202-217, :258-283 -- 65 281 at 256 taps, 261 632 at 513), so its chain is
    lost from the segment's start: up to a segment BEFORE the sample.  The kernels record the exact
    sample, the seal counts from the segment's start (osz_chain_zp_reach) and the generator holds
    chunks back for as long as that reach spans -- three more steps at 513 taps in chunks of
    65 536 --; a channel that first goes bad in the stream's last two chunks has its forward
    stream recomputed from a cleaned copy.  Masks equal to the oracle's (which runs the
    reference's segments), values to 1e-9 where finite: a sample mid-stream, samples on either
    side of a segment boundary, in chunk 0, in the last two chunks, an EDF-style tail, an Inf."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    nchunks = 9
    total = cs * (nchunks - 1) + 4321
    taps = sps.firwin(taps_n, 0.3)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    step = orc.oa_plan(total, taps_n)[1]
    where = [3 * cs + 777,                         # mid-stream
             (4 * cs // step + 1) * step - 3,      # the last samples of a segment ...
             (4 * cs // step + 1) * step + 2,      # ... and the first of the next
             5,                                    # chunk 0
             (nchunks - 2) * cs + 9 * cs // 10,    # chunk n-2, behind the head the zero-phase steps see
             (nchunks - 1) * cs + 17,              # the last chunk, first block
             (nchunks - 1) * cs + 4000]            # the last chunk (an EDF-style tail from here on)
    C = len(where) + 2
    rng = np.random.default_rng(77)
    x = rng.standard_normal((C, total))
    for c, at in enumerate(where):
        x[c, at] = np.nan if c != 2 else np.inf
    x[len(where) - 1, where[-1]:] = np.nan
    x[C - 2, (nchunks - 2) * cs + 100] = np.nan    # chunk n-2, inside that head
    src_data = x if fed == "host" else torch.from_numpy(x).cuda()
    steps, plain = [], dev_module().chain_zp_step
    dev_module().chain_zp_step = lambda *a, **k: (steps.append(a[2].shape[1]), plain(*a, **k))[1]
    try:
        src = producer(src_data, cs, -1)
        fir = producer(partial(nm.oaconvolve, src, taps, -1, "same"), cs, -1, shape=src.shape)
        pieces = list(nm.sosfiltfilt(fir, sos, -1))
    finally:
        dev_module().chain_zp_step = plain
    assert sum(steps) == (nchunks - 2) * cs, steps                       # the one-kernel route
    got = np.concatenate([p if isinstance(p, np.ndarray) else p.cpu().numpy() for p in pieces], -1)
    want = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(x, taps, "same"), -1), sos, cs)
    ok = np.isfinite(want)
    bad_chunks = lambda a: [[bool(~np.isfinite(a[c, k * cs:(k + 1) * cs]).all()) for k in range(nchunks)] for c in range(C)]
    assert bad_chunks(got) == bad_chunks(want)
    if (taps_n, cs) == (256, 65664):          # the REFERENCE's own record of these placements (g19)
        from conftest import load_golden
        g = load_golden("g19_fir_chain_nonfinite.npz")
        assert list(g["where"]) == where and np.array_equal(np.array(bad_chunks(got)), g["lost_chunks"])
    assert np.array_equal(ok, np.isfinite(got))
    assert ok[C - 1].all() and not ok[3].any()            # the clean channel; a NaN in chunk 0 loses everything
    assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))


def dev_module():
    from openseize_amd import _device
    return _device


def _fir_case(seed, C, n, where):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, n))
    for c, at, kind in where:
        if kind == "tail":
            x[c, at:] = np.nan
        elif kind == "all":
            x[c, :] = np.nan
        else:
            x[c, at] = np.inf if kind == "inf" else np.nan
    return x


@pytest.mark.parametrize("fed", ["resident", "host"])
@pytest.mark.parametrize("taps_n, n, cs", [(256, 300_000, 30_000), (65, 100_001, 20_000), (101, 5000, 1000),
                                            (1024, 1_200_000, 1 << 18), (3001, 700_000, 100_000),
                                            (3, 15_967, 1107), (1024, 5166, 4865)])
def test_oaconvolve_nan_reach_is_the_references(nm, fed, taps_n, n, cs):
    """``oaconvolve`` alone: a non-finite input sample costs the reference the whole output of
    the SEGMENT it sits in (core/numerical.py:202-217, 258-283) -- and so it does here: the same
    samples non-finite, the others as the oracle has them, in every mode, resident and host-fed,
    for samples at the stream's ends, runs to the end, a channel without a finite sample and a
    block of this library's transforms that straddles a segment boundary or begins whole segments
    before the sample (three taps: segments of 1022 samples, blocks of 7000)."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    h = sps.firwin(taps_n, 0.2)
    _, step = orc.oa_plan(n, taps_n, 32)
    C = 5
    where = [(0, 0, "nan"), (1, n - 1, "inf"), (2, min(step, n) - 1, "nan"), (2, min(2 * step + 7, n - 2), "nan"),
             (3, (n * 2) // 3, "tail")]
    if n > 3 * step:
        where.append((1, 3 * step, "nan"))           # the first sample of a segment
    x = _fir_case(taps_n + n, C, n, where)
    x[4, :] = np.nan if taps_n == 65 else x[4, :]    # (a dead channel in one of the cases)
    for mode in ("same", "full", "valid"):
        want = np.concatenate(orc.oaconvolve(x, h, mode), -1)
        src = torch.from_numpy(x).cuda() if fed == "resident" else x
        got = [p.cpu().numpy() if torch.is_tensor(p) else p
               for p in nm.oaconvolve(producer(src, cs, -1), h, -1, mode)]
        got = np.concatenate(got, -1)
        assert got.shape == want.shape, (mode, got.shape, want.shape)
        ok = np.isfinite(want)
        assert np.array_equal(ok, np.isfinite(got)), (mode, np.argwhere(ok != np.isfinite(got))[:5])
        scale = np.max(np.abs(want[ok]))
        assert np.max(np.abs(got[ok] - want[ok])) < RTOL * scale, mode


def test_oaconvolve_reach_holds_nothing_back_for_ever(nm):
    """A clean stream comes out piece for piece as before, two pieces late at most, and the
    class API (FIR.__call__) inherits the reach."""
    import torch
    from oracle import oracle as orc
    from openseize_amd.filtering.fir import Kaiser
    x = _fir_case(5, 3, 400_000, [(1, 123_456, "nan")])
    filt = Kaiser(fpass=500, fstop=600, fs=5000)
    y = filt(x, chunksize=50_000, axis=-1, mode="same")
    want = np.concatenate(orc.oaconvolve(x, filt.coeffs, "same"), -1)
    assert np.array_equal(np.isfinite(want), np.isfinite(y))
    ok = np.isfinite(want)
    assert np.max(np.abs(y[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))
    # pieces arrive in order, chunk-sized, while the source is still being read
    xs = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 600_000))).cuda()
    pulled, lens = [], []

    def source():
        for k in range(0, 600_000, 100_000):
            pulled.append(k)
            yield xs[:, k:k + 100_000].clone()       # (buffers of their own: adjacent views would be joined per push)

    taps = filt.coeffs
    src = producer(source, 100_000, -1, shape=(2, 600_000))
    first_after = None
    for piece in nm.oaconvolve(src, taps, -1, "same"):
        if first_after is None:
            first_after = len(pulled)
        lens.append(piece.shape[-1])
    assert sum(lens) == 600_000 and first_after <= 5, (lens, first_after)


@pytest.mark.parametrize("fed", ["resident", "host"])
@pytest.mark.parametrize("taps_n, cs, nchunks", [(256, 65664, 7), (1024, 1 << 17, 9), (65, 70000, 5)])
def test_fir_sosfilt_chain_nan_reach_is_the_references(nm, fed, taps_n, cs, nchunks):
    """``sosfilt(oaconvolve(x, 'same'))`` on the fused launch (osz_chain_forward): a non-finite
    input sample costs the reference the channel from the START of the FIR's segment that holds it
    (core/numerical.py:202-217 into :334) -- and so it does here, resident and host-fed: the same
    samples non-finite as in the oracle's restatement, the others equal; a clean stream still takes
    one fused launch per chunk."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev
    sos = sps.butter(4, [0.05, 0.3], "bandpass", output="sos")
    h = sps.firwin(taps_n, 0.2)
    total = cs * (nchunks - 1) + cs // 3 + 17
    _, step = orc.oa_plan(total, taps_n, 32)
    C = 6
    where = [(0, 3 * cs + 8000, "nan"), (1, total - 5, "inf"), (2, min(2 * step + 7, total - 9), "nan"),
             (3, (total * 2) // 3, "tail"), (4, 0, "nan")]
    x = _fir_case(taps_n + cs, C, total, where)
    launches, plain = [], dev.chain_forward
    dev.chain_forward = lambda *a, **k: (launches.append(1), plain(*a, **k))[1]
    try:
        def through(data):
            src = producer(data, cs, -1)
            fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
            return np.concatenate([p.cpu().numpy() if torch.is_tensor(p) else p for p in nm.sosfilt(fir, sos, -1)], -1)

        got = through(torch.from_numpy(x).cuda() if fed == "resident" else x)
        assert launches, "the fused launch did not run"
        with np.errstate(invalid="ignore"):
            want = orc.sosfilt(np.concatenate(orc.oaconvolve(x, h, "same"), -1), sos, cs)[0]
        ok = np.isfinite(want)
        assert got.shape == want.shape
        assert np.array_equal(ok, np.isfinite(got)), np.argwhere(ok != np.isfinite(got))[:5]
        assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))
        # a clean stream: every chunk but the first on the fused launch, nothing recomputed
        del launches[:]
        xc = np.random.default_rng(1).standard_normal((C, total))
        got = through(torch.from_numpy(xc).cuda() if fed == "resident" else xc)
        assert len(launches) == nchunks - 1
        want = orc.sosfilt(np.concatenate(orc.oaconvolve(xc, h, "same"), -1), sos, cs)[0]
        assert np.max(np.abs(got - want)) < RTOL * np.max(np.abs(want))
    finally:
        dev.chain_forward = plain


@pytest.mark.parametrize("LM", [(1, 5), (3, 2), (2, 1), (1, 25), (2, 7), (1, 2)])
def test_resample_nan_reach_is_the_references(nm, golden, LM):
    """The polyphase resampler (core/numerical.py:523-632: scipy.signal.resample_poly chunk by chunk)
    is a sum in time: a non-finite sample is lost to the outputs whose taps touch it -- the taps of
    SciPy's PADDED window, 0 x NaN being NaN (tests/golden/g20: the reference's own masks).  The
    generator hands the kernels that window (numerical._resample_padded, osz_poly_create_centred),
    and they multiply every tap of it and no other: the same outputs lost, sample for sample, as
    the reference (g20), SciPy and the oracle, resident and host-fed, whatever the chunking."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd.resampling.resampling import resample
    L, M = LM
    # --- the reference's own masks (g20's input: make_golden.g20_input)
    g = golden("g20_resample_nonfinite.npz")
    if f"nout_L{L}_M{M}" in g:
        n = int(g["n"])
        x = np.random.default_rng(2020).standard_normal((4, n))
        x[0, 0] = np.nan
        x[0, 10_000] = np.nan
        x[1, n - 1] = np.inf
        x[1, 13_333] = np.nan
        x[2, 17_000:] = np.nan
        nout = int(g[f"nout_L{L}_M{M}"])
        for cs in (5_000, 7_321):
            lost = np.unpackbits(g[f"lost_L{L}_M{M}_cs{cs}"], axis=-1)[:, :nout].astype(bool)
            for data in (x, torch.from_numpy(x).cuda()):
                got = resample(data, L, M, 5000, chunksize=cs, axis=-1)
                got = got.cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
                assert np.array_equal(~np.isfinite(got), lost), (LM, cs, np.argwhere(~np.isfinite(got) != lost)[:4])
    # --- a longer stream against SciPy and the oracle
    n = 200_003
    x = _fir_case(L * 100 + M, 4, n, [(0, 0, "nan"), (0, 77_777, "inf"), (1, n - 1, "nan"), (2, 150_000, "tail")])
    h = orc.resample_filter(L, M, 5000)
    with np.errstate(invalid="ignore"):
        want = sps.resample_poly(x, L, M, axis=-1, window=h)
        assert np.array_equal(np.isfinite(want), np.isfinite(orc.polyphase_resample(x, L, M, h)))
    ok = np.isfinite(want)
    assert ok[3].all() and not ok[0].all()
    for data in (x, torch.from_numpy(x).cuda()):
        for cs in (30_000, 70_001):
            got = resample(data, L, M, 5000, chunksize=cs, axis=-1)
            got = got.cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
            assert got.shape == want.shape
            assert np.array_equal(ok, np.isfinite(got)), (LM, cs, np.argwhere(ok != np.isfinite(got))[:4])
            assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))


@pytest.mark.parametrize("fed", ["resident", "host"])
@pytest.mark.parametrize("kind", ["five chunks", "nine sections"])
def test_fir_sosfiltfilt_off_the_one_kernel_route_keeps_the_references_reach(nm, fed, kind):
    """FIR -> sosfiltfilt where the one-kernel route does not apply (a stream of five chunks; a
    cascade it refuses): the two generators apart since round 5 -- the reference FIR's segments
    and the cascade's own reach, sample for sample the oracle's masks; the two-kernel step
    (osz_chain_step, whose FIR loses only its own 4096-point blocks) runs when asked for only."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    from openseize_amd import _device as dev
    h = sps.firwin(256, 0.2)
    if kind == "five chunks":
        sos, cs, nchunks = sps.butter(6, [0.05, 0.3], "bandpass", output="sos"), 70_000, 5
    else:
        sos, cs, nchunks = sps.cheby1(9, 0.5, [0.01, 0.1], "bandpass", output="sos"), 70_000, 8
    total = cs * (nchunks - 1) + 12_345
    x = _fir_case(11, 4, total, [(0, 3 * cs + 100, "nan"), (1, total - 3, "inf"), (2, 150_000, "nan")])
    ran, plain_step, plain_zp = [], dev.chain_step, dev.chain_zp_step
    dev.chain_step = lambda *a, **k: (ran.append("step"), plain_step(*a, **k))[1]
    dev.chain_zp_step = lambda *a, **k: (ran.append("zp"), plain_zp(*a, **k))[1]
    try:
        src = producer(torch.from_numpy(x).cuda() if fed == "resident" else x, cs, -1)
        fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, -1, shape=src.shape)
        got = np.concatenate([p.cpu().numpy() if torch.is_tensor(p) else p for p in nm.sosfiltfilt(fir, sos, -1)], -1)
    finally:
        dev.chain_step, dev.chain_zp_step = plain_step, plain_zp
    assert "step" not in ran, ran
    with np.errstate(invalid="ignore"):
        want = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(x, h, "same"), -1), sos, cs)
    ok = np.isfinite(want)
    assert np.array_equal(ok, np.isfinite(got)), np.argwhere(ok != np.isfinite(got))[:4]
    assert ok[3].all() and not ok[0].all()
    tol = RTOL if kind == "five chunks" else 1e-7        # (nine narrow-band sections: the cascade's own conditioning)
    assert np.max(np.abs(got[ok] - want[ok])) < tol * np.max(np.abs(want[ok]))


@pytest.mark.parametrize("nfft", [4096, 1024, 1000, 50000, 347, 4099])
def test_spectra_nan_reach_is_the_segments(nm, nfft):
    """Welch / STFT (core/numerical.py:635-1087): a non-finite sample costs the reference the
    segments that hold it (detrend, window, rfft of a segment) -- the whole PSD of the channel in the
    average, the segment's column in the STFT -- on every transform route here (cube, fft8, mixed
    radix, pairs of sub-transforms, chirp, rocFFT staging): the same entries non-finite as the oracle's."""
    import torch
    from oracle import oracle as orc
    from openseize_amd.spectra.estimators import psd, stft
    rng = np.random.default_rng(nfft)
    n = 7 * nfft + 123
    x = rng.standard_normal((4, n))
    x[0, 3 * nfft + 5] = np.nan                  # in two overlapping segments
    x[1, n - 1] = np.inf                         # behind the last whole segment of the PSD
    x[2, 0] = np.nan
    for data in (x, torch.from_numpy(x).cuda()):
        cnt, f, p = psd(data, fs=nfft, axis=-1, resolution=1.0)
        p = p.cpu().numpy() if torch.is_tensor(p) else p
        with np.errstate(invalid="ignore"):
            rc, rf, rp = orc.psd(x, nfft, resolution=1.0)
        assert cnt == rc and np.array_equal(np.isfinite(p), np.isfinite(rp))
        ok = np.isfinite(rp)
        assert ok[3].all() and not ok[0].any()
        assert np.max(np.abs(p[ok] - rp[ok])) < RTOL * np.max(np.abs(rp[ok]))
        ft, tt, X = stft(data, fs=nfft, axis=-1, resolution=1.0, overlap=0.5, boundary=True, padded=True)
        X = X.cpu().numpy() if torch.is_tensor(X) else X
        with np.errstate(invalid="ignore"):
            _, _, rX = orc.stft(x, nfft, resolution=1.0)
        assert X.shape == rX.shape and np.array_equal(np.isfinite(X), np.isfinite(rX))
        ok = np.isfinite(rX)
        assert np.max(np.abs(X[ok] - rX[ok])) < RTOL * np.max(np.abs(rX[ok]))
        # segments that do not overlap: the bad one's neighbour -- its partner in a transform that
        # carries two segments (cube, fft8, chirp kernels) -- stays what it is
        ft, tt, X = stft(data, fs=nfft, axis=-1, resolution=1.0, overlap=0.0, boundary=False, padded=False)
        X = X.cpu().numpy() if torch.is_tensor(X) else X
        with np.errstate(invalid="ignore"):
            _, _, rX = orc.stft(x, nfft, resolution=1.0, overlap=0.0, boundary=False, padded=False)
        assert X.shape == rX.shape and np.array_equal(np.isfinite(X), np.isfinite(rX))
        ok = np.isfinite(rX)
        assert ok[0][:, 2].all() and not ok[0][:, 3].any() and ok[0][:, 4].all()      # (sample 3 nfft + 5: segment 3 alone)
        assert np.max(np.abs(X[ok] - rX[ok])) < RTOL * np.max(np.abs(rX[ok]))


def test_linear_trend_refuses_nonfinite_data_as_the_reference_does(nm):
    """detrend='linear' is scipy.signal.detrend's least-squares fit, and SciPy's lstsq refuses
    non-finite data: the reference raises ValueError('array must not contain infs or NaNs') at
    core/numerical.py:691 as soon as a segment of ANY channel holds such a sample (the segments
    before it have been handed on).  So does this library, on every entry point; a constant trend
    lets the NaN through (test_spectra_nan_reach_is_the_segments)."""
    import scipy.signal as sps
    import torch
    from openseize_amd.spectra.estimators import psd, stft
    nfft = 1000
    x = np.random.default_rng(3).standard_normal((3, 9 * nfft))
    x[1, 5 * nfft + 7] = np.nan
    with pytest.raises(ValueError, match="infs or NaNs"):
        sps.detrend(x[:, 5 * nfft:6 * nfft], axis=-1, type="linear")          # what the reference calls
    for data in (x, torch.from_numpy(x).cuda()):
        with pytest.raises(ValueError, match="infs or NaNs"):
            psd(data, fs=nfft, axis=-1, resolution=1.0, detrend="linear")
        with pytest.raises(ValueError, match="infs or NaNs"):
            stft(data, fs=nfft, axis=-1, resolution=1.0, detrend="linear")
        with pytest.raises(ValueError, match="infs or NaNs"):
            nm.periodogram(data[:, 5 * nfft:6 * nfft], fs=nfft, nfft=nfft, detrend="linear")
        with pytest.raises(ValueError, match="infs or NaNs"):
            nm.modified_dft(data[:, 5 * nfft:6 * nfft], nfft, nfft, "hann", -1, "linear", "density")
        # the per-segment generator hands on the segments before the refused one (no overlap: five)
        freqs, pro = nm.welch(producer(data, 2 * nfft, -1), nfft, nfft, "hann", 0.0, -1, "linear", "density")
        got = []
        with pytest.raises(ValueError, match="infs or NaNs"):
            for seg in pro:
                got.append(seg)
        assert len(got) == 5
        # finite data: nothing refused
        psd(data[:, :5 * nfft], fs=nfft, axis=-1, resolution=1.0, detrend="linear")


@pytest.mark.parametrize("fed", ["resident", "host"])
def test_oaconvolve_reach_behind_a_masked_producer(nm, fed):
    """A MaskedProducer in front (its shape names the kept samples; core/producer.py:399-408): the
    reach of a non-finite sample is counted in the samples the FIR sees, as in the reference."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    rng = np.random.default_rng(17)
    n, cs = 400_000, 30_000
    x = rng.standard_normal((3, n))
    mask = rng.random(n) > 0.3
    kept = np.flatnonzero(mask)
    x[0, kept[100_000]] = np.nan
    x[1, kept[-1]] = np.inf
    x[2, kept[5] + 1 if not mask[kept[5] + 1] else kept[5]] = np.nan      # (a dropped sample's NaN must not matter when it is dropped)
    h = sps.firwin(256, 0.2)
    src = torch.from_numpy(x).cuda() if fed == "resident" else x
    got = np.concatenate([p.cpu().numpy() if torch.is_tensor(p) else p
                          for p in nm.oaconvolve(producer(src, cs, -1, mask=mask), h, -1, "same")], -1)
    with np.errstate(invalid="ignore"):
        want = np.concatenate(orc.oaconvolve(x[:, mask], h, "same"), -1)
    assert got.shape == want.shape
    ok = np.isfinite(want)
    assert np.array_equal(ok, np.isfinite(got)), np.argwhere(ok != np.isfinite(got))[:4]
    assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))


@pytest.mark.parametrize("fed", ["resident", "host"])
def test_oaconvolve_reach_on_any_axis(nm, fed):
    """The sample axis first, and in the middle of three: the pieces the reach is laid over are
    the caller's N-D arrays."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    rng = np.random.default_rng(23)
    h = sps.firwin(65, 0.25)
    for shape, axis in (((150_000, 3), 0), ((2, 90_001, 3), 1)):
        x = rng.standard_normal(shape)
        idx = [slice(None)] * len(shape)
        idx[axis] = 60_000
        x[tuple(idx)].flat[0] = np.nan                      # one channel, one sample
        idx[axis] = shape[axis] - 2
        x[tuple(idx)].flat[-1] = np.inf
        src = torch.from_numpy(x).cuda() if fed == "resident" else x
        got = np.concatenate([p.cpu().numpy() if torch.is_tensor(p) else p
                              for p in nm.oaconvolve(producer(src, 20_000, axis), h, axis, "same")], axis)
        x2 = np.moveaxis(x, axis, -1)
        with np.errstate(invalid="ignore"):
            want = np.concatenate(orc.oaconvolve(x2.reshape(-1, shape[axis]), h, "same"), -1)
        want = np.moveaxis(want.reshape(x2.shape[:-1] + (-1,)), -1, axis)
        assert got.shape == want.shape
        ok = np.isfinite(want)
        assert not ok.all() and np.array_equal(ok, np.isfinite(got)), (shape, np.argwhere(ok != np.isfinite(got))[:4])
        assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))


def test_fir_sosfilt_chain_reach_with_the_sample_axis_first(nm):
    """The fused FIR -> sosfilt launch with (samples, channels) data: the reach is laid over 2-D
    pieces inside the step, the caller gets its own layout back."""
    import scipy.signal as sps
    import torch
    from oracle import oracle as orc
    sos = sps.butter(4, [0.05, 0.3], "bandpass", output="sos")
    h = sps.firwin(256, 0.2)
    cs, total = 70_000, 70_000 * 5 + 999
    x = np.random.default_rng(29).standard_normal((total, 3))
    x[200_000, 1] = np.nan
    for data in (x, torch.from_numpy(x).cuda()):
        src = producer(data, cs, 0)
        fir = producer(partial(nm.oaconvolve, src, h, 0, "same"), cs, 0, shape=src.shape)
        got = np.concatenate([p.cpu().numpy() if torch.is_tensor(p) else p for p in nm.sosfilt(fir, sos, 0)], 0)
        with np.errstate(invalid="ignore"):
            want = orc.sosfilt(np.concatenate(orc.oaconvolve(np.ascontiguousarray(x.T), h, "same"), -1), sos, cs)[0].T
        ok = np.isfinite(want)
        assert got.shape == want.shape and not ok[:, 1].all() and ok[:, 0].all()
        assert np.array_equal(ok, np.isfinite(got))
        assert np.max(np.abs(got[ok] - want[ok])) < RTOL * np.max(np.abs(want[ok]))

"""Pins the CPU oracle (oracle/) against the golden vectors the reference
produced (tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from oracle import oracle as orc

RTOL = 1e-9   # oracle vs reference: same algorithm, only summation order differs


def close(a, b, rtol=RTOL):
    scale = max(1.0, float(np.max(np.abs(b))))
    return np.max(np.abs(a - b)) <= rtol * scale


def test_producer_lengths(golden):
    g = golden("g1_producer.npz")
    x = g["x"]
    for cs in (1000, 1024, 10007, 20000):
        assert orc.array_chunk_lengths(x.shape[-1], cs) == list(g[f"array_len_cs{cs}"])
    for cs in (1000, 4096, 20000):
        assert orc.rechunk_lengths(x.shape[-1], cs) == list(g[f"gen_len_cs{cs}"])
    assert np.array_equal(g["gen_cat_cs1000"], x)


@pytest.mark.parametrize("name", ["rand", "hole", "short"])
def test_masked(golden, name):
    g = golden("g1_producer.npz")
    x, m = g["x"], g[f"mask_{name}"]
    for cs in (1000, 1024):
        y = orc.masked_stream(x, m, cs)
        assert orc.rechunk_lengths(y.shape[-1], cs) == list(g[f"masked_{name}_len_cs{cs}"])
    assert np.array_equal(orc.masked_stream(x, m, 1000), g[f"masked_{name}_cat_cs1000"])


@pytest.mark.parametrize("taps", [76, 255, 256, 1024])
@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_oaconvolve(golden, taps, mode):
    g = golden("g2_fir.npz")
    x, h = g["x"], g[f"h{taps}"]
    pieces = orc.oaconvolve(x, h, mode)
    assert [p.shape[-1] for p in pieces] == list(g[f"pieces_t{taps}_{mode}"])
    y = np.concatenate(pieces, axis=-1)
    assert close(y, g[f"y_t{taps}_{mode}"], 1e-12)
    # and the segmentation-independent meaning: np.convolve
    assert close(orc.convolve_direct(x, h, mode), g[f"y_t{taps}_{mode}"], 1e-12)


def test_oaconvolve_long_and_quirk(golden):
    g = golden("g2_fir.npz")
    x, h = g["x_long"], g["h76"]
    for mode in ("full", "same", "valid"):
        pieces = orc.oaconvolve(x, h, mode)
        assert [p.shape[-1] for p in pieces] == list(g[f"ylong_{mode}_pieces"])
        y = np.concatenate(pieces, axis=-1)
        assert close(y[:, :400], g[f"ylong_{mode}_head"], 1e-12)
        assert close(y[:, -400:], g[f"ylong_{mode}_tail"], 1e-12)
        assert close(y[:, ::37], g[f"ylong_{mode}_dec"], 1e-12)
    assert bool(g["quirk_odd_fallback_raises"])
    with pytest.raises(ValueError):
        orc.oaconvolve(g["x"][:, :5003], g["h1024"], "same")


FILTERS = ["butter_lp", "butter_bp6", "cheby1_bp", "butter_cls6"]


@pytest.mark.parametrize("name", FILTERS)
def test_sosfilt(golden, name):
    g = golden("g3_sosfilt.npz")
    x, sos = g["x"], g[f"sos_{name}"]
    for cs in (1000, 4096):
        y, _ = orc.sosfilt(x, sos, cs)
        assert np.array_equal(y, g[f"y_{name}_cs{cs}"])      # bit-exact DF2T
    zi = g[f"zi_{name}"]
    y, _ = orc.sosfilt(x, sos, 1000, zi=zi)
    assert np.array_equal(y, g[f"yzi_{name}"])


@pytest.mark.parametrize("name", FILTERS)
def test_sosfilt_zi(golden, name):
    g = golden("g9_design.npz")
    assert np.allclose(orc.sosfilt_zi(g[f"sos_{name}"]), g[f"zi_{name}"],
                       rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("name", FILTERS)
@pytest.mark.parametrize("cs", [200, 1000, 4096, 6007])
def test_sosfiltfilt(golden, name, cs):
    g = golden("g4_sosfiltfilt.npz")
    y = orc.sosfiltfilt(g["x"], g[f"sos_{name}"], cs)
    assert close(y, g[f"y_{name}_cs{cs}"], 1e-9)


@pytest.mark.parametrize("LM", [(1, 5), (3, 1), (3, 2), (2, 7), (3, 11)])
def test_resample(golden, LM):
    g = golden("g5_resample.npz")
    L, M = LM
    x = g["x"]
    h = orc.resample_filter(L, M, 5000)
    assert np.allclose(h, g[f"h_L{L}_M{M}"], rtol=0, atol=1e-15)
    y = orc.polyphase_resample(x, L, M, h)
    for cs in (3000, 7001):
        ref = g[f"y_L{L}_M{M}_cs{cs}"]
        assert y.shape == ref.shape
        assert close(y, ref, 1e-12)
    assert orc.rechunk_lengths(y.shape[-1], 3000) == list(g[f"len_L{L}_M{M}"])
    assert tuple(g[f"shape_L{L}_M{M}"]) == y.shape


def test_periodogram(golden):
    g = golden("g6_periodogram.npz")
    x = g["x"]
    for window in ("hann", "hamming", "boxcar", "blackman"):
        for detrend in ("constant", "linear"):
            for scaling in ("density", "spectrum"):
                f, p = orc.periodogram(x, 500, None, window, detrend, scaling)
                assert close(p, g[f"p_{window}_{detrend}_{scaling}"], 1e-10)
    assert np.array_equal(f, g["freqs"])
    _, X = orc.modified_dft(x, 500, 1024, "hann", "constant", "density")
    assert close(np.abs(X - g["dft_hann"]), 0 * np.abs(X), 1e-12)
    f, p = orc.periodogram(x[:, :1023], 500)
    assert close(p, g["p_odd"], 1e-10) and np.array_equal(f, g["freqs_odd"])
    f, p = orc.periodogram(x, 500, 2048)
    assert close(p, g["p_pad2048"], 1e-10)
    f, p = orc.periodogram(x, 500, 512, "hann", "linear", "spectrum")
    assert close(p, g["p_crop512"], 1e-10)
    with pytest.raises(ValueError):
        orc.modified_dft(x, 500, 1024, "hann", "constant", "power")


def test_welch_psd(golden):
    g = golden("g7_welch.npz")
    x = g["x"]
    for ov in (0.0, 0.5, 0.6):
        cnt, f, p = orc.psd(x, 1024, resolution=1.0, overlap=ov)
        assert cnt == int(g[f"cnt_ov{ov}"])
        assert close(p, g[f"psd_ov{ov}"], 1e-10)
    assert np.array_equal(f, g["freqs"])
    cnt, f, p = orc.psd(x, 1024, resolution=0.5, window="hamming",
                        detrend="linear", scaling="spectrum")
    assert cnt == int(g["cnt_hamming"]) and close(p, g["psd_hamming_linear_spectrum"], 1e-10)
    f, segs = orc.welch_segments(x, 1024, 1024, "hann", 0.5, "constant", "density")
    assert len(segs) == int(g["welch_nseg"])
    assert close(segs[0], g["welch_seg0"], 1e-10) and close(segs[-1], g["welch_seg_last"], 1e-10)
    assert orc.welch_reported_nsegs(x.shape[-1], 1024, 0.5) == int(g["welch_shape"][-1])
    cnt, f, p = orc.psd(np.ascontiguousarray(x[:2]), 1000, resolution=2.0)
    assert cnt == int(g["cnt_axis0"]) and close(p.T, g["psd_axis0_nfft500"], 1e-10)


def test_stft(golden):
    g = golden("g8_stft.npz")
    x = g["x"]
    for b in (True, False):
        for p in (True, False):
            for scaling in ("density", "spectrum"):
                f, t, X = orc.stft(x, 256, resolution=1.0, boundary=b, padded=p,
                                   scaling=scaling)
                key = f"b{int(b)}_p{int(p)}_{scaling}"
                assert X.shape == g[f"X_{key}"].shape
                assert np.allclose(t, g[f"t_{key}"], rtol=0, atol=1e-12)
                assert np.max(np.abs(X - g[f"X_{key}"])) < 1e-12
    f, t, X = orc.stft(x, 256, resolution=0.5, overlap=0.75, detrend="linear",
                       window="hamming")
    assert np.allclose(t, g["pro_t"]) and np.max(np.abs(X - g["pro_X"])) < 1e-12


BA = ["butter_lp", "cheby1_bp", "ellip_lp", "butter_bp", "notch"]


@pytest.mark.parametrize("name", BA)
def test_ba_filters(golden, name):
    g = golden("g10_ba.npz")
    x, coeffs = g["x"], (g[f"b_{name}"], g[f"a_{name}"])
    for cs in (1000, 4000):
        y, _ = orc.lfilter(x, coeffs, cs)
        assert np.array_equal(y, g[f"lfilter_{name}_cs{cs}"])       # bit-exact DF2T
        # filtfilt starts from lfilter_zi: closed form here, a linear solve in
        # SciPy -- the 18th-order direct form amplifies that to ~1e-8
        assert close(orc.filtfilt(x, coeffs, cs), g[f"filtfilt_{name}_cs{cs}"], 1e-7)
    if name == "notch":
        y, _ = orc.lfilter(x, coeffs, 1000, zi=g["notch_zi"])
        assert close(y, g["notch_lfilter_zi"], 1e-12)
        assert close(orc.filtfilt(x, coeffs, 1500).T, g["notch_axis0"], 1e-12)


def test_nonfinite_reach(golden):
    """G15: where the reference's IIR passes are non-finite after a NaN / an Inf in the
    input (forward: to the end of the stream; backward: the chunk and the one before)."""
    from conftest import nonfinite_inputs, nonfinite_runs
    g = golden("g15_nonfinite.npz")
    sos, cs = g["sos"], int(g["chunksize"])
    for tag, data in zip(("nan", "inf"), nonfinite_inputs()):
        y, _ = orc.sosfilt(data, sos, cs)
        yy = orc.sosfiltfilt(data, sos, cs)
        for name, arr in (("sosfilt", y), ("sosfiltfilt", yy)):
            assert np.array_equal(nonfinite_runs(arr), g[f"{name}_{tag}_runs"])
            assert bool(np.all(np.isnan(arr[~np.isfinite(arr)]))) == bool(g[f"{name}_{tag}_isnan_all"])
            dec, want = arr[:, ::97], g[f"{name}_{tag}_dec"]
            ok = np.isfinite(want)
            assert np.array_equal(ok, np.isfinite(dec))
            assert np.max(np.abs(dec[ok] - want[ok])) < 1e-12 * np.max(np.abs(want[ok]))


def test_oracle_chain_at_the_zero_phase_geometry(golden):
    """g18: the reference's own outputs for the headline chain at a geometry the build's
    zero-phase route takes (1024 taps -> 6-section band-pass sosfiltfilt, chunks of 65 536,
    channel 1 on an offset of 10^4).  The oracle's restatement (core/numerical.py:158-298,
    :338-411) reproduces them: what the GPU route is held against is pinned here too."""
    g = golden("g18_chain_long.npz")
    x, want, h, sos, cs = g["x32"].astype(np.float64), g["y"], g["h"], g["sos"], int(g["chunksize"])
    u = np.concatenate(orc.oaconvolve(x, h, "same"), axis=-1)
    got = orc.sosfiltfilt(u, sos, cs)
    for c in range(x.shape[0]):
        assert np.max(np.abs(got[c] - want[c])) < 1e-10 * np.max(np.abs(want[c])), c


def test_oracle_nonfinite_reach_through_the_fir(golden):
    """g19: which output chunks the REFERENCE's FIR -> sosfiltfilt chain loses to a non-finite input
    sample -- its overlap-add makes the sample's whole segment non-finite (core/numerical.py:258-283),
    so the chain is lost from up to a segment before the sample.  The oracle's restatement runs the
    same segments: same chunks lost, for every placement."""
    import scipy.signal as sps
    g = golden("g19_fir_chain_nonfinite.npz")
    where, cs, total, taps_n = g["where"], int(g["chunksize"]), int(g["total"]), int(g["taps"])
    assert orc.oa_plan(total, taps_n)[1] == int(g["step"])
    nchunks = -(-total // cs)
    C = len(where) + 2
    x = np.random.default_rng(5).standard_normal((C, total))       # (a mask depends on where the bad samples are only)
    for c, at in enumerate(where):
        x[c, at] = np.nan if c != 2 else np.inf
    x[len(where) - 1, where[-1]:] = np.nan
    x[C - 2, (nchunks - 2) * cs + 100] = np.nan
    h = sps.firwin(taps_n, 0.3)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    with np.errstate(invalid="ignore"):
        y = orc.sosfiltfilt(np.concatenate(orc.oaconvolve(x, h, "same"), axis=-1), sos, cs)
    lost = np.array([[bool((~np.isfinite(y[c, k * cs:(k + 1) * cs])).all()) for k in range(nchunks)] for c in range(C)])
    assert np.array_equal(lost, g["lost_chunks"])


def test_oracle_resample_nonfinite_reach_is_the_references(golden):
    """A non-finite sample through the reference's resampler (g20: its own outputs' masks for
    samples at the stream's ends, at a chunk boundary, an Inf, a run to the end; five ratios,
    two chunkings): the outputs SciPy's PADDED window touches -- `oracle.resample_lost` -- and
    the finite ones still the definition's."""
    import scipy.signal as sps
    g = golden("g20_resample_nonfinite.npz")
    n = int(g["n"])
    x = np.random.default_rng(2020).standard_normal((4, n))       # (g20_input of make_golden.py)
    x[0, 0] = np.nan
    x[0, 10_000] = np.nan
    x[1, n - 1] = np.inf
    x[1, 13_333] = np.nan
    x[2, 17_000:] = np.nan
    for (L, M) in ((1, 5), (3, 2), (2, 1), (1, 25), (2, 7)):
        h = orc.resample_filter(L, M, 5000)
        nout = int(g[f"nout_L{L}_M{M}"])
        with np.errstate(invalid="ignore"):
            y = orc.polyphase_resample(x, L, M, h)
        assert y.shape[-1] == nout
        for cs in (5_000, 7_321):
            lost = np.unpackbits(g[f"lost_L{L}_M{M}_cs{cs}"], axis=-1)[:, :nout].astype(bool)
            assert np.array_equal(~np.isfinite(y), lost), (L, M, cs)
        clean = np.where(np.isfinite(x), x, 0.0)
        ok = np.isfinite(y)
        ref = sps.resample_poly(clean, L, M, axis=-1, window=h)
        # (a finite output touches no non-finite sample: the cleaned stream gives the same number)
        assert np.max(np.abs(y[ok] - ref[ok])) < 1e-12 * np.max(np.abs(ref))


def test_oracle_linear_trend_refuses_nonfinite_data():
    """scipy.signal.detrend(type='linear') -- what the reference calls at core/numerical.py:691 --
    raises on non-finite data (scipy.linalg.lstsq's check); so does the oracle, and a constant
    trend lets the NaN through."""
    import scipy.signal as sps
    x = np.random.default_rng(0).standard_normal((2, 3000))
    x[0, 1500] = np.nan
    with pytest.raises(ValueError, match="infs or NaNs"):
        sps.detrend(x, axis=-1, type="linear")
    with pytest.raises(ValueError, match="infs or NaNs"):
        orc.psd(x, 1000, resolution=1.0, detrend="linear")
    cnt, f, p = orc.psd(x, 1000, resolution=1.0, detrend="constant")
    assert np.isnan(p[0]).all() and np.isfinite(p[1]).all()

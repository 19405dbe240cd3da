"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (the reference does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src \
        python3 /root/repo/tests/golden/make_golden.py

Every file holds seeded float64 inputs and the outputs the reference produced
for them.  Only data is written: no reference
source text is stored.  Keys are documented next to each block.
"""

import os
from functools import partial

import numpy as np
import scipy.signal as sps

from openseize import producer
from openseize.core import numerical as nm
from openseize.filtering import fir as ref_fir
from openseize.filtering import iir as ref_iir
from openseize.resampling import resampling as ref_rs
from openseize.spectra import estimators as ref_est

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e3:.0f} kB, {len(arrays)} arrays")


def lengths(pro, axis=-1):
    return np.array([a.shape[axis] for a in pro], dtype=np.int64)


# --------------------------------------------------------------------------
# G1 producers (rows a1-a4): chunk-length lists + bit-exact contents
# --------------------------------------------------------------------------
def g1_producer():
    rng = np.random.default_rng(101)
    x = rng.standard_normal((3, 10007))
    out = {"x": x}
    for cs in (1000, 1024, 10007, 20000):
        out[f"array_len_cs{cs}"] = lengths(producer(x, cs, axis=-1))

    # generator of ragged pieces -> GenProducer re-chunking (a2)
    cuts = np.array([0, 13, 700, 701, 2900, 2900, 6000, 9999, 10007])

    def ragged(arr, cuts):
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                yield arr[:, a:b]

    out["gen_cuts"] = cuts
    for cs in (1000, 4096, 20000):
        pro = producer(ragged, cs, axis=-1, shape=x.shape, arr=x, cuts=cuts)
        out[f"gen_len_cs{cs}"] = lengths(pro)
        if cs == 1000:
            out[f"gen_cat_cs{cs}"] = np.concatenate(list(pro), axis=-1)

    # masked producers (a3): random mask, an all-False chunk, a short mask
    m1 = rng.random(10007) < 0.2
    m2 = m1.copy()
    m2[2000:3000] = False
    m3 = rng.random(7000) < 0.5
    for name, m in (("rand", m1), ("hole", m2), ("short", m3)):
        out[f"mask_{name}"] = m
        for cs in (1000, 1024):
            pro = producer(x, cs, axis=-1, mask=m)
            out[f"masked_{name}_len_cs{cs}"] = lengths(pro)
            if cs == 1000:
                out[f"masked_{name}_cat_cs{cs}"] = np.concatenate(
                    list(pro), axis=-1)
            out[f"masked_{name}_shape_cs{cs}"] = np.array(pro.shape)
    # sample axis first (axis=0) masked producer
    xt = np.ascontiguousarray(x.T)
    pro = producer(xt, 1000, axis=0, mask=m1)
    out["masked_axis0_cat"] = np.concatenate(list(pro), axis=0)
    save("g1_producer.npz", **out)


# --------------------------------------------------------------------------
# G2 oaconvolve (row a5) + class API (row a13)
# --------------------------------------------------------------------------
def g2_fir():
    rng = np.random.default_rng(202)
    x = rng.standard_normal((2, 5004))
    out = {"x": x}
    # Reference quirk (found while generating): when the nfft fallback picks
    # nfft = N (numerical.py:210-211) and N is odd, np.fft.irfft returns N-1
    # samples and the overlap add raises a broadcasting ValueError.  Recorded
    # here as a flag; the even-N case below is the one with golden outputs.
    try:
        list(nm.oaconvolve(producer(x[:, :5003], 1000, axis=-1),
                           sps.firwin(1024, 0.2), -1, "same"))
        out["quirk_odd_fallback_raises"] = np.array(False)
    except ValueError:
        out["quirk_odd_fallback_raises"] = np.array(True)
    for taps in (76, 255, 256, 1024):
        h = sps.firwin(taps, 0.2)
        out[f"h{taps}"] = h
        for mode in ("full", "same", "valid"):
            pro = producer(x, 1000, axis=-1)
            pieces = list(nm.oaconvolve(pro, h, -1, mode))
            out[f"y_t{taps}_{mode}"] = np.concatenate(pieces, axis=-1)
            out[f"pieces_t{taps}_{mode}"] = np.array(
                [p.shape[-1] for p in pieces], dtype=np.int64)
    # long input: the non-fallback nfft (= 8*128*32 = 32768 for 76 taps)
    xl = rng.standard_normal((2, 70001))
    h = out["h76"]
    out["x_long"] = xl
    for mode in ("full", "same", "valid"):
        pro = producer(xl, 16384, axis=-1)
        pieces = list(nm.oaconvolve(pro, h, -1, mode))
        # decimated + head/tail keeps the file small; lengths pin the indexing
        y = np.concatenate(pieces, axis=-1)
        out[f"ylong_{mode}_head"] = y[:, :400]
        out[f"ylong_{mode}_tail"] = y[:, -400:]
        out[f"ylong_{mode}_dec"] = y[:, ::37]
        out[f"ylong_{mode}_pieces"] = np.array(
            [p.shape[-1] for p in pieces], dtype=np.int64)
    # GenProducer-wrapped lengths through the class API, ndarray and producer
    kais = ref_fir.Kaiser(fpass=200, fstop=400, fs=5000, gpass=0.5, gstop=40)
    out["kaiser_h"] = kais.coeffs
    xk = rng.standard_normal((2, 6001))
    out["xk"] = xk
    for mode in ("full", "same", "valid"):
        out[f"kaiser_arr_{mode}"] = kais(xk, chunksize=2000, axis=-1, mode=mode)
        res = kais(producer(xk, 2000, axis=-1), chunksize=2000, axis=-1, mode=mode)
        out[f"kaiser_pro_len_{mode}"] = lengths(res)
        out[f"kaiser_pro_shape_{mode}"] = np.array(res.shape)
    # sample axis not last (axis=1 of a 3-D array), as tests/test_oaconvolve.py
    x3 = rng.standard_normal((2, 2001, 3))
    out["x3"] = x3
    # (FIR.__call__ itself raises IndexError for 0 < axis: bases.py:411 indexes
    # the 1-D window's shape with the data axis; the generator is used directly)
    out["kaiser_axis1_same"] = np.concatenate(list(nm.oaconvolve(
        producer(x3, 700, axis=1), kais.coeffs, 1, "same")), axis=1)
    save("g2_fir.npz", **out)


# --------------------------------------------------------------------------
# G3 sosfilt (a6) and G4 sosfiltfilt (a7)
# --------------------------------------------------------------------------
def filters():
    f = {}
    f["butter_lp"] = ref_iir.Butter(fpass=100, fstop=200, fs=500).coeffs
    f["butter_bp6"] = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    f["cheby1_bp"] = ref_iir.Cheby1(
        fpass=[200, 600], fstop=[150, 650], fs=2500).coeffs
    f["butter_cls6"] = ref_iir.Butter(
        fpass=[8, 30], fstop=[3, 60], fs=500, gpass=1, gstop=40).coeffs
    return f


def g3_sosfilt():
    rng = np.random.default_rng(303)
    x = rng.standard_normal((2, 6007))
    out = {"x": x}
    for name, sos in filters().items():
        out[f"sos_{name}"] = sos
        for cs in (1000, 4096):
            pro = producer(x, cs, axis=-1)
            out[f"y_{name}_cs{cs}"] = np.concatenate(
                list(nm.sosfilt(pro, sos, -1)), axis=-1)
        zi = rng.standard_normal((sos.shape[0], 2, 2))
        out[f"zi_{name}"] = zi
        pro = producer(x, 1000, axis=-1)
        out[f"yzi_{name}"] = np.concatenate(
            list(nm.sosfilt(pro, sos, -1, zi=zi)), axis=-1)
    # sample axis in the middle: shape (3, L, 6), axis=1 as tests/test_iir.py
    x3 = rng.standard_normal((3, 2001, 2))
    out["x3"] = x3
    sos = out["sos_butter_lp"]
    out["y3_butter_lp"] = np.concatenate(
        list(nm.sosfilt(producer(x3, 1000, axis=1), sos, 1)), axis=1)
    save("g3_sosfilt.npz", **out)


def g4_sosfiltfilt():
    rng = np.random.default_rng(404)
    x = rng.standard_normal((2, 6007))
    out = {"x": x}
    for name, sos in filters().items():
        out[f"sos_{name}"] = sos
        for cs in (200, 1000, 4096, 6007):
            pro = producer(x, cs, axis=-1)
            out[f"y_{name}_cs{cs}"] = np.concatenate(
                list(nm.sosfiltfilt(pro, sos, -1)), axis=-1)
    # class API, ndarray in -> ndarray out, and causal variant
    butter = ref_iir.Butter(fpass=[8, 30], fstop=[3, 60], fs=500, gpass=1, gstop=40)
    out["cls_dephase"] = butter(x, chunksize=2000, axis=-1, dephase=True)
    out["cls_causal"] = butter(x, chunksize=2000, axis=-1, dephase=False)
    save("g4_sosfiltfilt.npz", **out)


# --------------------------------------------------------------------------
# G5 polyphase resampling (a8)
# --------------------------------------------------------------------------
def g5_resample():
    rng = np.random.default_rng(505)
    x = rng.standard_normal((2, 9011))
    out = {"x": x}
    fs = 5000
    for (L, M) in ((1, 5), (3, 1), (3, 2), (2, 7), (3, 11)):
        for cs in (3000, 7001):
            y = ref_rs.resample(x, L, M, fs, chunksize=cs, axis=-1)
            out[f"y_L{L}_M{M}_cs{cs}"] = y
        pro = ref_rs.resample(producer(x, 3000, axis=-1), L, M, fs, 3000, axis=-1)
        out[f"len_L{L}_M{M}"] = lengths(pro)
        out[f"shape_L{L}_M{M}"] = np.array(pro.shape)
        cutoff = fs / (2 * max(L, M))
        h = ref_fir.Kaiser(cutoff - cutoff / 10, cutoff + cutoff / 10, fs,
                           gpass=0.1, gstop=40).coeffs
        out[f"h_L{L}_M{M}"] = h
    out["down5"] = ref_rs.downsample(x, 5, fs, chunksize=4000, axis=-1)
    out["up3"] = ref_rs.upsample(x[:, :6000], 3, fs, chunksize=2000, axis=-1)
    # sample axis first
    xt = np.ascontiguousarray(x[:2, :5000].T)
    out["down5_axis0"] = ref_rs.downsample(xt, 5, fs, chunksize=1000, axis=0)
    save("g5_resample.npz", **out)


# --------------------------------------------------------------------------
# G6 periodogram / modified_dft (a9, a10); G7 welch / psd; G8 stft (a11, a12)
# --------------------------------------------------------------------------
def g6_periodogram():
    rng = np.random.default_rng(606)
    x = rng.standard_normal((3, 1024)) + np.linspace(0, 3, 1024)
    out = {"x": x}
    fs = 500
    for window in ("hann", "hamming", "boxcar", "blackman"):
        for detrend in ("constant", "linear"):
            for scaling in ("density", "spectrum"):
                f, p = nm.periodogram(x, fs, None, window, -1, detrend, scaling)
                out[f"p_{window}_{detrend}_{scaling}"] = p
                out["freqs"] = f
    f, X = nm.modified_dft(x, fs, 1024, "hann", -1, "constant", "density")
    out["dft_hann"] = X
    # odd nfft, zero-padded nfft, cropped nfft
    f, p = nm.periodogram(x[:, :1023], fs, None, "hann", -1, "constant", "density")
    out["p_odd"], out["freqs_odd"] = p, f
    f, p = nm.periodogram(x, fs, 2048, "hann", -1, "constant", "density")
    out["p_pad2048"], out["freqs_pad2048"] = p, f
    f, p = nm.periodogram(x, fs, 512, "hann", -1, "linear", "spectrum")
    out["p_crop512"] = p
    save("g6_periodogram.npz", **out)


def g7_welch():
    rng = np.random.default_rng(707)
    x = rng.standard_normal((3, 20000))
    out = {"x": x}
    fs = 1024
    for overlap in (0.0, 0.5, 0.6):
        cnt, f, p = ref_est.psd(x, fs, axis=-1, resolution=1.0, overlap=overlap)
        out[f"psd_ov{overlap}"] = p
        out[f"cnt_ov{overlap}"] = np.array(cnt)
        out["freqs"] = f
    cnt, f, p = ref_est.psd(x, fs, axis=-1, resolution=0.5, window="hamming",
                            detrend="linear", scaling="spectrum")
    out["psd_hamming_linear_spectrum"] = p
    out["cnt_hamming"] = np.array(cnt)
    out["freqs_hamming"] = f
    # the per-segment producer: lengths, reported shape (quirk Q7) and segments
    pro = producer(x, 5000, axis=-1)
    f, wp = nm.welch(pro, fs, 1024, "hann", 0.5, -1, "constant", "density")
    segs = list(wp)
    out["welch_shape"] = np.array(wp.shape)
    out["welch_nseg"] = np.array(len(segs))
    out["welch_seg0"], out["welch_seg_last"] = segs[0], segs[-1]
    # non power-of-two nfft and sample axis first
    cnt, f, p = ref_est.psd(np.ascontiguousarray(x[:2].T), 1000, axis=0,
                            resolution=2.0)
    out["psd_axis0_nfft500"], out["cnt_axis0"] = p, np.array(cnt)
    save("g7_welch.npz", **out)


def g8_stft():
    rng = np.random.default_rng(808)
    x = rng.standard_normal((2, 6100))
    out = {"x": x}
    fs = 256
    for boundary in (True, False):
        for padded in (True, False):
            for scaling in ("density", "spectrum"):
                f, t, X = ref_est.stft(x, fs, axis=-1, resolution=1.0,
                                       boundary=boundary, padded=padded,
                                       scaling=scaling, asarray=True)
                key = f"b{int(boundary)}_p{int(padded)}_{scaling}"
                out[f"X_{key}"], out[f"t_{key}"] = X, t
                out["freqs"] = f
    f, t, pro = ref_est.stft(producer(x, 1000, axis=-1), fs, axis=-1,
                             resolution=0.5, overlap=0.75, detrend="linear",
                             window="hamming", asarray=False)
    out["pro_shape"] = np.array(pro.shape)
    out["pro_t"] = t
    out["pro_X"] = np.stack(list(pro), axis=-1)
    save("g8_stft.npz", **out)


# --------------------------------------------------------------------------
# G9 design constants
# --------------------------------------------------------------------------
def g9_design():
    out = {}
    for name, sos in filters().items():
        out[f"sos_{name}"] = sos
        out[f"zi_{name}"] = sps.sosfilt_zi(sos)
    cutoff = 5000 / 10
    out["kaiser_down5_fs5000"] = ref_fir.Kaiser(
        cutoff - cutoff / 10, cutoff + cutoff / 10, 5000, gpass=0.1, gstop=40).coeffs
    out["cheby2_lp"] = ref_iir.Cheby2(fpass=100, fstop=150, fs=1000).coeffs
    out["ellip_hp"] = ref_iir.Ellip(fpass=200, fstop=150, fs=1000).coeffs
    out["hamming_bp"] = ref_fir.Hamming(
        fpass=[100, 200], fstop=[50, 250], fs=1000).coeffs
    save("g9_design.npz", **out)


# --------------------------------------------------------------------------
# G10 transfer-function (ba) filters: lfilter / filtfilt
# --------------------------------------------------------------------------
def g10_ba():
    rng = np.random.default_rng(1010)
    x = rng.random((2, 9001))
    out = {"x": x}
    filts = {
        "butter_lp": ref_iir.Butter(fpass=100, fstop=200, fs=500, fmt="ba"),
        "cheby1_bp": ref_iir.Cheby1(fpass=[200, 600], fstop=[150, 650], fs=2500,
                                    fmt="ba"),
        "ellip_lp": ref_iir.Ellip(fpass=100, fstop=200, fs=500, fmt="ba"),
        "butter_bp": ref_iir.Butter(fpass=[300, 900], fstop=[150, 1050], fs=3000,
                                    fmt="ba"),
        "notch": ref_iir.Notch(60, 8, 500),
    }
    for name, f in filts.items():
        b, a = f.coeffs
        out[f"b_{name}"], out[f"a_{name}"] = b, a
        for cs in (1000, 4000):
            out[f"lfilter_{name}_cs{cs}"] = f(x, chunksize=cs, axis=-1, dephase=False)
            out[f"filtfilt_{name}_cs{cs}"] = f(x, chunksize=cs, axis=-1, dephase=True)
    # user zi for the second-order notch, sample axis first
    b, a = filts["notch"].coeffs
    zi = rng.standard_normal((2, 2))
    out["notch_zi"] = zi
    out["notch_lfilter_zi"] = np.concatenate(list(nm.lfilter(
        producer(x, 1000, axis=-1), (b, a), -1, zi=zi)), axis=-1)
    xt = np.ascontiguousarray(x.T)
    out["notch_axis0"] = filts["notch"](xt, chunksize=1500, axis=0, dephase=True)
    save("g10_ba.npz", **out)


# --------------------------------------------------------------------------
# G11 EDF decode: a synthetic EDF written byte by byte from
# the published EDF layout (256-byte fixed header, 256 bytes per signal,
# little-endian int16 records) and read back with the reference's Reader.
# --------------------------------------------------------------------------
def write_synthetic_edf(path, rng):
    names = ["EEG Fp1", "EEG Cz", "EMG slow", "EEG O2", "EDF Annotations"]
    spr = [500, 500, 250, 500, 30]
    pmin = [-3276.8, -500.0, -1000.0, -200.0, -1.0]
    pmax = [3276.7, 500.0, 1000.0, 250.0, 1.0]
    dmin, dmax = [-32768] * 5, [32767] * 5
    nrec, ns = 20, len(names)

    def field(val, width):
        return str(val).ljust(width)[:width].encode("ascii")

    head = b"".join([
        field("0", 8), field("synthetic patient", 80), field("synthetic recording", 80),
        field("01.01.26", 8), field("00.00.00", 8), field(256 + 256 * ns, 8),
        field("", 44), field(nrec, 8), field(1, 8), field(ns, 4),
        b"".join(field(n, 16) for n in names),
        b"".join(field("AgAgCl", 80) for _ in names),
        b"".join(field("uV", 8) for _ in names),
        b"".join(field(v, 8) for v in pmin), b"".join(field(v, 8) for v in pmax),
        b"".join(field(v, 8) for v in dmin), b"".join(field(v, 8) for v in dmax),
        b"".join(field("HP:0.1Hz", 80) for _ in names),
        b"".join(field(v, 8) for v in spr), b"".join(field("", 32) for _ in names)])
    assert len(head) == 256 + 256 * ns
    recs = rng.integers(-32768, 32768, size=(nrec, sum(spr)), dtype=np.int16)
    with open(path, "wb") as fp:
        fp.write(head)
        fp.write(recs.astype("<i2").tobytes())


def g11_edf():
    from openseize.file_io import edf as ref_edf
    rng = np.random.default_rng(1111)
    path = os.path.join(OUT, "synthetic.edf")
    write_synthetic_edf(path, rng)
    out = {}
    with ref_edf.Reader(path) as reader:
        hdr = reader.header
        out["channels"] = np.array(hdr.channels)
        out["samples"] = np.array(hdr.samples)
        out["slopes"] = np.array(hdr.slopes)
        out["offsets"] = np.array(hdr.offsets)
        out["shape"] = np.array(reader.shape)
        out["read_all"] = reader.read(0)
        out["read_123_4567"] = reader.read(123, 4567)
        out["read_4900_5200"] = reader.read(4900, 5200)     # slow channel runs out
        out["read_9990_end"] = reader.read(9990)
        out["read_pad0"] = reader.read(4000, 6000, padvalue=0.0)
        reader.channels = [0, 3]
        out["read_ch03"] = reader.read(250, 2750)
        reader.channels = [2]
        out["read_ch2"] = reader.read(100, 4000)
        reader.channels = hdr.channels
    # ReaderProducer chunk lengths
    rd = ref_edf.Reader(path)
    pro = producer(rd, chunksize=1700, axis=-1)
    out["pro_shape"] = np.array(pro.shape)
    out["pro_len"] = lengths(pro)
    pro2 = producer(ref_edf.Reader(path), chunksize=1700, axis=-1, start=300, stop=8000)
    out["pro2_len"] = lengths(pro2)
    out["pro2_cat"] = np.concatenate(list(pro2), axis=-1)
    save("g11_edf.npz", **out)


# --------------------------------------------------------------------------
# G12 protools glue
# --------------------------------------------------------------------------
def g12_protools():
    from openseize.core import protools as ref_pt
    rng = np.random.default_rng(1212)
    x = rng.standard_normal((3, 1, 4001))
    x[1, 0, 77] = np.nan
    out = {"x": x}
    pro = producer(x, 900, axis=-1)
    sq = ref_pt.squeeze(pro)
    out["squeeze_shape"], out["squeeze_axis"] = np.array(sq.shape), np.array(sq.axis)
    ex = ref_pt.expand_dims(producer(x[:, 0], 900, axis=-1), (0, -1))
    out["expand_shape"], out["expand_axis"] = np.array(ex.shape), np.array(ex.axis)
    out["expand_arr"] = ex.to_array()
    other = rng.standard_normal((3, 1, 1))
    out["other"] = other
    out["add_arr"] = ref_pt.add(pro, other).to_array()
    out["mul_pro"] = ref_pt.multiply(pro, producer(2 * x, 500, axis=-1)).to_array()
    w = rng.standard_normal(4001)
    out["w"] = w
    out["mul_along_prod"] = ref_pt.multiply_along_axis(pro, w, -1).to_array()
    out["mul_along_other"] = ref_pt.multiply_along_axis(pro, np.array([1.0, 2.0, 3.0]), 0).to_array()
    out["slice_prod"] = ref_pt.slice_along_axis(pro, 10, 3000, 3, axis=-1).to_array()
    out["slice_other"] = ref_pt.slice_along_axis(pro, 1, None, None, axis=0).to_array()
    for ignore in (True, False):
        out[f"mean_prod_{int(ignore)}"] = ref_pt.mean(pro, -1, ignore, keepdims=True)
        out[f"std_prod_{int(ignore)}"] = ref_pt.std(pro, -1, ignore, keepdims=True)
    out["mean_other"] = ref_pt.mean(pro, 0)
    out["std_other"] = ref_pt.std(pro, 0)
    out["standardize_prod"] = ref_pt.standardize(pro, -1).to_array()
    out["standardize_other"] = ref_pt.standardize(pro, 0).to_array()
    # multiply_along_axis along a non-production axis: the reference zips the producer with
    # the reshaped multiplier ARRAY (core/protools.py:418-426), which iterates its axis 0 --
    # chunk k is scaled by the single value arr[k] only when the multiplied axis IS axis 0;
    # along a middle axis (the docstring's own example) it is the plain broadcast
    x3 = rng.standard_normal((2, 4, 1250))
    w4, w2 = np.array([0.0, -1.0, 1.0, 0.5]), np.array([3.0, -2.0])
    out["x3"], out["w4"], out["w2"] = x3, w4, w2
    out["mul3_middle"] = ref_pt.multiply_along_axis(producer(x3, 100, axis=-1), w4, 1).to_array()
    out["mul3_first"] = ref_pt.multiply_along_axis(producer(x3, 100, axis=-1), w2, 0).to_array()
    w1250 = rng.standard_normal(1250)
    out["w1250"] = w1250
    out["mul3_last_prod0"] = ref_pt.multiply_along_axis(producer(x3, 1, axis=0), w1250, 2).to_array()
    save("g12_protools.npz", **out)


# --------------------------------------------------------------------------
# G13 Hilbert transformer
# --------------------------------------------------------------------------
def g13_hilbert():
    from openseize.filtering.special import Hilbert
    rng = np.random.default_rng(1313)
    x = rng.standard_normal((2, 8000))
    filt = Hilbert(width=12.5, fs=500)
    out = {"x": x, "coeffs": filt.coeffs}
    out["imag_same"] = filt(x, chunksize=3000, axis=-1, mode="same")
    save("g13_hilbert.npz", **out)


# --------------------------------------------------------------------------
# G14 band metrics and the analytic transform
# --------------------------------------------------------------------------
def g14_metrics_analytic():
    from openseize.experimental.coupling.transforms import Analytic
    from openseize.spectra import metrics
    rng = np.random.default_rng(1414)
    psd = rng.random((3, 501)) + 0.1
    freqs = np.linspace(0, 250, 501)
    out = {"psd": psd, "freqs": freqs}
    out["power_all"] = metrics.power(psd, freqs)
    out["power_0_40"] = metrics.power(psd, freqs, start=0, stop=40)
    out["power_7p3_33p1"] = metrics.power(psd, freqs, start=7.3, stop=33.1)   # even sample count
    out["power_axis0"] = metrics.power(psd.T, freqs, start=2, stop=100, axis=0)
    out["power_norm_4_30"] = metrics.power_norm(psd, freqs, start=4, stop=30)
    ci = metrics.confidence_interval(psd, n_estimates=47, alpha=0.05)
    out["ci_lower"] = np.stack([c[0] for c in ci])
    out["ci_upper"] = np.stack([c[1] for c in ci])
    x = rng.standard_normal((2, 9000))
    out["x"] = x
    tr = Analytic(x, fs=500, chunksize=2500, axis=-1, width=12.5)
    out["signal"] = tr.signal.to_array(dtype=complex)
    out["amplitudes"] = tr.amplitudes.to_array()
    out["phases"] = tr.phases.to_array()
    save("g14_metrics_analytic.npz", **out)


# --------------------------------------------------------------------------
# G15 non-finite samples through the IIR passes (rows a6, a7): a NaN never
# leaves a section's state -- forward: NaN to the end of the stream; backward:
# the chunk that holds it, and the chunk before it (whose warm-up runs over it).
# Inputs are regenerated from the seed; stored are the non-finite masks as run
# boundaries and a decimation of the finite values.
# --------------------------------------------------------------------------
def g15_inputs():
    rng = np.random.default_rng(1515)
    x = rng.standard_normal((3, 600000))
    x[1, 250123] = np.nan                 # one NaN in the middle of chunk 1
    x[2, 590000:] = np.nan                # EDF-style NaN padding of the last record
    xi = x.copy()
    xi[0, 100] = np.inf                   # an Inf: NaN within a sample or two
    return x, xi


def g15_nonfinite():
    x, xi = g15_inputs()
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    cs = 200000
    out = {"sos": sos, "chunksize": np.array(cs)}
    for tag, data in (("nan", x), ("inf", xi)):
        y = np.concatenate(list(nm.sosfilt(producer(data, cs, -1), sos, -1)), -1)
        yy = np.concatenate(list(nm.sosfiltfilt(producer(data, cs, -1), sos, -1)), -1)
        for name, arr in (("sosfilt", y), ("sosfiltfilt", yy)):
            bad = ~np.isfinite(arr)
            # rows of (channel, start, stop) runs of non-finite samples
            runs = []
            for c in range(arr.shape[0]):
                edges = np.flatnonzero(np.diff(np.concatenate(([0], bad[c].astype(np.int8), [0]))))
                runs += [(c, a, b) for a, b in zip(edges[::2], edges[1::2])]
            out[f"{name}_{tag}_runs"] = np.array(runs, dtype=np.int64).reshape(-1, 3)
            out[f"{name}_{tag}_isnan_all"] = np.array(bool(np.all(np.isnan(arr[bad]))))
            out[f"{name}_{tag}_dec"] = arr[:, ::97]
    save("g15_nonfinite.npz", **out)


# --------------------------------------------------------------------------
# G16 Remez designs (row a13; filtering/fir.py:483-662): taps, tap-count estimate,
# band type and the derived edges for low / high / band-pass / band-stop /
# multiband specifications and keyword overrides; one filtered stream through the
# class API.
# --------------------------------------------------------------------------
REMEZ_CASES = [
    dict(bands=[0, 300, 400, 800, 900, 2500], desired=[0, 1, 0], fs=5000, gpass=.5, gstop=40),
    dict(bands=[0, 300, 400, 2500], desired=[1, 0], fs=5000),
    dict(bands=[0, 100, 200, 2500], desired=[0, 1], fs=5000, gpass=1, gstop=60),
    dict(bands=[0, 200, 300, 600, 700, 2500], desired=[1, 0, 1], fs=5000),
    dict(bands=[0, 100, 150, 400, 450, 800, 850, 1200, 1250, 2500], desired=[0, 1, 0, 1, 0],
         fs=5000, gpass=1, gstop=30),
    dict(bands=[0, 300, 400, 2500], desired=[1, 0], fs=5000, numtaps=101, grid_density=32),
]


def g16_remez():
    out = {}
    for i, kw in enumerate(REMEZ_CASES):
        filt = ref_fir.Remez(**kw)
        out[f"coeffs{i}"] = filt.coeffs
        out[f"numtaps{i}"] = np.array(filt.numtaps)
        out[f"btype{i}"] = np.array(filt.btype)
        out[f"fpass{i}"], out[f"fstop{i}"] = filt.fpass, filt.fstop
        out[f"cutoff{i}"], out[f"width{i}"] = filt.cutoff, np.array(filt.width)
        out[f"delta{i}"] = filt.delta
    rng = np.random.default_rng(1616)
    x = rng.standard_normal((3, 30011))
    out["x"] = x
    filt = ref_fir.Remez(**REMEZ_CASES[0])
    out["y0_same"] = filt(x, chunksize=4000, axis=-1, mode="same")
    res = filt(producer(x, 4000, axis=-1), chunksize=4000, axis=-1, mode="full")
    out["y0_full_lens"] = lengths(res)
    save("g16_remez.npz", **out)


# --------------------------------------------------------------------------
# G17 design-time inspection (filtering/mixins.py:226-317): impulse and frequency
# responses of an sos IIR, a ba IIR and a FIR.
# --------------------------------------------------------------------------
def g17_responses():
    out = {}
    filts = {"butter": ref_iir.Butter(fpass=[8, 30], fstop=[3, 60], fs=500, gpass=1, gstop=40),
             "notch": ref_iir.Notch(60, 8, 500),
             "kaiser": ref_fir.Kaiser(fpass=200, fstop=400, fs=5000, gpass=0.5, gstop=40)}
    for name, filt in filts.items():
        out[f"{name}_impulse"] = filt.impulse_response()
        for scale in ("dB", "abs", "complex"):
            freqs, gain, _ = filt.frequency_response(scale, 512, -100)
            out[f"{name}_freqs"] = freqs
            out[f"{name}_{scale}"] = gain
    save("g17_responses.npz", **out)


# --------------------------------------------------------------------------
# G18 the headline chain at a geometry where the build's zero-phase route engages: the
# reference's oaconvolve('same') -> sosfiltfilt (core/numerical.py:158-298 into :338-411) as
# chained producers, 1024 taps, the 6-section Butterworth band-pass of cfg-3, chunksize 65 536,
# six whole chunks + a ragged one.  Channel 1 rides an offset of 10^4 with a slow drift.
# x is float32-representable (stored as float32, exactly the float64 values the reference saw).
# --------------------------------------------------------------------------
def g18_chain_long():
    rng = np.random.default_rng(1801)
    cs, total = 65536, 6 * 65536 + 12345
    x = rng.standard_normal((2, total))
    x[1] += 1e4 + 2e3 * np.linspace(0.0, 1.0, total)
    x = x.astype(np.float32).astype(np.float64)
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    src = producer(x, cs, axis=-1)
    fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, axis=-1, shape=x.shape)
    pieces = list(nm.sosfiltfilt(fir, sos, -1))
    y = np.concatenate(pieces, axis=-1)
    assert y.shape == x.shape
    save("g18_chain_long.npz", x32=x.astype(np.float32), y=y, h=h, sos=sos, chunksize=np.int64(cs),
         piece_lengths=np.array([p.shape[-1] for p in pieces], dtype=np.int64))


# --------------------------------------------------------------------------
# G19 how far a non-finite input sample reaches in the reference's FIR -> sosfiltfilt chain: the FIR's
# overlap-add makes the whole SEGMENT (nfft - wlen + 1 input samples) of a non-finite sample
# non-finite (core/numerical.py:258-283), the cascade everything behind it, sosfiltfilt whole chunks.
# 256 taps (segments of 65 281 samples), chunksize 65 664, nine chunks; one placement per channel (the
# placements of tests/test_gpu_nonfinite.py::test_fir_chain_nan_reach_is_the_references).  Stored:
# the positions and, per channel and chunk, whether the reference's output chunk is non-finite (a
# mask depends on where the bad samples are, not on the other values).
# --------------------------------------------------------------------------
def g19_fir_chain_nonfinite():
    taps_n, cs, nchunks = 256, 65664, 9
    total = cs * (nchunks - 1) + 4321
    h = sps.firwin(taps_n, 0.3)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    step = 8 * 256 * 32 - taps_n + 1
    where = [3 * cs + 777, (4 * cs // step + 1) * step - 3, (4 * cs // step + 1) * step + 2, 5,
             (nchunks - 2) * cs + 9 * cs // 10, (nchunks - 1) * cs + 17, (nchunks - 1) * cs + 4000]
    C = len(where) + 2
    x = np.random.default_rng(77).standard_normal((C, total))
    for c, at in enumerate(where):
        x[c, at] = np.nan if c != 2 else np.inf
    x[len(where) - 1, where[-1]:] = np.nan
    x[C - 2, (nchunks - 2) * cs + 100] = np.nan
    src = producer(x, cs, axis=-1)
    fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), cs, axis=-1, shape=x.shape)
    y = np.concatenate(list(nm.sosfiltfilt(fir, sos, -1)), axis=-1)
    lost = np.array([[bool((~np.isfinite(y[c, k * cs:(k + 1) * cs])).all()) for k in range(nchunks)] for c in range(C)])
    partly = np.array([[bool((~np.isfinite(y[c, k * cs:(k + 1) * cs])).any()) for k in range(nchunks)] for c in range(C)])
    assert np.array_equal(lost, partly)          # sosfiltfilt loses whole chunks
    save("g19_fir_chain_nonfinite.npz", where=np.array(where, dtype=np.int64), chunksize=np.int64(cs),
         taps=np.int64(taps_n), total=np.int64(total), step=np.int64(step), lost_chunks=lost)


# --------------------------------------------------------------------------
# G20 how far a non-finite input sample reaches in the reference's resampler (core/numerical.py:523-632:
# scipy.signal.resample_poly chunk by chunk): the outputs whose taps touch it -- SciPy's zero padding
# of the window included (0 x NaN is NaN) -- whatever the chunking.  Per ratio: the reference's output
# for one input with samples at the stream's ends, at a chunk boundary, an Inf and a run to the end.
# --------------------------------------------------------------------------
def g20_resample_nonfinite():
    n = 20_011
    x = g20_input(n)
    out = {"n": np.int64(n)}
    for (L, M) in ((1, 5), (3, 2), (2, 1), (1, 25), (2, 7)):
        for cs in (5_000, 7_321):
            y = ref_rs.resample(x, L, M, 5000, chunksize=cs, axis=-1)
            out[f"lost_L{L}_M{M}_cs{cs}"] = np.packbits(~np.isfinite(y), axis=-1)      # (a mask: where, not what)
            out[f"nout_L{L}_M{M}"] = np.int64(y.shape[-1])
    save("g20_resample_nonfinite.npz", **out)


def g20_input(n):
    """The input of G20 (tests/test_oracle.py rebuilds it: the masks depend on where the bad samples are)."""
    x = np.random.default_rng(2020).standard_normal((4, n))
    x[0, 0] = np.nan
    x[0, 10_000] = np.nan            # a chunk boundary of chunksize 5 000
    x[1, n - 1] = np.inf
    x[1, 13_333] = np.nan
    x[2, 17_000:] = np.nan
    return x


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:                      # regenerate the named blocks only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_producer()
    g2_fir()
    g3_sosfilt()
    g4_sosfiltfilt()
    g5_resample()
    g6_periodogram()
    g7_welch()
    g8_stft()
    g9_design()
    g10_ba()
    g11_edf()
    g12_protools()
    g13_hilbert()
    g14_metrics_analytic()
    g15_nonfinite()
    g16_remez()
    g17_responses()
    g18_chain_long()
    g19_fir_chain_nonfinite()
    g20_resample_nonfinite()

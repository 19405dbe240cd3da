#!/usr/bin/env python3
"""Secondary measurements: the other kernels of the path at the shapes of
BASELINE.json configs 4 and 5 (per-GPU shard), against their own algorithmic
bytes.  Prints one JSON object per workload.  The
headline metric is bench.py's; this script feeds DESIGN.md / profiles/."""

import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    from openseize_amd.filtering.fir import Kaiser
    CH, N = 256, 1 << 20
    x = dev.synth_normal(CH, N, seed=3)
    out = []
    # --pmc-subset: only the device kernels the counter passes are collected for
    # (benchmarks/collect_pmc.sh), a few launches each
    subset = "--pmc-subset" in sys.argv

    # cfg-4: Welch PSD, nperseg 4096, 50 % overlap, hann, density (8 B / sample)
    nfft, fs = 4096, 4096.0
    w = sps.get_window("hann", nfft)
    scale = float(np.sqrt(1 / (fs * np.sum(w ** 2))))
    spec = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant",
                          _lib.SPEC_PSD_MEAN, CH)
    dt = timed(lambda: spec.push(x), 5)
    out.append({"workload": "cfg-4 Welch PSD 256 ch x 2^20, nfft 4096, 50 %",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 8 * CH * N / dt / 1e9})
    spec.close()

    # the same through the 512-thread fft8 kernel (OSZ_SPEC_V8=2), and the other
    # on-chip sizes (fs = nfft, 50 % overlap)
    os.environ["OSZ_SPEC_V8"] = "2"
    for nf in (4096, 512, 1024, 2048, 8192):
        wn = sps.get_window("hann", nf)
        sc = float(np.sqrt(1 / (float(nf) * np.sum(wn ** 2))))
        sp8 = dev.SpecStream(nf, nf, nf // 2, wn, sc, "constant", _lib.SPEC_PSD_MEAN, CH)
        dt = timed(lambda: sp8.push(x), 5)
        out.append({"workload": f"Welch PSD 256 ch x 2^20, nfft {nf}, 50 % (fft8 kernel)",
                    "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                    "algorithmic_GBps": 8 * CH * N / dt / 1e9})
        sp8.close()
    xs8 = x[:, : 1 << 18].contiguous()
    st8 = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant", _lib.SPEC_DFT_SEGMENTS, CH)
    dt = timed(lambda: st8.push(xs8), 5)
    out.append({"workload": "cfg-5 STFT 256 ch x 2^18, nfft 4096, 50 % (fft8 kernel)",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * xs8.shape[1] / dt / 1e6,
                "algorithmic_GBps": 24 * CH * xs8.shape[1] / dt / 1e9})
    st8.close()
    os.environ.pop("OSZ_SPEC_V8")

    # nfft = 2 fs of the reference's default resolution at fs = 500 ... 10 000 Hz:
    # the mixed-radix on-chip kernel (specmix.h) against the rocFFT staging route
    # (50 000, 65 536: halves beyond the LDS, pairs of sub-transforms -- specsplit.h)
    for nf in ((10000, 50000) if subset else (1000, 2000, 5000, 10000, 20000, 50000, 65536)):
        wn = sps.get_window("hann", nf)
        sc = float(np.sqrt(1 / (float(nf) * np.sum(wn ** 2))))
        for route in ("specmix kernel" if nf <= 20412 else "specsplit kernel", "rocFFT route"):
            if route == "rocFFT route":
                if subset or nf not in (1000, 10000, 50000):
                    continue
                os.environ["OSZ_SPEC_MIX"] = "0"
            spm = dev.SpecStream(nf, nf, nf // 2, wn, sc, "constant", _lib.SPEC_PSD_MEAN, CH)
            os.environ.pop("OSZ_SPEC_MIX", None)
            dt = timed(lambda: spm.push(x), 5)
            out.append({"workload": f"Welch PSD 256 ch x 2^20, nfft {nf}, 50 % ({route})",
                        "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                        "algorithmic_GBps": 8 * CH * N / dt / 1e9})
            spm.close()

    # cfg-5 part 1: polyphase downsample 5 -> 1, default Kaiser (113 taps)
    cutoff = 20480 / 10
    h = Kaiser(cutoff - cutoff / 10, cutoff + cutoff / 10, 20480, gpass=0.1, gstop=40).coeffs
    poly = dev.PolyStream(h, 1, 5, CH)
    dt = timed(lambda: poly.push(x, final=False), 5)
    out.append({"workload": f"cfg-5 downsample M=5 ({len(h)} taps) 256 ch x 2^20",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 9.6 * CH * N / dt / 1e9})
    poly.close()

    # decimation by 10 (5 kHz -> 500 Hz, the reference's demo): smaller tiles
    cutoff = 5000 / 20
    h10 = Kaiser(cutoff - cutoff / 10, cutoff + cutoff / 10, 5000, gpass=0.1, gstop=40).coeffs
    poly = dev.PolyStream(h10, 1, 10, CH)
    dt = timed(lambda: poly.push(x, final=False), 5)
    out.append({"workload": f"downsample M=10 ({len(h10)} taps) 256 ch x 2^20",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 8.8 * CH * N / dt / 1e9})
    poly.close()

    # rational resampling 3/2 (upsampling path of the same kernel)
    cut = 5000 / 6
    h32 = Kaiser(cut - cut / 10, cut + cut / 10, 5000, gpass=0.1, gstop=40).coeffs
    xs2 = x[:, : 1 << 19].contiguous()
    poly = dev.PolyStream(h32, 3, 2, CH)
    dt = timed(lambda: poly.push(xs2, final=False), 5)
    out.append({"workload": f"resample 3/2 ({len(h32)} taps) 256 ch x 2^19",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * xs2.shape[1] / dt / 1e6,
                "algorithmic_GBps": 20 * CH * xs2.shape[1] / dt / 1e9})
    poly.close()

    # cfg-5 part 2: STFT segments (complex128 out), nfft 4096, 50 % (24 B / sample)
    xs = x[:, : 1 << 18].contiguous()
    stft = dev.SpecStream(nfft, nfft, nfft // 2, w, scale, "constant",
                          _lib.SPEC_DFT_SEGMENTS, CH)
    dt = timed(lambda: stft.push(xs), 5)
    out.append({"workload": "cfg-5 STFT 256 ch x 2^18, nfft 4096, 50 %",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * xs.shape[1] / dt / 1e6,
                "algorithmic_GBps": 24 * CH * xs.shape[1] / dt / 1e9})
    stft.close()

    # mask compaction (K7), 50 % density
    idx = torch.arange(0, N, 2, dtype=torch.int64, device="cuda")
    dt = timed(lambda: dev.take(x, idx), 5)
    out.append({"workload": "K7 take 256 ch x 2^20, every 2nd sample",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 12 * CH * N / dt / 1e9})
    # FIR feeding the forward SOS pass: one fused kernel (16 B / sample) against
    # the two separate kernels (32 B / sample), cfg-3 filters
    hfir = sps.firwin(1024, 0.2)
    sosb = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    fir, iir = dev.FirStream(hfir, CH), dev.SosStream(sosb, CH)
    fo, so = torch.empty_like(x), torch.empty_like(x)
    dt = timed(lambda: dev.chain_forward(fir, iir, x, out=so), 10)
    out.append({"workload": "fused FIR(1024) -> forward SOS(6), 256 ch x 2^20 (osz_chain_forward)",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 16 * CH * N / dt / 1e9})

    def separate():
        fir.push(x, 0, out=fo)
        iir.forward(fo, out=so)
    dt = timed(separate, 10)
    out.append({"workload": "separate FIR(1024), forward SOS(6), 256 ch x 2^20",
                "ms_per_chunk": dt * 1e3, "Msamples_s": CH * N / dt / 1e6,
                "algorithmic_GBps": 32 * CH * N / dt / 1e9})
    fir.close()
    iir.close()
    if subset:
        for o in out:
            print(json.dumps(o))
        return

    # practical ceiling: device-to-device copy of one chunk (read 8 + write 8 B)
    y = torch.empty_like(x)
    dt = timed(lambda: y.copy_(x), 10)
    out.append({"workload": "d2d copy 256 ch x 2^20 (practical HBM ceiling)",
                "ms_per_chunk": dt * 1e3, "algorithmic_GBps": 16 * CH * N / dt / 1e9})

    # EDF record decode on the device: 64 channels x 1000
    # samples per 1 s record, 4000 records of little-endian int16 already in
    # HBM -> (64, 4e6) float64: 2 B read + 8 B written per sample
    nch_e, spr_e, nrec_e = 64, 1000, 4000
    raw = torch.randint(-2000, 2000, (nrec_e * nch_e * spr_e,), dtype=torch.int16, device="cuda")
    as_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    choff = as_dev((np.arange(nch_e) * spr_e).astype(np.int32))
    spr = as_dev(np.full(nch_e, spr_e, np.int32))
    lens = as_dev(np.full(nch_e, spr_e * nrec_e, np.int64))
    slope, offs = as_dev(np.full(nch_e, 0.25)), as_dev(np.full(nch_e, -3.0))
    width = spr_e * nrec_e
    dec = torch.empty((nch_e, width), dtype=torch.float64, device="cuda")
    lib = _lib.load()

    def edf():
        _lib.check(lib.osz_edf_decode(dev.ptr(raw), nch_e * spr_e, nch_e, dev.ptr(choff), dev.ptr(spr),
                                      dev.ptr(slope), dev.ptr(offs), dev.ptr(lens), 0, 0, width,
                                      ctypes.c_double(float("nan")), dev.ptr(dec), dec.stride(0),
                                      dev.stream_ptr()))
    dt = timed(edf, 5)
    out.append({"workload": "EDF decode 64 ch x 4e6 samples (int16 records in HBM -> f64)",
                "ms_per_chunk": dt * 1e3, "Msamples_s": nch_e * width / dt / 1e6,
                "algorithmic_GBps": 10 * nch_e * width / dt / 1e9})

    # transfer-function filter through the public API, device-resident:
    # Notch (order 2, one section) zero-phase on 64 ch x 2^20
    from openseize_amd import producer as make_producer
    from openseize_amd.filtering.iir import Notch
    xs = x[:64]
    notch = Notch(fstop=60, width=4, fs=5000)

    def ba():
        for _ in notch(make_producer(xs, 1 << 18, -1), 1 << 18, -1, dephase=True):
            pass
    dt = timed(ba, 3)
    out.append({"workload": "Notch filtfilt (ba, order 2) 64 ch x 2^20 via the producer API, chunksize 2^18",
                "ms_per_chunk": dt * 1e3, "Msamples_s": 64 * N / dt / 1e6,
                "algorithmic_GBps": 32 * 64 * N / dt / 1e9})

    # cfg-1, HOST-FED through the public API (ndarray in -> ndarray out):
    # 16 ch x 1e6, 256-tap FIR, chunksize 30000, mode same.  PCIe + per-chunk
    # launch overhead included; the reference needs 0.309 s for this.
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    xh = np.random.default_rng(0).standard_normal((16, 1_000_000))
    hh = sps.firwin(256, 0.2)

    def cfg1():
        return np.concatenate(list(nm.oaconvolve(producer(xh, 30000, -1), hh, -1, "same")), -1)

    cfg1()
    t0 = time.perf_counter()
    for _ in range(3):
        yh = cfg1()
    dt = (time.perf_counter() - t0) / 3
    out.append({"workload": "cfg-1 host-fed oaconvolve 16 ch x 1e6, 256 taps, chunksize 30000",
                "seconds": dt, "Msamples_s": xh.size / dt / 1e6,
                "max_abs_err_vs_numpy": float(np.max(np.abs(
                    yh[0] - np.convolve(xh[0], hh, "same"))))})
    # same data, one big chunk (what a user with enough memory would do)
    def cfg1_big():
        return np.concatenate(list(nm.oaconvolve(producer(xh, 1_000_000, -1), hh, -1, "same")), -1)
    cfg1_big()
    t0 = time.perf_counter()
    for _ in range(3):
        cfg1_big()
    dt = (time.perf_counter() - t0) / 3
    out.append({"workload": "cfg-1 host-fed, chunksize 1e6", "seconds": dt,
                "Msamples_s": xh.size / dt / 1e6})
    # host-fed steady state: 16 ch x 2^23 samples (1 GiB) of ndarray through the
    # 256-tap FIR in chunks of 2^18 ... 2^20 -- pinned ring + H2D / compute / D2H
    # streams (dev.HostPipe); PCIe Gen5 x16 bounds it at ~63 GB/s / 16 B
    xl = np.random.default_rng(1).standard_normal((16, 1 << 23))
    for cs in (1 << 18, 1 << 20):
        def fed():
            total = 0
            for piece in nm.oaconvolve(producer(xl, cs, -1), hh, -1, "same"):
                total += piece.shape[-1]
            return total
        fed()
        t0 = time.perf_counter()
        got = fed()
        dt = time.perf_counter() - t0
        assert got == xl.shape[-1]
        out.append({"workload": f"host-fed oaconvolve 16 ch x 2^23, 256 taps, chunksize {cs}",
                    "seconds": dt, "Msamples_s": xl.size / dt / 1e6,
                    "pcie_GBps_each_way": 8 * xl.size / dt / 1e9})
    # the same stream through sosfiltfilt (6 sections) and through the FIR -> sosfiltfilt
    # chain of cfg-3, host in / host out
    sos6 = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    from functools import partial
    from openseize_amd.core.producer import producer as mkpro
    for name, run in (
        ("host-fed sosfiltfilt (6 sections)",
         lambda: nm.sosfiltfilt(producer(xl, 1 << 18, -1), sos6, -1)),
        ("host-fed FIR(256) -> sosfiltfilt(6) chain",
         lambda: nm.sosfiltfilt(
             mkpro(partial(nm.oaconvolve, window=hh, axis=-1, mode="same"), 1 << 18, -1,
                   shape=xl.shape, pro=producer(xl, 1 << 18, -1)), sos6, -1))):
        def fed2():
            return sum(piece.shape[-1] for piece in run())
        fed2()
        t0 = time.perf_counter()
        got = fed2()
        dt = time.perf_counter() - t0
        assert got == xl.shape[-1], (name, got)
        out.append({"workload": f"{name} 16 ch x 2^23, chunksize 262144",
                    "seconds": dt, "Msamples_s": xl.size / dt / 1e6,
                    "pcie_GBps_each_way": 8 * xl.size / dt / 1e9})
    # a three-stage host chain: sosfilt(4 sections) -> FIR(256) -> downsample by 5, every stage
    # a producer; the stages hand CUDA tensors on (dev.chain_aware), OSZ_HOST_CHAIN=0: ndarrays
    from openseize_amd.filtering.fir import Kaiser as _Kaiser
    sos4 = sps.butter(4, 0.4, output="sos")

    def three_stage():
        a = producer(xl, 1 << 18, -1)
        b = mkpro(partial(nm.sosfilt, a, sos4, -1), 1 << 18, -1, shape=a.shape)
        c = mkpro(partial(nm.oaconvolve, b, hh, -1, "same"), 1 << 18, -1, shape=a.shape)
        d = mkpro(partial(nm.polyphase_resample, c, 1, 5, 5000.0, _Kaiser, -1), 1 << 18, -1,
                  shape=(xl.shape[0], -(-xl.shape[1] // 5)))
        return sum(piece.shape[-1] for piece in d)

    for label, env in (("stages hand CUDA tensors on", None), ("stages hand ndarrays on (OSZ_HOST_CHAIN=0)", "0")):
        if env is not None:
            os.environ["OSZ_HOST_CHAIN"] = env
        try:
            three_stage()
            t0 = time.perf_counter()
            got = three_stage()
            dt = time.perf_counter() - t0
        finally:
            os.environ.pop("OSZ_HOST_CHAIN", None)
        assert got == -(-xl.shape[1] // 5)
        out.append({"workload": f"host-fed sosfilt -> FIR -> downsample 5 chain 16 ch x 2^23, {label}",
                    "seconds": dt, "Msamples_s": xl.size / dt / 1e6})
    # the README's pipeline from ndarrays: Kaiser low-pass -> Butterworth band-pass (zero phase)
    # -> psd, every stage through the class API
    from openseize_amd.filtering.iir import Butter as _Butter
    from openseize_amd.spectra.estimators import psd as _psd
    xr = xl[:, :3_000_000]

    def readme():
        pro = producer(xr, 1 << 18, -1)
        lp = _Kaiser(fpass=500, fstop=600, fs=5000)(pro, chunksize=1 << 18, axis=-1)
        bp = _Butter(fpass=[8, 30], fstop=[3, 60], fs=5000)(lp, chunksize=1 << 18, axis=-1, dephase=True)
        return _psd(bp, fs=5000, axis=-1)

    for label, envs in (("chain-aware", {}), ("stages apart (OSZ_HOST_CHAIN=0 OSZ_CHAIN_API=0)",
                                              {"OSZ_HOST_CHAIN": "0", "OSZ_CHAIN_API": "0"})):
        os.environ.update(envs)
        try:
            readme()
            t0 = time.perf_counter()
            readme()
            dt = time.perf_counter() - t0
        finally:
            for k_ in envs:
                os.environ.pop(k_, None)
        out.append({"workload": f"README pipeline from ndarrays, 16 ch x 3e6: FIR -> zero-phase IIR -> psd, {label}",
                    "seconds": dt, "Msamples_s": xr.size / dt / 1e6})
    # host-fed Welch PSD at the reference's default resolution, fs = 5 kHz (nfft 10 000)
    from openseize_amd.spectra.estimators import psd
    psd(xl, fs=5000.0, axis=-1)
    t0 = time.perf_counter()
    cnt, _, _ = psd(xl, fs=5000.0, axis=-1)
    dt = time.perf_counter() - t0
    out.append({"workload": f"host-fed psd 16 ch x 2^23, fs 5000 (nfft 10000), {cnt} segments",
                "seconds": dt, "Msamples_s": xl.size / dt / 1e6,
                "pcie_GBps_up": 8 * xl.size / dt / 1e9})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()

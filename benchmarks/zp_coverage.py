#!/usr/bin/env python3
"""Which route do EEG-realistic cascades take, and what does a 256 x 2^20 chunk cost there?
One JSON line per (filter, setting):

    PYTHONPATH=. python benchmarks/zp_coverage.py > profiles/rNN_zp_coverage.jsonl

  chain        1024-tap FIR -> the cascade's sosfiltfilt (the headline's shape) at the C ABI: `zp` = ONE kernel per
               chunk (osz_chain_zp_step + seal), else `chain_step` = the fused FIR + forward kernel
               with the backward pass beside it (the public generators run FIR and cascade APART there
               since round 5: no slower, and the reference FIR's NaN reach; osz_chain_step: `forward_kernel` names which fused
               kernel it is, the spectral one or the time-domain one)
  sosfiltfilt  the cascade alone: `zp` (the identity as the FIR) or `dual` (osz_sosfiltfilt_step)
Filters: the headline; Butter(fpass=[8, 30], fstop=[3, 60], fs=500) (SURVEY 8d's class-API cfg-3);
Cheby1([200, 600] / [150, 650] Hz at 2500 Hz, the reference's tests/test_iir.py:132-158); 0.5-4 Hz
at 5 kHz (SURVEY 7's stress filter); an 8-section Butterworth band-pass; Notch(60, 8, 500)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))

CHUNK, C = 1 << 20, 256


def filters():
    import scipy.signal as sps
    from openseize_amd.core import numerical as nm
    from openseize_amd.filtering import iir
    return [
        ("headline: butter(6, [0.05, 0.3]) band-pass", sps.butter(6, [0.05, 0.3], "bandpass", output="sos")),
        ("Butter(fpass=[8, 30], fstop=[3, 60], fs=500)", iir.Butter(fpass=[8, 30], fstop=[3, 60], fs=500, gpass=1, gstop=40).coeffs),
        ("Cheby1(fpass=[200, 600], fstop=[150, 650], fs=2500)", iir.Cheby1(fpass=[200, 600], fstop=[150, 650], fs=2500).coeffs),
        ("butter(4, 0.5-4 Hz at 5 kHz) band-pass", sps.butter(4, [0.0002, 0.0016], "bandpass", output="sos")),
        ("butter(8, [0.05, 0.3]) band-pass, 8 sections", sps.butter(8, [0.05, 0.3], "bandpass", output="sos")),
        ("Notch(fstop=60, width=8, fs=500)", nm._ba_to_sos(iir.Notch(60, 8, 500).coeffs)[0]),
    ]


def main():
    import numpy as np
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    from sweep_chain import timed
    lib = _lib.load()
    h = sps.firwin(1024, 0.2)
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    ys = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
    for name, sos in filters():
        sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
        for setting, taps in (("chain", h), ("sosfiltfilt", np.array([1.0, 0.0]))):
            fir, iir = dev.FirStream(taps, C), dev.SosStream(sos, C)
            row = {"filter": name, "setting": setting, "sections": int(sos.shape[0]), "channels": C, "chunksize": CHUNK}
            try:
                lag = dev.chain_zp_lag(fir, iir)
                if setting == "chain":       # what sosfilt behind the FIR runs on (osz_chain_forward_route)
                    f2, i2 = dev.FirStream(taps, C), dev.SosStream(sos, C)
                    row["forward_route"] = ("time scan", "pair of blocks", "one block")[dev.chain_forward_route(f2, i2)]
                    f2.close()
                    i2.close()
                iir.set_state_scaled(ring[0], 0)
                if lag >= 0:
                    dev.chain_zp_open(fir, iir, 0)

                    def step(k):
                        dev.chain_zp_step(fir, iir, ring[k % 3], out=ys[k % 4][:, :CHUNK - lag],
                                          tail=ys[(k - 1) % 4][:, CHUNK - lag:])
                        if k >= 2:
                            dev.chain_zp_seal(fir, iir, ys[(k - 2) % 4], (k - 2) * CHUNK, 0, CHUNK)
                    row.update(route="zp", lag=lag)
                elif setting == "chain":
                    for k in range(2):
                        dev.chain_forward(fir, iir, ring[k % 3], out=ys[k % 4])

                    def step(k):
                        dev.chain_step(fir, iir, ring[k % 3], ys[(k - 2) % 4], ys[(k - 1) % 4], f_out=ys[k % 4],
                                       y_out=ys[3] if False else out_y, defer=True)
                    out_y = torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda")
                    row.update(route="chain_step")
                else:
                    def step(k):
                        iir.step(ring[k % 3], ys[(k + 1) % 3], ys[(k + 2) % 3], f_out=ys[k % 3], y_out=ys[3])
                    row.update(route="dual")
                _lib.check(lib.osz_profile_reset())
                _lib.check(lib.osz_profile_enable(1))
                dt = timed(step if lag >= 0 or setting != "chain" else (lambda k: step(k + 2)), 24, 6)
                _lib.check(lib.osz_profile_enable(0))
                if row["route"] == "chain_step":
                    dev.chain_wait(iir)
                    import ctypes
                    for kn in ("chain_fwd", "chain_spec", "chain_kernel"):
                        n, ms = ctypes.c_int64(), ctypes.c_double()
                        if lib.osz_profile_query(kn.encode(), ctypes.byref(n), ctypes.byref(ms)) == 0 and n.value:
                            row.setdefault("members_ms", {})[kn] = ms.value / n.value
                row.update(ms_per_chunk=dt * 1e3, Gsamples_s=C * CHUNK / dt / 1e9)
            except Exception as exc:
                row["error"] = str(exc)[:200]
            finally:
                fir.close()
                iir.close()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

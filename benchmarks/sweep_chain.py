#!/usr/bin/env python3
"""Channel sweep of the three BASELINE workloads with this round's kernels, steady state,
inputs resident: how the kernels hold up away from the headline shape.  The 8-GPU channel
split of the metric's 256 channels leaves 32 per GPU; the reference's own
tutorials run 4 channels.  One JSON line per (workload, channels):

    PYTHONPATH=. python benchmarks/sweep_chain.py > profiles/rNN_channel_sweep.jsonl

  chain  FIR(1024) -> sosfiltfilt(6 sections) on osz_chain_zp_step (+ a seal per chunk): 256 / C chunks of
         2^20 samples per launch, as numerical.sosfiltfilt steps a resident stream of C channels (round 5:
         `chunks_per_launch`); `chain_1` is the same with every chunk its own launch (rounds 3-4)
  fir    FIR(1024) overlap-add alone (osz_fir_push), 256 / C chunks per push as numerical.oaconvolve joins the
         adjacent views of a resident stream (round 5)
  welch  Welch PSD nperseg 4096, 50 % overlap, segment average (osz_spec_push), pushes of 2^28 / C samples per
         channel as psd() makes them of a resident array (round 5; rounds 3-4: 2^20)
  sosfiltfilt  the 6-section cascade alone, zero phase: osz_chain_zp_step with the identity as its FIR,
         256 / C chunks per launch (what numerical.sosfiltfilt runs on long resident streams; `dual_ms`:
         the separate kernels' osz_sosfiltfilt_step per chunk on the same box)
`rel_256` is the rate relative to the same workload at 256 channels (printed last)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNK = 1 << 20


def timed(step, steps, warm):
    import torch
    for k in range(warm):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def chain(C, steps=24, warm=6, group=None):
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    from openseize_amd.core import numerical as nm
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    g = nm._zp_group(C) if group is None else group       # chunks per launch (the generators' own rule)
    n = g * CHUNK
    ring = [dev.synth_normal(C, n, seed=0, n0=k * n) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    lag = dev.chain_zp_lag(fir, iir)
    ys = [torch.zeros((C, n), dtype=torch.float64, device="cuda") for _ in range(3)]
    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)

    def step(k):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=ys[k % 3][:, :n - lag], tail=ys[(k - 1) % 3][:, n - lag:])
        if k >= 1:                                   # the chunks that have just become complete, one seal each
            for j in range(g):
                dev.chain_zp_seal(fir, iir, ys[(k - 1) % 3][:, j * CHUNK:(j + 1) * CHUNK], ((k - 1) * g + j) * CHUNK, 0, CHUNK)

    dt = timed(step, steps, warm) / g
    fir.close()
    iir.close()
    return dt, 48, {"chunks_per_launch": g}


def chain_1(C, steps=24, warm=6):
    dt, bps, extra = chain(C, steps, warm, group=1)
    return dt, bps, extra


def sosfiltfilt(C, steps=24, warm=6):
    import numpy as np
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    from openseize_amd.core import numerical as nm
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    g = nm._zp_group(C)                                  # chunks per launch, as the generator steps a resident stream
    n = g * CHUNK
    ring = [dev.synth_normal(C, n, seed=0, n0=k * n) for k in range(3)]
    fir, iir = dev.FirStream(np.array([1.0, 0.0]), C), dev.SosStream(sos, C)
    lag = dev.chain_zp_lag(fir, iir)
    ys = [torch.zeros((C, n), dtype=torch.float64, device="cuda") for _ in range(3)]
    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)

    def step(k):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=ys[k % 3][:, :n - lag], tail=ys[(k - 1) % 3][:, n - lag:])
        if k >= 1:
            for j in range(g):
                dev.chain_zp_seal(fir, iir, ys[(k - 1) % 3][:, j * CHUNK:(j + 1) * CHUNK], ((k - 1) * g + j) * CHUNK, 0, CHUNK)

    dt = timed(step, steps, warm) / g
    fir.close()
    iir.close()
    del ring, ys
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    ys = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
    iir = dev.SosStream(sos, C)
    iir.set_state_scaled(ring[0], 0)
    dt2 = timed(lambda k: iir.step(ring[k % 3], ys[(k + 1) % 3], ys[(k + 2) % 3], f_out=ys[k % 3], y_out=ys[3]), steps, warm)
    iir.close()
    return dt, 32, {"dual_ms": dt2 * 1e3, "chunks_per_launch": g}


def fir_only(C, steps=24, warm=6):
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    from openseize_amd.core import numerical as nm
    h = sps.firwin(1024, 0.2)
    g = nm._zp_group(C)                  # chunks per push, as numerical.oaconvolve joins a resident stream's views
    n = g * CHUNK
    ring = [dev.synth_normal(C, n, seed=0, n0=k * n) for k in range(3)]
    fir = dev.FirStream(h, C)
    out = torch.zeros((C, n), dtype=torch.float64, device="cuda")
    dt = timed(lambda k: fir.push(ring[k % 3], 0, out=out), steps, warm) / g
    fir.close()
    return dt, 16, {"chunks_per_launch": g}


def welch(C, steps=24, warm=6):
    import scipy.signal as sps
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    win = sps.get_window("hann", 4096)
    # what psd() pushes of a resident array: views of up to 2^28 elements (numerical._batched)
    g = max(1, min(1 << 24, (1 << 28) // C) // CHUNK)
    ring = [dev.synth_normal(C, g * CHUNK, seed=0, n0=k * g * CHUNK) for k in range(3)]
    spec = dev.SpecStream(4096, 4096, 2048, win, 1.0 / (4096.0 * float((win ** 2).sum())), "constant",
                          _lib.SPEC_PSD_MEAN, C)
    dt = timed(lambda k: spec.push(ring[k % 3]), steps, warm) / g
    spec.close()
    return dt, 8, {"chunks_per_launch": g}


if __name__ == "__main__":
    rows = []
    for name, fn in (("chain", chain), ("chain_1", chain_1), ("fir", fir_only), ("welch", welch), ("sosfiltfilt", sosfiltfilt)):
        for C in (4, 8, 16, 32, 64, 128, 256):
            try:
                dt, bps, *extra = fn(C)
            except Exception as exc:      # a workload this build does not offer in that form
                print(json.dumps({"workload": name, "channels": C, "error": str(exc)[:200]}), flush=True)
                continue
            rows.append({"workload": name, "channels": C, "chunksize": CHUNK, "ms_per_chunk": dt * 1e3,
                         "Gsamples_s": C * CHUNK / dt / 1e9, "algorithmic_TBps": bps * C * CHUNK / dt / 1e12})
            for e in extra:
                rows[-1].update(e)
    base = {r["workload"]: r["Gsamples_s"] for r in rows if r["channels"] == 256}
    for r in rows:
        r["rel_256"] = r["Gsamples_s"] / base[r["workload"]] if r["workload"] in base else None
        print(json.dumps(r), flush=True)

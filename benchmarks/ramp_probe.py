"""Per-step durations of the headline step from the first launch on (HIP events on the launch
stream): how long the first launches of a fresh process run slower than the steady state, and
whether device work in front of them (the input ring's synthesis) changes that.
    python benchmarks/ramp_probe.py [idle_seconds_before_the_first_step]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CHUNK, C = 1 << 20, 256


def main():
    idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    lag = dev.chain_zp_lag(fir, iir)
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    fwd = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle)
    n = 80
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=fwd[k % 4][:, :CHUNK - lag], tail=fwd[(k - 1) % 4][:, CHUNK - lag:])
        if k >= 2:
            dev.chain_zp_seal(fir, iir, fwd[(k - 2) % 4], (k - 2) * CHUNK, 0, CHUNK)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
    if len(sys.argv) > 2:
        # a second batch on the same buffers after an idle gap / on fresh buffers
        time.sleep(float(sys.argv[2]))
        if len(sys.argv) > 3:
            ring = [dev.synth_normal(C, CHUNK, seed=1, n0=k * CHUNK) for k in range(3)]
            fwd = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
        ev[0].record()
        for k in range(n, 2 * n):
            dev.chain_zp_step(fir, iir, ring[k % 3], out=fwd[k % 4][:, :CHUNK - lag], tail=fwd[(k - 1) % 4][:, CHUNK - lag:])
            dev.chain_zp_seal(fir, iir, fwd[(k - 2) % 4], (k - 2) * CHUNK, 0, CHUNK)
            ev[k + 1 - n].record()
        torch.cuda.synchronize()
        ms2 = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
        print(json.dumps({"second_batch_after_idle_s": float(sys.argv[2]), "fresh_buffers": len(sys.argv) > 3,
                          "ms_steps_0_9": [round(v, 3) for v in ms2[:10]], "mean_5_24": round(sum(ms2[5:25]) / 20, 4),
                          "mean_60_79": round(sum(ms2[60:80]) / 20, 4)}))
    print(json.dumps({"idle_s": idle, "ms_steps_0_9": [round(v, 3) for v in ms[:10]],
                      "mean_5_24": round(sum(ms[5:25]) / 20, 4), "mean_25_44": round(sum(ms[25:45]) / 20, 4),
                      "mean_60_79": round(sum(ms[60:80]) / 20, 4), "every_10th": [round(v, 3) for v in ms[::10]]}))


if __name__ == "__main__":
    main()

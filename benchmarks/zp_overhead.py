"""Fixed cost of one zero-phase chain launch: time per step over chunk lengths at a fixed number of
workgroups, fitted as  a + b * (pairs of blocks per run).  `a` is what a launch costs besides its
whole pairs (start-up, opening and closing pair, carry export, the tail of the slowest run).

    PYTHONPATH=. python benchmarks/zp_overhead.py
"""
import json
import time

import numpy as np
import scipy.signal as sps
import torch

from openseize_amd import _device as dev


def step_time(C, cs, steps=30, warm=6):
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, cs, seed=0, n0=k * cs) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)
    y = torch.zeros((C, cs), dtype=torch.float64, device="cuda")
    for k in range(warm):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        dev.chain_zp_step(fir, iir, ring[(warm + k) % 3], out=y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fir.close()
    iir.close()
    return dt


if __name__ == "__main__":
    S2 = 27 * 256      # samples per transform: a block of 27 rows (the pair kernel: 2 * 2816)
    for C in (32, 256):
        nruns = 512 // C
        rows = []
        for cs in (1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20, 3 << 19, 1 << 21):
            if C * cs * 8 * 4 > 40e9:
                continue
            dt = step_time(C, cs)
            pairs = cs / S2 / nruns
            rows.append((pairs, dt))
            print(json.dumps({"channels": C, "chunksize": cs, "pairs_per_run": pairs, "us_per_step": dt * 1e6}), flush=True)
        A = np.array([[1.0, p] for p, _ in rows])
        a, b = np.linalg.lstsq(A, np.array([d for _, d in rows]), rcond=None)[0]
        print(json.dumps({"channels": C, "fixed_us": a * 1e6, "us_per_pair": b * 1e6}), flush=True)

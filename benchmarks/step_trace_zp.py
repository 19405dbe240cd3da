"""Per-step durations of the default bench loop (osz_chain_zp_step + osz_chain_zp_seal) right after a
short warm-up: events on the caller's stream after every step.
    PYTHONPATH=. python benchmarks/step_trace_zp.py [warmup] [steps]"""
import sys
import time

import scipy.signal as sps
import torch

import bench
from openseize_amd import _device as dev

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
C, CHUNK = bench.C_PER_GPU, bench.CHUNK
h = sps.firwin(bench.NTAPS, 0.2)
sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
lag = dev.chain_zp_lag(fir, iir)
ys = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
iir.set_state_scaled(ring[0], 0)
dev.chain_zp_open(fir, iir, 0)


def step(k):
    dev.chain_zp_step(fir, iir, ring[k % 3], out=ys[k % 4][:, :CHUNK - lag], tail=ys[(k - 1) % 4][:, CHUNK - lag:])
    if k >= 2:
        dev.chain_zp_seal(fir, iir, ys[(k - 2) % 4], (k - 2) * CHUNK, 0, CHUNK)


k = 0
for _ in range(W):
    step(k)
    k += 1
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
t0 = time.perf_counter()
evs[0].record()
for i in range(K):
    step(k)
    k += 1
    evs[i + 1].record()
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"W={W} K={K}: total {total * 1e3:.3f} ms = {total / K * 1e3:.4f} ms/step")
print("gpu ms between step ends:", " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.3f}" for i in range(K)))

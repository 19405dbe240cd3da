"""Per-step durations of the default bench loop (osz_chain_step, OSZ_CHAIN_DEFER) right after a
short warm-up: events on the caller's stream after every step."""
import os
import sys
import time

import scipy.signal as sps

sys.path.insert(0, ".")
import bench                                              # noqa: E402


def main():
    import torch
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    lib = _lib.load()
    W, K = int(sys.argv[1]), int(sys.argv[2])
    prof = len(sys.argv) > 3 and sys.argv[3] == "prof"
    C, CHUNK = bench.C_PER_GPU, bench.CHUNK
    h = sps.firwin(bench.NTAPS, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    fwd = [torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda") for _ in range(4)]
    y_out = torch.zeros_like(fwd[0])

    def step(k):
        if k < 2:
            dev.chain_forward(fir, iir, ring[k % 3], out=fwd[k % 4])
        else:
            dev.chain_step(fir, iir, ring[k % 3], fwd[(k - 2) % 4], fwd[(k - 1) % 4],
                           f_out=fwd[k % 4], y_out=y_out, defer=True)

    iir.set_state_scaled(ring[0], 0)
    k = 0
    for _ in range(W):
        step(k)
        k += 1
    torch.cuda.synchronize()
    if prof:
        _lib.check(lib.osz_profile_reset())
        _lib.check(lib.osz_profile_enable(1))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(K):
        step(k)
        k += 1
        evs[i + 1].record()
        host.append(time.perf_counter() - t0)
    dev.chain_wait(iir)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"W={W} K={K} prof={prof}: total {total * 1e3:.3f} ms = {total / K * 1e3:.4f} ms/step")
    print("gpu ms between step ends :", " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.2f}" for i in range(K)))
    print("host ms at step issue end:", " ".join(f"{t * 1e3:.2f}" for t in host))


if __name__ == "__main__":
    main()

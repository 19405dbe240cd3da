#!/usr/bin/env python3
"""Experiment: the fused FIR -> forward-SOS kernel of chunk k (compute / latency
bound, 2.2 TB/s) BESIDE the backward pass of chunk k-2 (memory bound) on a
second HIP stream -- do the two fill each other's gaps?  cfg-3 shapes, steady
state, three variants per box:
  unfused   fir_oa + fir_seam + sos_dual                (the bench's step)
  fused     chain_forward ; backward                    (one stream)
  overlap   chain_forward || backward of chunk k-2      (two streams, from Python)
  abi       the same as ONE call, osz_chain_step        (the handle's own side stream)
  abi_defer with OSZ_CHAIN_DEFER and four forward buffers (y one step late)
    python benchmarks/overlap_fused_probe.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.signal as sps
    import torch
    from openseize_amd import _device as dev

    C, CHUNK, steps = 256, 1 << 20, 100
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
    out = []
    for variant in ("unfused", "overlap", "abi", "abi_defer") * 3:
        fir = dev.FirStream(h, C)
        iir = dev.SosStream(sos, C)
        fir_out = torch.empty((C, CHUNK), dtype=torch.float64, device="cuda")
        fwd = [torch.empty_like(fir_out) for _ in range(4)]
        y_out = torch.empty_like(fir_out)
        main_s = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        done_fwd = [None] * 4        # event: forward chunk written
        done_bwd = [None] * 4        # event: last backward pass that read this buffer

        def step(k):
            if variant == "unfused":
                fir.push(ring[k % 3], 0, out=fir_out)
                if k < 2:
                    iir.forward(fir_out, out=fwd[k % 4])
                else:
                    iir.step(fir_out, fwd[(k - 2) % 4], fwd[(k - 1) % 4], f_out=fwd[k % 4], y_out=y_out)
                return
            if variant == "fused":
                dev.chain_forward(fir, iir, ring[k % 3], out=fwd[k % 4])
                if k >= 2:
                    iir.backward(fwd[(k - 2) % 4], fwd[(k - 1) % 4], out=y_out)
                return
            if variant in ("abi", "abi_defer"):   # through the library's own side stream (osz_chain_step)
                if k < 2:
                    dev.chain_forward(fir, iir, ring[k % 3], out=fwd[k % 4])
                else:
                    dev.chain_step(fir, iir, ring[k % 3], fwd[(k - 2) % 4], fwd[(k - 1) % 4],
                                   f_out=fwd[k % 4], y_out=y_out, defer=variant == "abi_defer")
                return
            # overlap: backward(k-2) on the side stream while chain(k) runs on the main one
            if k >= 2:
                side.wait_event(done_fwd[(k - 1) % 4])
                with torch.cuda.stream(side):
                    iir.backward(fwd[(k - 2) % 4], fwd[(k - 1) % 4], out=y_out)
                    ev = torch.cuda.Event()
                    ev.record(side)
                done_bwd[(k - 2) % 4] = done_bwd[(k - 1) % 4] = ev
            if done_bwd[k % 4] is not None:
                main_s.wait_event(done_bwd[k % 4])
            dev.chain_forward(fir, iir, ring[k % 3], out=fwd[k % 4])
            ev = torch.cuda.Event()
            ev.record(main_s)
            done_fwd[k % 4] = ev

        iir.set_state_scaled(ring[0], 0)
        k = 0
        for _ in range(6):
            step(k)
            k += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(k)
            k += 1
        if variant == "abi_defer":
            dev.chain_wait(iir)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        bits, fsum = dev.checksum(y_out)
        out.append({"variant": variant, "ms_per_step": dt * 1e3, "Gsamples_s": C * CHUNK / dt / 1e9,
                    "checksum": f"{bits:#018x}"})
        fir.close()
        iir.close()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()

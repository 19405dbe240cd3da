#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path.

Workload (BASELINE.json configs[2], the configuration the metric is quoted
on): 256 channels x float64, chunksize 2^20, 1024-tap FIR overlap-add
(`firwin(1024, 0.2)`) chained into a 6-section Butterworth band-pass
`sosfiltfilt` (`butter(6, [0.05, 0.3], 'bandpass')`).  A "step" is one chunk
(256 x 2^20 channel-samples) through the whole chain in steady state:

    FIR push(chunk k) -> [SOS forward(chunk k) + SOS backward(chunk k-2;
    chunk-local warm-up over forward chunk k-1)] in one launch

Inputs are synthesised on the device before the timed region (a ring of
resident chunks keyed by (seed, channel, sample)); outputs land in a resident
buffer.  With N GPUs every rank runs the same 256-channel shard workload on its
own GPU (channels are independent: no data-path collective, weak scaling).

Prints ONE JSON line (see README / the driver contract) with `roofline` for
the dominant kernel and, at N=1 on rank 0, `cpu_baseline` (the CPU oracle).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C_PER_GPU = 256
CHUNK = 1 << 20
NTAPS = 1024
# algorithmic HBM bytes per channel-sample of one launch
KERNEL_BYTES = {"fir_oa": 16, "sos_dual": 32, "sos_fwd": 16, "sos_bwd": 16,
                "sos_warmup": 0, "fir_seam": 0}
CHAIN_BYTES = 48         # FIR 16 + sosfiltfilt 32 (unfused)


def _cpu_worker(args):
    """One host core: the CPU oracle chain on `ch` channels x `n` samples."""
    h, sos, ch, n, seed = args
    from oracle import oracle as orc
    x = np.random.default_rng(seed).standard_normal((ch, n))
    t0 = time.perf_counter()
    y = np.concatenate(orc.oaconvolve(x, h, "same"), axis=-1)
    orc.sosfiltfilt(y, sos, CHUNK)
    return time.perf_counter() - t0


def cpu_baseline(h, sos):
    """The CPU oracle (oracle/: NumPy FFT overlap-add + C DF2T loops) on a
    bounded sample of the same workload: once on one core, once with one
    process per available core (at most 16) over channel shards.  Runs BEFORE
    the GPU is initialised (worker processes are forked)."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    ch, n = 16, 1 << 21
    avail = len(os.sched_getaffinity(0))
    procs = max(1, min(16, avail))
    single = _cpu_worker((h, sos, ch, n, 0))
    value_1 = ch * n / single / 1e6
    value_p, wall = value_1, single
    if procs > 1:
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(procs) as pool:
            pool.map(_cpu_worker, [(h, sos, ch, n, 100 + i) for i in range(procs)])
        wall = time.perf_counter() - t0
        value_p = procs * ch * n / wall / 1e6
    return {"value": value_p, "unit": "Msamples/s", "cores": procs, "kind": "port",
            "single_core_value": value_1,
            "sample": f"{procs} processes x ({ch} ch x 2^21 samples), same "
                      f"FIR(1024)+sosfiltfilt(6) chain, chunksize 2^20, "
                      f"{wall:.1f} s wall; 1 core alone: {value_1:.1f} Msamples/s; "
                      f"host exposes {avail} cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import scipy.signal as sps
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    h = sps.firwin(NTAPS, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(h, sos)          # before any GPU initialisation
    import torch
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from openseize_amd import _device as dev
    from openseize_amd import _lib
    lib = _lib.load()

    C = C_PER_GPU
    ring = [dev.synth_normal(C, CHUNK, seed=0, ch0=rank * C, n0=k * CHUNK)
            for k in range(3)]
    fir = dev.FirStream(h, C)
    iir = dev.SosStream(sos, C)
    fir_out = torch.empty((C, CHUNK), dtype=torch.float64, device="cuda")
    fwd = [torch.empty_like(fir_out) for _ in range(3)]
    y_out = torch.empty_like(fir_out)

    def step(k):
        # chunk k: FIR, then ONE launch = forward(chunk k) + backward(chunk
        # k-2, warmed up over forward chunk k-1)  [osz_sosfiltfilt_step]
        fir.push(ring[k % len(ring)], 0, out=fir_out)
        if k < 2:
            iir.forward(fir_out, out=fwd[k % 3])
        else:
            iir.step(fir_out, fwd[(k - 2) % 3], fwd[(k - 1) % 3],
                     f_out=fwd[k % 3], y_out=y_out)

    # start of the stream: steady-state init as sosfiltfilt does, then warm up
    iir.set_state_scaled(ring[0], 0)
    k = 0
    for _ in range(max(args.warmup, 2)):
        step(k)
        k += 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    _lib.check(lib.osz_profile_reset())
    _lib.check(lib.osz_profile_enable(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(k)
        k += 1
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.osz_profile_enable(0))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier()

    # order-independent checksum of the last output chunk (also the 8 B/lane
    # streaming read of known size that calibrates FETCH_SIZE in the PMC runs)
    bits, fsum = dev.checksum(y_out)

    samples_per_step = C * CHUNK
    value = samples_per_step * args.steps * world / elapsed / 1e6

    # per-kernel durations measured with HIP events on the launch stream
    kernels = {}
    for name, bps in KERNEL_BYTES.items():
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.osz_profile_query(name.encode(), ctypes.byref(n),
                                         ctypes.byref(ms)))
        if n.value:
            avg = ms.value / n.value
            kernels[name] = {"launches": n.value, "avg_ms": avg,
                             "total_ms": ms.value,
                             "achieved_gbps": bps * samples_per_step / (avg * 1e-3) / 1e9}
    dom = max(kernels, key=lambda nm: kernels[nm]["total_ms"])
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        traffic = json.load(open(tfile)).get(dom)
    roofline = {"kernel": dom, "bound": "hbm",
                "achieved": kernels[dom]["achieved_gbps"], "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": kernels[dom]["achieved_gbps"] / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": KERNEL_BYTES[dom] * samples_per_step,
                "avg_launch_ms": kernels[dom]["avg_ms"]}

    out = {
        "metric": "Msamples/sec/node (FIR+IIR chain, 256ch f64); HBM GB/s vs roofline at 1/2/4/8 GPU",
        "value": value, "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cfg-3: 256 ch/GPU x 2^20-sample chunks, FIR "
                               "overlap-add 1024 taps -> 6-section Butterworth "
                               "band-pass sosfiltfilt, steady-state stream",
                   "channels_per_gpu": C, "chunksize": CHUNK, "fir_taps": NTAPS,
                   "sos_sections": int(sos.shape[0]), "parallelism": f"channel-shard x{world}"},
        "chain_hbm_gbps": value * 1e6 * CHAIN_BYTES / 1e9 / world,
        "chain_hbm_frac": value * 1e6 * CHAIN_BYTES / 1e9 / world / HBM_PEAK_GBPS,
        "roofline": roofline, "kernels": kernels,
        "output_checksum": {"bits": f"{bits:#018x}", "sum": fsum},
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

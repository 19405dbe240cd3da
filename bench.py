#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path.

Default workload `chain` (BASELINE.json configs[2], the configuration the
metric is quoted on): 256 channels x float64, chunksize 2^20, 1024-tap FIR
overlap-add (`firwin(1024, 0.2)`) chained into a 6-section Butterworth
band-pass `sosfiltfilt` (`butter(6, [0.05, 0.3], 'bandpass')`).  A "step" is one
chunk (256 x 2^20 channel-samples) through the whole chain in steady state, and it is
ONE kernel:

    osz_chain_zp_step(chunk k): FIR, forward and backward cascade as one multiplication per
    bin of the FIR's own transform, plus the cascade's mode bursts (csrc/chain_zpn_body.h: one
    real block of 27 rows x 256 samples per 4096-point transform at the odd frequencies);
    the stream runs `lag` samples late, the step writes the tail of output chunk k-1 and the
    head of chunk k; osz_chain_zp_seal(chunk k-2) (NaN reach of sosfiltfilt) is in the step.

`--two-kernel`, `--fused`, `--unfused` time the older sequences (rounds 2 and 1).
`roofline` prices the kernel by what the fused operation must move -- 8 B read + 8 B written
per channel-sample -- over its measured duration: `frac` is a physical fraction of the HBM
peak (<= 1); the 48 B per sample the unfused chain would move (FIR 16 +
sosfiltfilt 32) appear only as `unfused_equivalent_*`.  `value` is W warm-up steps, then K
timed steps, as the driver's contract says; `steady_state` (not `value`) is the same K steps
timed once more 40 steps further into the stream: the first ~30 launches after the GPU was idle
run slower while its clock / power controller settles (benchmarks/ramp_probe.py), which is what
K = 20 behind W = 5 times.

Inputs are synthesised on the device before the timed region (a ring of
resident chunks keyed by (seed, channel, sample)); outputs land in a resident
buffer.  With N GPUs every rank runs the same 256-channel shard workload on its
own GPU (channels are independent: no data-path collective, weak scaling).

Workload `welch` (BASELINE.json configs[3]): Welch PSD, nperseg 4096, 50 %
overlap, 256 channels; the stream is split in TIME across the ranks (each rank
pushes its own block of 2^20-sample chunks) and the per-rank periodogram sums
meet in ONE RCCL all-reduce of (256 x 2049) float64 + the segment count -- the
"segment-average reduce" of cfg-4, inside the timed region.  A small untimed
pass then checks the reduced estimate against the single-rank PSD.

Workloads `fir` (configs[1]: 128 channels per GPU through the overlap-add FIR alone, 16 B per
sample) and `stft` (configs[4]: 128 channels per GPU, polyphase downsample 5 -> 1 then STFT
nfft 4096 / 50 %, 14.4 B per input sample): channel shards, no collective.

Launching: `python bench.py --gpus N` with N > 1 starts N ranks itself (fresh
child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before
anything touches the GPU in the parent); under `torch.distributed.run` the
ranks are already there and the script runs as one of them.  `--dry` runs the
launcher, the barriers and the collectives on CPU (gloo) without any kernel.

Prints ONE JSON line (see README / the driver contract) with `roofline` for
the dominant kernel and, at N=1 on rank 0, `cpu_baseline` (the CPU oracle).
"""

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
C_PER_GPU = 256
CHUNK = 1 << 20
NTAPS = 1024
NFFT = 4096              # cfg-4: nperseg 4096 (fs 4096, resolution 1.0), 50 % overlap
RAGGED = 100_000_000 - 95 * CHUNK     # last chunk of the literal 1e8-sample stream
# algorithmic HBM bytes per channel-sample of one launch: what the operation the launch
# performs must move, so that roofline.frac is a physical fraction of the peak.
# chain_fwd = FIR + forward sosfilt in one kernel: a sample in, a sample out, 16 B (the FIR's
# output never reaches HBM); chain_step = that kernel and the backward pass of an earlier chunk
# side by side on two streams, one osz_chain_step: 16 + 16; chain_zp: ONE kernel for the whole
# chain, 16 B.  The 48 B the unfused chain would move are reported apart
# (unfused_equivalent_*).  What the launches really move is roofline.traffic.
KERNEL_BYTES = {"fir_oa": 16, "sos_dual": 32, "sos_fwd": 16, "sos_bwd": 16, "sos_fwd_split": 16,
                "sos_bwd_split": 16, "chain_fwd": 16, "chain_step": 32, "chain_zp": 16, "sos_warmup": 0,
                "fir_seam": 0, "spec_fused": 8, "poly_block": 9.6}
CHAIN_BYTES = 48         # FIR 16 + sosfiltfilt 32 (the unfused accounting of the metric)
METRIC = "Msamples/sec/node (FIR+IIR chain, 256ch f64); HBM GB/s vs roofline at 1/2/4/8 GPU"


# --------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    """One host core: the CPU oracle chain on `ch` channels x `n` samples."""
    h, sos, ch, n, seed = args
    from oracle import oracle as orc
    x = np.random.default_rng(seed).standard_normal((ch, n))
    t0 = time.perf_counter()
    y = np.concatenate(orc.oaconvolve(x, h, "same"), axis=-1)
    orc.sosfiltfilt(y, sos, CHUNK)
    return time.perf_counter() - t0


def _cpu_welch_worker(args):
    ch, n, seed = args
    from oracle import oracle as orc
    x = np.random.default_rng(seed).standard_normal((ch, n))
    t0 = time.perf_counter()
    orc.psd(x, NFFT, resolution=1.0)
    return time.perf_counter() - t0


def _cpu_stft_worker(args):
    ch, n, seed = args
    from oracle import oracle as orc
    x = np.random.default_rng(seed).standard_normal((ch, n))
    t0 = time.perf_counter()
    y = orc.polyphase_resample(x, 1, 5, orc.resample_filter(1, 5, 20480.0))
    orc.stft(y, 4096.0, resolution=1.0)
    return time.perf_counter() - t0


def _cpu_share():
    """(processes to run, how that number came about): one per core this process may run on
    (os.sched_getaffinity), capped by the container's CPU quota (cgroup cpu.max / cfs_quota):
    the GPU boxes expose all 256 host cores in the affinity mask but schedule the job on a
    quota of a few of them -- 256 workers on that quota run slower than 16."""
    avail = len(os.sched_getaffinity(0))
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2: "max 100000" | "1600000 100000"
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < avail:
        n = max(1, int(quota + 0.5))
        return n, f"affinity lists {avail} cores, the container's CPU quota is {quota:.1f}: {n} processes"
    return avail, f"affinity lists {avail} cores, no smaller CPU quota: {avail} processes"


def cpu_baseline(workload, h, sos):
    """The CPU oracle (oracle/: NumPy FFT overlap-add + C DF2T loops; NumPy
    windowed rFFT for Welch) on a bounded sample of the same workload: once on
    one core, once with one process per core this process may run on
    (os.sched_getaffinity; fewer only if the host's free memory does not hold
    that many workers, and the line says so) over channel shards.  Runs BEFORE
    the GPU is initialised (workers are forked)."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    ch, n = 16, 1 << 21
    avail = len(os.sched_getaffinity(0))
    procs, share_note = _cpu_share()
    capped = ""
    try:
        import psutil
        per_worker = 10 * ch * n * 8            # input, FIR pieces, forward / backward copies
        fit = int(0.5 * psutil.virtual_memory().available // per_worker)
        if fit < procs:
            procs, capped = max(1, fit), f" (memory holds {max(1, fit)} workers of {per_worker >> 20} MiB)"
    except ImportError:
        pass
    if workload == "welch":
        worker, work = _cpu_welch_worker, lambda seed: (ch, n, seed)
        what = "Welch PSD nfft 4096, 50 % overlap"
    elif workload == "fir":
        worker, work = _cpu_fir_worker, lambda seed: (h, ch, n, seed)
        what = "FIR(1024) overlap-add, mode same"
    elif workload == "stft":
        worker, work = _cpu_stft_worker, lambda seed: (ch, n, seed)
        what = "downsample by 5 (113 taps) -> STFT nfft 4096, 50 % overlap"
    else:
        worker, work = _cpu_worker, lambda seed: (h, sos, ch, n, seed)
        what = "FIR(1024)+sosfiltfilt(6) chain, chunksize 2^20"
    single = worker(work(0))
    value_1 = ch * n / single / 1e6
    value_p, wall = value_1, single
    if procs > 1:
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(procs) as pool:
            pool.map(worker, [work(100 + i) for i in range(procs)])
        wall = time.perf_counter() - t0
        value_p = procs * ch * n / wall / 1e6
    return {"value": value_p, "unit": "Msamples/s", "cores": procs, "kind": "port",
            "single_core_value": value_1,
            "sample": f"{procs} processes x ({ch} ch x 2^21 samples), same {what}, "
                      f"{wall:.1f} s wall; 1 core alone: {value_1:.1f} Msamples/s; "
                      f"{share_note}{capped}"}


# ------------------------------------------------------------------- launcher
def launch(args):
    """--gpus N without a rank environment: start N ranks as fresh child
    processes (one per GPU) and wait for them.  The parent never initialises
    HIP; nothing is exec'ed from a process that touched the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:          # a dead rank leaves the others in a collective
                    q.terminate()
        time.sleep(0.05)
    return rc


# --------------------------------------------------------------------- ranks
class Ranks:
    """Rank environment + the few collective helpers the bench needs."""

    def __init__(self, dry):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dry = dry
        self.dist = None
        import torch
        self.torch = torch
        if not dry:
            torch.cuda.set_device(self.local)
        if self.world > 1 or "RANK" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            self.dist = dist
        self.device = "cpu" if dry else "cuda"

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def ranks_seen(self):
        return self.dist.get_world_size() if self.dist is not None else 1

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def kernel_table(lib, samples_per_step):
    """Per-kernel durations measured with HIP events on the launch stream."""
    from openseize_amd import _lib
    kernels = {}
    for name, bps in KERNEL_BYTES.items():
        n, ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.osz_profile_query(name.encode(), ctypes.byref(n), ctypes.byref(ms)))
        if n.value:
            avg = ms.value / n.value
            kernels[name] = {"launches": n.value, "avg_ms": avg, "total_ms": ms.value,
                             "achieved_gbps": bps * samples_per_step / (avg * 1e-3) / 1e9}
    return kernels


def kernel_sources_sha():
    import hashlib
    csrc = os.path.join(ROOT, "openseize_amd", "csrc")
    hsh = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            hsh.update(name.encode())
            hsh.update(open(os.path.join(csrc, name), "rb").read())
    return hsh.hexdigest()


def committed_traffic(key):
    """(HBM bytes per launch of `key` from the committed counter passes, where it comes from);
    None when the kernel sources are not the ones those passes ran."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, "profiles/traffic.json missing"
    tj = json.load(open(tfile))
    if tj.get("_kernel_sources_sha256") != kernel_sources_sha():
        return None, "stale: kernel sources changed since the --pmc passes (re-run refresh_profiles.sh)"
    return tj.get(key), "rocprofv3 --pmc passes of these kernel sources"


def roofline_of(kernels, samples_per_step):
    # one osz_chain_step is the unit when the step ran: its members (fused kernel on
    # the caller's stream, backward pass on the handle's) overlap inside it, and a
    # member's own bracket may include its wait for the other
    dom = ("chain_zp" if "chain_zp" in kernels else "chain_step" if "chain_step" in kernels
           else max(kernels, key=lambda nm: kernels[nm]["total_ms"]))
    # PMC counters cannot be read from inside a timed run: the bytes come from the
    # committed rocprofv3 --pmc passes, and only while the kernel sources are the ones
    # those passes ran (fingerprint written by benchmarks/summarise_profiles.py)
    traffic, traffic_note = committed_traffic(dom)
    out = {"kernel": dom, "bound": "hbm",
           "achieved": kernels[dom]["achieved_gbps"], "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": kernels[dom]["achieved_gbps"] / HBM_PEAK_GBPS,
           "traffic": traffic, "traffic_source": traffic_note,
           "algorithmic_bytes_per_launch": KERNEL_BYTES[dom] * samples_per_step,
           "avg_launch_ms": kernels[dom]["avg_ms"]}
    if traffic:
        # what the launch really moves against the peak, beside the algorithmic `frac`
        out["frac_physical"] = traffic / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
    if dom == "chain_zp":
        out["kernel"] = ("chain_zp (FIR, forward and backward cascade as one spectrum multiply: 8 B read + 8 B "
                         "written per channel-sample)")
        # what the three passes of the unfused chain would have had to move in the same time
        # (SURVEY 8d's 48 B per sample): a rate of work done, not of bytes moved -- may exceed 1
        ueq = CHAIN_BYTES * samples_per_step / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        out["unfused_equivalent_gbps"] = ueq
        out["unfused_equivalent_frac"] = ueq / HBM_PEAK_GBPS
        # the limiter is not HBM: vector issue + LDS latency at two waves per SIMD;
        # `bound` names the roofline the kernel's bytes are priced against
        out["limiter"] = ("power-held: the shader clock runs at 2.0 GHz under this kernel's 4.1 TB/s of HBM traffic and at "
                          "2.4 GHz with either half of it removed (profiles/r05_zpn_ablation.txt); vector issue 57 % at "
                          "2 waves per SIMD (profiles/r05_pmc_bench.json)")
    if dom == "chain_step":
        # two kernels side by side under one call: the duration is that of the pair
        # (HIP events on the caller's stream around osz_chain_step), the bytes what the two
        # fused halves must move: 16 + 16 per sample (counters: ~32.3)
        out["kernel"] = "chain_step (chain_fwd || sos_bwd_split, two streams, one osz_chain_step)"
        out["members"] = {nm: kernels[nm]["avg_ms"] for nm in ("chain_fwd", "sos_bwd_split", "sos_warmup")
                          if nm in kernels}
        if traffic:
            out["hbm_gbps_measured_traffic"] = traffic / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
    return out


# ------------------------------------------------------------ workload: chain
def run_chain(args, R, h, sos):
    torch = R.torch
    if R.dry:
        # no kernels: the partition of the channels, the barriers and the max-over-ranks timing
        C, ch0 = chain_channels(args, R)
        blocks = torch.zeros(R.world, 2, dtype=torch.int64)
        blocks[R.rank, 0], blocks[R.rank, 1] = ch0, ch0 + C
        if R.dist is not None:
            R.dist.all_reduce(blocks)
        R.barrier()
        t0 = time.perf_counter()
        time.sleep(1e-3 * args.steps)
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
        R.barrier()
        extra = {"channel_blocks": blocks.tolist()}
        if R.world > 1 and args.scaling == "weak":
            # the strong leg's partition and gather (strong_leg below), without kernels
            from openseize_amd import sharding
            from openseize_amd.core import numerical as nm
            lo, hi = sharding.channel_block(C_PER_GPU, R.rank, R.world)
            per_rank = torch.zeros(R.world, dtype=torch.float64)
            per_rank[R.rank] = 1.0 + R.rank
            R.dist.all_reduce(per_rank)
            extra["strong"] = {"channels_per_gpu": hi - lo, "chunks_per_launch": nm._zp_group(hi - lo),
                               "ms_per_step_by_rank": per_rank.tolist(), "dry": True}
        return elapsed, {}, None, extra
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    lib = _lib.load()
    C, ch0 = chain_channels(args, R)
    # host-side setup first (handles, filter tables), the device-side one (the synthesised input
    # ring, the zero-filled output ring) directly in front of the warm-up steps
    fir = dev.FirStream(h, C)
    iir = dev.SosStream(sos, C)
    zp = not (args.unfused or args.fused or args.two_kernel)
    if args.zp_tol:
        dev.chain_zp_tolerance(fir, iir, args.zp_tol)       # A/B only: the default line runs the library's default cut
    lag = dev.chain_zp_lag(fir, iir) if zp else -1
    if zp and lag < 0:
        raise RuntimeError("the zero-phase chain kernel refused the benchmark's filters")
    # chunks per launch: 1 at 256 channels; a rank with fewer channels (--scaling strong: 256 / N)
    # steps its resident stream 256 / C chunks at a time, as numerical.sosfiltfilt does
    # (numerical._zp_group) -- a launch is always 2^28 channel-samples
    from openseize_amd.core import numerical as nm
    g = nm._zp_group(C) if zp else 1
    NS = g * CHUNK
    ring = [dev.synth_normal(C, NS, seed=0, ch0=ch0, n0=k * NS) for k in range(3)]
    # resident buffers with defined contents (zero-filled, so every page of the ring
    # has been written once before the first step reads or overwrites it)
    fir_out = torch.zeros((C, NS), dtype=torch.float64, device="cuda")
    nf = 3 if (args.unfused or args.fused) else 4
    fwd = [torch.zeros_like(fir_out) for _ in range(nf)]
    y_out = torch.zeros_like(fir_out)

    def step(k):
        if zp:
            # default: ONE kernel per chunk, osz_chain_zp_step = FIR, forward and backward
            # cascade of input chunk k as one multiplication per bin; its outputs (the stream
            # runs `lag` samples late) are the tail of output chunk k-1 and the head of chunk
            # k; output chunk k-2 is sealed (NaN reach of sosfiltfilt) and the caller's
            yb = fwd[k % nf]
            dev.chain_zp_step(fir, iir, ring[k % len(ring)], out=yb[:, :NS - lag],
                              tail=fwd[(k - 1) % nf][:, NS - lag:])
            if k >= 2:
                for j in range(g):       # (one seal per output CHUNK, as the generator issues them)
                    dev.chain_zp_seal(fir, iir, fwd[(k - 2) % nf][:, j * CHUNK:(j + 1) * CHUNK],
                                      ((k - 2) * g + j) * CHUNK, 0, CHUNK)
            return
        if args.unfused:
            # chunk k: FIR, then ONE launch = forward(chunk k) + backward(chunk
            # k-2, warmed up over forward chunk k-1)  [osz_sosfiltfilt_step]
            fir.push(ring[k % len(ring)], 0, out=fir_out)
            if k < 2:
                iir.forward(fir_out, out=fwd[k % nf])
            else:
                iir.step(fir_out, fwd[(k - 2) % nf], fwd[(k - 1) % nf],
                         f_out=fwd[k % nf], y_out=y_out)
            return
        if args.fused or k < 2:
            # FIR + forward SOS of chunk k in ONE kernel (osz_chain_forward: the FIR
            # output never reaches HBM), then the backward pass of chunk k-2
            dev.chain_forward(fir, iir, ring[k % len(ring)], out=fwd[k % nf])
            if args.fused and k >= 2:
                iir.backward(fwd[(k - 2) % nf], fwd[(k - 1) % nf], out=y_out)
            return
        # --two-kernel: ONE call, osz_chain_step = the fused forward half of chunk k on this
        # stream and, beside it on the handle's own stream, the backward pass of chunk
        # k-2 (warmed up over forward chunk k-1); OSZ_CHAIN_DEFER + four forward buffers:
        # the pass may finish under the next step's forward kernel, y is taken one step late
        dev.chain_step(fir, iir, ring[k % len(ring)], fwd[(k - 2) % nf], fwd[(k - 1) % nf],
                       f_out=fwd[k % nf], y_out=y_out, defer=not args.ordered)

    # start of the stream: steady-state init as sosfiltfilt does, then warm up
    iir.set_state_scaled(ring[0], 0)
    if zp:
        dev.chain_zp_open(fir, iir, 0)
    k = 0
    launches, warm_launches = -(-args.steps // g), max(-(-args.warmup // g), 2)
    for _ in range(warm_launches):
        step(k)
        k += 1
    R.barrier()
    _lib.check(lib.osz_profile_reset())
    _lib.check(lib.osz_profile_enable(1))
    t0 = time.perf_counter()
    for _ in range(launches):
        step(k)
        k += 1
    dev.chain_wait(iir)                 # the last deferred backward pass (no-op otherwise)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.osz_profile_enable(0))
    elapsed = R.max_over_ranks(elapsed)
    R.barrier()
    # The same K steps once more, further into the stream (NOT `value`): the first ~30 launches
    # after the GPU was idle run up to 40 % slower while its clock / power controller settles
    # (benchmarks/ramp_probe.py: the same on the same buffers after a 2 s pause), so K = 20
    # steps behind W = 5 time that transient; a stream of hours runs at the rate below.
    steady = None
    if zp and not args.no_steady:
        for _ in range(max(launches, 40 - launches - warm_launches)):
            step(k)
            k += 1
        R.barrier()
        t1 = time.perf_counter()
        for _ in range(launches):
            step(k)
            k += 1
        torch.cuda.synchronize()
        el2 = R.max_over_ranks(time.perf_counter() - t1)
        R.barrier()
        steady = {"ms_per_step": el2 / (launches * g) * 1e3,
                  "Msamples_s": R.ranks_seen() * C * NS * launches / el2 / 1e6,
                  "untimed_steps_before": (k - launches) * g,
                  "note": "the same steps later in the stream, the GPU's power state settled; not `value`"}
    # order-independent checksum of the last output chunk (also the 8 B/lane
    # streaming read of known size that calibrates FETCH_SIZE in the PMC runs)
    bits, fsum = dev.checksum(fwd[(k - 3) % nf] if zp else y_out)
    kernels = kernel_table(lib, C * NS)
    if not args.unfused and not zp:
        # beside the fused kernel and the backward pass, a step has the head of the chunk
        # (the 4096 of 2^20 samples that are not whole block pairs) on the plain kernels,
        # plus, with --fused, small warm-up launches: durations only, a rate per full
        # chunk would mean nothing
        for name, rec in kernels.items():
            if name not in ("chain_fwd", "sos_bwd_split", "chain_step"):
                rec["achieved_gbps"] = None
                rec["note"] = "small launch: head of the chunk / warm-up"
    extra = {"output_checksum": {"bits": f"{bits:#018x}", "sum": fsum}}
    if zp:
        extra["output_lag_samples"] = lag
        extra["steps_done"] = launches * g            # (steps rounded up to whole launches)
        extra["chunks_per_launch"] = g
    if steady:
        extra["steady_state"] = steady
    if args.full_stream:
        del fwd, y_out, fir_out
        fir.close()
        iir.close()
        extra["full_stream"] = full_stream_leg(R, ring, h, sos)
    # (OSZ_BENCH_STRONG=1: the leg also at one rank -- how its code is exercised on a one-GPU box)
    if zp and (R.world > 1 or os.environ.get("OSZ_BENCH_STRONG") == "1") and args.scaling == "weak":
        del fwd, y_out, fir_out, ring
        fir.close()
        iir.close()
        extra["strong"] = strong_leg(args, R, h, sos, R.world * C * CHUNK * launches / elapsed / 1e6,
                                     steady["Msamples_s"] if steady else None)
    return elapsed, kernels, roofline_of(kernels, C * NS), extra


def strong_leg(args, R, h, sos, weak_value, weak_steady=None):
    """Beside the weak line of an N > 1 run: the metric's OWN job -- 256 channels, split
    channel_block-wise over the ranks (32 per GPU at 8) -- timed the same way (barrier,
    K steps, max over ranks).  A rank steps its C = 256 / N channels 256 / C chunks per launch (what
    numerical.sosfiltfilt does with a resident stream of few channels).  `efficiency` = this job's rate
    over the weak line's: 1 when N GPUs run the 256 channels N times as fast as one GPU runs them."""
    torch = R.torch
    from openseize_amd import _device as dev
    from openseize_amd import sharding
    from openseize_amd.core import numerical as nm
    lo, hi = sharding.channel_block(C_PER_GPU, R.rank, R.world)
    C = hi - lo
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    lag = dev.chain_zp_lag(fir, iir)
    g = nm._zp_group(C)
    NS = g * CHUNK
    ring = [dev.synth_normal(C, NS, seed=0, ch0=lo, n0=k * NS) for k in range(3)]
    ys = [torch.zeros((C, NS), dtype=torch.float64, device="cuda") for _ in range(3)]

    def step(k):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=ys[k % 3][:, :NS - lag], tail=ys[(k - 1) % 3][:, NS - lag:])
        if k >= 1:
            for j in range(g):
                dev.chain_zp_seal(fir, iir, ys[(k - 1) % 3][:, j * CHUNK:(j + 1) * CHUNK], ((k - 1) * g + j) * CHUNK, 0, CHUNK)

    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)
    launches, warm = -(-args.steps // g), max(-(-args.warmup // g), 2)
    k = 0
    for _ in range(warm):
        step(k)
        k += 1
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(launches):
        step(k)
        k += 1
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    elapsed = R.max_over_ranks(mine)
    R.barrier()
    per_rank = torch.zeros(R.world, dtype=torch.float64)
    per_rank[R.rank] = mine / (launches * g) * 1e3
    if R.dist is not None:
        dev_t = per_rank.cuda()
        R.dist.all_reduce(dev_t)
        per_rank = dev_t.cpu()
    fir.close()
    iir.close()
    value = C_PER_GPU * CHUNK * launches * g / elapsed / 1e6
    return {"job": "256 channels split channel_block-wise over the ranks", "channels_per_gpu": C,
            "chunks_per_launch": g, "steps": launches * g, "ms_per_step": elapsed / (launches * g) * 1e3,
            "ms_per_step_by_rank": [float(v) for v in per_rank], "value": value, "unit": "Msamples/s",
            "efficiency": value / (weak_steady or weak_value),
            "efficiency_vs_value": value / weak_value,
            "note": "efficiency = this rate / the weak job's rate (N x one GPU's 256-channel rate) with the GPU's power "
                    "state settled in both (the weak line's `steady_state`; this leg runs behind it); "
                    "efficiency_vs_value: against the weak line's `value`, which times the first launches after idle"}


def chain_channels(args, R):
    """(channels of this rank, first channel).  weak: 256 per rank (BASELINE cfg-3's shard);
    strong: the metric's 256 channels split channel_block-wise over the ranks (cfg-4's 32 per
    GPU at 8 ranks)."""
    if args.scaling == "strong":
        from openseize_amd import sharding
        if args.shard_of > 1 and R.world == 1:      # one GPU standing in for rank 0 of an N-way split
            lo, hi = sharding.channel_block(C_PER_GPU, 0, args.shard_of)
            return hi - lo, lo
        lo, hi = sharding.channel_block(C_PER_GPU, R.rank, R.world)
        return hi - lo, lo
    return C_PER_GPU, R.rank * C_PER_GPU


def full_stream_leg(R, ring, h, sos):
    """The literal stream length of cfg-3 once, through the PUBLIC producer API:
    96 chunks (95 x 2^20 + 385 280 samples = 1e8 per channel) of 256 channels
    from the resident synth ring -> FIR(1024, 'same') producer -> sosfiltfilt
    generator, outputs consumed and dropped.  Secondary number: `value` stays
    the steady-state step above."""
    from functools import partial
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    torch = R.torch
    C = ring[0].shape[0]
    lengths = [CHUNK] * 95 + [RAGGED]
    total = sum(lengths)

    def through_the_api(lens):
        n = sum(lens)

        def source():
            for k, m in enumerate(lens):
                yield ring[k % len(ring)][:, :m]

        src = producer(source, CHUNK, -1, shape=(C, n))
        fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), CHUNK, -1, shape=(C, n))
        got = 0
        for out in nm.sosfiltfilt(fir, sos, -1):
            got += out.shape[-1]
        torch.cuda.synchronize()
        assert got == n, (got, n)

    through_the_api([CHUNK] * 6 + [RAGGED])      # untimed: handles, allocator, first-call costs
    R.barrier()
    t0 = time.perf_counter()
    through_the_api(lengths)
    secs = time.perf_counter() - t0
    return {"chunks": len(lengths), "samples_per_channel": total, "channels": C,
            "seconds": secs, "Msamples_s": C * total / secs / 1e6,
            "path": "producer(gen) -> oaconvolve('same') -> GenProducer -> sosfiltfilt (the public "
                    "generators; sosfiltfilt recognises the FIR producer and runs one "
                    "osz_chain_zp_step per chunk, the stream's last two chunks on the separate kernels; "
                    "OSZ_CHAIN_ZP=0: osz_chain_step, OSZ_CHAIN_API=0: the two generators apart), "
                    "device-resident ring of 3 synthesised chunks"}


# ------------------------------------------------------------ workload: welch
def run_welch(args, R):
    torch = R.torch
    import scipy.signal as sps
    from openseize_amd import sharding
    C = C_PER_GPU
    stride = NFFT // 2
    if R.dry:
        # the partition + reduce logic without kernels: "periodograms" that are
        # a known function of the global segment index
        nseg_total = 40 * R.world + 3
        nsamples = (nseg_total - 1) * stride + NFFT
        a, b = sharding.time_block(nsamples, NFFT, stride, R.rank, R.world)
        mine = range(a // stride, a // stride + (b - a - NFFT) // stride + 1) if b > a else range(0)
        base = np.arange(4 * 33, dtype=np.float64).reshape(4, 33)
        R.barrier()
        t0 = time.perf_counter()
        total = torch.from_numpy(sum((base * (g + 1) for g in mine), np.zeros_like(base)))
        mean, cnt = sharding.reduce_segment_sums(total, len(mine))
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
        R.barrier()
        want = base * (nseg_total + 1) / 2
        err = float(np.max(np.abs(mean.numpy() - want)) / np.max(want))
        assert cnt == nseg_total and err < 1e-13, (cnt, nseg_total, err)
        return elapsed, {}, None, {"welch_check": {"segments": cnt, "max_rel_err": err,
                                                   "against": "closed form (dry run)"}}
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    lib = _lib.load()
    w = sps.get_window("hann", NFFT)
    scale = float(np.sqrt(1 / (float(NFFT) * np.sum(w ** 2))))      # fs = 4096, density
    # this rank's time block: chunk k of the block starts at sample
    # (rank * steps + k) * CHUNK of the stream; the ring repeats for timing only
    ring = [dev.synth_normal(C, CHUNK, seed=0, n0=(R.rank * 3 + k) * CHUNK) for k in range(3)]
    spec = dev.SpecStream(NFFT, NFFT, stride, w, scale, "constant", _lib.SPEC_PSD_MEAN, C)
    comm = None
    if args.reduce == "abi":
        # an RCCL communicator made through the C ABI; the unique id travels over
        # the torch.distributed group that exists anyway
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if R.rank == 0:
            uid.copy_(torch.frombuffer(bytearray(dev.RcclComm.unique_id()), dtype=torch.uint8))
        if R.dist is not None:
            R.dist.broadcast(uid, src=0)
        comm = dev.RcclComm(R.world, R.rank, bytes(uid.cpu().numpy().tobytes()))
    for k in range(max(args.warmup, 1)):
        spec.push(ring[k % 3])
    if comm is not None:
        spec.welch_reduce(comm)
        spec.mean_device()
    else:
        sharding.reduce_segment_sums(*spec.export_sum())
    _lib.check(lib.osz_spec_reset(spec.h, dev.stream_ptr()))
    R.barrier()
    _lib.check(lib.osz_profile_reset())
    _lib.check(lib.osz_profile_enable(1))
    t0 = time.perf_counter()
    for k in range(args.steps):
        spec.push(ring[k % 3])
    if comm is not None:
        spec.welch_reduce(comm)                         # osz_welch_reduce (RCCL, C ABI)
        cnt, mean = spec.mean_device()
    else:
        total, cnt = spec.export_sum()
        mean, cnt = sharding.reduce_segment_sums(total, cnt)   # RCCL via torch.distributed
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.osz_profile_enable(0))
    elapsed = R.max_over_ranks(elapsed)
    R.barrier()
    spec.close()
    kernels = kernel_table(lib, C * CHUNK)
    # ---- untimed check: time split over the ranks == single-rank PSD
    Cv, nv = 64, R.world * (1 << 18) + 1234

    def block(a, b):
        return dev.synth_normal(Cv, b - a, seed=7, n0=a)

    cntv, _, pv = sharding.psd_time_split(block, 4096, R.rank, R.world, resolution=1.0,
                                          nsamples=nv, shape=(Cv, nv), chunksize=1 << 18)
    check = None
    if R.rank == 0:
        from openseize_amd.spectra.estimators import psd
        c1, _, p1 = psd(block(0, nv), fs=4096, axis=-1, resolution=1.0)
        err = float((pv - p1).abs().max() / p1.abs().max())
        assert cntv == c1 and err < 1e-9, (cntv, c1, err)
        check = {"segments": int(cntv), "max_rel_err": err, "reduce": args.reduce,
                 "against": "single-rank psd() of the same synthetic stream"}
    if comm is not None:
        comm.close()
    extra = {"welch_check": check, "segments_per_rank": int(cnt) // max(R.world, 1),
             "reduce_bytes_per_rank": C * (NFFT // 2 + 1) * 8}
    return elapsed, kernels, roofline_of(kernels, C * CHUNK), extra


# -------------------------------------------------------------- workload: fir
FIR_CH = 128             # cfg-2: 128 channels, 1024 taps, chunksize 2^20, one GPU


def run_fir(args, R, h):
    """cfg-2 (secondary): the overlap-add FIR alone, 128 channels per GPU, 1024 taps, a step = one
    2^20-sample chunk through osz_fir_push (16 algorithmic bytes per sample)."""
    torch = R.torch
    if R.dry:
        R.barrier()
        t0 = time.perf_counter()
        time.sleep(1e-3 * args.steps)
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
        R.barrier()
        return elapsed, {}, None, None
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    lib = _lib.load()
    C = FIR_CH
    ring = [dev.synth_normal(C, CHUNK, seed=0, ch0=R.rank * C, n0=k * CHUNK) for k in range(3)]
    fir = dev.FirStream(h, C)
    out = torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda")
    for k in range(max(args.warmup, 1)):
        fir.push(ring[k % 3], 0, out=out)
    R.barrier()
    _lib.check(lib.osz_profile_reset())
    _lib.check(lib.osz_profile_enable(1))
    t0 = time.perf_counter()
    for k in range(args.steps):
        fir.push(ring[(args.warmup + k) % 3], 0, out=out)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.osz_profile_enable(0))
    elapsed = R.max_over_ranks(elapsed)
    R.barrier()
    bits, fsum = dev.checksum(out)
    fir.close()
    kernels = kernel_table(lib, C * CHUNK)
    roof = roofline_of({k: v for k, v in kernels.items() if k == "fir_oa"}, C * CHUNK) if "fir_oa" in kernels else None
    if roof is not None and roof.get("traffic"):
        roof["traffic"] = roof["traffic"] * C / C_PER_GPU      # the counters were collected at 256 channels
        roof["traffic_source"] += " (scaled from the 256-channel launch)"
        roof["frac_physical"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
    return elapsed, kernels, roof, {"output_checksum": {"bits": f"{bits:#018x}", "sum": fsum}}


def _cpu_fir_worker(args):
    h, ch, n, seed = args
    from oracle import oracle as orc
    x = np.random.default_rng(seed).standard_normal((ch, n))
    t0 = time.perf_counter()
    orc.oaconvolve(x, h, "same")
    return time.perf_counter() - t0


# ------------------------------------------------------------- workload: stft
STFT_CH = 128            # cfg-5: 1024 channels over 8 GPUs


def run_stft(args, R):
    """cfg-5 (secondary): polyphase downsample 5 -> 1 (default Kaiser anti-alias filter, 113
    taps) feeding the STFT (nfft 4096, 50 % overlap, complex128 segments), 128 channels per
    GPU, channels sharded with no collective.  A step = one 2^20-sample input chunk through
    osz_poly_push and osz_spec_push; the STFT segments go to a fresh buffer per step and are
    dropped.  14.4 algorithmic bytes per INPUT sample (9.6 + 24 / 5)."""
    torch = R.torch
    if R.dry:
        R.barrier()
        t0 = time.perf_counter()
        time.sleep(1e-3 * args.steps)
        elapsed = R.max_over_ranks(time.perf_counter() - t0)
        R.barrier()
        return elapsed, {}, None, None
    import scipy.signal as sps
    from openseize_amd import _device as dev
    from openseize_amd import _lib
    from openseize_amd.filtering.fir import Kaiser
    lib = _lib.load()
    C = STFT_CH
    cutoff = 20480 / 10
    hk = Kaiser(cutoff - cutoff / 10, cutoff + cutoff / 10, 20480, gpass=0.1, gstop=40).coeffs
    w = sps.get_window("hann", NFFT)
    scale = float(np.sqrt(1 / (4096.0 * np.sum(w ** 2))))
    ring = [dev.synth_normal(C, CHUNK, seed=0, ch0=R.rank * C, n0=k * CHUNK) for k in range(3)]
    poly = dev.PolyStream(hk, 1, 5, C)
    spec = dev.SpecStream(NFFT, NFFT, NFFT // 2, w, scale, "constant", _lib.SPEC_DFT_SEGMENTS, C)
    nseg = 0

    def step(k):
        y = poly.push(ring[k % 3], final=False)
        seg = spec.push(y)
        return 0 if seg is None else seg.shape[0]

    for k in range(max(args.warmup, 1)):
        step(k)
    R.barrier()
    _lib.check(lib.osz_profile_reset())
    _lib.check(lib.osz_profile_enable(1))
    t0 = time.perf_counter()
    for k in range(args.steps):
        nseg += step(args.warmup + k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.osz_profile_enable(0))
    elapsed = R.max_over_ranks(elapsed)
    R.barrier()
    poly.close()
    spec.close()
    kernels = kernel_table(lib, C * CHUNK)
    if "spec_fused" in kernels:       # the STFT kernel sees a fifth of the input samples, 24 B each
        rec = kernels["spec_fused"]
        rec["achieved_gbps"] = 24 * C * CHUNK / 5 / (rec["avg_ms"] * 1e-3) / 1e9
    roof = None
    if "poly_block" in kernels:
        rec = kernels["poly_block"]
        roof = {"kernel": "poly_block (downsample 5 -> 1; the STFT kernel behind it moves half as much)",
                "bound": "hbm", "achieved": rec["achieved_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": rec["achieved_gbps"] / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                "algorithmic_bytes_per_launch": 9.6 * C * CHUNK, "avg_launch_ms": rec["avg_ms"]}
    if roof is not None:
        tr, note = committed_traffic("poly_block")
        roof["traffic"] = tr * C / C_PER_GPU if tr else None      # collected at 256 channels
        roof["traffic_source"] = note + (" (scaled from the 256-channel launch)" if tr else "")
    return elapsed, kernels, roof, {"stft_segments": nseg, "stft_segments_per_step": nseg / max(args.steps, 1)}


# ----------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the two-stream step needs ~8 steps after a cold start to settle into its
    # steady interleaving (benchmarks/step_trace.py: 3.1, 2.9, 2.8 ... 2.6 ms); 100 steps
    # of 2.6 ms are a quarter of a second
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("chain", "fir", "welch", "stft"), default="chain")
    ap.add_argument("--reduce", choices=("torch", "abi"), default="torch",
                    help="welch: all-reduce through torch.distributed (RCCL) or through "
                         "osz_welch_reduce of the C ABI")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--zp-tol", type=float, default=0.0,
                    help="A/B: where the chain kernel cuts its bursts (0: the library's default, 1e-15; "
                         "rounds 3-4 ran 1e-12 on zero-mean data)")
    ap.add_argument("--no-steady", action="store_true",
                    help="skip the second, later timing of the same steps (steady_state)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="chain: weak = 256 channels per rank; strong = the metric's 256 channels "
                         "split over the ranks (32 per GPU at 8 ranks)")
    ap.add_argument("--shard-of", type=int, default=1,
                    help="with --scaling strong on ONE GPU: run rank 0's shard of an N-way split of the 256 "
                         "channels (N = 8: 32 channels, 8 chunks per launch) -- the one-GPU stand-in for a "
                         "point of the strong curve; `value` is then that shard's rate x N")
    ap.add_argument("--two-kernel", action="store_true",
                    help="chain: osz_chain_step, the fused FIR + forward SOS kernel with the backward "
                         "pass of chunk k-2 beside it on a second stream (the round-2 step; the "
                         "default is the single zero-phase kernel, osz_chain_zp_step)")
    ap.add_argument("--fused", action="store_true",
                    help="chain: FIR + forward SOS as one kernel (osz_chain_forward), then the "
                         "backward pass, on ONE stream")
    ap.add_argument("--ordered", action="store_true",
                    help="chain: osz_chain_step without OSZ_CHAIN_DEFER (every step stream-ordered)")
    ap.add_argument("--unfused", action="store_true",
                    help="chain: the three-launch sequence fir_oa, fir_seam, sos_dual "
                         "(48 B per sample through HBM; the round-1 step)")
    ap.add_argument("--full-stream", action="store_true",
                    help="chain: also time the literal 96-chunk (1e8-sample) stream once "
                         "through the public producer API")
    ap.add_argument("--dry", action="store_true",
                    help="launcher, barriers and collectives on CPU (gloo), no kernels")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args))

    import scipy.signal as sps
    h = sps.firwin(NTAPS, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = None
    if int(os.environ.get("RANK", "0")) == 0 and world_env == 1 and not (args.no_cpu or args.dry):
        cpu = cpu_baseline(args.workload, h, sos)       # before any GPU initialisation
    R = Ranks(args.dry)
    if args.workload == "welch":
        elapsed, kernels, roofline, extra = run_welch(args, R)
        metric = "Msamples/sec/node (Welch PSD nperseg 4096, 50 % overlap, 256ch f64)"
        bytes_per_sample, label = 8, (
            "cfg-4: Welch PSD nperseg 4096, 50 % overlap, hann, 256 ch x 2^20-sample chunks per "
            "GPU, stream split in time across ranks, one all-reduce of (256 x 2049) f64 + count")
        parallelism = f"time-split x{R.world} + all-reduce"
    elif args.workload == "fir":
        elapsed, kernels, roofline, extra = run_fir(args, R, h)
        metric = "Msamples/sec/node (FIR overlap-add 1024 taps, 128ch/GPU f64)"
        bytes_per_sample, label = 16, (
            "cfg-2: 128 ch x 2^20-sample chunks per GPU, FIR overlap-add 1024 taps, steady-state stream")
        parallelism = f"channel-shard x{R.world}"
    elif args.workload == "stft":
        elapsed, kernels, roofline, extra = run_stft(args, R)
        metric = "Msamples/sec/node (downsample 5->1 then STFT nfft 4096, 128ch/GPU f64, input samples)"
        bytes_per_sample, label = 14.4, (
            "cfg-5: polyphase downsample 5 -> 1 (113-tap Kaiser) -> STFT nfft 4096, 50 % overlap, "
            "hann, complex128 segments; 128 ch x 2^20-sample input chunks per GPU, channels sharded")
        parallelism = f"channel-shard x{R.world}"
    else:
        elapsed, kernels, roofline, extra = run_chain(args, R, h, sos)
        metric = METRIC
        how = ("three launches per chunk: fir_oa, fir_seam, sos_dual" if args.unfused else
               "osz_chain_forward then the backward pass, one stream" if args.fused else
               "one osz_chain_step per chunk: fused FIR + forward SOS kernel with the backward "
               "pass of chunk k-2 beside it on a second stream" if args.two_kernel else
               "one osz_chain_zp_step per chunk: FIR, forward and backward cascade as ONE spectrum "
               "multiply in the FIR's transform (+ osz_chain_zp_seal of the finished output chunk)")
        bytes_per_sample, label = CHAIN_BYTES, (
            "cfg-3: 256 ch/GPU x 2^20-sample chunks, FIR overlap-add 1024 taps -> 6-section "
            "Butterworth band-pass sosfiltfilt, steady-state stream; " + how)
        parallelism = f"channel-shard x{R.world}"
    ch_per_gpu = {"stft": STFT_CH, "fir": FIR_CH}.get(args.workload, C_PER_GPU)
    strong = args.workload == "chain" and args.scaling == "strong"
    steps_done = (extra or {}).get("steps_done", args.steps)
    if strong:
        # the job is the metric's 256 channels whatever the rank count
        samples_per_step = C_PER_GPU * CHUNK
        value = samples_per_step * steps_done / elapsed / 1e6
        ch_per_gpu = C_PER_GPU / R.world
        parallelism = f"256 channels split over {R.world} ranks (channel_block)"
        if args.shard_of > 1 and R.world == 1:
            ch_per_gpu = C_PER_GPU / args.shard_of
            parallelism = (f"rank 0's shard of a {args.shard_of}-way channel_block split, on one GPU; value = "
                           f"that shard's rate x {args.shard_of} (a stand-in, not a {args.shard_of}-GPU run)")
    else:
        samples_per_step = ch_per_gpu * CHUNK
        value = samples_per_step * steps_done * R.world / elapsed / 1e6
    out = {
        "metric": metric, "value": value, "unit": "Msamples/s", "n_gpus": R.world,
        "steps": steps_done, "warmup": args.warmup,
        "ms_per_step": elapsed / steps_done * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": label,
                   "channels_per_gpu": ch_per_gpu, "chunksize": CHUNK,
                   "fir_taps": NTAPS, "sos_sections": int(sos.shape[0]),
                   "parallelism": parallelism},
        "rccl_ranks": R.ranks_seen(),
        "roofline": roofline, "kernels": kernels,
    }
    # whole-job rate by the accounting of the workload (chain: SURVEY 8d's 48 B per sample for
    # the UNFUSED passes -- a measure of work, not of bytes the fused step moves)
    key = "unfused_equivalent_hbm_gbps" if args.workload == "chain" else "algorithmic_hbm_gbps"
    out[key] = value * 1e6 * bytes_per_sample / 1e9 / R.world
    if args.dry:
        out["dry"] = True
    if extra:
        out.update(extra)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if R.rank == 0:
        print(json.dumps(out), flush=True)
    R.close()


if __name__ == "__main__":
    main()

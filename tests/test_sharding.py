"""Multi-rank logic on CPU: gloo backend, world_size 2 (spawned processes).
Covers the channel/time partition and the Welch segment-average reduce that
runs over RCCL on the GPUs."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openseize_amd import sharding


def test_channel_and_time_blocks():
    for nch, world in ((256, 8), (10, 4), (3, 8), (1024, 8)):
        blocks = [sharding.channel_block(nch, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == nch
        assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1
    # time split: every segment exactly once, halo included
    n, nfft, stride = 100000, 4096, 2048
    nseg = (n - nfft) // stride + 1
    seen = []
    for r in range(8):
        a, b = sharding.time_block(n, nfft, stride, r, 8)
        assert a % stride == 0 and (b - a - nfft) % stride == 0 and b <= n
        seen += list(range(a // stride, a // stride + (b - a - nfft) // stride + 1))
    assert seen == list(range(nseg))
    assert sharding.time_block(100, 4096, 2048, 0, 2) == (0, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _segments(rank):
    return np.random.default_rng(100 + rank).random((5 + rank, 4, 33))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # per-rank periodogram sums and counts, as the SpecStream would export
        seg = _segments(rank)                          # (segments, ch, freq)
        total = torch.from_numpy(seg.sum(0))
        mean, cnt = sharding.reduce_segment_sums(total, seg.shape[0])
        # channel all-gather with unequal blocks (5 channels over 2 ranks)
        a, b = sharding.channel_block(5, rank, world)
        local = (torch.arange(a, b, dtype=torch.float64).reshape(-1, 1)
                 * torch.ones(1, 3, dtype=torch.float64))
        full = sharding.gather_channels(local, 5)
        q.put((rank, mean.numpy(), cnt, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_welch_reduce_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    allseg = np.concatenate([_segments(r) for r in range(world)], 0)
    # the reference's running mean over all segments (estimators.py:149-152)
    running = 0
    for c, arr in enumerate(allseg, 1):
        running = running + 1 / c * (arr - running)
    for rank, mean, cnt, full in results:
        assert cnt == allseg.shape[0]
        assert np.max(np.abs(mean - running)) < 1e-14
        assert np.array_equal(full[:, 0], np.arange(5.0))


def _run_bench(*flags):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags],
                         capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout          # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_launcher_world2_dry():
    """`python bench.py --gpus 2` starts two ranks itself (no torchrun): the
    launcher, the barriers and the max-over-ranks timing on CPU (gloo), no
    kernels; then the cfg-4 time split + segment-average all-reduce logic."""
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry")
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 2
    assert out["scaling"] == "weak" and out["dry"] is True
    # beside the weak line: the metric's own 256 channels split over the ranks (128 each, two
    # chunks per launch), every rank's time gathered
    assert out["strong"]["channels_per_gpu"] == 128 and out["strong"]["chunks_per_launch"] == 2
    assert out["strong"]["ms_per_step_by_rank"] == [1.0, 2.0]
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry",
                     "--workload", "welch")
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["welch_check"]["segments"] == 83 and out["welch_check"]["max_rel_err"] < 1e-13
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry",
                     "--workload", "fir")
    assert out["n_gpus"] == 2 and out["config"]["channels_per_gpu"] == 128
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry",
                     "--workload", "stft")
    assert out["n_gpus"] == 2 and out["config"]["channels_per_gpu"] == 128
    assert out["config"]["parallelism"] == "channel-shard x2"


def test_bench_strong_scaling_world3_dry():
    """`bench.py --scaling strong`: the metric's 256 channels split channel_block-wise over the
    ranks (32 per GPU at 8 ranks) instead of 256 per rank; `value` is the job's 256
    channels over the slowest rank's time.  Three ranks on CPU (gloo), no kernels."""
    out = _run_bench("--gpus", "3", "--steps", "2", "--warmup", "1", "--dry", "--scaling", "strong")
    assert out["n_gpus"] == 3 and out["rccl_ranks"] == 3 and out["scaling"] == "strong"
    assert out["channel_blocks"] == [[0, 86], [86, 171], [171, 256]]
    assert abs(out["value"] - 256 * (1 << 20) * 2 / (out["ms_per_step"] * 2e-3) / 1e6) < 1e-6 * out["value"]
    weak = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry")
    assert weak["scaling"] == "weak" and weak["channel_blocks"] == [[0, 256], [256, 512]]


def test_bench_under_torchrun_world2_dry():
    """The driver's launch line: torch.distributed.run starts the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry"],
        capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.gpu
def test_psd_time_split_callable_source_and_abi_reduce(golden):
    """A rank that synthesises only its own time block (callable source), and
    the C-ABI collective on a one-rank RCCL communicator: osz_rccl_* +
    osz_welch_reduce leave sum and count unchanged and the mean equals psd()."""
    from openseize_amd import _device as dev, _lib
    from openseize_amd.spectra.estimators import psd
    C, n = 8, 300_000

    def block(a, b):
        return dev.synth_normal(C, b - a, seed=3, n0=a)

    whole = block(0, n)
    c1, f1, p1 = psd(whole, fs=4096, axis=-1, resolution=1.0)
    acc = None
    total = 0
    for r in range(3):                                  # three "ranks", one after the other
        a, b = sharding.time_block(n, 4096, 2048, r, 3)
        c, _, p = psd(block(a, b), fs=4096, axis=-1, resolution=1.0)
        acc = c * p if acc is None else acc + c * p
        total += c
    assert total == c1
    assert float((acc / total - p1).abs().max()) < 1e-9 * float(p1.abs().max())
    cnt, f, p = sharding.psd_time_split(block, 4096, 0, 1, resolution=1.0, nsamples=n,
                                        shape=(C, n), chunksize=70_000)
    assert cnt == c1 and np.array_equal(f, f1)
    assert float((p - p1).abs().max()) < 1e-12 * float(p1.abs().max())
    # C-ABI collective, one rank
    import scipy.signal as sps
    w = sps.get_window("hann", 4096)
    scale = float(np.sqrt(1 / (4096.0 * np.sum(w ** 2))))
    spec = dev.SpecStream(4096, 4096, 2048, w, scale, "constant", _lib.SPEC_PSD_MEAN, C)
    spec.push(whole)
    before, cb = spec.export_sum()
    comm = dev.RcclComm(1, 0, dev.RcclComm.unique_id())
    assert comm.size() == 1
    spec.welch_reduce(comm)
    after, ca = spec.export_sum()
    assert ca == cb == c1 and torch.equal(before, after)
    cm, mean = spec.mean_device()
    assert cm == c1 and float((mean - p1).abs().max()) < 1e-12 * float(p1.abs().max())
    comm.close()
    spec.close()


@pytest.mark.gpu
def test_psd_time_split_single_gpu(golden):
    """world_size 1 on the GPU: the time-split driver equals psd(); two time
    blocks run back to back recombine to the same mean."""
    from openseize_amd.spectra.estimators import psd
    g = golden("g7_welch.npz")
    x = g["x"]
    ref = g["psd_ov0.5"]
    cnt, f, p = sharding.psd_time_split(torch.from_numpy(x).cuda(), 1024, 0, 1,
                                        resolution=1.0)
    assert cnt == int(g["cnt_ov0.5"])
    assert np.max(np.abs(p.cpu().numpy() - ref)) < 1e-9 * np.max(ref)
    parts = []
    for r in range(2):
        a, b = sharding.time_block(x.shape[-1], 1024, 512, r, 2)
        c, _, pr = psd(x[:, a:b], 1024, resolution=1.0)
        parts.append((c, pr))
    tot = sum(c * pr for c, pr in parts) / sum(c for c, _ in parts)
    assert sum(c for c, _ in parts) == cnt
    assert np.max(np.abs(tot - ref)) < 1e-9 * np.max(ref)

"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI), channels sharded across ranks.

Every operator on the path acts along the sample axis only, so a contiguous
block of channels per rank needs no exchange at all (FIR, SOS, resampling,
STFT: outputs stay sharded or are all-gathered).  The one real collective is
the Welch segment average when the *time* axis of a channel block is split
across ranks: each rank holds the sum of its periodograms and its segment
count, and the estimate is  all_reduce(sum) / all_reduce(count)  -- numerically
the running mean of the reference (spectra/estimators.py:149-152) in a
different summation order (O(1e-16) relative).

Nothing here touches the data path of a single GPU; with world_size == 1 all
functions are the identity.  The functions take a process group so they are
testable on CPU with the gloo backend.
"""

import numpy as np

from openseize_amd import _device as dev
from openseize_amd import _lib
from openseize_amd.core import numerical as nm
from openseize_amd.core.producer import producer


def channel_block(nch, rank, world):
    """Contiguous, balanced block of channels owned by `rank`:
    the first nch % world ranks get one extra channel."""
    base, extra = divmod(int(nch), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def time_block(nsamples, nfft, stride, rank, world):
    """Segment-aligned split of the sample axis for Welch: rank r takes a
    contiguous range of whole segments; returns (start_sample, stop_sample)
    including the nfft - stride halo so no segment is lost or duplicated."""
    nseg = (nsamples - nfft) // stride + 1 if nsamples >= nfft else 0
    s0, s1 = channel_block(nseg, rank, world)
    if s1 <= s0:
        return 0, 0
    return s0 * stride, (s1 - 1) * stride + nfft


def reduce_segment_sums(total, count, group=None):
    """all_reduce(SUM) of the periodogram sums and of the segment counts over
    the group (RCCL on GPUs, gloo in the CPU tests); returns (mean, count).
    `total` is a tensor (nch, nfreq); it is reduced in place."""
    import torch
    import torch.distributed as dist
    cnt = torch.tensor([float(count)], dtype=torch.float64, device=total.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    n = int(round(float(cnt.item())))
    if n and total.is_cuda:
        # the divide of the segment average is a kernel of this library too
        from openseize_amd import _lib as lib
        return dev.ew(lib.EW_DIV, total.reshape(total.shape[0], -1), cnt).reshape(total.shape), n
    return (total / n if n else total), n


def gather_channels(local, nch, axis=0, group=None):
    """all_gather of per-rank channel blocks (unequal blocks allowed) back to
    the full (nch, ...) tensor on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [b - a for a, b in (channel_block(nch, r, world) for r in range(world))]
    # collectives want equal shapes: pad every block to the largest, trim after
    big = max(sizes)
    local = torch.movedim(local, axis, 0).contiguous()
    if local.shape[0] < big:
        pad = torch.zeros((big - local.shape[0],) + tuple(local.shape[1:]),
                          dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    pieces = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(pieces, local, group=group)
    full = torch.cat([p[:n] for p, n in zip(pieces, sizes)], dim=0)
    return torch.movedim(full, 0, axis)


def psd_time_split(data, fs, rank, world, resolution=0.5, window="hann",
                   overlap=0.5, detrend="constant", scaling="density",
                   group=None, nsamples=None, shape=None, chunksize=None):
    """Welch PSD of a (channels..., samples) stream with the SEGMENTS split
    across the ranks of ``group`` and ONE all-reduce of the per-rank
    periodogram sums (cfg-4's "RCCL segment-average reduce",
    reference spectra/estimators.py:149-152 for the whole stream).

    ``data`` is either an array / device tensor (every rank passes a view of
    the same stream and reads only its own time block), or a callable
    ``data(start, stop)`` returning this rank's samples [start, stop) as an
    array, tensor or producer -- a rank then never holds or even addresses the
    rest of the stream (204.8 GB per rank at cfg-4); ``nsamples`` and ``shape``
    (the shape of the whole stream) are required with a callable.
    Returns (cnt, freqs, psd), identical on every rank."""
    nfft = int(fs / resolution)
    stride = nfft - int(nfft * overlap)
    if callable(data):
        if nsamples is None or shape is None:
            raise ValueError("a callable source needs nsamples and shape")
        full_shape = tuple(shape)
    else:
        full_shape = tuple(data.shape)
        nsamples = full_shape[-1]
    a, b = time_block(int(nsamples), nfft, stride, rank, world)
    freqs = np.fft.rfftfreq(nfft, 1 / fs)
    coeffs, scale = nm._window_and_scale(window, nfft, fs, scaling)
    layout = dev.Layout(full_shape[:-1] + (max(b - a, 0),), -1)
    spec = dev.SpecStream(nfft, nfft, stride, coeffs, scale, detrend,
                          _lib.SPEC_PSD_MEAN, layout.nch)
    try:
        if b > a:
            block = data(a, b) if callable(data) else data[..., a:b]
            cs = int(chunksize) if chunksize else int(fs) * 64
            for arr in producer(block, cs, axis=-1):
                x2d, _ = layout.to2d(arr)
                spec.push(x2d)
        total, cnt = spec.export_sum()
    finally:
        spec.close()
    mean, cnt = reduce_segment_sums(total, cnt, group=group)
    return cnt, freqs, mean.reshape(full_shape[:-1] + (len(freqs),))

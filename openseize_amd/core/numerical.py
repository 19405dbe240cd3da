"""The chunked numerics behind the Producer iterator -- host side.

Same generator functions, argument meaning, shapes and exceptions as the
reference's ``core/numerical.py`` (cited per function), but every sample is
processed by the HIP kernels of ``libosz_hip.so`` through the C ABI
(``include/osz_hip.h``); nothing here computes on the CPU and nothing falls
back to SciPy/NumPy for the filtering or transforms.  SciPy is used only for
design-time constants (window coefficients), as the reference does.

Each generator owns one C-ABI handle (the iterator's carried state: overlap
tail, zi, resampler history, segment FIFO) created when iteration starts, so
producers stay re-iterable and picklable.  Chunks that arrive as ndarrays are
sent to the device and results come back as ndarrays; chunks that arrive as
CUDA tensors stay in HBM.
"""

import math
import os
from functools import partial

from collections import deque

import numpy as np
import scipy.signal as sps

from openseize_amd import _device as dev
from openseize_amd import _lib
from openseize_amd.core import protools
from openseize_amd.core.arraytools import normalize_axis, slice_along_axis
from openseize_amd.core.producer import ArrayProducer, producer


# ---------------------------------------------------------------------------
# overlap-add FIR (reference core/numerical.py:19-298)
# ---------------------------------------------------------------------------
def optimal_nffts(arr):
    """FFT length the reference would pick per segment
    (core/numerical.py:19-38).  Kept for API parity: the device kernel uses a
    fixed on-chip 4096-point transform, which gives the same linear
    convolution (the result does not depend on the segmentation)."""
    return int(8 * 2 ** np.ceil(np.log2(len(arr))))


def convolved_shape(shape1, shape2, mode, axis):
    """Shape of the convolution of arrays of shape1 and shape2 along axis
    (same results as core/numerical.py:41-73).  The array with more dimensions
    lends the other axes (the second one on a tie, as the reference's stable sort
    has it); an unknown mode leaves the length alone.  A 1-D second shape is taken
    as the window length whatever ``axis`` is (the reference indexes it with
    ``axis`` and raises IndexError for 0 < axis, filtering/bases.py:411)."""
    lengths = sorted((shape1[axis], shape2[0] if len(shape2) == 1 else shape2[axis]))
    short, long_ = lengths
    by_mode = {"full": long_ + short - 1, "same": long_, "valid": long_ - short + 1}
    lender = shape2 if len(shape2) >= len(shape1) else shape1
    out = list(lender)
    if mode in by_mode:
        out[axis] = by_mode[mode]
    return tuple(out)


def _chain_first(first, rest):
    """The stream again after its first array has been peeked at."""
    yield first
    yield from rest


def _oa_cuts(wlen, mode):
    """Samples dropped left of the first and right of the last segment
    (core/numerical.py:143-150)."""
    if mode == "full":
        return 0, 0
    if mode == "same":
        return (wlen - 1) // 2, int(math.ceil((wlen - 1) / 2))
    if mode == "valid":
        return wlen - 1, wlen - 1
    raise KeyError(mode)


def _joined_resident(source, layout, gmax):
    """The chunks of ``source`` as 2-D arrays, those of a resident stream that lie one behind the
    other in memory joined up to ``gmax`` at a time (``_row_joined``): yields (2-D chunk, came from
    the host, lengths of the produced chunks it holds).  Empty chunks are dropped."""
    held, sizes = None, []
    for arr in source:
        x2d, from_host = layout.to2d(arr)
        if x2d.shape[1] == 0:
            continue
        if from_host or gmax == 1:
            if held is not None:
                yield held, False, sizes
                held, sizes = None, []
            yield x2d, from_host, [x2d.shape[1]]
            continue
        both = _row_joined(held, x2d) if held is not None and len(sizes) < gmax else None
        if both is None:
            if held is not None:
                yield held, False, sizes
            held, sizes = x2d, [x2d.shape[1]]
        else:
            held, sizes = both, sizes + [x2d.shape[1]]
    if held is not None:
        yield held, False, sizes


_PROBE_AT = {}
_PROBE_SLOTS = 8192            # pinned slots of one oaconvolve stream's probes (far more than it ever has pending)
_PROBE_POOL = {}               # device -> [(pinned slots, side stream)] of finished streams


def _probe_sum(v):
    """Sum of one sample in 2048 along the last axis and of the last one: non-finite if one of
    them is (a block of the kernels' transforms is non-finite as a whole).  Two launches."""
    import torch
    n, key = v.shape[-1], (v.shape[-1], v.device)
    idx = _PROBE_AT.get(key)
    if idx is None:
        if len(_PROBE_AT) > 32:
            _PROBE_AT.clear()
        at = list(range(0, n, 2048))
        if at[-1] != n - 1:
            at.append(n - 1)
        idx = _PROBE_AT[key] = torch.tensor(at, dtype=torch.int64, device=v.device)
    return torch.index_select(v, -1, idx).sum()


class _ProbeWatch:
    """Probes (0-d CUDA tensors, ``_probe_sum``) looked at from the host without waiting for anything
    but the probe: each is copied into a pinned slot on a side stream behind an event recorded
    right after it -- ``float(tensor)`` would queue its copy behind the pushes launched since and
    park the host until they are done (a bubble per piece: 4 % at 256 channels x 2^20)."""

    def __init__(self):
        self.ring = self.side = None
        self.watch, self.slot = {}, 0

    def add(self, t):
        """Key of the probe `t` (pieces of one buffer share theirs), its copy to the host started."""
        import torch
        key = id(t)
        if key in self.watch:
            return key
        if self.ring is None:
            # (pinned memory and a stream cost a third of a millisecond to make: kept for the process)
            self.device = torch.cuda.current_device()          # (a stream belongs to its device)
            pool = _PROBE_POOL.setdefault(self.device, [])
            self.ring, self.side = pool.pop() if pool else (
                torch.empty(_PROBE_SLOTS, dtype=torch.float64, pin_memory=True), torch.cuda.Stream())
        while len(self.watch) >= _PROBE_SLOTS:
            self.watch.pop(next(iter(self.watch)))           # (long handed on: far more slots than pieces ever pending)
        slot = self.slot = (self.slot + 1) % _PROBE_SLOTS
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            self.ring[slot:slot + 1].copy_(t.reshape(1), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        t.record_stream(self.side)
        self.watch[key] = [t, slot, done, None]              # (the tensor kept: its id stays its own)
        return key

    def tripped(self, key):
        """Is the probe non-finite?  Waits for the probe's own copy, nothing else."""
        if isinstance(key, bool):
            return key
        w = self.watch.get(key)
        if w is None:                                        # (never the case for a pending piece)
            raise RuntimeError("a pending piece lost its probe")
        if w[3] is None:
            w[2].synchronize()
            w[3] = not math.isfinite(float(self.ring[w[1]]))
        return w[3]

    def close(self):
        if self.ring is not None:
            self.side.synchronize()
            _PROBE_POOL.setdefault(self.device, []).append((self.ring, self.side))
            self.ring = self.side = None


class _Tee:
    """A producer as ``_oaconvolve_stream`` sees it (shape, chunksize, iteration) that remembers the
    chunks it hands on: ``seen`` gets (first sample, samples, chunk) -- references, no copies."""

    def __init__(self, pro, axis, seen):
        self.pro, self.axis, self.seen = pro, axis, seen
        self.shape = pro.shape
        self.chunksize = int(getattr(pro, "chunksize", 1 << 20))

    def __iter__(self):
        at = 0
        for chunk in self.pro:
            n = chunk.shape[self.axis]
            if n:
                self.seen.append((at, n, chunk))
            at += n
            yield chunk


@dev.chain_aware
def oaconvolve(pro, window, axis, mode, nfft_factor=32):
    """Streaming overlap-add convolution of a producer with a 1-D window
    (core/numerical.py:158-298) on the device (K1, ``osz_fir_*``): ``_oaconvolve_stream`` below,
    and around it the REACH of non-finite input samples as the reference has it.

    The reference transforms segments of ``step = nfft - wlen + 1`` input samples (nfft =
    8 * 2^ceil(log2 wlen) * 32, :202-217); a non-finite sample makes the whole output of ITS
    segment non-finite -- 'full' samples [k step, (k + 1) step + wlen - 1) -- and nothing else
    (:258-283).  The kernels here transform blocks of a few thousand samples, so their own
    non-finite runs are shorter and start elsewhere.  The pieces are therefore held back until
    the inputs that can still reach them have gone by (one segment: two pieces at chunks of 2^20
    and 1024 taps), every piece is probed -- one output sample in 2048; a block of the kernels is
    non-finite as a whole -- and only when a probe trips: the exact non-finite inputs are looked
    up in the chunks kept by reference (``_Tee``), samples that are non-finite here but finite in
    the reference are computed again from a cleaned copy, and the reference's segments are laid
    over the piece.  ``OSZ_FIR_REACH=0``: the kernels' own reach (rounds 1-4).
    """
    if os.environ.get("OSZ_FIR_REACH", "1") == "0":
        yield from _oaconvolve_stream(pro, window, axis, mode)
        return
    import torch
    taps = np.asarray(window, dtype=np.float64)
    nsamples = pro.shape[axis]
    if taps.ndim != 1 or nsamples < len(taps):
        yield from _oaconvolve_stream(pro, window, axis, mode)      # (raises, in its own words)
        return
    wlen = len(taps)
    lcut, _ = _oa_cuts(wlen, mode)
    step = _oa_reference_step(nsamples, wlen)
    layout = dev.Layout(pro.shape, axis)
    seen = deque()
    inner = _oaconvolve_stream(_Tee(pro, axis, seen), window, axis, mode, probed=True)
    pending = deque()                 # [piece, its first sample ('full' numbering), samples, probe]
    produced = lcut                   # 'full' sample behind the last piece produced
    flagged_to = 0                    # 'full' sample behind the last piece whose probe tripped

    def probe(arr):
        """Does the piece hold a non-finite sample?  One sample in 2048 and the last one (a
        block of the kernels' transforms is non-finite as a whole): their sum, a 0-d CUDA tensor
        nobody waits for yet -- the stream's own, over the whole buffer a piece is a view of, where
        it made one (``_probe_sum``) -- or, for an ndarray, a bool."""
        if dev.is_tensor(arr):
            held = getattr(arr, "_osz_probe", None)
            return held if held is not None else _probe_sum(torch.movedim(arr, axis, -1))
        v = np.moveaxis(arr, axis, -1)
        return not math.isfinite(float(v[..., ::2048].sum()) + float(v[..., -1].sum()))

    watch = _ProbeWatch()

    def tripped(rec):
        rec[3] = watch.tripped(rec[3])
        return rec[3]

    def segments(a, n):
        """The reference's segments that reach 'full' samples [a, a + n): first, last."""
        return max((a - wlen + 1) // step, 0), (a + n - 1) // step

    def gather(i0, i1):
        parts = []
        for at, n, chunk in seen:
            lo, hi = max(i0 - at, 0), min(i1 - at, n)
            if lo < hi:
                parts.append(layout.to2d(chunk)[0][:, lo:hi])
        if not parts:
            return torch.zeros((layout.nch, 0), dtype=torch.float64, device="cuda")
        return torch.cat(parts, 1)

    def settle(arr, a, n):
        """The piece as the reference leaves it (the rare path)."""
        b = a + n
        k_lo, k_hi = segments(a, n)
        i0, i1 = k_lo * step, min(max((k_hi + 1) * step, b), nsamples)
        X = gather(i0, i1)
        if X.shape[1] < i1 - i0:          # (a MaskedProducer's shape may name more samples than it yields)
            X = torch.nn.functional.pad(X, (0, i1 - i0 - X.shape[1]))
        bad_in = ~torch.isfinite(X)
        y2d, host = layout.to2d(arr)
        mine = ~torch.isfinite(y2d)
        if not bool(bad_in.any()) and not bool(mine.any()):
            return arr              # (the probe that tripped was another piece's)
        # (no non-finite input in reach and non-finite here all the same: a block of the kernels
        # that began before the reference's segment did)
        nseg = k_hi - k_lo + 1
        pad = nseg * step - X.shape[1]
        seg_bad = torch.nn.functional.pad(bad_in, (0, max(pad, 0)))[:, :nseg * step]
        seg_bad = seg_bad.reshape(seg_bad.shape[0], nseg, step).any(2)            # (channels, segments)
        at = torch.arange(a, b, device=X.device)
        k1 = torch.div(at, step, rounding_mode="floor")
        lost = seg_bad[:, (k1 - k_lo).clamp(0, nseg - 1)]
        behind = ((at - k1 * step) < wlen - 1) & (k1 - 1 >= k_lo)                  # the previous segment's overhang
        lost = lost | (seg_bad[:, (k1 - 1 - k_lo).clamp(0, nseg - 1)] & behind[None, :])
        if bool((mine & ~lost).any()):
            # non-finite here, finite there: once more, from a cleaned copy
            clean = torch.where(bad_in, torch.zeros((), dtype=X.dtype, device=X.device), X)
            s, e = max(a - (wlen - 1), 0), min(b, nsamples)
            again = dev.FirStream(taps, layout.nch)
            try:
                parts = []
                if e > s:
                    parts.append(again.push(clean[:, s - i0:e - i0].contiguous(), min(a - s, e - s)))
                if b > nsamples:
                    parts.append(again.flush(X.device, skip=max(a - nsamples, 0), drop=nsamples + wlen - 1 - b))
                y2d = torch.where(mine, torch.cat(parts, 1), y2d)
            finally:
                again.close()
        y2d = y2d.masked_fill(lost, float("nan"))
        return layout.from2d(y2d, host)

    def release(final):
        nonlocal flagged_to
        while pending:
            arr, a, n, _ = pending[0]
            k_lo, k_hi = segments(a, n)
            need = min((k_hi + 1) * step, nsamples)            # inputs that can still reach the piece
            if not final and (produced < need or pending[-1][1] < need or len(pending) < 2):
                return              # (the newest piece's probe is still on its way: not waited for)
            rec = pending.popleft()
            hit = flagged_to > k_lo * step or tripped(rec)
            for other in pending:
                if hit or other[1] >= need:
                    break
                hit = tripped(other)
            if tripped(rec):
                flagged_to = max(flagged_to, a + n)
            yield settle(arr, a, n) if hit else arr
            # inputs no pending piece can want any more
            keep = segments(pending[0][1], pending[0][2])[0] * step if pending else segments(produced, 1)[0] * step
            while seen and seen[0][0] + seen[0][1] <= keep:
                seen.popleft()

    try:
        for piece in inner:
            n = piece.shape[axis]
            if n == 0:
                continue
            flag = probe(piece)
            pending.append([piece, produced, n, flag if isinstance(flag, bool) else watch.add(flag)])
            produced += n
            yield from release(False)
        yield from release(True)
    finally:
        inner.close()
        watch.close()


def _oaconvolve_stream(pro, window, axis, mode, nfft_factor=32, probed=False):
    """The stream of ``oaconvolve`` as the kernels produce it.

    Yields one piece per produced chunk plus the final overhang; concatenated
    they are the ``np.convolve(x, window, mode)`` of every channel, i.e. what
    the reference yields in ``step``-sized pieces.  ``nfft_factor`` is accepted
    for signature parity and ignored (see ``optimal_nffts``).

    Differences from the reference, both on inputs where it misbehaves: data
    shorter than the window raises a clear ValueError (the reference fails
    with a broadcasting ValueError), and the lengths ``N == nfft-wlen+1`` /
    odd fallback nfft (where the reference skips the left cut or raises) give
    the plain ``np.convolve`` answer.
    """
    window = np.asarray(window, dtype=np.float64)
    if window.ndim != 1:
        raise ValueError("window must be 1-D")
    wlen = len(window)
    lcut, rcut = _oa_cuts(wlen, mode)
    nsamples = pro.shape[axis]
    chunk_len = int(getattr(pro, "chunksize", 1 << 20))
    if nsamples < wlen:
        raise ValueError(
            f"operands could not be convolved: data has {nsamples} samples "
            f"along axis {axis}, fewer than the {wlen} window taps")
    layout = dev.Layout(pro.shape, axis)
    fir = dev.FirStream(window, layout.nch)
    # Output pieces are written straight into buffers as long as the incoming
    # chunks: the left cut shifts the output against the input by `lcut`
    # samples, so every incoming chunk is pushed in two parts -- its head
    # completes the open buffer, its body starts the next one.  Downstream
    # re-chunking (GenProducer) then finds chunk-aligned arrays and copies
    # nothing; for a resident stream no sample is moved twice.  (Host-fed
    # streams keep one push per chunk: their pieces cross PCIe anyway.)
    pos, host, device = 0, True, "cuda"
    cur, fill = None, 0                      # open output buffer and its filled columns
    import torch

    chunks = iter(pro)
    first = next(chunks, None)
    if first is not None and not dev.is_tensor(first):
        # host-fed stream: transfers and kernels overlapped (dev.HostPipe); every
        # piece crosses PCIe as it is, one push per chunk
        state = {"pos": 0}

        def push(x2d):
            n = x2d.shape[1]
            if n == 0:
                return None
            skip = min(max(lcut - state["pos"], 0), n)
            state["pos"] += n
            return fir.push(x2d, skip)

        try:
            pipe = dev.HostPipe(layout)
            yield from pipe.run(_chain_first(first, chunks), push)
            if state["pos"] > 0:
                skip = min(max(lcut - state["pos"], 0), wlen - 1)
                if wlen - 1 - skip - rcut > 0:
                    tail = fir.flush("cuda", skip=skip, drop=rcut)
                    yield layout.from2d(tail, True)
        finally:
            fir.close()
        return
    pro = _chain_first(first, chunks) if first is not None else ()
    # Few channels or short chunks: chunks of a resident source that lie one behind the other in
    # memory (the views an ArrayProducer cuts from one tensor) are pushed several at a time -- as
    # many as make 2^28 channel-samples, as the zero-phase chain does (_zp_group) -- and handed on
    # chunk by chunk: `cuts` are the lengths the open buffer is handed on in.
    gmax = _zp_group(layout.nch, chunk_len)
    cuts = []

    def emit(buf, cols):
        if len(cuts) <= 1:
            yield layout.from2d(buf if cols == buf.shape[1] else buf[:, :cols], host)
            return
        # (one probe for the whole buffer, oaconvolve's reach: its pieces carry it along)
        held = _probe_sum(buf[:, :cols]) if probed and not host else None
        at = 0
        for m in cuts:                       # (a last buffer may be short of its end: mode 'valid')
            if at >= cols:
                break
            piece = layout.from2d(buf[:, at:min(at + m, cols)], host)
            if held is not None and dev.is_tensor(piece):
                piece._osz_probe = held
            yield piece
            at += m

    try:
        for x2d, host, sizes in _joined_resident(pro, layout, gmax):
            device = x2d.device
            n = x2d.shape[1]
            skip = min(max(lcut - pos, 0), n)
            pos += n
            if host:
                # host-fed: every piece goes back over PCIe as it is; one push per chunk
                y = fir.push(x2d, skip)
                if y.shape[1] > 0:
                    yield layout.from2d(y, host)
                continue
            done = 0                          # input columns of this chunk already pushed
            if cur is not None:
                part = min(n - skip, cur.shape[1] - fill)
                if part > 0 or skip > 0:
                    fir.push(x2d[:, :skip + part], skip, out=cur[:, fill:fill + part])
                    fill += part
                    done = skip + part
                    skip = 0
                if fill == cur.shape[1]:
                    yield from emit(cur, fill)
                    cur, fill = None, 0
            if done < n:
                if cur is None:
                    cur = torch.empty((layout.nch, n), dtype=torch.float64, device=device)
                    fill = 0
                    cuts = sizes
                cnt = n - done - skip
                fir.push(x2d[:, done:], skip, out=cur[:, fill:fill + cnt])
                fill += cnt
                if fill == cur.shape[1]:
                    yield from emit(cur, fill)
                    cur, fill = None, 0
        if pos > 0:
            skip = min(max(lcut - pos, 0), wlen - 1)
            cnt = max(wlen - 1 - skip - rcut, 0)
            if cur is not None and cnt > 0:
                part = min(cnt, cur.shape[1] - fill)
                if part > 0:
                    fir.flush(device, skip=skip, drop=rcut + cnt - part,
                              out=cur[:, fill:fill + part])
                    fill += part
                    skip += part
                    cnt -= part
            if cur is not None and fill > 0:
                yield from emit(cur, fill)
            if cnt > 0:
                tail = fir.flush(device, skip=skip, drop=rcut)
                yield layout.from2d(tail, host)
    finally:
        fir.close()


# ---------------------------------------------------------------------------
# SOS IIR (reference core/numerical.py:301-411)
# ---------------------------------------------------------------------------
def _zi_to_2d(zi, nsec, layout):
    """User zi of shape (nsec, ..., 2 on axis, ...) -> (nsec, nch, 2)."""
    zi = np.asarray(zi, dtype=np.float64)
    want = list(layout.other)
    want.insert(layout.axis, 2)
    if zi.shape != (nsec, *want):
        raise ValueError(
            f"Invalid zi shape. With axis={layout.axis} and sos of {nsec} "
            f"sections, expected {(nsec, *want)}, got {zi.shape}.")
    zi = np.moveaxis(zi, layout.axis + 1, -1)
    return np.ascontiguousarray(zi.reshape(nsec, layout.nch, 2))


@dev.chain_aware
def sosfilt(pro, sos, axis, zi=None):
    """Forward cascaded-biquad filter with the state carried from chunk to
    chunk (core/numerical.py:301-335) on the device (K2,
    ``osz_sos_forward``).  ``zi``: None (zeros) or an array of shape
    (nsections, ..., 2 along axis, ...)."""
    sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
    fused = _fir_feeding(pro, axis)
    if fused is not None:
        # a resident FIR producer feeding this filter: one kernel per chunk for both
        gen = _sosfilt_after_fir(pro, *fused, sos, zi)
        if gen is not None:
            yield from gen
            return
    layout = dev.Layout(pro.shape, axis)
    stream = dev.SosStream(sos, layout.nch)
    try:
        if zi is not None:
            stream.set_state(_zi_to_2d(zi, sos.shape[0], layout))
        chunks = iter(pro)
        first = next(chunks, None)
        if first is None:
            return
        if not dev.is_tensor(first):
            # host-fed: transfers overlapped with the kernels (dev.HostPipe)
            yield from dev.HostPipe(layout).run(
                _chain_first(first, chunks),
                lambda x2d: stream.forward(x2d) if x2d.shape[1] else None)
            return
        # (adjacent views of a resident stream go through the kernel several at a time, _zp_group)
        gmax = _zp_group(layout.nch, int(getattr(pro, "chunksize", 1 << 20)))
        for x2d, host, sizes in _joined_resident(_chain_first(first, chunks), layout, gmax):
            y = stream.forward(x2d)
            if len(sizes) == 1:
                yield layout.from2d(y, host)
                continue
            at = 0
            for m in sizes:
                yield layout.from2d(y[:, at:at + m], host)
                at += m
    finally:
        stream.close()


def _fir_feeding(pro, axis):
    """(source producer, taps) when ``pro`` is what ``FIR.__call__`` /
    ``producer(partial(oaconvolve, ...), ...)`` builds: a generating producer over
    ``oaconvolve`` in mode 'same' along the same axis with the source's
    chunksize; None otherwise."""
    import os
    from openseize_amd.core.producer import GenProducer, Producer
    if os.environ.get("OSZ_CHAIN_API") == "0":     # A/B and tests: the two generators apart
        return None
    if not isinstance(pro, GenProducer):
        return None
    gen = pro.data
    if not isinstance(gen, partial) or gen.func is not oaconvolve:
        return None
    # however the call was spelled: positional, keywords of the partial, kwargs of the producer
    import inspect
    try:
        bound = inspect.signature(oaconvolve).bind(*gen.args, **{**(gen.keywords or {}), **pro.kwargs})
    except TypeError:
        return None
    source, taps = bound.arguments["pro"], bound.arguments["window"]
    fir_axis, mode = bound.arguments["axis"], bound.arguments["mode"]
    if not isinstance(source, Producer) or mode != "same":
        return None
    from openseize_amd.core.producer import MaskedProducer
    if isinstance(source, MaskedProducer):
        # its shape may name more samples than it yields (core/producer.py:399-408): the fused
        # flows plan their chunks from the shape
        return None
    ndim = len(pro.shape)
    # the fused flows filter along the axis the chunks are cut along: the FIR's axis, the
    # cascade's axis and both producers' chunking axes must be one and the same
    ax = normalize_axis(axis, ndim)
    if (normalize_axis(fir_axis, ndim) != ax or normalize_axis(pro.axis, ndim) != ax
            or normalize_axis(source.axis, ndim) != ax):
        return None
    taps = np.asarray(taps, dtype=np.float64)
    if taps.ndim != 1 or not 2 <= len(taps) <= 2049 or tuple(source.shape) != tuple(pro.shape):
        return None
    if int(source.chunksize) != int(pro.chunksize):
        return None
    return source, taps


def _sosfilt_after_fir(pro, source, taps, sos, zi):
    """``sosfilt(oaconvolve(source, taps, 'same'))``, chunk for chunk what the two generators
    yield one after the other (reference core/numerical.py:158-298 feeding :301-335), on
    ``osz_chain_forward``: FIR and forward cascade of an input chunk in one launch, the FIR's
    output never in HBM (16 instead of 32 B per channel-sample).

    'same' drops the first ``lcut = (taps-1)//2`` outputs, so the piece input chunk k produces
    is output samples [k cs - lcut, (k+1) cs - lcut).  Piece k is written at the front of a
    fresh buffer ``lcut`` columns wider than a chunk; its first ``lcut`` columns are copied
    behind piece k-1 (C x lcut samples: nothing beside a chunk), whose buffer then holds
    output chunk k-1 as one view -- memory of its own, as every array the separate generators
    yield.  Input chunk 0 (the left cut) and the overhang of the convolution go through the
    separate kernels.  A host-fed source goes up and its results come down through
    ``dev.HostPipe``: ONE trip over PCIe each way for the chain, where the two generators
    apart make two and re-chunk on the host in between.  Returns None (the caller runs the
    two generators apart) for CPU tensors, short streams, short chunks and a last chunk
    shorter than the cut."""
    import torch
    axis = pro.axis
    cs, total = int(pro.chunksize), int(pro.shape[axis])
    wlen = len(taps)
    lcut, rcut = _oa_cuts(wlen, "same")
    nchunks = -(-total // cs)
    last = total - (nchunks - 1) * cs
    if nchunks < 3 or cs < 65536 or total < wlen or last < max(lcut, 1):
        return None
    chunks = dev.pull_resident(source, source)         # a source of this library hands CUDA tensors
    first = next(chunks, None)
    if first is None or first.shape[axis] != cs or (dev.is_tensor(first) and not first.is_cuda):
        chunks.close()                                 # (a reader behind the source is released)
        return None

    def run():
        layout = dev.Layout(pro.shape, axis)
        C = layout.nch
        fir, iir = dev.FirStream(taps, C), dev.SosStream(sos, C)
        watch = _ProbeWatch()
        try:
            dev.chain_zp_tolerance(fir, iir, dev.zp_tolerance())   # (the forward link's cut too)
            if zi is not None:
                iir.set_state(_zi_to_2d(zi, sos.shape[0], layout))
            resident = dev.is_tensor(first)
            device = first.device if resident else "cuda"
            pipe = None if resident else dev.HostPipe(layout)
            at = {"k": 0, "open": None}                # chunks taken; the buffer of the last piece

            def fused(x2d):
                """Input chunk k in; output chunk k - 1 (complete now) out."""
                k, m = at["k"], x2d.shape[1]
                if k >= nchunks or m != (cs if k < nchunks - 1 else last):
                    raise RuntimeError("sosfilt after oaconvolve: an inner chunk of the source "
                                       f"is not chunksize = {cs} long")
                at["k"] = k + 1
                cur = torch.empty((C, m + lcut), dtype=torch.float64, device=device)
                prev, at["open"] = at["open"], cur
                if k == 0:                             # the left cut: separate kernels
                    iir.forward(fir.push(x2d, lcut), out=cur[:, lcut:cs])
                    return None
                dev.chain_forward(fir, iir, x2d, out=cur[:, :m])
                if lcut:
                    prev[:, cs:cs + lcut].copy_(cur[:, :lcut])
                return prev[:, lcut:cs + lcut]

            # ---- the reference FIR's reach of a non-finite sample (oaconvolve's docstring): its
            # whole SEGMENT of `step` samples is lost, and behind a cascade the channel from the
            # segment's start on.  Pieces are held back (inside `op`: a host-fed stream's results
            # stay in HBM until they are final) until the inputs of their last segment have gone by,
            # each is probed, the pushes they came from are kept by reference with the handles'
            # states before them (device snapshots) -- and when a probe trips the stream goes on, from
            # the oldest piece still here, the slow exact way: FIR of the cleaned chunk, the
            # reference's segments laid over its output, the cascade behind that (which loses the
            # channel by itself).  OSZ_CHAIN_REACH=0: the kernels' own reach.
            reach = os.environ.get("OSZ_CHAIN_REACH", "1") != "0"
            step = _oa_reference_step(total, wlen)
            pend, kept = deque(), deque()              # [j, piece, first 'full' sample, probe]; [k, chunk, states before it]
            slow = {"on": False, "out": deque(), "queue": deque(), "seen": 0, "parts": [], "have": 0,
                    "piece": 0, "skip": 0, "drop": 0, "bad": None}

            def lost(a, b):
                """'full' samples [a, b) the reference's FIR has lost, per channel."""
                at_ = torch.arange(a, b, device=device)
                k1 = torch.div(at_, step, rounding_mode="floor")
                bad = slow["bad"]
                hit = bad[:, k1.clamp(max=bad.shape[1] - 1)]
                behind = ((at_ - k1 * step) < wlen - 1) & (k1 >= 1)
                return hit | (bad[:, (k1 - 1).clamp(min=0)] & behind[None, :])

            def slow_emit(f, a):
                """Cascade output for 'full' samples from `a` on -> whole output chunks."""
                if slow["skip"]:
                    f = f[:, slow["skip"]:]
                    slow["skip"] = 0
                slow["parts"].append(f)
                slow["have"] += f.shape[1]
                while True:
                    j = slow["piece"]
                    want = cs if j < nchunks - 1 else last
                    if j >= nchunks or slow["have"] < want:
                        return
                    cat = slow["parts"][0] if len(slow["parts"]) == 1 else torch.cat(slow["parts"], 1)
                    if slow["drop"] > 0:           # (handed on before the trip, from the fused launches)
                        slow["drop"] -= 1
                    else:
                        slow["out"].append(cat[:, :want])
                    slow["parts"] = [cat[:, want:]] if cat.shape[1] > want else []
                    slow["have"] -= want
                    slow["piece"] = j + 1

            def slow_drain(final):
                while slow["queue"]:
                    a, u = slow["queue"][0]
                    b = a + u.shape[1]
                    if not final and slow["seen"] < min(((b - 1) // step + 1) * step, total):
                        return
                    slow["queue"].popleft()
                    slow_emit(iir.forward(u.masked_fill(lost(a, b), float("nan"))), a)

            def slow_feed(k, x2d):
                m = x2d.shape[1]
                bad = ~torch.isfinite(x2d)
                where_ = torch.nonzero(bad)
                if where_.shape[0]:
                    slow["bad"][where_[:, 0], torch.div(k * cs + where_[:, 1], step, rounding_mode="floor")] = True
                    x2d = torch.where(bad, torch.zeros((), dtype=x2d.dtype, device=x2d.device), x2d)
                u = fir.push(x2d.contiguous(), lcut if k == 0 else 0)
                slow["queue"].append((lcut if k == 0 else k * cs, u))
                slow["seen"] = k * cs + m
                slow_drain(False)

            def go_slow():
                j0 = pend[0][0]
                trim_kept(j0)
                if not kept or kept[0][0] > j0 or kept[0][2] is None:
                    raise RuntimeError("sosfilt after oaconvolve: the pushes of the pending pieces are gone")
                ks = kept[0][0]                    # the last push at or before j0 with the states before it
                fir.restore(kept[0][2][0])
                iir.restore(kept[0][2][1])
                slow.update(on=True, piece=ks, skip=lcut if ks > 0 else 0, drop=j0 - ks,
                            bad=torch.zeros((C, -(-(total + wlen) // step) + 1), dtype=torch.bool, device=device))
                pend.clear()
                pushes = list(kept)
                kept.clear()
                for k, xk, _ in pushes:
                    slow_feed(k, xk)

            def trim_kept(j0):
                """Pushes before the last one at or before j0 that has the states before it are let go."""
                keep_from = None
                for rec in kept:
                    if rec[0] > j0:
                        break
                    if rec[2] is not None:
                        keep_from = rec[0]
                while kept and keep_from is not None and kept[0][0] < keep_from:
                    kept.popleft()

            def release(final):
                """The oldest piece if it is final (None: not yet)."""
                if slow["on"]:
                    return slow["out"].popleft() if slow["out"] else None
                if not pend:
                    return None
                j, piece, a, _ = pend[0]
                need = min(((a + piece.shape[1] - 1) // step + 1) * step, total)
                if not final and pend[-1][2] < need:
                    return None          # (and the newest piece's probe is still on its way: not waited for)
                for rec in pend:
                    if rec[2] >= need and rec is not pend[0]:
                        break
                    if watch.tripped(rec[3]):
                        go_slow()
                        return slow["out"].popleft() if slow["out"] else None
                pend.popleft()
                trim_kept(pend[0][0] if pend else at["k"] - 1)     # (the next piece's body is the last push's)
                return piece

            def op(x2d):
                k = at["k"]
                if slow["on"]:
                    if k >= nchunks or x2d.shape[1] != (cs if k < nchunks - 1 else last):
                        raise RuntimeError("sosfilt after oaconvolve: an inner chunk of the source "
                                           f"is not chunksize = {cs} long")
                    at["k"] = k + 1
                    slow_feed(k, x2d)
                    return release(False)
                if not reach:
                    return fused(x2d)
                # (a host-fed chunk sits in a staging buffer that is written again: a copy is kept.  The
                # handles' states before every fourth push: reading them makes the forward link settle its
                # carry into them -- five small launches -- and the slow way can start a few chunks early)
                kept.append([k, x2d if resident else x2d.clone(),
                             (fir.snapshot(), iir.snapshot()) if k % 4 == 0 else None])
                y = fused(x2d)
                if y is not None:
                    pend.append([k - 1, y, (k - 1) * cs + lcut, watch.add(_probe_sum(y))])
                return release(False)

            def hand_on(y):
                if resident or dev.emit_resident():
                    return layout.from2d(y, False)
                out, done = pipe.download(y)
                done.synchronize()
                return pipe.restore(out)

            seq = (c for c in _chain_first(first, chunks) if c.shape[axis] > 0)
            if resident:
                for arr in seq:
                    y = op(layout.to2d(arr)[0])
                    if y is not None:
                        yield layout.from2d(y, False)
            else:
                yield from pipe.run(seq, op)
            if at["k"] != nchunks:
                raise RuntimeError(f"sosfilt after oaconvolve: {at['k']} of {nchunks} chunks")
            # ---- the overhang of the convolution, behind the last piece
            if not slow["on"]:
                cur = at["open"]
                if lcut:
                    iir.forward(fir.flush(device, skip=0, drop=rcut), out=cur[:, last:last + lcut])
                y = cur[:, lcut:last + lcut]
                if not reach:
                    yield hand_on(y)
                    return
                pend.append([nchunks - 1, y, (nchunks - 1) * cs + lcut, watch.add(_probe_sum(y))])
                while pend and not slow["on"]:
                    y = release(True)
                    if y is not None:
                        yield hand_on(y)
            if slow["on"]:
                # (the slow way's own end: what is queued, the overhang with the segments over it, the rest)
                slow_drain(True)
                if lcut:
                    slow["queue"].append((total, fir.flush(device, skip=0, drop=rcut)))
                    slow_drain(True)
                while slow["out"]:
                    yield hand_on(slow["out"].popleft())
                if slow["piece"] != nchunks:
                    raise RuntimeError(f"sosfilt after oaconvolve: {slow['piece']} of {nchunks} output chunks")
        finally:
            watch.close()
            fir.close()
            iir.close()

    return run()


def _sosfiltfilt_after_fir(pro, source, taps, sos):
    """``sosfiltfilt(oaconvolve(source, taps, 'same'))`` for a device-resident
    source, chunk for chunk what the two generators yield one after the other
    (reference core/numerical.py:158-298 feeding :338-411), on the steady-state
    step of the C ABI: ``osz_chain_step`` -- the fused FIR + forward-cascade kernel
    of input chunk k with the backward pass of output chunk k-2 beside it.

    The forward stream goes into a ring of four chunks per channel.  'same' drops
    the first ``lcut = (taps-1)//2`` outputs, so the piece input chunk k produces
    is output samples [k cs - lcut, (k+1) cs - lcut): pieces are written slot by
    slot (never across the ring's end), output chunk j is the view ``lcut``
    columns further on, and the ``lcut`` columns by which the last slot's chunk
    reaches past the ring are copied behind it when slot 0 is rewritten.
    A host-fed source goes up through the pinned ring of ``dev.HostPipe`` while the
    previous step runs and the results leave on its D2H stream: ONE trip over PCIe
    each way for the chain, where the two generators apart make two and re-chunk
    on the host in between.  Returns None (the caller falls back to the two
    separate generators) for short streams and CPU tensors."""
    import torch
    axis = pro.axis
    cs, total = int(pro.chunksize), int(pro.shape[axis])
    wlen = len(taps)
    lcut, rcut = _oa_cuts(wlen, "same")
    nchunks = -(-total // cs)
    if nchunks < 5 or cs < 65536 or total < wlen:
        return None
    chunks = dev.pull_resident(source, source)         # a source of this library hands CUDA tensors
    first = next(chunks, None)
    resident = first is not None and dev.is_tensor(first) and first.is_cuda
    if first is None or first.shape[axis] != cs or (dev.is_tensor(first) and not resident):
        chunks.close()                                 # (a reader behind the source is released)
        return None

    def run():
        layout = dev.Layout(pro.shape, axis)
        C, R = layout.nch, 4
        fir, iir = dev.FirStream(taps, C), dev.SosStream(sos, C)
        try:
            warm = iir.warm_len
            if warm > cs - lcut:                       # the warm-up would need samples not yet there
                yield None
                return
            dev.chain_zp_tolerance(fir, iir, dev.zp_tolerance())
            lag = dev.chain_zp_lag(fir, iir)
            one_kernel = (lag >= 0 and nchunks >= 6 and os.environ.get("OSZ_CHAIN_ZP", "1") != "0"
                          and cs >= max(4 * (lcut + lag), warm + lcut + lag, 2 * dev.chain_zp_min_chunk(fir, iir)))
            if not one_kernel and os.environ.get("OSZ_CHAIN_ZP", "1") != "0":
                # Streams and cascades the one-kernel route does not take: the two generators apart.
                # They carry the reference FIR's NaN reach (oaconvolve's docstring), which the
                # two-kernel step below does not, and since round 5 (chunks joined per push, the
                # cascade's own zero-phase route) they are no slower on resident data: 3.3 against
                # 3.9 ms per 256 x 2^20 chunk on a five-chunk stream, 4.5 against 4.8 behind nine
                # sections.  OSZ_CHAIN_ZP=0 asks for the two-kernel step (A/B runs, tests).
                yield None
                return
            yield True
            device = first.device if resident else "cuda"
            pipe = None if resident else dev.HostPipe(layout)
            flying = deque()
            if one_kernel:
                # (the reference FIR's NaN reach, segment by segment; OSZ_ZP_REACH=0: the kernels' own,
                # from the sample itself)
                reach = 0 if os.environ.get("OSZ_ZP_REACH") == "0" else _oa_reference_step(total, wlen)
                yield from _zero_phase_stream(fir, iir, layout, pipe, flying, first, chunks, taps, cs, total,
                                              lcut, rcut, lag, device, ref_step=reach)
                return

            def feed(arr):
                return layout.to2d(arr)[0] if pipe is None else pipe.feed(arr)

            def emit(y):
                """Results in order: resident ones as they are, host-bound ones once their
                transfer (behind the kernels queued so far) has been two steps in flight."""
                if pipe is None or dev.emit_resident():
                    yield layout.from2d(y, False)
                    return
                flying.append(pipe.download(y))
                while len(flying) > 2:
                    out, done = flying.popleft()
                    done.synchronize()
                    yield pipe.restore(out)

            shift = lcut + (lcut & 1)                  # even: chunk views keep 16-byte alignment
            span = R * cs
            off = shift - lcut                         # piece k starts at column (k % R) cs + off
            ring = torch.empty((C, span + shift + 2), dtype=torch.float64, device=device)
            # Output sample s lives at column (s + shift) mod span; the columns
            # [span, span + shift) repeat [0, shift) for the chunk of the last slot.  A
            # piece written by the kernel into the last slot with off = 1 puts its last
            # sample at column `span` itself (column 0 is then never read).

            def chunk(j, n):                           # first n samples of output chunk j
                c0 = (j % R) * cs + shift
                return ring[:, c0:c0 + n]

            def mend(j):
                # chunk j in the last slot reaches `shift` columns past the ring
                if j % R == R - 1 and shift > off:
                    ring[:, span + off:span + shift].copy_(ring[:, off:shift])

            def seal(y, j):
                """NaN reach of the backward pass of chunk j (core/numerical.py:397-403: it
                warms up over ALL of forward chunk j + 1): the step probed the last sample of
                what it had of that chunk, cs - lcut samples -- the last lcut arrive with the
                same step's own piece.  Its true last sample is there by now."""
                col = ((j + 2) * cs - 1 + shift) % span
                return y.masked_fill_(~torch.isfinite(ring[:, col:col + 1]), float("nan"))

            def store(s0, data):                       # host-placed samples (the overhang)
                n, c0 = data.shape[1], (s0 + shift) % span
                part = min(n, span - c0)
                ring[:, c0:c0 + part].copy_(data[:, :part])
                if part < n:
                    ring[:, :n - part].copy_(data[:, part:])
                    if off:
                        ring[:, span:span + 1].copy_(data[:, part:part + 1])

            # ---- input chunk 0 on the plain kernels: the left cut, the forward state
            x0 = feed(first)
            head = fir.push(x0, lcut)
            iir.set_state_scaled(head, 0)
            iir.forward(head, out=ring[:, shift:shift + cs - lcut])
            produced = cs - lcut                       # output samples in the ring
            late, k = None, 1
            for arr in chunks:
                if arr.shape[axis] == 0:
                    continue
                x2d = feed(arr)
                m = x2d.shape[1]
                if produced + lcut != k * cs or m > cs:
                    raise RuntimeError("sosfiltfilt after oaconvolve: an inner chunk of the source "
                                       f"is not chunksize = {cs} long")
                dst = ring[:, (k % R) * cs + off:(k % R) * cs + off + m]
                j = k - 2
                if j >= 0:
                    mend(j)
                    y = torch.empty((C, cs), dtype=torch.float64, device=device)
                    dev.chain_step(fir, iir, x2d, chunk(j, cs), chunk(j + 1, cs - lcut), f_out=dst,
                                   y_out=y, defer=True)
                    if late is not None:
                        yield from emit(seal(late, j - 1))   # the previous step's: ours now
                    late = y
                else:
                    dev.chain_forward(fir, iir, x2d, out=dst)
                produced += m
                k += 1
            # ---- the overhang of the convolution: plain kernels, behind the last piece
            cnt = max(wlen - 1 - rcut, 0)
            if cnt > 0:
                store(produced, iir.forward(fir.flush(device, skip=0, drop=rcut)))
                produced += cnt
            if produced != total:
                raise RuntimeError(f"sosfiltfilt after oaconvolve: {produced} of {total} samples")
            dev.chain_wait(iir)
            if late is not None:
                yield from emit(seal(late, k - 3))
            # ---- the chunks the steady state has not reached: plain backward passes
            for j in range(max(k - 2, 0), nchunks):
                n = min(cs, total - j * cs)
                mend(j)
                fb = None
                if j + 1 < nchunks:
                    mend(j + 1)
                    fb = chunk(j + 1, min(cs, total - (j + 1) * cs))
                yield from emit(iir.backward(chunk(j, n), fb))
            while flying:
                out, done = flying.popleft()
                done.synchronize()
                yield pipe.restore(out)
        finally:
            fir.close()
            iir.close()

    gen = run()
    ok = next(gen)                                     # handles exist, the plan holds?
    if ok is None:
        gen.close()
        chunks.close()
        return None
    return gen


def _zp_group(nch, cs=1 << 20):
    """Chunks of ``cs`` samples per launch for a resident stream of ``nch`` channels: as many as make
    a launch of 2^28 channel-samples -- the headline's 256 x 2^20, 2 GiB in and out: few channels
    OR short chunks both leave a one-chunk launch too few blocks per workgroup to hide what it pays
    once -- at most 64 (``OSZ_ZP_GROUP`` overrides; 1: every chunk its own launch)."""
    env = os.environ.get("OSZ_ZP_GROUP")
    if env:
        return max(1, int(env))
    return max(1, min(64, (1 << 28) // max(int(nch) * max(int(cs), 1), 1)))


def _row_joined(a, b):
    """The (C, na + nb) view of two 2-D device views of ONE storage whose rows continue each
    other (b's row r starts where a's row r ends), or None."""
    if (a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0] or a.stride(1) != 1 or b.stride(1) != 1
            or a.dtype != b.dtype or a.device != b.device
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + a.shape[1]):
        return None
    n = a.shape[1] + b.shape[1]
    if a.shape[0] == 1:                    # (a single row: its pitch is whatever a view says -- make it the length)
        return a.as_strided((1, n), (n, 1), a.storage_offset())
    if a.stride(0) != b.stride(0) or a.stride(0) < n:
        return None                        # rows of different arrays, or rows that would run into each other
    return a.as_strided((a.shape[0], n), a.stride(), a.storage_offset())


def _oa_reference_step(nsamples, wlen):
    """Input samples per FFT segment of the reference's overlap-add (core/numerical.py:202-217):
    nfft = 8 * 2^ceil(log2 wlen) * 32 unless a segment of that size is longer than the data, then
    min(8 * 2^ceil(log2 wlen), N); step = nfft - wlen + 1.  A non-finite input sample makes its whole
    segment's output non-finite there (:258-283): the NaN reach of a chain behind the FIR counts from
    the segment's start."""
    base = int(8 * 2 ** math.ceil(math.log2(wlen)))
    nfft = base * 32
    if nfft - wlen + 1 > nsamples:
        nfft = min(base, int(nsamples))
    return max(nfft - wlen + 1, 1)


def _zero_phase_stream(fir, iir, layout, pipe, flying, first, chunks, taps, cs, total, lcut, rcut, lag, device,
                       what="sosfiltfilt after oaconvolve", ref_step=0):
    """The body of ``_sosfiltfilt_after_fir`` on the zero-phase kernel (C ABI: osz_chain_zp_*,
    csrc/chain_zp.hip): FIR, forward and backward cascade of an input chunk in ONE launch.

    The reference's backward pass of chunk i starts from what back-filtering forward chunk
    i + 1 leaves (core/numerical.py:397-403); for chunks much longer than the cascade's
    memory -- the caller has checked chunksize against ``warm_len`` -- that is, to 1e-18,
    what back-filtering the whole rest of the stream leaves, i.e. the zero-phase filter
    |H_iir|^2 on the FIR output.  Only the two ends of the stream differ:
      * the start: the cascade's forward pass begins at FIR output sample ``lcut`` from
        ``sosfilt_zi * u[lcut]`` (:374-386); what the ``lcut`` samples before -- which it
        never sees -- would have left in its state is taken off that start state;
      * the end: the last two output chunks come from the separate kernels (forward stream
        from the states ``osz_chain_zp_finish`` leaves, then the chunk-local backward
        passes with the reference's own start states, :397-411).
    The kernel's output runs ``lag`` samples late, 'same' drops ``lcut`` more: input chunk
    k delivers output samples [k cs - shift, (k + 1) cs - shift), i.e. the tail of output
    chunk k - 1 and the head of chunk k, which is where the kernel writes them.  Output
    chunk j is handed on two steps later, after ``osz_chain_zp_seal`` (NaN reach: a NaN in
    the forward stream of chunk j + 1 makes chunk j NaN as a whole).

    ``ref_step`` > 0 (a real FIR in front): the reference's FIR turns a non-finite input sample
    into a whole non-finite SEGMENT of ``ref_step`` samples (``_oa_reference_step``), so its
    forward stream is bad from the segment's START -- up to ``ref_step`` samples before the
    sample.  The kernels record the exact sample, ``osz_chain_zp_reach`` makes the seal count
    from the segment's start, and output chunks are held back for as many more steps as that
    reach spans chunks, so that the chunks the reference loses are still here to be lost.  In
    the stream's last two chunks (separate kernels, whose FIR makes its own 4096-point blocks
    non-finite) the forward stream is recomputed from a cleaned copy and poisoned by the
    reference's rule when a channel first goes bad there."""
    import torch
    C = layout.nch
    wlen = len(taps)
    nchunks = -(-total // cs)
    shift = lcut + lag

    def feed(arr):
        return layout.to2d(arr)[0] if pipe is None else pipe.feed(arr)

    def emit(y):
        if pipe is None or dev.emit_resident():
            yield layout.from2d(y, False)
            return
        flying.append(pipe.download(y))
        while len(flying) > 2:
            out, done = flying.popleft()
            done.synchronize()
            yield pipe.restore(out)

    def fresh(j):
        return torch.empty((C, min(cs, total - j * cs)), dtype=torch.float64, device=device)

    # ---- the start state of the forward cascade
    x0 = feed(first)
    u = fir.push(x0[:, :lcut + 1].contiguous(), 0)            # FIR output samples 0 .. lcut
    seen = 0.0
    if lcut > 0:
        iir.set_state(None)
        iir.forward(u[:, :lcut].contiguous())
        seen = iir.get_state()
    iir.set_state_scaled(u, lcut)
    iir.set_state(iir.get_state() - seen)
    fir.reset()
    dev.chain_zp_open(fir, iir, lcut)
    dev.chain_zp_reach(fir, iir, ref_step)
    # chunks the reach of a non-finite sample spans beyond the two the backward pass's does
    extra = max(-(-(ref_step + lcut) // cs) - 1, 0) if ref_step else 0
    junk = torch.empty((C, shift), dtype=torch.float64, device=device)     # outputs before sample 0
    ys = {0: fresh(0)}
    dev.chain_zp_step(fir, iir, x0, out=ys[0][:, :cs - shift], tail=junk)
    # Few channels (or short chunks): a launch over ONE chunk has too few blocks per workgroup to
    # hide what it pays once (tables, the pre-roll block of every run, the launch itself: 32
    # channels x 2^20 run at 0.7 of the 256-channel rate).  Chunks of a resident source
    # that lie one behind the other in memory -- the views an ArrayProducer cuts from one tensor
    # -- therefore go through the kernel several at a time: `gmax` of them make one step of as
    # many channel-samples as a 256-channel chunk of 2^20; the results are handed on chunk by
    # chunk as before (views of the step's output buffer).
    gmax = _zp_group(C, cs) if pipe is None else 1
    group, emitted = [], 0                                    # [(k, 2-D view)]; chunks handed on so far

    def run_group():
        nonlocal emitted
        a, g = group[0][0], len(group)
        X = group[0][1]
        for _, nxt in group[1:]:
            X = _row_joined(X, nxt)
        Y = ys[a] = fresh(a) if g == 1 else torch.empty((C, g * cs), dtype=torch.float64, device=device)
        for i in range(g if g > 1 else 0):
            ys[a + i] = Y[:, i * cs:(i + 1) * cs]
        dev.chain_zp_step(fir, iir, X, out=Y[:, :g * cs - shift], tail=ys[a - 1][:, cs - shift:])
        group.clear()
        while emitted <= a + g - 3 - extra:                   # complete: the forward stream of chunk j + 1 is known
            j = emitted
            dev.chain_zp_seal(fir, iir, ys[j], lcut + j * cs, lcut, cs)
            yield from emit(ys.pop(j))
            emitted += 1

    k, x2d = 1, None
    for arr in chunks:
        if arr.shape[layout.axis] == 0:
            continue
        x2d = feed(arr)
        if k * cs >= total or x2d.shape[1] != min(cs, total - k * cs):
            raise RuntimeError(f"{what}: an inner chunk of the source is not chunksize = {cs} long")
        if k == nchunks - 2:
            break
        if group and (len(group) >= gmax or _row_joined(group[-1][1], x2d) is None):
            yield from run_group()
        group.append((k, x2d))
        k += 1
    if group:
        yield from run_group()
    if k != nchunks - 2 or x2d is None:
        raise RuntimeError(f"{what}: the source ended after {k} of {nchunks} chunks")
    # ---- the end: chunks n-2 and n-1 on the separate kernels.  The rest of output chunk
    # n-3 needs the head of input chunk n-2; the handles' own states are those at its start.
    m = shift + dev.chain_zp_min_chunk(fir, iir)
    dev.chain_zp_finish(fir, iir, x2d[:, :m], out=ys[k - 1][:, cs - shift:])
    off = lcut & 1                                            # chunk views on even columns
    n_last = total - (nchunks - 1) * cs
    last = next(chunks, None)
    while last is not None and last.shape[layout.axis] == 0:
        last = next(chunks, None)
    if last is None:
        raise RuntimeError(f"{what}: the source ended before its last chunk")
    xl = feed(last)
    if xl.shape[1] != n_last:
        raise RuntimeError(f"{what}: the last chunk has {xl.shape[1]} of {n_last} samples")
    cnt = max(wlen - 1 - rcut, 0)
    F = torch.empty((C, off + lcut + cs + n_last + 2), dtype=torch.float64, device=device)
    # (the states at the start of chunk n-2, should the rare path below want them: device copies, no wait)
    kept = (fir.snapshot(), iir.snapshot()) if ref_step else None

    def forward_tail(xa, xb):
        """The forward stream of the last two chunks and the overhang, FIR 'full' sample
        (n - 2) cs + j at column off + j."""
        dev.chain_forward(fir, iir, xa, out=F[:, off:off + cs])
        dev.chain_forward(fir, iir, xb, out=F[:, off + cs:off + cs + n_last])
        if cnt > 0:
            F[:, off + cs + n_last:off + cs + n_last + cnt].copy_(iir.forward(fir.flush(device, skip=0, drop=rcut)))

    forward_tail(x2d, xl)
    late = None
    if ref_step:
        # A channel that FIRST goes bad in these two chunks: the separate kernels' FIR makes its
        # own 4096-point blocks non-finite, the reference its segments of ref_step samples.  Rare,
        # and only then: the exact sample from the inputs, the forward stream once more from a
        # cleaned copy (every sample the bad one does not reach is then what it should be), and the
        # reference's reach laid over it.  (Channels bad from before arrive with poisoned states:
        # their forward stream is non-finite from column `off` on, the seal below knows the rest.)
        first_bad = torch.isfinite(F[:, off]) & ~torch.isfinite(F[:, off + cs + n_last + cnt - 1])
        if bool(first_bad.any()):
            xa = torch.cat([x2d, xl], 1)
            nonfinite = ~torch.isfinite(xa)
            at = nonfinite.to(torch.int8).argmax(1).to(torch.int64) + (nchunks - 2) * cs     # the first one per channel
            seg = torch.div(at, ref_step, rounding_mode="floor") * ref_step                   # its segment's first sample
            fir.restore(kept[0])
            iir.restore(kept[1])
            xa = torch.where(nonfinite & first_bad[:, None], torch.zeros((), dtype=xa.dtype, device=xa.device), xa)
            forward_tail(xa[:, :cs], xa[:, cs:])
            col = off + (seg - (nchunks - 2) * cs).clamp(min=0)
            cols = torch.arange(F.shape[1], device=F.device)
            F.masked_fill_(first_bad[:, None] & (cols[None, :] >= col[:, None]), float("nan"))
            # the first output chunk the reference loses: the one before the chunk the segment starts in
            late = (first_bad, torch.div((seg - lcut).clamp(min=0), cs, rounding_mode="floor") - 1)
    fa = F[:, off + lcut:off + lcut + cs]
    fb = F[:, off + lcut + cs:off + lcut + cs + n_last]
    # NaN reach across the seam: the zero-phase steps have seen chunk n-2 only up to its head
    for j in range(emitted, nchunks - 2):
        dev.chain_zp_seal(fir, iir, ys[j], lcut + j * cs, lcut, cs)
        if j == nchunks - 3:
            ys[j].masked_fill_(~torch.isfinite(fa[:, -1:]), float("nan"))
        if late is not None:
            ys[j].masked_fill_((late[0] & (late[1] <= j))[:, None], float("nan"))
        yield from emit(ys.pop(j))
    yield from emit(iir.backward(fa, fb))
    yield from emit(iir.backward(fb, None))
    while flying:
        out, done = flying.popleft()
        done.synchronize()
        yield pipe.restore(out)


@dev.chain_aware
def sosfiltfilt(pro, sos, axis):
    """Forward-backward (zero-phase) cascaded-biquad filter
    (core/numerical.py:338-411) on the device (K2 + K3).

    Reproduces the reference's chunk-local scheme exactly: the forward pass
    starts from ``sosfilt_zi * x[0]`` (:374-386); the backward pass of chunk i
    starts from the state left by back-filtering forward chunk i+1 alone,
    itself started at ``sosfilt_zi * (its last sample)`` (:397-403); the last
    chunk starts at ``sosfilt_zi * (its own last sample)`` (:408-411).  Hence,
    like the reference, the result depends on ``pro.chunksize``.  The forward
    pass of every chunk is computed once and reused (the reference runs it
    twice, quirk Q11).
    """
    sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
    fused = _fir_feeding(pro, axis)
    if fused is not None:
        # a resident FIR producer feeding this filter: the fused steady-state step
        gen = _sosfiltfilt_after_fir(pro, *fused, sos)
        if gen is not None:
            yield from gen
            return
    layout = dev.Layout(pro.shape, axis)
    stream = dev.SosStream(sos, layout.nch)
    try:
        # zero-length arrays carry no samples (sosfilt skips them too)
        chunks = (c for c in pro if c.shape[layout.axis] > 0)
        first = next(chunks, None)
        if first is None:
            return
        # host-fed: every chunk goes up through the pinned ring on the H2D stream
        # while the previous step runs, results leave on the D2H stream and are
        # handed over two steps later (dev.HostPipe); resident chunks are views
        pipe = None if dev.is_tensor(first) else dev.HostPipe(layout)
        flying = deque()
        # Long streams of long chunks: the zero-phase kernel with the identity as its FIR --
        # both passes of a chunk as ONE spectrum multiply |H|^2 instead of two recurrences
        # over HBM (256 x 2^20: 1.2-1.45 instead of 1.7-1.85 ms per chunk) -- where its tables
        # take the cascade; the stream's two ends on the separate kernels, as behind a FIR.
        cs, total = int(pro.chunksize), int(pro.shape[layout.axis])
        from openseize_amd.core.producer import MaskedProducer
        # (a MaskedProducer's shape may name more samples than it yields, core/producer.py:399-408:
        # the flow below wants the chunk count up front)
        if (not isinstance(pro, MaskedProducer)
                and -(-total // cs) >= 6 and cs >= 65536 and first.shape[layout.axis] == cs
                and (pipe is not None or first.is_cuda) and os.environ.get("OSZ_CHAIN_ZP", "1") != "0"
                and stream.warm_len <= cs):
            ident = dev.FirStream(np.array([1.0, 0.0]), layout.nch)
            try:
                dev.chain_zp_tolerance(ident, stream, dev.zp_tolerance())
                lag = dev.chain_zp_lag(ident, stream)
                if lag >= 0 and cs >= max(4 * lag, stream.warm_len + lag, 2 * dev.chain_zp_min_chunk(ident, stream)):
                    yield from _zero_phase_stream(ident, stream, layout, pipe, flying, first, chunks,
                                                  np.array([1.0, 0.0]), cs, total, 0, 1, lag,
                                                  first.device if pipe is None else "cuda", what="sosfiltfilt")
                    return
            finally:
                ident.close()

        def put(chunk):
            if pipe is None or dev.is_tensor(chunk):
                return layout.to2d(chunk)
            x2d, ready = pipe.upload(chunk)
            pipe.compute.wait_event(ready)
            return x2d, True

        x2d, host = put(first)
        stream.set_state_scaled(x2d, 0)
        fwd, hosts = [stream.forward(x2d)], [host]
        second = next(chunks, None)
        if second is not None:
            b2d, host_b = put(second)
            fwd.append(stream.forward(b2d))
            hosts.append(host_b)
        n = int(np.ceil(pro.shape[axis] / pro.chunksize))
        idx = 1
        while fwd:
            # fwd[0] = forward chunk idx, fwd[1] = forward chunk idx+1 (if any).
            # The backward pass of chunk idx warms up over forward chunk idx+1;
            # the reference takes its "last chunk" branch for every idx >= n.
            fa = fwd[0]
            fb = fwd[1] if (len(fwd) > 1 and idx < n) else None
            nxt = next(chunks, None) if len(fwd) > 1 else None
            if nxt is not None:
                # steady state: forward of chunk idx+2 and backward of chunk
                # idx share one launch (osz_sosfiltfilt_step)
                c2d, host_c = put(nxt)
                fnew, y = stream.step(c2d, fa, fb)
                fwd.append(fnew)
                hosts.append(host_c)
            else:
                y = stream.backward(fa, fb)
            if pipe is not None and hosts[0] and not dev.emit_resident():
                flying.append(pipe.download(y))
                while len(flying) > 2:
                    out, done = flying.popleft()
                    done.synchronize()
                    yield pipe.restore(out)
            else:
                yield layout.from2d(y, hosts[0])
            fwd.pop(0)
            hosts.pop(0)
            idx += 1
        while flying:
            out, done = flying.popleft()
            done.synchronize()
            yield pipe.restore(out)
    finally:
        stream.close()


def _ba_to_sos(coeffs):
    """(b, a) -> (sos, order).  Order <= 2 is one biquad verbatim (the same
    DF2T recurrence scipy.signal.lfilter runs); higher orders are factored into
    second-order sections with scipy.signal.tf2sos (design-time host math).
    The cascade realises the same transfer function; on the reference's own
    ba test filters it agrees with the direct form to <= 3e-8 relative, the
    direct form being the less accurate of the two."""
    b, a = (np.atleast_1d(np.asarray(c, dtype=np.float64)) for c in coeffs)
    order = max(len(b), len(a)) - 1
    if order <= 2:
        bb = np.zeros(3)
        aa = np.zeros(3)
        bb[:len(b)] = b / a[0]
        aa[:len(a)] = a / a[0]
        return np.concatenate([bb, aa])[None, :], order
    return sps.tf2sos(b, a), order


def _cascade_state_map(sos):
    """Matrix M (2S x 2S) with  M @ s = n  where s = (z_00, z_01, z_10, ...)
    are the DF2T states of the sections of a cascade and n the coefficients
    (powers of q = z^-1, constant first) of the numerator of its ZERO-INPUT
    response  Y(q) = N(q) / A(q),  A = prod A_j.

    A section in state (z0, z1) emits (z0 + z1 q) / A_s(q) on its own; the
    sections after it filter that and the ones before it contribute their
    denominators to the common A, so section s adds
    (z_s0 + z_s1 q) * prod_{j > s} B_j(q) * prod_{j < s} A_j(q)  to N.
    A direct-form II transposed filter (b, a) in state z has N(q) = sum z_i q^i,
    so equating the two numerators maps one state onto the other."""
    nsec = sos.shape[0]
    cols = []
    for s in range(nsec):
        poly = np.ones(1)
        for j in range(nsec):
            if j < s:
                poly = np.convolve(poly, sos[j, 3:] / sos[j, 3])
            elif j > s:
                poly = np.convolve(poly, sos[j, :3] / sos[j, 3])
        base = np.zeros(2 * nsec)
        base[:len(poly)] = poly
        cols += [base, np.roll(base, 1)]          # z_s0 * P_s(q), z_s1 * q P_s(q)
    return np.stack(cols, axis=1)


def _ba_zi_to_sos_zi(zi, order, sos, axis):
    """User ``zi`` of a (b, a) filter (``order`` entries along axis, the state
    scipy.signal.lfilter takes) -> the equivalent ``zi`` of its biquad cascade,
    shaped (nsections, ..., 2 along axis, ...) for ``sosfilt``."""
    zi = np.asarray(zi, dtype=np.float64)
    ax = normalize_axis(axis, zi.ndim)
    if zi.shape[ax] != order:
        raise ValueError(f"zi must have {order} entries along axis {axis}")
    nsec = sos.shape[0]
    z = np.moveaxis(zi, ax, -1)
    rhs = np.zeros(z.shape[:-1] + (2 * nsec,))
    rhs[..., :order] = z
    M = _cascade_state_map(sos)
    try:
        states = np.linalg.solve(M, rhs[..., None])[..., 0]
    except np.linalg.LinAlgError:                 # sections sharing a root: minimum norm
        states = np.einsum("ij,...j->...i", np.linalg.pinv(M), rhs)
    states = states.reshape(z.shape[:-1] + (nsec, 2))
    states = np.moveaxis(states, -2, 0)           # (nsec, ..., 2)
    return np.moveaxis(states, -1, ax + 1)


def lfilter(pro, coeffs, axis, zi=None):
    """Transfer-function (b, a) forward filter with carried state
    (core/numerical.py:414-446), run on the device as a biquad cascade (see
    ``_ba_to_sos``).  ``zi`` has ``max(len(a), len(b)) - 1`` entries along axis
    (:437-440) and is the direct-form state scipy.signal.lfilter takes; it is
    mapped onto the cascade's section states (``_ba_zi_to_sos_zi``), so the
    output continues exactly as the direct form would."""
    sos, order = _ba_to_sos(coeffs)
    if zi is not None:
        zi = _ba_zi_to_sos_zi(zi, order, sos, axis)
    yield from sosfilt(pro, sos, axis, zi=zi)


def filtfilt(pro, coeffs, axis):
    """Transfer-function forward-backward filter (core/numerical.py:449-520):
    the same chunk-local scheme as ``sosfiltfilt`` started from the
    steady state (``lfilter_zi * x0`` there, the cascade's steady state
    here -- the same state of the same filter)."""
    sos, _ = _ba_to_sos(coeffs)
    yield from sosfiltfilt(pro, sos, axis)


# ---------------------------------------------------------------------------
# polyphase resampling (reference core/numerical.py:523-632)
# ---------------------------------------------------------------------------
def _resample_plan(pro, L, M, fs, fir, axis, kwargs):
    """(chunksize, anti-aliasing taps) of a rational resampling of ``pro`` -- the rules of the
    reference's preamble (core/numerical.py:566-587), which its outputs depend on:
      * a decimation factor that is not smaller than the stream is refused (ValueError);
      * at least three chunks: the chunksize is capped at a third of the stream's length, then
        raised to the next multiple of M (every chunk decimates without a remainder);
      * the interpolation / anti-aliasing filter is a low-pass at the tighter of the two Nyquist
        rates, fs / (2 max(L, M)), with a transition band of a tenth of that on either side,
        0.1 dB ripple and 40 dB attenuation unless ``fpass`` / ``fstop`` / ``gpass`` / ``gstop``
        say otherwise (they are consumed from ``kwargs``)."""
    nsamples = pro.shape[axis]
    if M >= nsamples:
        raise ValueError(f"cannot decimate by M = {M}: the data have only {nsamples} samples along "
                         f"axis {axis} (M must be smaller)")
    edge = fs / (2 * max(L, M))
    band = {"fpass": kwargs.pop("fpass", edge - edge / 10), "fstop": kwargs.pop("fstop", edge + edge / 10)}
    ripple = {"gpass": kwargs.pop("gpass", 0.1), "gstop": kwargs.pop("gstop", 40)}
    taps = fir(band["fpass"], band["fstop"], fs, ripple["gpass"], ripple["gstop"]).coeffs
    csize = min(pro.chunksize, nsamples // 3)
    csize = -(-csize // M) * M if csize % M else csize
    return max(int(csize), 1), taps


def _resample_padded(h, L, M, n):
    """(taps, centre) for ``dev.PolyStream``: the window with the zeros
    ``scipy.signal.resample_poly`` puts around it before it filters a stream of ``n`` samples
    (the reference: core/numerical.py:610, :631) -- ``n_pre_pad = M - half % M`` in front, so that
    the delay is whole outputs; ``n_post_pad`` behind, until upfirdn's output is long enough; every
    phase to one count of taps.  The numbers do not change (zeros), the reach of a non-finite
    sample does: 0 x NaN is NaN, and the kernels multiply every tap they are given."""
    h = np.asarray(h, dtype=np.float64)
    nh = len(h)
    nout = -(-n * L // M)
    half = (nh - 1) // 2
    pre = M - half % M
    remove = (half + pre) // M
    post = 0
    while ((n - 1) * L + nh + pre + post - 1) // M + 1 < nout + remove:      # (upfirdn's output length)
        post += 1
    K = -(-(nh + pre + post) // L)
    taps = np.zeros(K * L)
    taps[pre:pre + nh] = h
    return taps, half + pre


@dev.chain_aware
def polyphase_resample(pro, L, M, fs, fir, axis, **kwargs):
    """Rational L/M resampling of a producer (core/numerical.py:523-632) on
    the device (K4, ``osz_poly_*``).

    The reference resamples chunk by chunk with overhangs borrowed from the
    neighbouring chunks, which reproduces
    ``scipy.signal.resample_poly(x, L, M, window=h)`` of the whole stream; the
    device kernel evaluates that definition directly and carries the input
    history it needs.  What the outputs depend on in the reference's preamble
    is kept by ``_resample_plan`` (the refusal of M >= N, at least three chunks
    of a multiple of M samples, the default low-pass); the new chunksize is
    applied to ``pro`` in place, as ``producer(pro, csize, axis)`` does at :590.
    """
    csize, h = _resample_plan(pro, L, M, fs, fir, axis, kwargs)
    pro = producer(pro, csize, axis)

    layout = dev.Layout(pro.shape, axis)
    # the window as resample_poly filters with it: zeros in front and behind (a non-finite sample is
    # then lost to the outputs SciPy -- the reference -- loses it to; OSZ_POLY_PAD=0: the bare window)
    if os.environ.get("OSZ_POLY_PAD", "1") != "0":
        taps, centre = _resample_padded(h, int(L), int(M), int(pro.shape[axis]))
        stream = dev.PolyStream(taps, int(L), int(M), layout.nch, centre=centre)
    else:
        stream = dev.PolyStream(h, int(L), int(M), layout.nch)
    try:
        chunks = iter(pro)
        cur = next(chunks, None)
        if cur is not None and not dev.is_tensor(cur):
            # host-fed: transfers overlapped with the kernels (dev.HostPipe)
            yield from dev.HostPipe(layout).run(
                _chain_first(cur, chunks), lambda x2d, last: stream.push(x2d, final=last),
                tell_last=True)
            return
        while cur is not None:
            nxt = next(chunks, None)
            x2d, host = layout.to2d(cur)
            y = stream.push(x2d, final=nxt is None)
            if y.shape[1] > 0:
                yield layout.from2d(y, host)
            cur = nxt
    finally:
        stream.close()


# ---------------------------------------------------------------------------
# windowed DFT / periodogram / Welch / STFT
# (reference core/numerical.py:635-1087)
# ---------------------------------------------------------------------------
def _window_and_scale(window, nwin, fs, scaling):
    """Window coefficients (periodic, scipy.signal.get_window) and
    sqrt(norm) exactly as core/numerical.py:694, :703-716."""
    coeffs = sps.get_window(window, nwin)
    if scaling == "spectrum":
        norm = 1 / np.sum(coeffs) ** 2
    elif scaling == "density":
        norm = 1 / (fs * np.sum(coeffs ** 2))
    else:
        raise ValueError("Unknown scaling: {}".format(scaling))
    return coeffs, float(np.sqrt(norm))


def _linear_trend_refuses(out, detrend):
    """``scipy.signal.detrend(type='linear')`` is a least-squares fit, and SciPy's ``lstsq`` REFUSES
    non-finite data: the reference raises ``ValueError: array must not contain infs or NaNs`` at
    core/numerical.py:691 when a segment of any channel holds such a sample (type='constant'
    subtracts a mean: NaN goes through, there and here).  The kernels give such a segment back as
    NaN; this looks at one bin per segment and channel of what a push produced -- linear trend
    only -- and returns the index of the first segment the reference would have refused (None: none).
    ``out``: (segments, channels, bins)."""
    if detrend != "linear" or out.shape[0] == 0:
        return None
    probe = out[..., 0]                                  # (segments, channels): a lost segment is lost in every bin
    if dev.is_tensor(out):
        import torch
        bad = ~torch.isfinite(probe).reshape(out.shape[0], -1).all(1)
        return int(torch.nonzero(bad)[0, 0]) if bool(bad.any()) else None
    bad = ~np.isfinite(probe).reshape(out.shape[0], -1).all(1)
    return int(np.flatnonzero(bad)[0]) if bad.any() else None


_REFUSED = "array must not contain infs or NaNs"


def _one_shot(arr, fs, nfft, window, axis, detrend, scaling, mode):
    axis = normalize_axis(axis, arr.ndim)
    nsamples = arr.shape[axis]
    if nsamples == 0 or nfft < 1:
        raise ValueError(
            f"Invalid number of FFT data points ({min(nsamples, nfft)}) "
            "specified.")
    if detrend not in _lib.DETREND:
        raise ValueError("Trend type must be 'linear' or 'constant'.")
    if nfft < nsamples:
        # the reference crops along axis=-1 whatever `axis` is
        # (core/numerical.py:688); the sample axis is used here
        arr = slice_along_axis(arr, 0, nfft, axis=axis)
        nsamples = nfft
    coeffs, scale = _window_and_scale(window, nsamples, fs, scaling)
    layout = dev.Layout(arr.shape, axis)
    spec = dev.SpecStream(nsamples, nfft, nsamples, coeffs, scale, detrend,
                          mode, layout.nch)
    try:
        x2d, host = layout.to2d(arr)
        out = spec.push(x2d)          # (1, nch, nfreq)
        if _linear_trend_refuses(out, detrend) is not None:
            raise ValueError(_REFUSED)
        res = layout.from2d(out[0], host)
    finally:
        spec.close()
    return np.fft.rfftfreq(nfft, d=1 / fs), res


def modified_dft(arr, fs, nfft, window, axis, detrend, scaling):
    """Windowed DFT of a real array along axis (core/numerical.py:635-718):
    detrend, window, rFFT to ``nfft`` points (cropping or zero padding), scale
    by sqrt(norm).  Returns (freqs, complex array with nfft//2+1 along axis)."""
    return _one_shot(arr, fs, int(nfft), window, axis, detrend, scaling,
                     _lib.SPEC_DFT_SEGMENTS)


def periodogram(arr, fs, nfft=None, window="hann", axis=-1,
                detrend="constant", scaling="density"):
    """Windowed periodogram (core/numerical.py:721-796): |modified DFT|^2 with
    every bin but DC (and Nyquist for even nfft) doubled."""
    nfft = arr.shape[axis] if not nfft else int(nfft)
    return _one_shot(arr, fs, nfft, window, axis, detrend, scaling,
                     _lib.SPEC_PSD_SEGMENTS)


def _coarse(pro, nch):
    """A shallow copy of ``pro`` that yields the same stream in pieces of up
    to ~2^25 elements, a whole multiple of the original chunksize.  The
    estimators set chunksize = int(fs) on the caller's producer
    (spectra/estimators.py:141) -- that side effect is kept -- but iterating
    at one second of signal per array costs a Python round trip and a launch
    per second; the segment sequence does not depend on where the stream is
    cut, so the work is done on the coarse copy."""
    import copy
    from openseize_amd.core.producer import MaskedProducer
    target = max(1, min(1 << 20, (1 << 25) // max(int(nch), 1)))
    if target <= pro.chunksize:
        return pro
    big = (target // pro.chunksize) * pro.chunksize

    def clone(p):
        c = copy.copy(p)
        if isinstance(p, MaskedProducer):
            c.data, c.mask = clone(p.data), copy.copy(p.mask)
        c.chunksize = big
        return c

    return clone(pro)


def _batched(pro, axis, nch):
    """Joins consecutive produced arrays into pieces of up to ~2^25 elements
    before they go to the device.  The estimators force chunksize = int(fs)
    (spectra/estimators.py:141), i.e. a launch per second of signal; the
    segment sequence -- and so every estimate -- does not depend on where the
    stream is cut, so the cuts are moved, not the result."""
    target = max(1, min(1 << 20, (1 << 25) // max(int(nch), 1)))
    if isinstance(pro, ArrayProducer):
        # an in-memory array: larger views of it, no copies -- and of a RESIDENT array views of up
        # to 2^28 elements (the size of the headline's 256 x 2^20 chunk), whatever the channel
        # count: a launch over 2^25 elements of 32 channels is too short to hide what it pays once
        # (cfg-4's 8-GPU shard ran at 0.84 of the 256-channel rate)
        if dev.is_tensor(pro.data) and pro.data.is_cuda:
            target = max(1, min(1 << 24, (1 << 28) // max(int(nch), 1)))
        n = pro.data.shape[axis]
        for start in range(0, n, target):
            yield slice_along_axis(pro.data, start, min(start + target, n), axis=axis)
        return
    buf, count = [], 0
    for arr in pro:
        buf.append(arr)
        count += arr.shape[axis]
        if count >= target:
            yield buf[0] if len(buf) == 1 else dev.concatenate(buf, axis)
            buf, count = [], 0
    if buf:
        yield buf[0] if len(buf) == 1 else dev.concatenate(buf, axis)


@dev.chain_aware
def _spectra_estimatives(pro, fs, nfft, window, overlap, axis, detrend,
                         scaling, func, **kwargs):
    """One estimate per nfft-sample segment, ``stride = nfft - int(nfft *
    overlap)`` apart; a trailing partial segment is dropped
    (core/numerical.py:799-849).  ``func`` selects the estimate: periodogram
    -> PSD, modified_dft -> complex DFT.  The reference's FIFO lives on the
    device inside the ``osz_spec`` handle."""
    noverlap = int(nfft * overlap)
    stride = nfft - noverlap
    mode = (_lib.SPEC_PSD_SEGMENTS if func is periodogram
            else _lib.SPEC_DFT_SEGMENTS)
    coeffs, scale = _window_and_scale(window, nfft, fs, scaling)
    layout = dev.Layout(pro.shape, axis)
    spec = dev.SpecStream(nfft, nfft, stride, coeffs, scale, detrend, mode,
                          layout.nch)
    pipe = None
    try:
        for arr in _batched(pro, layout.axis, layout.nch):
            if dev.is_tensor(arr):
                x2d, host = layout.to2d(arr)
            else:
                pipe = pipe or dev.HostPipe(layout)   # pinned ring + H2D stream
                x2d, host = pipe.feed(arr), True
            if x2d.shape[1] == 0:
                continue
            out = spec.push(x2d)               # (nseg, nch, nfreq)
            nseg = out.shape[0]
            if nseg == 0:
                continue
            refused = _linear_trend_refuses(out, detrend)      # (nseg, nch, nfreq): before the axes move
            out = out.reshape((nseg,) + layout.other + (out.shape[-1],))
            out = out.movedim(-1, layout.axis + 1)
            out = out.cpu().numpy() if host else out
            for s in range(nseg if refused is None else refused):
                yield out[s]
            if refused is not None:
                raise ValueError(_REFUSED)
    finally:
        spec.close()


def welch(pro, fs, nfft, window, overlap, axis, detrend, scaling):
    """Producer of per-segment PS(D) estimates (core/numerical.py:852-947).
    Returns (freqs, producer); the producer's ``shape[axis]`` reports the
    segment count in the reference's float arithmetic (:941) although every
    yielded array has nfft//2+1 entries along axis (quirk Q7)."""
    if scaling not in ("spectrum", "density"):
        raise ValueError("Unknown scaling: {}".format(scaling))
    genfunc = partial(_spectra_estimatives, pro, fs, nfft, window, overlap,
                      axis, detrend, scaling, func=periodogram)
    freqs = np.fft.rfftfreq(nfft, 1 / fs)
    nsegs = int((pro.shape[axis] - nfft) // (nfft * (1 - overlap)) + 1)
    shape = list(pro.shape)
    shape[axis] = nsegs
    result = producer(genfunc, chunksize=len(freqs), axis=axis, shape=shape)
    return freqs, result


def stft(pro, fs, nfft, window, overlap, axis, detrend, scaling, boundary,
         padded):
    """Producer of per-segment modified DFTs (core/numerical.py:950-1087).
    ``boundary`` zero-extends nfft//2 both sides (:1041-1044); ``padded``
    appends a whole stride of zeros when N % stride != 0 (:1046-1051, quirk
    Q8); segment times as :1076-1083.  Returns (freqs, time, producer)."""
    if scaling not in ("spectrum", "density"):
        raise ValueError("Unknown scaling: {}".format(scaling))
    noverlap = int(nfft * overlap)
    stride = nfft - noverlap
    data = pro
    if boundary:
        data = protools.pad(data, nfft // 2, axis=pro.axis)
    if padded:
        nsamples = pro.shape[axis]
        amt = stride if nsamples % stride else 0
        data = protools.pad(data, [0, amt], axis=pro.axis)

    genfunc = partial(_spectra_estimatives, data, fs, nfft, window, overlap,
                      axis, detrend, scaling, func=modified_dft)
    freqs = np.fft.rfftfreq(nfft, 1 / fs)
    nsegs = int((data.shape[axis] - nfft) // (nfft * (1 - overlap)) + 1)
    shape = list(data.shape)
    shape[axis] = nsegs
    if boundary:
        time = 1 / fs * np.arange(0, data.shape[axis] - nfft + 1,
                                  nfft - noverlap)
    else:
        time = 1 / fs * np.arange(nfft // 2,
                                  data.shape[axis] + 1 - nfft // 2,
                                  nfft - noverlap)
    result = producer(genfunc, chunksize=len(freqs), axis=axis, shape=shape)
    return freqs, time, result

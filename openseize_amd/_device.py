"""Device plumbing: array-kind helpers and thin owners of the C-ABI handles.

PyTorch is used here only for device memory and the current HIP stream; all
numerics are the HIP kernels behind ``_lib``.  Host ndarrays go to the device
chunk by chunk and results come back as ndarrays; CUDA(HIP) tensors stay
resident end to end (a device-resident producer chain never touches the host).
"""

import ctypes
import os

import numpy as np

from openseize_amd import _lib

try:  # torch is plumbing; the import itself works without a GPU
    import torch
except Exception:  # pragma: no cover
    torch = None


def is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


def is_arraylike(a):
    return isinstance(a, np.ndarray) or is_tensor(a)


def concatenate(arrs, axis):
    """np.concatenate for ndarrays, torch.cat for device tensors."""
    if arrs and is_tensor(arrs[0]):
        return torch.cat(list(arrs), dim=axis)
    return np.concatenate(arrs, axis=axis)


def stack(arrs, axis):
    if arrs and is_tensor(arrs[0]):
        return torch.stack(list(arrs), dim=axis)
    return np.stack(arrs, axis=axis)


def size(a):
    return a.numel() if is_tensor(a) else a.size


def zeros_like_kind(ref, shape, value=0.0):
    """Constant array of `shape` of the same kind (host/device) as `ref`."""
    if is_tensor(ref):
        return torch.full(tuple(shape), float(value), dtype=ref.dtype,
                          device=ref.device)
    return value * np.ones(shape)


# -- small kind-agnostic array helpers for the producer-level glue (protools)
def expand_dims(a, axes):
    if is_tensor(a):
        for ax in sorted(ax % (a.ndim + len(axes)) for ax in axes):
            a = a.unsqueeze(ax)
        return a
    return np.expand_dims(a, axes)


def squeeze(a, axis=None):
    if is_tensor(a):
        return a.squeeze() if axis is None else a.squeeze(axis)
    return np.squeeze(a, axis)


def require_gpu():
    lib = _lib.load()          # raises OszLibraryError when the .so is missing
    if torch is None or not torch.cuda.is_available():
        raise RuntimeError(
            "openseize_amd needs a HIP device (MI355X): no GPU is visible. "
            "There is no CPU fallback.")
    return lib


def stream_ptr():
    """The calling thread's current stream, asked per call (a caller may switch streams
    between chunks); the raw query costs a fraction of building a torch.cuda.Stream."""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:   # pragma: no cover - other torch builds
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def host_dp(a):
    return a.ctypes.data_as(_lib.c_dp)


# ---------------------------------------------------------------------------
# chains of this library's generators over HOST data
# ---------------------------------------------------------------------------
# The reference's producers hand ndarrays from stage to stage.  Two stages of this
# library in a row would each cross PCIe: results down, then up again through the
# next stage's staging ring (a 256-tap FIR into sosfiltfilt from ndarrays: 0.95
# instead of 5 Gsamples/s).  Instead a generator of this library that is being
# pulled by ANOTHER generator of this library hands CUDA tensors on, and only the
# last stage of the chain -- the one the caller iterates -- goes back to the host.
#   _PULL:  the producer object a generator of this library is iterating right now (set
#           around every next() it makes on its source);
#   _GRANT: set by a GenProducer that finds ITSELF in _PULL -- its direct consumer is one
#           of ours -- to the generating function it is about to resume: that function,
#           if it is one of ours, hands CUDA tensors on.  (Identity on both hops: a stage
#           written by the user in between sees ndarrays from its source and is handed
#           ndarrays' worth of behaviour, whatever runs around it.)
#   _EMIT:  "the generator running right now hands CUDA tensors on" -- what the staging
#           code (HostPipe.run, Layout.from2d) asks.
import contextvars as _contextvars
import functools as _functools

_PULL = _contextvars.ContextVar("osz_pull", default=None)
_GRANT = _contextvars.ContextVar("osz_grant", default=None)
_EMIT = _contextvars.ContextVar("osz_emit_resident", default=False)
_END = object()


def emit_resident():
    return _EMIT.get()


def _looks_like_producer(obj):
    return hasattr(obj, "chunksize") and hasattr(obj, "shape") and hasattr(obj, "__iter__") \
        and not is_arraylike(obj)


def origin_is_host(pro):
    """Does the chain of producers under ``pro`` start at host data (an ndarray, a CPU
    tensor, a file reader)?  Static walk: array producers by their data, masked producers
    through their data producer, generating producers through the producer among the
    arguments of their (partial) function."""
    for _ in range(64):
        if not _looks_like_producer(pro):
            return False
        data = getattr(pro, "data", None)
        if isinstance(data, np.ndarray):
            return True
        if is_tensor(data):
            return not data.is_cuda
        if type(pro).__name__ == "ReaderProducer":
            # decoded on the device when asked to (file_io/edf.py: device=True)
            return not (getattr(pro, "kwargs", None) or {}).get("device", False)
        if _looks_like_producer(data):
            pro = data
            continue
        cands = list(getattr(data, "args", ()) or ())
        cands += list((getattr(data, "keywords", None) or {}).values())
        cands += list((getattr(pro, "kwargs", None) or {}).values())
        pro = next((c for c in cands if _looks_like_producer(c)), None)
    return False


def pull_resident(iterable, pro):
    """Iterate ``iterable`` -- chunks drawn from the producer ``pro`` -- as a consumer that
    takes CUDA tensors (the estimators and other non-generator consumers of a chain)."""
    it = iter(iterable)
    try:
        while True:
            token = _PULL.set(pro)
            try:
                item = next(it, _END)
            finally:
                _PULL.reset(token)
            if item is _END:
                return
            yield item
    finally:
        close = getattr(it, "close", None)       # an abandoned consumer releases its source
        if close is not None:
            close()


def relay_pull(outer, inner):
    """Iterator over ``inner`` for a producer ``outer`` that merely passes chunks on (a
    masked producer over its data producer): when ``outer``'s direct consumer is a
    generator of this library, ``inner`` is named in its place."""
    if _PULL.get() is outer:
        return pull_resident(inner, inner)
    return iter(inner)


def run_generating(producer, func, kwargs):
    """What a GenProducer iterates: ``func(**kwargs)``, resumed with the grant in place
    when the producer's direct consumer is a generator of this library."""
    direct = _PULL.get() is producer
    t1, t2 = _GRANT.set(func if direct else None), _PULL.set(None)
    try:
        gen = func(**kwargs)
    finally:
        _PULL.reset(t2)
        _GRANT.reset(t1)
    try:
        while True:
            t1, t2 = _GRANT.set(func if direct else None), _PULL.set(None)
            try:
                item = next(gen, _END)
            finally:
                _PULL.reset(t2)
                _GRANT.reset(t1)
            if item is _END:
                return
            yield item
    finally:
        close = getattr(gen, "close", None)
        if close:
            close()


def to_host_async(t):
    """CUDA tensor -> (pinned host tensor, event): the copy runs on a stream of its own
    behind everything queued on the current stream."""
    cur = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(cur)
    side = _d2h_stream()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        out = torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)
        out.copy_(t, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    t.record_stream(side)
    return out, done


_D2H = None


def _d2h_stream():
    global _D2H
    if _D2H is None:
        _D2H = torch.cuda.Stream()
    return _D2H


def chain_aware(fn):
    """Decorator for the generator functions ``fn(pro, ...)`` of core/numerical.py: see
    the note above.  Resident chains pass through untouched."""
    from collections import deque

    @_functools.wraps(fn)
    def wrapper(pro, *args, **kwargs):          # a generator function itself (producer() asks)
        grant = _GRANT.get()                    # does whoever pulls me take CUDA tensors?
        outer = grant is not None and getattr(grant, "func", grant) is wrapper
        import os
        if not origin_is_host(pro) or os.environ.get("OSZ_HOST_CHAIN") == "0":   # (A/B knob)
            yield from fn(pro, *args, **kwargs)
            return
        it = fn(pro, *args, **kwargs)
        flying = deque()
        try:
            while True:
                t1, t2, t3 = _PULL.set(pro), _EMIT.set(outer), _GRANT.set(None)
                try:
                    item = next(it, _END)
                finally:
                    _GRANT.reset(t3)
                    _EMIT.reset(t2)
                    _PULL.reset(t1)
                if item is _END:
                    break
                if outer or not (is_tensor(item) and item.is_cuda):
                    # results leave in the order they were produced: whatever is still on
                    # its way down goes first (a generator that hands out an ndarray behind
                    # CUDA tensors, e.g. a tail assembled on the host)
                    while flying:
                        out, done = flying.popleft()
                        done.synchronize()
                        yield out.numpy()
                    yield item
                    continue
                # the caller's stage: back to the host, two transfers in flight
                flying.append(to_host_async(item))
                while len(flying) > 2:
                    out, done = flying.popleft()
                    done.synchronize()
                    yield out.numpy()
            while flying:
                out, done = flying.popleft()
                done.synchronize()
                yield out.numpy()
        finally:
            it.close()
    return wrapper


class Layout:
    """Maps an N-D chunk with a sample axis to the (channels, samples)
    row-major layout of the C ABI and back.  (The reference's functions take
    any axis: core/numerical.py passes ``axis`` straight to SciPy.)"""

    def __init__(self, shape, axis):
        shape = tuple(int(s) for s in shape)
        self.ndim = len(shape)
        self.axis = axis % self.ndim
        self.other = shape[:self.axis] + shape[self.axis + 1:]
        self.nch = int(np.prod(self.other)) if self.other else 1

    def to2d(self, arr):
        """-> (float64 CUDA tensor (nch, n), was_host)."""
        if is_tensor(arr):
            host = False
            t = arr
            if t.dtype != torch.float64:
                t = t.to(torch.float64)
            if not t.is_cuda:
                t = t.cuda()
                host = True
        else:
            host = True
            a = np.asarray(arr, dtype=np.float64)
            t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        t = torch.movedim(t, self.axis, -1)
        n = t.shape[-1]
        t = t.reshape(self.nch, n)
        # the C ABI takes a row pitch: a chunk that is a column range of a larger
        # resident array (what ArrayProducer yields) goes in as the view it is
        if n > 0 and (t.stride(1) != 1 or t.stride(0) < n):
            t = t.contiguous()
        return t, host

    def from2d(self, t, host):
        """(nch, m) tensor -> array of the original kind with the sample axis
        restored."""
        m = t.shape[-1]
        out = t.reshape(self.other + (m,))
        out = torch.movedim(out, -1, self.axis)
        if host and not _EMIT.get():
            return out.cpu().numpy()
        # (a resident result whose sample axis is the last one is handed on as the view it is:
        # the chunks of a grouped zero-phase step are column ranges of the step's buffer, and a
        # copy of each would move as many bytes again as the kernel did)
        return out if self.axis == self.ndim - 1 else out.contiguous()


# ---------------------------------------------------------------------------
# host-fed streams: pinned staging + H2D / compute / D2H on three streams
# ---------------------------------------------------------------------------
_COPY_POOL = None


def _copy_pool():
    global _COPY_POOL
    if _COPY_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 4) // 2)))
    return _COPY_POOL


def _threaded_copy(dst, src):
    """dst[...] = src for 2-D arrays, split over a few threads (numpy releases
    the GIL in the copy): one core moves ~10 GB/s, PCIe wants more."""
    rows, cols = dst.shape
    if dst.size < (1 << 16):
        np.copyto(dst, src)
        return
    fast = (src.ndim == 2 and src.dtype == dst.dtype and src.shape == dst.shape and cols > 0
            and src.strides[1] == src.itemsize and dst.strides[1] == dst.itemsize
            and src.strides[0] >= cols * src.itemsize and dst.strides[0] >= cols * dst.itemsize)
    if fast:
        # a column range of a C-ordered array: rows a pitch apart -- the library's own threads
        # (Python's pool spends most of a 4 MB copy waking its workers)
        _lib.check(_lib.load().osz_host_copy2d(dst.ctypes.data, dst.strides[0], src.ctypes.data, src.strides[0],
                                               rows, cols * src.itemsize))
        return
    if dst.size < (1 << 18):       # (Python's pool costs more than it gains below a couple of MB)
        np.copyto(dst, src)
        return
    parts = 8
    if rows >= parts:
        cuts = [(slice(r * rows // parts, (r + 1) * rows // parts), slice(None)) for r in range(parts)]
    else:
        cuts = [(slice(None), slice(c * cols // parts, (c + 1) * cols // parts)) for c in range(parts)]
    list(_copy_pool().map(lambda sl: np.copyto(dst[sl], src[sl]), cuts))


def gather_along_axis(pieces, counts, axis):
    """One new array = the first counts[i] samples of pieces[i] along ``axis``,
    joined (what a FIFO pop needs).  Device tensors: torch.cat of views.  Host
    arrays: one allocation filled by a few threads, the destination split
    across the leading rows so every thread writes its own cache lines."""
    if is_tensor(pieces[0]):
        return torch.cat([p.narrow(axis, 0, c) for p, c in zip(pieces, counts) if c > 0], dim=axis)
    first = pieces[0]
    axis = axis % first.ndim
    shape = list(first.shape)
    shape[axis] = int(sum(counts))
    out = np.empty(shape, dtype=np.result_type(*[p.dtype for p in pieces]))
    jobs, at = [], 0
    for p, c in zip(pieces, counts):
        if c > 0:
            idx = [slice(None)] * first.ndim
            idx[axis] = slice(at, at + c)
            src = [slice(None)] * first.ndim
            src[axis] = slice(0, c)
            jobs.append((tuple(idx), p, tuple(src)))
            at += c
    if out.size < (1 << 18):
        for idx, p, src in jobs:
            out[idx] = p[src]
        return out
    # split every job along the longest other axis (or the sample axis itself)
    tasks = []
    for idx, p, src in jobs:
        d, s_ = out[idx], p[src]
        cut_axis = max(range(d.ndim), key=lambda k: d.shape[k] if k != axis else -1) if d.ndim > 1 else 0
        if d.ndim == 1 or d.shape[cut_axis] < 4:
            cut_axis = axis
        n = d.shape[cut_axis]
        parts = min(8, n) or 1
        for q in range(parts):
            sl = [slice(None)] * d.ndim
            sl[cut_axis] = slice(q * n // parts, (q + 1) * n // parts)
            tasks.append((d[tuple(sl)], s_[tuple(sl)]))
    list(_copy_pool().map(lambda t: np.copyto(t[0], t[1]), tasks))
    return out


class HostPipe:
    """Overlaps the transfers of a host-fed stream with its kernels.  The
    reference's real sources are host ndarrays and EDF files
    (core/producer.py:289-295, file_io/edf.py:558-586); taken one chunk at a
    time, synchronously and from pageable memory, they reach ~1/8 of what PCIe
    allows.  Here every chunk is copied (threaded) into a pinned ring buffer and
    sent on an H2D stream, the kernels run on the compute stream, results leave
    on a D2H stream into pinned arrays that are handed to the caller as
    ndarrays (torch's caching pinned allocator recycles them once dropped), and
    the generators built on ``run`` read one or two chunks ahead so the three
    stages of neighbouring chunks overlap."""

    def __init__(self, layout, slots=3):
        require_gpu()
        self.layout = layout
        self.h2d = torch.cuda.Stream()
        self.d2h = torch.cuda.Stream()
        self.compute = torch.cuda.current_stream()
        self.slots = [None] * slots          # (pinned tensor, last H2D event)
        self.turn = 0

    def upload(self, arr):
        """ndarray chunk -> (float64 CUDA tensor (nch, n), ready event)."""
        lay = self.layout
        a = np.asarray(arr, dtype=np.float64)
        moved = np.moveaxis(a, lay.axis, -1)
        n = moved.shape[-1]
        src = moved.reshape(lay.nch, n)
        k = self.turn % len(self.slots)
        self.turn += 1
        slot = self.slots[k]
        if slot is not None:
            slot[1].synchronize()                        # its previous transfer has left the buffer
        if slot is None or slot[0].numel() < lay.nch * n:
            slot = [torch.empty(max(lay.nch * n, 1), dtype=torch.float64, pin_memory=True), None]
        stage = slot[0][:lay.nch * n].view(lay.nch, n)
        _threaded_copy(stage.numpy(), src)
        with torch.cuda.stream(self.h2d):
            x = torch.empty((lay.nch, n), dtype=torch.float64, device="cuda")
            x.copy_(stage, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.h2d)
        x.record_stream(self.compute)
        slot[1] = ev
        self.slots[k] = slot
        return x, ev

    def feed(self, arr):
        """``upload`` for consumers that launch on the compute stream right away:
        the compute stream is made to wait for the transfer, the CPU is not."""
        x2d, ready = self.upload(arr)
        self.compute.wait_event(ready)
        return x2d

    def download(self, y2d):
        """Device result -> (pinned host tensor, done event), behind the kernels
        queued so far on the compute stream."""
        done = torch.cuda.Event()
        done.record(self.compute)
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(done)
            out = torch.empty(tuple(y2d.shape), dtype=y2d.dtype, pin_memory=True)
            out.copy_(y2d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.d2h)
        y2d.record_stream(self.d2h)
        return out, ev

    def restore(self, out):
        """Pinned (nch, m) result -> ndarray with the sample axis back in place."""
        lay = self.layout
        arr = out.numpy().reshape(lay.other + (out.shape[-1],))
        return np.moveaxis(arr, -1, lay.axis)

    def run(self, chunks, op, lookahead=2, tell_last=False):
        """Generator: for every ndarray chunk, ``op(x2d) -> y2d`` (a CUDA tensor,
        or None / zero columns for "nothing to emit") in stream order; yields
        the results as ndarrays.  The next chunk is uploaded before the current
        one is computed, and up to ``lookahead`` results are in flight.  With
        ``tell_last`` the call is ``op(x2d, is_last_chunk)``."""
        from collections import deque
        it = iter(chunks)
        flying = deque()
        nxt = next(it, None)
        staged = self.upload(nxt) if nxt is not None else None
        while staged is not None:
            x2d, ready = staged
            nxt = next(it, None)
            staged = self.upload(nxt) if nxt is not None else None
            self.compute.wait_event(ready)
            y2d = op(x2d, staged is None) if tell_last else op(x2d)
            if y2d is not None and y2d.shape[-1] > 0:
                if _EMIT.get():
                    # pulled by another generator of this library: the result stays in HBM
                    yield self.layout.from2d(y2d, False)
                    continue
                flying.append(self.download(y2d))
            while len(flying) > lookahead:
                out, ev = flying.popleft()
                ev.synchronize()
                yield self.restore(out)
        while flying:
            out, ev = flying.popleft()
            ev.synchronize()
            yield self.restore(out)


class _Handle:
    _destroy = None
    _state = None          # prefix of the osz_*_state_size / _get_state / _set_state trio

    def __init__(self):
        self.h = ctypes.c_void_p()
        self.lib = require_gpu()

    def get_state(self):
        """The iterator's carried state as a flat host array (checkpoint)."""
        n = getattr(self.lib, self._state + "_state_size")(self.h)
        buf = np.empty(int(n))
        _lib.check(getattr(self.lib, self._state + "_get_state")(self.h, host_dp(buf), stream_ptr()))
        return buf

    def set_state(self, state):
        """Resume from a state taken with ``get_state`` on an identical handle."""
        n = getattr(self.lib, self._state + "_state_size")(self.h)
        buf = np.ascontiguousarray(state, dtype=np.float64)
        if buf.shape != (int(n),):
            raise ValueError(f"state must have {int(n)} entries, got {buf.shape}")
        _lib.check(getattr(self.lib, self._state + "_set_state")(self.h, host_dp(buf), stream_ptr()))

    def _state_count(self):
        return int(getattr(self.lib, self._state + "_state_size")(self.h))

    _device_state = False  # osz_fir_* / osz_sos_*: the state calls take device arrays too

    def snapshot(self):
        """The carried state as a device tensor: a copy ordered on the current stream that
        nothing waits for (the C ABI takes a device array in place of the host one)."""
        if not self._device_state:
            raise TypeError(f"{type(self).__name__}: the state holds host scalars, use get_state")
        buf = torch.empty(self._state_count(), dtype=torch.float64, device="cuda")
        _lib.check(getattr(self.lib, self._state + "_get_state")(
            self.h, ctypes.cast(buf.data_ptr(), _lib.c_dp), stream_ptr()))
        return buf

    def restore(self, snap):
        """Back to a ``snapshot`` of this handle, ordered on the current stream."""
        if (not self._device_state or snap.shape != (self._state_count(),) or snap.dtype != torch.float64
                or not snap.is_cuda or not snap.is_contiguous()):
            raise ValueError("restore: not a snapshot of this handle")
        _lib.check(getattr(self.lib, self._state + "_set_state")(
            self.h, ctypes.cast(snap.data_ptr(), _lib.c_dp), stream_ptr()))

    def close(self):
        if self.h:
            getattr(self.lib, self._destroy)(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover - interpreter teardown order
        try:
            self.close()
        except Exception:
            pass


class SosStream(_Handle):
    """One iterator's cascade state (C ABI: osz_sos_*)."""
    _destroy = "osz_sos_destroy"

    def __init__(self, sos, nch):
        super().__init__()
        sos = np.ascontiguousarray(sos, dtype=np.float64)
        if sos.ndim != 2 or sos.shape[1] != 6:
            raise ValueError("sos array must be shape (n_sections, 6)")
        self.nsec, self.nch = sos.shape[0], nch
        _lib.check(self.lib.osz_sos_create(ctypes.byref(self.h), host_dp(sos),
                                           self.nsec, nch))

    def set_state(self, zi):
        """zi: None or host array (nsec, nch, 2)."""
        if zi is None:
            _lib.check(self.lib.osz_sos_set_state(self.h, None, stream_ptr()))
            return
        zi = np.ascontiguousarray(zi, dtype=np.float64)
        if zi.shape != (self.nsec, self.nch, 2):
            raise ValueError(
                f"Invalid zi shape. Expected {(self.nsec, self.nch, 2)}, "
                f"got {zi.shape}")
        _lib.check(self.lib.osz_sos_set_state(self.h, host_dp(zi), stream_ptr()))

    _state = "osz_sos"
    _device_state = True

    def _state_count(self):
        return self.nsec * self.nch * 2

    def get_state(self):
        zf = np.empty((self.nsec, self.nch, 2))
        _lib.check(self.lib.osz_sos_get_state(self.h, host_dp(zf), stream_ptr()))
        return zf

    def set_zi_unit(self, zi_unit):
        zi_unit = np.ascontiguousarray(zi_unit, dtype=np.float64)
        _lib.check(self.lib.osz_sos_set_zi_unit(self.h, host_dp(zi_unit)))

    def set_state_scaled(self, x2d, col):
        _lib.check(self.lib.osz_sos_set_state_scaled(
            self.h, ptr(x2d), x2d.stride(0), col, stream_ptr()))

    def forward(self, x2d, out=None):
        if x2d.dim() != 2 or x2d.shape[0] != self.nch:
            raise ValueError(f"SosStream.forward: {tuple(x2d.shape)} is not ({self.nch}, n)")
        y = torch.empty_like(x2d) if out is None else out
        _lib.check(self.lib.osz_sos_forward(
            self.h, ptr(x2d), x2d.stride(0), ptr(y), y.stride(0),
            x2d.shape[1], stream_ptr()))
        return y

    @property
    def warm_len(self):
        """Samples of the next chunk the backward warm-up reads (osz_sos_warm_len)."""
        return int(self.lib.osz_sos_warm_len(self.h))

    def backward(self, fa, fb=None, out=None):
        """Chunk-local backward sweep (osz_sosfiltfilt_chunk)."""
        y = torch.empty_like(fa) if out is None else out
        _lib.check(self.lib.osz_sosfiltfilt_chunk(
            self.h, ptr(fa), fa.stride(0), fa.shape[1],
            ptr(fb) if fb is not None else None,
            fb.stride(0) if fb is not None else 0,
            fb.shape[1] if fb is not None else 0,
            ptr(y), y.stride(0), stream_ptr()))
        return y


    def step(self, x2d, fa, fb=None, f_out=None, y_out=None):
        """Forward-filter x2d and back-filter fa (warm-up over fb) in one
        launch (osz_sosfiltfilt_step).  Returns (forward of x2d, backward of
        fa)."""
        f = torch.empty_like(x2d) if f_out is None else f_out
        y = torch.empty_like(fa) if y_out is None else y_out
        _lib.check(self.lib.osz_sosfiltfilt_step(
            self.h, ptr(x2d), x2d.stride(0), x2d.shape[1], ptr(f), f.stride(0),
            ptr(fa), fa.stride(0), fa.shape[1],
            ptr(fb) if fb is not None else None,
            fb.stride(0) if fb is not None else 0,
            fb.shape[1] if fb is not None else 0,
            ptr(y), y.stride(0), stream_ptr()))
        return f, y


class FirStream(_Handle):
    """One iterator's overlap-add state (C ABI: osz_fir_*)."""
    _destroy = "osz_fir_destroy"
    _state = "osz_fir"
    _device_state = True

    def __init__(self, taps, nch):
        super().__init__()
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.ntaps, self.nch = len(taps), nch
        _lib.check(self.lib.osz_fir_create(ctypes.byref(self.h), host_dp(taps),
                                           self.ntaps, nch))

    def push(self, x2d, skip=0, out=None):
        if x2d.dim() != 2 or x2d.shape[0] != self.nch:
            raise ValueError(f"FirStream.push: {tuple(x2d.shape)} is not ({self.nch}, n)")
        n = x2d.shape[1]
        if not 0 <= skip <= n:
            raise ValueError(f"skip={skip} not in [0, {n}]")
        y = out if out is not None else torch.empty(
            (self.nch, n - skip), dtype=torch.float64, device=x2d.device)
        _lib.check(self.lib.osz_fir_push(
            self.h, ptr(x2d), x2d.stride(0), n, ptr(y), max(y.stride(0), 1),
            skip, stream_ptr()))
        return y

    def reset(self):
        """Forget the carried overlap tail (C ABI: osz_fir_reset)."""
        _lib.check(self.lib.osz_fir_reset(self.h, stream_ptr()))

    def flush(self, device, skip=0, drop=0, out=None):
        cnt = self.ntaps - 1 - skip - drop
        y = out if out is not None else torch.empty(
            (self.nch, max(cnt, 0)), dtype=torch.float64, device=device)
        if cnt > 0:
            _lib.check(self.lib.osz_fir_flush(self.h, ptr(y), y.stride(0), skip,
                                              drop, stream_ptr()))
        return y


def chain_forward(fir, sos, x2d, out=None):
    """f = sosfilt(fir(x)) for the next chunk in one fused launch (C ABI:
    osz_chain_forward); both iterators' carried states advance."""
    n = x2d.shape[1]
    f = out if out is not None else torch.empty((fir.nch, n), dtype=torch.float64,
                                                 device=x2d.device)
    _lib.check(fir.lib.osz_chain_forward(fir.h, sos.h, ptr(x2d), x2d.stride(0), n, ptr(f),
                                         max(f.stride(0), 1), stream_ptr()))
    return f


def chain_forward_route(fir, sos):
    """Which kernel chain_forward runs this pair's whole blocks on (C ABI:
    osz_chain_forward_route): 2 one block per transform with the cascade in the spectrum,
    1 a pair of blocks per transform, 0 the cascade as a scan in time."""
    r = int(fir.lib.osz_chain_forward_route(fir.h, sos.h, stream_ptr()))
    if r < 0:
        raise RuntimeError(fir.lib.osz_last_error().decode("utf-8", "replace"))
    return r


def chain_step(fir, sos, x2d, fa, fb=None, f_out=None, y_out=None, defer=False):
    """One steady-state step of FIR -> sosfiltfilt (C ABI: osz_chain_step): the
    fused forward half of chunk ``x2d`` -> f, and beside it, on the SOS
    handle's own stream, the backward sweep of the earlier forward chunk
    ``fa`` (warmed up over ``fb``) -> y.  Stream-ordered for the caller; with
    ``defer`` y (and the right to overwrite fa / fb) is the caller's only after
    the next ``chain_step`` / ``chain_wait`` on these handles."""
    n = x2d.shape[1]
    f = f_out if f_out is not None else torch.empty((fir.nch, n), dtype=torch.float64,
                                                     device=x2d.device)
    y = torch.empty_like(fa) if y_out is None else y_out
    _lib.check(fir.lib.osz_chain_step(
        fir.h, sos.h, ptr(x2d), x2d.stride(0), n, ptr(f), max(f.stride(0), 1),
        ptr(fa), fa.stride(0), fa.shape[1],
        ptr(fb) if fb is not None else None,
        fb.stride(0) if fb is not None else 0,
        fb.shape[1] if fb is not None else 0,
        ptr(y), y.stride(0), _lib.CHAIN_DEFER if defer else 0, stream_ptr()))
    return f, y


def chain_wait(sos):
    """The current stream is ordered behind a deferred backward pass (osz_chain_wait)."""
    _lib.check(sos.lib.osz_chain_wait(sos.h, stream_ptr()))


def chain_zp_lag(fir, sos):
    """Delay of the zero-phase chain kernel in samples, or -1 when this pair of filters
    does not take it (C ABI: osz_chain_zp_lag)."""
    return int(fir.lib.osz_chain_zp_lag(fir.h, sos.h))


def chain_zp_min_chunk(fir, sos):
    return int(fir.lib.osz_chain_zp_min_chunk(fir.h, sos.h))


def chain_zp_tolerance(fir, sos, tol):
    """Where the spectral chain kernels cut their bursts, relative to the norm of the composite
    impulse response (C ABI: osz_chain_zp_tolerance; 0: the default 1e-15).  Before chain_zp_lag."""
    _lib.check(fir.lib.osz_chain_zp_tolerance(fir.h, sos.h, float(tol)))


def chain_zp_reach(fir, sos, step):
    """The reference FIR's segment length in input samples (C ABI: osz_chain_zp_reach): the NaN
    reach of the chain counts from the start of the segment that holds a non-finite sample."""
    _lib.check(fir.lib.osz_chain_zp_reach(fir.h, sos.h, int(step)))


def zp_tolerance():
    """The cut the generators ask for: the library's default (1e-15 of the response's norm -- what
    is cut scales with the INPUT's magnitude, about 0.3 tol max|x|, and at 1e-15 that is float64's
    own rounding on any stream, whatever offset, step or rail it carries and wherever) unless
    ``OSZ_ZP_TOL`` names another one for data known to be zero-mean (1e-12: one burst row less each
    way, 2 % less time, 6e-13 of the output scale on such data).  Rounds 3-4 chose between the two
    by a look at the first 8192 samples of the first chunk; an offset that appeared later got the
    loose cut, so nothing is decided from the data any more."""
    env = os.environ.get("OSZ_ZP_TOL")
    return float(env) if env else 0.0


def chain_zp_open(fir, sos, skip=0):
    """Starts a zero-phase stream (C ABI: osz_chain_zp_open): the forward cascade starts
    from the state on ``sos`` at stream sample ``skip``."""
    _lib.check(fir.lib.osz_chain_zp_open(fir.h, sos.h, int(skip), stream_ptr()))


def chain_zp_step(fir, sos, x2d, out=None, tail=None):
    """FIR -> sosfiltfilt of the next chunk in ONE kernel (C ABI: osz_chain_zp_step).  The
    chunk's n output samples -- stream samples ``pos - lag`` onwards -- go to ``tail`` (its
    width of them: the end of the caller's previous output chunk) and then to ``out``."""
    n = x2d.shape[1]
    n0 = 0 if tail is None else tail.shape[1]
    y = out if out is not None else torch.empty((fir.nch, n - n0), dtype=torch.float64, device=x2d.device)
    _lib.check(fir.lib.osz_chain_zp_step(
        fir.h, sos.h, ptr(x2d), x2d.stride(0), n, ptr(tail) if n0 else None, tail.stride(0) if n0 else 0, n0,
        ptr(y) if n > n0 else None, max(y.stride(0), 1), stream_ptr()))
    return y


def chain_zp_seal(fir, sos, y, s0, origin, cs):
    """NaN reach of sosfiltfilt on output samples [s0, s0 + n) (C ABI: osz_chain_zp_seal)."""
    _lib.check(fir.lib.osz_chain_zp_seal(fir.h, sos.h, ptr(y), max(y.stride(0), 1), y.shape[1], int(s0),
                                         int(origin), int(cs), stream_ptr()))


def chain_zp_finish(fir, sos, x2d=None, out=None):
    """Ends the zero-phase part of a stream (C ABI: osz_chain_zp_finish): the handles' own
    states are those at the end of the samples stepped so far, ``out`` receives the next
    output samples, computed from the head ``x2d`` of what follows."""
    if x2d is None or out is None:
        _lib.check(fir.lib.osz_chain_zp_finish(fir.h, sos.h, None, 0, 0, None, 0, 0, stream_ptr()))
        return None
    _lib.check(fir.lib.osz_chain_zp_finish(fir.h, sos.h, ptr(x2d), x2d.stride(0), x2d.shape[1], ptr(out),
                                           max(out.stride(0), 1), out.shape[1], stream_ptr()))
    return out


class PolyStream(_Handle):
    """One iterator's polyphase resampler state (C ABI: osz_poly_*)."""
    _destroy = "osz_poly_destroy"
    _state = "osz_poly"

    def __init__(self, taps, L, M, nch, centre=None):
        """``centre``: the tap that lines up with an output (default: the middle one) -- a window
        handed in with SciPy's zero padding around it (numerical._resample_padded)."""
        super().__init__()
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.nch = nch
        if centre is None:
            _lib.check(self.lib.osz_poly_create(ctypes.byref(self.h), host_dp(taps),
                                                len(taps), L, M, nch))
        else:
            _lib.check(self.lib.osz_poly_create_centred(ctypes.byref(self.h), host_dp(taps),
                                                        len(taps), int(centre), L, M, nch))

    def push(self, x2d, final):
        if x2d.dim() != 2 or x2d.shape[0] != self.nch:
            raise ValueError(f"PolyStream.push: {tuple(x2d.shape)} is not ({self.nch}, n)")
        n = x2d.shape[1]
        cnt = self.lib.osz_poly_out_count(self.h, n, int(final))
        y = torch.empty((self.nch, max(cnt, 0)), dtype=torch.float64,
                        device=x2d.device)
        nout = ctypes.c_int64()
        _lib.check(self.lib.osz_poly_push(
            self.h, ptr(x2d), x2d.stride(0), n, int(final), ptr(y),
            max(y.stride(0), 1), ctypes.byref(nout), stream_ptr()))
        return y


class SpecStream(_Handle):
    """One iterator's segmenter / windowed-DFT state (C ABI: osz_spec_*)."""
    _destroy = "osz_spec_destroy"
    _state = "osz_spec"

    def __init__(self, nwin, nfft, stride, window, scale, detrend, mode, nch):
        super().__init__()
        window = np.ascontiguousarray(window, dtype=np.float64)
        if detrend not in _lib.DETREND:
            raise ValueError("Trend type must be 'linear' or 'constant'.")
        self.nfreq, self.nch, self.mode = nfft // 2 + 1, nch, mode
        _lib.check(self.lib.osz_spec_create(
            ctypes.byref(self.h), nwin, nfft, stride, host_dp(window),
            float(scale), _lib.DETREND[detrend], mode, nch))

    def push(self, x2d):
        """Returns a (nseg, nch, nfreq) tensor (f64 or c128) or None in
        PSD_MEAN mode."""
        n = x2d.shape[1]
        nseg = self.lib.osz_spec_seg_count(self.h, n)
        out = None
        if self.mode == _lib.SPEC_PSD_SEGMENTS:
            out = torch.empty((nseg, self.nch, self.nfreq), dtype=torch.float64,
                              device=x2d.device)
        elif self.mode == _lib.SPEC_DFT_SEGMENTS:
            out = torch.empty((nseg, self.nch, self.nfreq),
                              dtype=torch.complex128, device=x2d.device)
        got = ctypes.c_int64()
        _lib.check(self.lib.osz_spec_push(
            self.h, ptr(x2d), x2d.stride(0), n,
            ptr(out) if out is not None else None, ctypes.byref(got),
            stream_ptr()))
        return out

    def sum_tensor(self):
        """(device tensor view of the running periodogram sum, count)."""
        dsum, cnt = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self.lib.osz_spec_sum(self.h, ctypes.byref(dsum),
                                         ctypes.byref(cnt)))
        return dsum, cnt.value

    def export_sum(self, device="cuda"):
        """(CUDA tensor copy of the running periodogram sum, segment count)."""
        out = torch.empty((self.nch, self.nfreq), dtype=torch.float64,
                          device=device)
        cnt = ctypes.c_int64()
        _lib.check(self.lib.osz_spec_export_sum(self.h, ptr(out),
                                                ctypes.byref(cnt), stream_ptr()))
        return out, cnt.value

    def mean(self):
        out = np.empty((self.nch, self.nfreq))
        cnt = ctypes.c_int64()
        _lib.check(self.lib.osz_spec_mean(self.h, host_dp(out),
                                          ctypes.byref(cnt), stream_ptr()))
        return cnt.value, out

    def mean_device(self, device="cuda"):
        """(count, CUDA tensor (nch, nfreq)): the segment average taken on the
        device (osz_spec_mean_device); nothing crosses PCIe."""
        out = torch.empty((self.nch, self.nfreq), dtype=torch.float64, device=device)
        cnt = ctypes.c_int64()
        _lib.check(self.lib.osz_spec_mean_device(self.h, ptr(out), ctypes.byref(cnt),
                                                 stream_ptr()))
        return cnt.value, out

    def welch_reduce(self, comm):
        """All-reduce (sum) of the periodogram accumulator and the segment
        count over the ranks of an ``RcclComm`` (osz_welch_reduce)."""
        _lib.check(self.lib.osz_welch_reduce(self.h, comm.comm, stream_ptr()))


class RcclComm:
    """An RCCL communicator created through the C ABI (osz_rccl_*), for hosts
    that do not run torch.distributed.  ``unique_id()`` on rank 0 gives the 128
    bytes every rank passes to the constructor."""

    def __init__(self, nranks, rank, unique_id):
        self.lib = require_gpu()
        self.comm = ctypes.c_void_p()
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        _lib.check(self.lib.osz_rccl_comm_create(ctypes.byref(self.comm), nranks, rank,
                                                 bytes(unique_id)))

    @staticmethod
    def unique_id():
        lib = require_gpu()
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.osz_rccl_unique_id(buf))
        return buf.raw

    def size(self):
        n = ctypes.c_int()
        _lib.check(self.lib.osz_rccl_comm_size(self.comm, ctypes.byref(n)))
        return n.value

    def close(self):
        if self.comm:
            _lib.check(self.lib.osz_rccl_comm_destroy(self.comm))
            self.comm = ctypes.c_void_p()


class MomentsStream(_Handle):
    """Streaming per-channel mean / std over the chunks of a producer
    (C ABI: osz_moments_*)."""
    _destroy = "osz_moments_destroy"

    def __init__(self, nch):
        super().__init__()
        self.nch = nch
        _lib.check(self.lib.osz_moments_create(ctypes.byref(self.h), nch))

    def push(self, x2d, ignore_nan=True):
        _lib.check(self.lib.osz_moments_push(self.h, ptr(x2d), x2d.stride(0), x2d.shape[1],
                                             int(bool(ignore_nan)), stream_ptr()))

    def finish(self, device="cuda"):
        """(mean, std) as CUDA tensors (nch,)."""
        mean = torch.empty(self.nch, dtype=torch.float64, device=device)
        sd = torch.empty_like(mean)
        _lib.check(self.lib.osz_moments_finish(self.h, ptr(mean), ptr(sd), stream_ptr()))
        return mean, sd


def col_moments(x2d, ignore_nan=True):
    """(mean, std) along the first axis of a (nred, ncols) CUDA tensor."""
    lib = require_gpu()
    mean = torch.empty(x2d.shape[1], dtype=torch.float64, device=x2d.device)
    sd = torch.empty_like(mean)
    _lib.check(lib.osz_col_moments(ptr(x2d), x2d.stride(0), x2d.shape[0], x2d.shape[1],
                                   int(bool(ignore_nan)), ptr(mean), ptr(sd), stream_ptr()))
    return mean, sd


def ew(op, x2d, a, b=None, kind=_lib.BCAST_SCALAR):
    """osz_ew: y = x (op) a [, b] with operands indexed per `kind`.  a / b are
    float64 CUDA tensors (one value, (nch,), (n,) or (nch, n))."""
    lib = require_gpu()
    y = torch.empty((x2d.shape[0], x2d.shape[1]), dtype=torch.float64, device=x2d.device)
    ldab = a.stride(0) if kind == _lib.BCAST_FULL else 0
    if kind == _lib.BCAST_FULL and b is not None and b.stride(0) != ldab:
        b = b.contiguous()
        a = a.contiguous()
        ldab = a.stride(0)
    _lib.check(lib.osz_ew(op, ptr(x2d), x2d.stride(0), x2d.shape[0], x2d.shape[1], ptr(a),
                          ptr(b) if b is not None else None, kind, ldab, ptr(y),
                          max(y.stride(0), 1), stream_ptr()))
    return y


def complex_join(re2d, im2d):
    """re + 1j * im as a complex128 CUDA tensor (osz_complex_join)."""
    lib = require_gpu()
    z = torch.empty(tuple(re2d.shape), dtype=torch.complex128, device=re2d.device)
    _lib.check(lib.osz_complex_join(ptr(re2d), re2d.stride(0), ptr(im2d), im2d.stride(0),
                                    re2d.shape[0], re2d.shape[1], ptr(z), max(z.stride(0), 1),
                                    stream_ptr()))
    return z


def magphase(z2d, want_mag=True, want_phase=True):
    """(|z|, angle(z) in [0, 2 pi)) of a complex128 CUDA tensor (nch, n)
    (osz_magphase); an output that is not wanted is None."""
    lib = require_gpu()
    shape = tuple(z2d.shape)
    mag = torch.empty(shape, dtype=torch.float64, device=z2d.device) if want_mag else None
    ph = torch.empty(shape, dtype=torch.float64, device=z2d.device) if want_phase else None
    _lib.check(lib.osz_magphase(ptr(z2d), z2d.stride(0), shape[0], shape[1],
                                ptr(mag) if want_mag else None, ptr(ph) if want_phase else None,
                                max(shape[1], 1), stream_ptr()))
    return mag, ph


def simpson(p2d, a, m, dx):
    """scipy.integrate.simpson(p[:, a:a+m], dx=dx) per row on the device."""
    lib = require_gpu()
    out = torch.empty(p2d.shape[0], dtype=torch.float64, device=p2d.device)
    _lib.check(lib.osz_simpson(ptr(p2d), p2d.stride(0), p2d.shape[0], int(a), int(m), float(dx),
                               ptr(out), stream_ptr()))
    return out


def take(x2d, idx):
    """osz_take: y[c, j] = x[c, idx[j]] (idx: int64 CUDA tensor)."""
    lib = require_gpu()
    y = torch.empty((x2d.shape[0], idx.numel()), dtype=torch.float64,
                    device=x2d.device)
    _lib.check(lib.osz_take(ptr(x2d), x2d.stride(0), x2d.shape[0], ptr(idx),
                            idx.numel(), ptr(y), max(y.stride(0), 1),
                            stream_ptr()))
    return y


def synth_normal(nch, n, seed=0, ch0=0, n0=0, out=None, device="cuda"):
    """Device-resident synthetic float64 N(0,1) block keyed by
    (seed, channel, sample) -- osz_synth_normal."""
    lib = require_gpu()
    x = out if out is not None else torch.empty((nch, n), dtype=torch.float64,
                                                device=device)
    _lib.check(lib.osz_synth_normal(ptr(x), x.stride(0), nch, n, seed, ch0, n0,
                                    stream_ptr()))
    return x


def checksum(x2d):
    lib = require_gpu()
    bits, fsum = ctypes.c_uint64(), ctypes.c_double()
    _lib.check(lib.osz_checksum(ptr(x2d), x2d.stride(0), x2d.shape[0],
                                x2d.shape[1], ctypes.byref(bits),
                                ctypes.byref(fsum), stream_ptr()))
    return bits.value, fsum.value

"""The iterator runtime in front of the hot path: ``producer()`` and the
Array / Gen / Masked / Reader producers.

Host-side mirror of the reference's ``core/producer.py`` (cited per symbol)
with identical signatures, chunk lengths, shapes and exceptions, written from
scratch.  Two things are new:

* data may be a CUDA (HIP) tensor as well as an ndarray -- an ArrayProducer
  over a device tensor yields device views and every downstream stage then
  stays resident in HBM (no host round trips);
* MaskedProducer gathers device chunks with the ``osz_take`` kernel (K7).

Producers are iterables, not iterators: every ``iter()`` restarts at sample 0
with fresh state, and they stay picklable (no device handle is stored on a
producer; handles are created inside ``__iter__``/generators).
"""

import abc
import inspect
from collections import abc as cabc

import numpy as np

from openseize_amd import _device as dev
from openseize_amd.core import resources
from openseize_amd.core.arraytools import normalize_axis
from openseize_amd.core.queues import FIFOArray


def _is_reader(obj):
    """Duck-typed file reader: the reference dispatches on its EDF ``Reader``
    class (core/producer.py:119-120); file I/O is out of this build's scope,
    so any object offering shape/read/open/close is accepted instead."""
    return all(hasattr(obj, name) for name in ("read", "shape", "open", "close"))


def _is_genfunc(obj):
    """A generating function, bare or wrapped in functools.partial."""
    return inspect.isgeneratorfunction(getattr(obj, "func", obj))


def _is_sequence(obj):
    return isinstance(obj, cabc.Sequence) and not isinstance(obj, (str, bytes))


def _adopt(existing, chunksize, axis, shape, kwargs):
    # an existing producer is re-chunked IN PLACE and handed back
    # (reference core/producer.py:114-117, quirk Q5)
    existing.chunksize = int(chunksize)
    existing.axis = normalize_axis(axis, len(existing.shape))
    return existing


def _from_genfunc(func, chunksize, axis, shape, kwargs):
    if shape is None:
        raise ValueError("A Producer from a generating function requires a shape.")
    return GenProducer(func, chunksize, normalize_axis(axis, len(shape)), shape, **kwargs)


def _from_array(arr, chunksize, axis, shape, kwargs):
    return ArrayProducer(arr, chunksize, normalize_axis(axis, len(arr.shape)), **kwargs)


def _from_sequence(seq, chunksize, axis, shape, kwargs):
    return _from_array(dev.concatenate(list(seq), axis), chunksize, axis, shape, kwargs)


def _from_reader(reader, chunksize, axis, shape, kwargs):
    # readers are (channels, samples): the sample axis is 1 whatever was asked
    return ReaderProducer(reader, chunksize, axis=1, **kwargs)


def producer(data, chunksize, axis, shape=None, mask=None, **kwargs):
    """Builds an iterable yielding arrays of ``chunksize`` samples along
    ``axis`` from an ndarray / device tensor, a sequence of arrays, a reader,
    a generating function or an existing producer.

    Same contract as reference core/producer.py:54-143: an existing producer
    is MUTATED (chunksize, axis) and returned (:114-117); generating functions
    need ``shape`` (ValueError); anything else raises
    ``TypeError("unproducible type")`` (:135-137); ``mask`` wraps the result in
    a MaskedProducer (:139-143).  The type dispatch is a table of (test,
    constructor) pairs tried in order.
    """
    dispatch = ((lambda d: isinstance(d, Producer), _adopt),
                (_is_genfunc, _from_genfunc),
                (dev.is_arraylike, _from_array),
                (_is_sequence, _from_sequence),
                (_is_reader, _from_reader))
    for accepts, build in dispatch:
        if accepts(data):
            made = build(data, chunksize, axis, shape, kwargs)
            break
    else:
        raise TypeError("unproducible type: {}".format(type(data)))
    if mask is not None:
        made = MaskedProducer(made, mask, chunksize, made.axis, **kwargs)
    return made


class Producer(cabc.Iterable):
    """ABC of all producers (reference core/producer.py:146-210): attributes
    ``data, chunksize (int-coerced property), axis, kwargs, shape, ndim`` and
    ``to_array``."""

    def __init__(self, data, chunksize, axis, **kwargs):
        self.data = data
        self._chunksize = int(chunksize)
        self.axis = axis
        self.kwargs = kwargs

    @property
    def chunksize(self):
        return self._chunksize

    @chunksize.setter
    def chunksize(self, value):
        self._chunksize = int(value)

    @property
    @abc.abstractmethod
    def shape(self):
        """Combined shape of everything this producer yields."""

    @property
    def ndim(self):
        return len(self.shape)

    def to_array(self, dtype=float, limit=None):
        """Concatenates all produced arrays along ``axis``; returns None (and
        prints) when the result would not fit in memory
        (core/producer.py:197-210)."""
        if resources.assignable(self.shape, dtype, limit=limit):
            return dev.concatenate(list(self), axis=self.axis)
        return None

    def __repr__(self):
        return (f"{type(self).__name__}(shape={tuple(self.shape)}, "
                f"chunksize={self.chunksize}, axis={self.axis})")


class ReaderProducer(Producer):
    """Producer over a reader object (core/producer.py:213-264): ``start`` /
    ``stop`` kwargs bound the samples (clamped like a slice), the reader is
    closed on construction (so the producer pickles) and reopened on
    iteration; remaining kwargs go to ``reader.read``."""

    def __init__(self, data, chunksize, axis, **kwargs):
        super().__init__(data, chunksize, axis, **kwargs)
        total = data.shape[axis]
        window = slice(self.kwargs.pop("start", 0), self.kwargs.pop("stop", total))
        self.start, self.stop = window.indices(total)[:2]
        data.close()

    @property
    def shape(self):
        dims = list(self.data.shape)
        dims[self.axis] = self.stop - self.start
        return tuple(dims)

    def __iter__(self):
        reader, step = self.data, self.chunksize
        reader.open()
        for first in range(self.start, self.stop, step):
            yield reader.read(first, min(first + step, self.stop), **self.kwargs)


class ArrayProducer(Producer):
    """Views of an in-memory array, ``chunksize`` samples at a time, the last
    one short (core/producer.py:267-295).  Never copies."""

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __iter__(self):
        n = self.data.shape[self.axis]
        index = [slice(None)] * len(self.data.shape)
        for start in range(0, n, self.chunksize):
            index[self.axis] = slice(start, start + self.chunksize)
            yield self.data[tuple(index)]


class GenProducer(Producer):
    """Re-chunks whatever a generating function yields into exactly
    ``chunksize`` samples (core/producer.py:298-376): pieces are collected
    until at least one chunk is available, full chunks are emitted, the
    leftover is kept; a non-empty remainder is yielded at the end."""

    def __init__(self, data, chunksize, axis, shape, **kwargs):
        if shape is None:
            raise ValueError(
                "A Producer from a generating function requires a shape.")
        super().__init__(data, chunksize, axis, **kwargs)
        self._shape = tuple(int(s) for s in shape)

    @property
    def shape(self):
        return self._shape

    def __iter__(self):
        fifo = FIFOArray(self.chunksize, self.axis)
        # (dev.run_generating: a generating function of this library hands CUDA tensors
        # on when this producer's direct consumer is another one of them)
        for piece in dev.run_generating(self, self.data, self.kwargs):
            fifo.put(piece)
            while fifo.full():
                yield fifo.get()
        if not fifo.empty():
            yield fifo.queue          # the short remainder, whole


class MaskedProducer(Producer):
    """Keeps only the samples where a boolean mask is True (behaviour of the
    reference's core/producer.py:379-444, formulated on global sample indices).

    The reference cuts data and mask with the same chunksize, zips the two
    streams and gathers ``flatnonzero`` of every mask piece.  The same stream
    is described once, up front, by the sorted list of kept SAMPLE INDICES
    ``keep = flatnonzero(mask[:limit])`` where ``limit`` is the end of the last
    chunk both streams still have (:427: zip stops with the shorter one).
    Data chunk k then contributes ``keep[lo:hi]``, the slice found by bisection
    for its window [k cs, (k+1) cs); windows without a kept sample are skipped
    (:429-430), and the survivors are re-chunked to ``chunksize``.  On the
    device the gather is the ``osz_take`` kernel (K7).

    ``shape`` reports min(data length, number of True in the WHOLE mask) along
    the axis (:399-408); ``chunksize`` is shared with the mask producer
    (:416-421)."""

    def __init__(self, pro, mask, chunksize, axis, **kwargs):
        super().__init__(pro, chunksize, axis, **kwargs)
        self.mask = producer(mask, chunksize, axis=0)

    def _mask_array(self):
        flags = self.mask.to_array(dtype=bool)
        if dev.is_tensor(flags):
            flags = flags.cpu().numpy()
        return np.asarray(flags, dtype=bool).reshape(-1)

    @property
    def shape(self):
        dims = list(self.data.shape)
        dims[self.axis] = min(dims[self.axis], int(self._mask_array().sum()))
        return tuple(dims)

    @property
    def chunksize(self):
        return self.data.chunksize

    @chunksize.setter
    def chunksize(self, value):
        for stream in (self.data, self.mask):
            stream.chunksize = int(value)

    def __iter__(self):
        cs = self.chunksize
        flags = self._mask_array()
        nsamples = self.data.shape[self.axis]
        # chunks both streams have: ceil(N / cs) and ceil(K / cs)
        paired = min(-(-nsamples // cs), -(-flags.size // cs))
        keep = np.flatnonzero(flags[:paired * cs])
        # keep[bounds[k]:bounds[k + 1]] falls into the window of data chunk k
        bounds = np.searchsorted(keep, np.arange(paired + 1) * cs)
        pending = FIFOArray(cs, self.axis)
        for k, arr in zip(range(paired), dev.relay_pull(self, self.data)):
            local = keep[bounds[k]:bounds[k + 1]] - k * cs
            if local.size == 0:
                continue
            if dev.is_tensor(arr) and arr.is_cuda:
                pending.put(_take_device(arr, local, self.axis))
            else:
                pending.put(np.take(arr, local, axis=self.axis))
            while pending.full():
                yield pending.get()
        if not pending.empty():
            yield pending.get()


def _take_device(arr, keep, axis):
    """K7: gather along the sample axis on the device (osz_take)."""
    import torch
    if keep.size and keep[-1] >= arr.shape[axis]:
        raise IndexError(
            f"index {keep[-1]} is out of bounds for axis {axis} with size "
            f"{arr.shape[axis]}")
    layout = dev.Layout(arr.shape, axis)
    x2d, _ = layout.to2d(arr)
    idx = torch.from_numpy(keep.astype(np.int64)).to(arr.device)
    return layout.from2d(dev.take(x2d, idx), host=False)

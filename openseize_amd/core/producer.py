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
from itertools import zip_longest

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


def producer(data, chunksize, axis, shape=None, mask=None, **kwargs):
    """Builds an iterable yielding arrays of ``chunksize`` samples along
    ``axis`` from an ndarray / device tensor, a sequence of arrays, a reader,
    a generating function or an existing producer.

    Mirrors reference core/producer.py:54-143, including: an existing producer
    is MUTATED (chunksize, axis) and returned (:114-117); generating functions
    need ``shape`` (ValueError); anything else raises
    ``TypeError("unproducible type")`` (:135-137); ``mask`` wraps the result in
    a MaskedProducer (:139-143).
    """
    if isinstance(data, Producer):
        data.chunksize = int(chunksize)
        data.axis = normalize_axis(axis, len(data.shape))
        result = data
    elif inspect.isgeneratorfunction(data) or (
            hasattr(data, "func") and inspect.isgeneratorfunction(data.func)):
        if shape is None:
            raise ValueError(
                "A Producer from a generating function requires a shape.")
        ax = normalize_axis(axis, len(shape))
        result = GenProducer(data, chunksize, ax, shape, **kwargs)
    elif dev.is_arraylike(data):
        ax = normalize_axis(axis, len(data.shape))
        result = ArrayProducer(data, chunksize, ax, **kwargs)
    elif isinstance(data, cabc.Sequence) and not isinstance(data, (str, bytes)):
        x = dev.concatenate(list(data), axis)
        ax = normalize_axis(axis, len(x.shape))
        result = ArrayProducer(x, chunksize, ax, **kwargs)
    elif _is_reader(data):
        result = ReaderProducer(data, chunksize, axis=1, **kwargs)
    else:
        raise TypeError("unproducible type: {}".format(type(data)))

    if mask is None:
        return result
    return MaskedProducer(result, mask, chunksize, result.axis, **kwargs)


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
    ``stop`` kwargs bound the samples, the reader is closed on construction
    (so the producer pickles) and reopened on iteration."""

    def __init__(self, data, chunksize, axis, **kwargs):
        super().__init__(data, chunksize, axis, **kwargs)
        a = self.kwargs.pop("start", 0)
        b = self.kwargs.pop("stop", self.data.shape[axis])
        self.start, self.stop, _ = slice(a, b).indices(data.shape[axis])
        self.data.close()

    @property
    def shape(self):
        s = list(self.data.shape)
        s[self.axis] = self.stop - self.start
        return tuple(s)

    def __iter__(self):
        self.data.open()
        for a in range(self.start, self.stop, self.chunksize):
            yield self.data.read(a, min(a + self.chunksize, self.stop),
                                 **self.kwargs)


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
        collector = FIFOArray(self.chunksize, self.axis)
        for subarr in self.data(**self.kwargs):
            collector.put(subarr)
            while collector.full():
                yield collector.get()
        if collector.qsize() > 0:
            yield collector.queue


class MaskedProducer(Producer):
    """Keeps only the samples where a boolean mask is True
    (core/producer.py:379-444).  Data and mask are cut with the same chunksize
    and zipped, so production ends with the shorter of the two (:427); chunks
    whose mask is all False are skipped (:429-430); survivors are gathered in
    index order (np.take(flatnonzero), :432) and re-chunked.  ``shape`` reports
    min(data length, count of True) along the axis (:399-408)."""

    def __init__(self, pro, mask, chunksize, axis, **kwargs):
        super().__init__(pro, chunksize, axis, **kwargs)
        self.mask = producer(mask, chunksize, axis=0)

    @property
    def shape(self):
        result = list(self.data.shape)
        included = int(np.count_nonzero(self.mask.to_array(dtype=bool)))
        result[self.axis] = min(self.data.shape[self.axis], included)
        return tuple(result)

    @property
    def chunksize(self):
        return self.data.chunksize

    @chunksize.setter
    def chunksize(self, value):
        self.data.chunksize = int(value)
        self.mask.chunksize = int(value)

    def __iter__(self):
        collector = FIFOArray(self.chunksize, self.axis)
        for arr, maskarr in zip(self.data, self.mask):
            maskarr = np.asarray(maskarr.cpu() if dev.is_tensor(maskarr)
                                 else maskarr)
            if not np.any(maskarr):
                continue
            keep = np.flatnonzero(maskarr)
            if dev.is_tensor(arr) and arr.is_cuda:
                filtered = _take_device(arr, keep, self.axis)
            else:
                filtered = np.take(arr, keep, axis=self.axis)
            collector.put(filtered)
            while collector.full():
                yield collector.get()
        if collector.qsize() > 0:
            yield collector.get()


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

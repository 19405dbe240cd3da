"""Producer-level glue: lazily evaluated array operations on producers.

Mirror of the reference's ``core/protools.py`` (cited per function), written
from scratch.  ``pad`` is on the hot path (STFT boundary handling, reference
core/numerical.py:1044,1051); the others are SURVEY section 8f rank 2 -- they
keep a chain of producers (filter -> standardize -> psd ...) lazy and, for
device-resident producers, resident in HBM.  None of them contains DSP
numerics: every function maps produced chunks with an elementwise operation or
folds them into per-channel moments, on whatever memory kind the chunks live in
(ndarray on the host exactly like the reference, CUDA tensor on the device).
"""

from functools import partial
from itertools import zip_longest

import numpy as np

from openseize_amd import _device as dev
from openseize_amd.core import arraytools
from openseize_amd.core.producer import Producer, producer


def pad(pro, amt, axis, value=0):
    amts = (amt, amt) if isinstance(amt, int) else tuple(amt)
    if arraytools.normalize_axis(axis, pro.ndim) == pro.axis:
        genfunc = _production_axis_padder
    else:
        genfunc = _other_axis_padder
    func = partial(genfunc, pro, amts, axis, value)
    new_shape = list(pro.shape)
    new_shape[axis] = pro.shape[axis] + sum(amts)
    return producer(func, pro.chunksize, pro.axis, shape=new_shape)


def _production_axis_padder(pro, amt, axis, value):
    """Only the first and last produced arrays change
    (core/protools.py:229-251)."""
    left_shape, right_shape = list(pro.shape), list(pro.shape)
    left_shape[axis], right_shape[axis] = amt[0], amt[1]
    it = iter(pro)
    first = next(it, None)
    ref = first if first is not None else None
    if ref is None:
        import numpy as np
        ref = np.zeros(0)
    yield dev.zeros_like_kind(ref, left_shape, value)
    if first is not None:
        yield first
        yield from it
    yield dev.zeros_like_kind(ref, right_shape, value)


def _other_axis_padder(pro, amt, axis, value):
    """Every produced array grows along a non-production axis
    (core/protools.py:254-264)."""
    for arr in pro:
        yield arraytools.pad_along_axis(arr, amt, axis, constant_values=value)


# ---------------------------------------------------------------------------
# the rest of the reference's protools
# ---------------------------------------------------------------------------
def squeeze(pro, axis=None):
    """Removes singleton axes, tracking where the production axis moves
    (core/protools.py:36-70)."""
    enumerated = list(enumerate(pro.shape))
    sample_axis = enumerated[pro.axis]
    if axis is None:
        reduced = [(idx, size) for idx, size in enumerated if size > 1]
    else:
        ax = arraytools.normalize_axis(axis, pro.ndim)
        if pro.shape[ax] != 1:
            raise ValueError("cannot select an axis to squeeze out which has "
                             "size not equal to one")
        reduced = [tup for tup in enumerated if tup[0] != ax]
    new_axis = reduced.index(sample_axis)
    new_shape = tuple(size for _, size in reduced)
    return producer(partial(_map_gen, pro, partial(dev.squeeze, axis=axis)),
                    pro.chunksize, new_axis, shape=new_shape)


def _map_gen(pro, func):
    for arr in pro:
        yield func(arr)


def _binary(pro, other, op, verb):
    """pro (op) other for a numeric, an array broadcastable to every produced
    chunk, or a producer of the same shape (core/protools.py:72-180)."""
    if isinstance(other, Producer):
        if tuple(pro.shape) != tuple(other.shape):
            raise ValueError(f"producers can not be {verb} with shapes"
                             f"{pro.shape} {other.shape}")
        if pro.chunksize != other.chunksize:
            other.chunksize = pro.chunksize
        for x, y in zip(pro, other):
            yield op(x, y)
    else:
        for arr in pro:
            yield op(arr, _like(other, arr))


def _like(value, ref):
    """Host constants follow device chunks onto the device."""
    if dev.is_tensor(ref) and isinstance(value, np.ndarray):
        import torch
        return torch.from_numpy(np.ascontiguousarray(value)).to(ref.device)
    return value


def add(pro, other):
    """core/protools.py:72-125."""
    func = partial(_binary, pro, other, lambda x, y: x + y, "added together")
    return producer(func, chunksize=pro.chunksize, axis=pro.axis, shape=pro.shape)


def multiply(pro, other):
    """core/protools.py:127-180."""
    func = partial(_binary, pro, other, lambda x, y: x * y, "multiplied")
    return producer(func, chunksize=pro.chunksize, axis=pro.axis, shape=pro.shape)


def expand_dims(pro, axis=0):
    """Inserts new axes, tracking the production axis (core/protools.py:266-312)."""
    axes = (axis,) if isinstance(axis, int) else tuple(axis)
    new_ndim = len(pro.shape) + len(axes)
    new_shape = np.ones(new_ndim, dtype=int)
    inserts = [arraytools.normalize_axis(ax, new_ndim) for ax in axes]
    complements = sorted(set(range(new_ndim)).difference(inserts))
    new_axis = complements[pro.axis]
    for idx, comp in enumerate(complements):
        new_shape[comp] = pro.shape[idx]
    func = partial(_map_gen, pro, partial(dev.expand_dims, axes=tuple(inserts)))
    return producer(func, pro.chunksize, new_axis, tuple(int(s) for s in new_shape))


def multiply_along_axis(pro, arr, axis):
    """Produced arrays times a 1-D array along one axis, the production axis
    included (core/protools.py:334-384)."""
    arr = np.array(arr)
    if arr.ndim > 1:
        raise ValueError("Dimensions of multiplier arr must be exactly 1.")
    if len(arr) != pro.shape[axis]:
        msg = "operands could not be broadcast together with shapes {} {}"
        raise ValueError(msg.format(pro.shape, arr.shape))
    shape = np.ones(len(pro.shape), dtype=int)
    shape[axis] = len(arr)
    x = arr.reshape(shape)
    if arraytools.normalize_axis(axis, pro.ndim) == pro.axis:
        x = producer(x, chunksize=pro.chunksize, axis=pro.axis)
    func = partial(_multiply_gen, pro, x)
    return producer(func, chunksize=pro.chunksize, axis=pro.axis, shape=pro.shape)


def _multiply_gen(pro, multiplier):
    factors = zip_longest(pro, multiplier, fillvalue=multiplier)
    if isinstance(multiplier, Producer):
        factors = zip(pro, multiplier)
    for arr, mult in factors:
        yield arr * _like(mult, arr)


def slice_along_axis(pro, start=None, stop=None, step=None, axis=-1):
    """Slices a producer; along the production axis this is a mask
    (core/protools.py:428-482)."""
    start, stop, step = slice(start, stop, step).indices(pro.shape[axis])
    if arraytools.normalize_axis(axis, pro.ndim) == pro.axis:
        mask = np.zeros(pro.shape[axis], dtype=bool)
        mask[start:stop:step] = True
        return producer(pro, pro.chunksize, pro.axis, mask=mask)
    new_shape = list(pro.shape)
    new_shape[axis] = (stop - start) // step
    func = partial(_map_gen, pro, partial(arraytools.slice_along_axis, start=start,
                                          stop=stop, step=step, axis=axis))
    return producer(func, pro.chunksize, pro.axis, shape=new_shape)


def mean(pro, axis=-1, ignore_nan=True, keepdims=False):
    """Mean along axis; along the production axis the chunk means are combined
    weighted by chunk length (core/protools.py:500-545)."""
    ax = arraytools.normalize_axis(axis, pro.ndim)
    if pro.axis == ax:
        sums, cnts = 0, 0
        for arr in pro:
            cnts += arr.shape[axis]
            sums = sums + arr.shape[axis] * dev.mean(arr, axis, keepdims, ignore_nan)
        return sums / cnts
    avgs = [dev.mean(x, ax, True, ignore_nan) for x in pro]
    result = dev.concatenate(avgs, pro.axis)
    return result if keepdims else dev.squeeze(result, ax)


def std(pro, axis=-1, ignore_nan=True, keepdims=False):
    """Standard deviation along axis, sqrt(E[x^2] - E[x]^2) over the chunks
    along the production axis (core/protools.py:547-592)."""
    ax = arraytools.normalize_axis(axis, pro.ndim)
    if ax == pro.axis:
        expected_squared = mean(pro, ax, ignore_nan, keepdims=keepdims) ** 2
        sum_squares, cnts = 0, 0
        for arr in pro:
            cnts += arr.shape[axis]
            sum_squares = sum_squares + arr.shape[axis] * dev.mean(
                arr ** 2, axis, keepdims, ignore_nan)
        return dev.sqrt(sum_squares / cnts - expected_squared)
    stds = [dev.std(x, ax, True, ignore_nan) for x in pro]
    result = dev.concatenate(stds, pro.axis)
    return result if keepdims else dev.squeeze(result, ax)


def standardize(pro, axis=-1, ignore_nan=True):
    """(x - mean) / std along axis as a producer (core/protools.py:594-671)."""
    means = mean(pro, axis, ignore_nan, keepdims=True)
    stds = std(pro, axis, ignore_nan, keepdims=True)
    func = partial(_standardize_gen, pro, means, stds, axis)
    return producer(func, pro.chunksize, pro.axis, shape=pro.shape)


def _standardize_gen(pro, means, stds, axis):
    if arraytools.normalize_axis(axis, pro.ndim) == pro.axis:
        for arr in pro:
            yield (arr - means) / stds
    else:
        mean_pro = producer(means, chunksize=pro.chunksize, axis=pro.axis)
        std_pro = producer(stds, chunksize=pro.chunksize, axis=pro.axis)
        for arr, mu, sd in zip(pro, mean_pro, std_pro):
            yield (arr - mu) / sd

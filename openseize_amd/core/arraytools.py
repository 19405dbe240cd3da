"""Axis helpers used inside the hot loops (host mirror of the reference's
``core/arraytools.py:6-134``: normalize_axis, pad/slice/split/multiply along
an axis).  They accept ndarrays and device tensors alike."""

import numpy as np

from openseize_amd import _device as dev


def normalize_axis(axis, ndim):
    """Positive index of ``axis`` for an ``ndim``-dimensional array
    (reference core/arraytools.py:6-18; IndexError when out of range)."""
    if not -ndim <= axis < ndim:
        raise IndexError(
            f"index {axis} is out of bounds for axis 0 with size {ndim}")
    return int(axis % ndim)


def _index(ndim, axis, sl):
    idx = [slice(None)] * ndim
    idx[axis] = sl
    return tuple(idx)


def slice_along_axis(arr, start=None, stop=None, step=None, axis=-1):
    """arr[..., start:stop:step, ...] along ``axis`` (core/arraytools.py:43-58)."""
    return arr[_index(arr.ndim, axis, slice(start, stop, step))]


def split_along_axis(arr, index, axis=-1):
    """(arr[..:index], arr[index:..]) along ``axis`` (core/arraytools.py:61-82)."""
    return (slice_along_axis(arr, 0, index, axis=axis),
            slice_along_axis(arr, index, None, axis=axis))


def pad_along_axis(arr, pad, axis=-1, **kwargs):
    """Constant padding before/after along one axis (core/arraytools.py:21-40)."""
    before, after = (pad, pad) if isinstance(pad, int) else pad
    value = kwargs.get("constant_values", 0)
    if dev.is_tensor(arr):
        shape_b, shape_a = list(arr.shape), list(arr.shape)
        shape_b[axis], shape_a[axis] = before, after
        return dev.concatenate([dev.zeros_like_kind(arr, shape_b, value), arr,
                                dev.zeros_like_kind(arr, shape_a, value)], axis)
    pads = [(0, 0)] * arr.ndim
    pads[axis] = (before, after)
    return np.pad(arr, pads, **kwargs)


def multiply_along_axis(x, y, axis=-1):
    """x times a 1-D array broadcast along ``axis`` (core/arraytools.py:118-134)."""
    shape = [1] * x.ndim
    shape[axis] = len(y)
    return x * y.reshape(shape)


# ---- host-side utilities off the hot path (core/arraytools.py:85-312), kept so that
# code importing them from the reference finds them here too
def expand_along_axis(arr, l, value=0, axis=-1):
    """Every sample along ``axis`` followed by ``l - 1`` copies of ``value``
    (zero stuffing for l-fold upsampling)."""
    moved = np.moveaxis(np.asarray(arr), axis, -1)
    out = np.full(moved.shape + (l,), value, dtype=np.result_type(moved, type(value)))
    out[..., 0] = moved
    return np.moveaxis(out.reshape(moved.shape[:-1] + (-1,)), -1, axis)


def filter1D(size, indices):
    """Boolean mask of length ``size`` that is True at every index or slice in ``indices``."""
    mask = np.zeros(int(size), dtype=bool)
    for where in np.atleast_1d(np.array(indices, dtype=object)):
        mask[where] = True
    return mask


def nearest1D(x, x0):
    """Index of the element of ``x`` closest to ``x0``."""
    return np.argmin(np.abs(x - x0))


def zero_extend(arr, n, axis=-1):
    """``n`` zeros on either side along ``axis``."""
    return pad_along_axis(arr, n, axis=axis)


def edge_extend(arr, n, axis=-1):
    """The first and the last sample repeated ``n`` times on their sides."""
    pads = [(0, 0)] * np.ndim(arr)
    pads[axis] = (n, n)
    return np.pad(arr, pads, mode="edge")


def _mirrored_ends(arr, n, axis):
    """The ``n`` samples next to either end, mirrored (end samples excluded); the
    reference's length check with SciPy's wording."""
    limit = arr.shape[axis] - 1
    if n > limit:
        raise ValueError("The extension length n ({}) is too big. It must not "
                         "exceed x.shape[axis] - 1, which is {}.".format(n, limit))
    head = slice_along_axis(arr, n, 0, -1, axis=axis)
    tail = slice_along_axis(arr, -2, -(n + 2), -1, axis=axis)
    return head, tail


def even_extend(arr, n, axis=-1):
    """Mirror images of the ``n`` samples next to either end."""
    head, tail = _mirrored_ends(arr, n, axis)
    return np.concatenate((head, arr, tail), axis=axis)


def odd_extend(arr, n, axis=-1):
    """Point reflections about the end samples of the ``n`` samples next to them."""
    head, tail = _mirrored_ends(arr, n, axis)
    first = slice_along_axis(arr, 0, 1, axis=axis)
    last = slice_along_axis(arr, -1, None, axis=axis)
    return np.concatenate((2 * first - head, arr, 2 * last - tail), axis=axis)

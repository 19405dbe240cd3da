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

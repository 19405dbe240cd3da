"""Producer-level padding -- the one ``protools`` function on the hot path
(STFT boundary handling, reference core/numerical.py:1044,1051).

Mirror of reference ``core/protools.py:182-264``: constant padding of a
producer before/after along one axis, returned as a new GenProducer with the
padded shape and the source's chunksize.  Pads are created in the same memory
kind (host / device) as the produced chunks.
"""

from functools import partial

from openseize_amd import _device as dev
from openseize_amd.core import arraytools
from openseize_amd.core.producer import producer


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

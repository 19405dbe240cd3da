"""downsample / upsample / resample: user API over the polyphase path.

Same signatures and return rules as reference resampling/resampling.py:72-311
(array in -> array out, producer in -> producer out; ``M == 1`` / ``L == 1`` /
``L == M`` return the input unchanged; L/M reduced by their gcd)."""

from functools import partial

import numpy as np

from openseize_amd import _device as dev
from openseize_amd.core.numerical import polyphase_resample
from openseize_amd.core.producer import producer
from openseize_amd.filtering.fir import Kaiser


def resampled_shape(pro, L, M, axis):
    """ceil(N * L / M) along axis (resampling/resampling.py:72-92)."""
    shape = list(pro.shape)
    shape[axis] = int(np.ceil(pro.shape[axis] * L / M))
    return tuple(shape)


def _run(data, L, M, shape_LM, fs, chunksize, axis, kwargs):
    pro = producer(data, chunksize, axis)
    genfunc = partial(polyphase_resample, pro, L, M, fs, Kaiser, axis, **kwargs)
    shape = resampled_shape(pro, L=shape_LM[0], M=shape_LM[1], axis=axis)
    result = producer(genfunc, chunksize, axis, shape=shape)
    return result.to_array() if dev.is_arraylike(data) else result


def downsample(data, M, fs, chunksize, axis=-1, **kwargs):
    """Polyphase decimation by M (resampling/resampling.py:95-161)."""
    if M == 1:
        return data
    return _run(data, 1, M, (1, M), fs, chunksize, axis, kwargs)


def upsample(data, L, fs, chunksize, axis=-1, **kwargs):
    """Polyphase expansion by L (resampling/resampling.py:164-230)."""
    if L == 1:
        return data
    return _run(data, L, 1, (L, 1), fs, chunksize, axis, kwargs)


def resample(data, L, M, fs, chunksize, axis=-1, **kwargs):
    """Rational L/M resampling, L and M reduced by their gcd
    (resampling/resampling.py:233-311)."""
    g = np.gcd(L, M)
    l, m = L // g, M // g
    if l == m == 1:
        return data
    return _run(data, int(l), int(m), (L, M), fs, chunksize, axis, kwargs)

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
    """The producer's shape with ceil(N * L / M) samples along axis
    (resampling/resampling.py:72-92), in exact integer arithmetic."""
    dims = list(pro.shape)
    dims[axis] = -((-int(dims[axis]) * int(L)) // int(M))
    return tuple(dims)


def _rate_change(data, up, down, fs, chunksize, axis, kwargs):
    """up / down (already in lowest terms, not 1 / 1) through the device
    polyphase generator: array in -> array out, producer in -> producer out."""
    source = producer(data, chunksize, axis)
    stage = partial(polyphase_resample, source, up, down, fs, Kaiser, axis, **kwargs)
    out = producer(stage, chunksize, axis, shape=resampled_shape(source, up, down, axis))
    return out.to_array() if dev.is_arraylike(data) else out


def resample(data, L, M, fs, chunksize, axis=-1, **kwargs):
    """Rational L/M resampling (resampling/resampling.py:233-311): the ratio is
    reduced to lowest terms; an identity ratio returns the input itself."""
    common = int(np.gcd(L, M))
    up, down = int(L) // common, int(M) // common
    if (up, down) == (1, 1):
        return data
    return _rate_change(data, up, down, fs, chunksize, axis, kwargs)


def downsample(data, M, fs, chunksize, axis=-1, **kwargs):
    """Polyphase decimation by M (resampling/resampling.py:95-161)."""
    return resample(data, 1, M, fs, chunksize, axis, **kwargs)


def upsample(data, L, fs, chunksize, axis=-1, **kwargs):
    """Polyphase expansion by L (resampling/resampling.py:164-230)."""
    return resample(data, L, 1, fs, chunksize, axis, **kwargs)

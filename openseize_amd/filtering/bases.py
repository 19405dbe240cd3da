"""Callable filter bases: the boundary between user code and the hot path.

``IIR.__call__`` / ``FIR.__call__`` keep the reference's signatures and
return rules (filtering/bases.py:153-213, :363-421): an ndarray (or device
tensor) in gives an array out, a producer in gives a producer out.  Filter
*design* stays on SciPy exactly as in the reference (O(taps) host math,
SURVEY section 2 row 10); plotting mixins are out of scope.
"""

import abc
from functools import partial

import numpy as np
import scipy.signal as sps

from openseize_amd import _device as dev
from openseize_amd.core import numerical as nm
from openseize_amd.core.producer import producer


def _check_bands(fpass, fstop):
    fpass, fstop = np.atleast_1d(fpass), np.atleast_1d(fstop)
    if len(fpass) != len(fstop):
        msg = "fpass and fstop must have the same shape, got {} and {}"
        raise ValueError(msg.format(fpass.shape, fstop.shape))
    return fpass, fstop


class IIR(abc.ABC):
    """Base of the IIR filters (filtering/bases.py:19-213)."""

    def __init__(self, fpass, fstop, gpass, gstop, fs, fmt):
        self.fs = fs
        self.nyq = fs / 2
        self.fpass, self.fstop = _check_bands(fpass, fstop)
        self.gpass = gpass
        self.gstop = gstop
        self.fmt = "sos" if fmt == "zpk" else fmt
        self.coeffs = self._build()

    @property
    def ftype(self):
        return type(self).__name__.lower()

    @property
    def btype(self):
        fp, fs = self.fpass, self.fstop
        if len(fp) < 2:
            return "lowpass" if fp < fs else "highpass"
        return "bandstop" if fp[0] < fs[0] else "bandpass"

    @property
    @abc.abstractmethod
    def order(self):
        """(order, critical frequency) of this filter."""

    def _build(self):
        N, Wn = self.order
        return sps.iirfilter(N, Wn, rp=self.gpass, rs=self.gstop,
                             btype=self.btype, ftype=self.ftype,
                             output=self.fmt, fs=self.fs)

    def __call__(self, data, chunksize, axis=-1, dephase=True, zi=None,
                 **kwargs):
        """Applies this filter (filtering/bases.py:153-213).  ``dephase`` runs
        the forward-backward ``sosfiltfilt``; otherwise the causal ``sosfilt``
        with optional ``zi`` (ignored when dephasing)."""
        pro = producer(data, chunksize, axis, **kwargs)
        if self.fmt == "sos":
            if dephase:
                genfunc = partial(nm.sosfiltfilt, pro, self.coeffs, axis)
            else:
                genfunc = partial(nm.sosfilt, pro, self.coeffs, axis, zi)
        elif self.fmt == "ba":
            if dephase:
                genfunc = partial(nm.filtfilt, pro, self.coeffs, axis)
            else:
                genfunc = partial(nm.lfilter, pro, self.coeffs, axis, zi)
        else:
            # the reference leaves genfunc unbound here (quirk Q10)
            raise ValueError(f"unknown coefficient format {self.fmt!r}")
        result = producer(genfunc, chunksize, axis, shape=pro.shape)
        if dev.is_arraylike(data):
            result = result.to_array()
        return result


class FIR(abc.ABC):
    """Base of the windowed FIR filters (filtering/bases.py:216-421)."""

    def __init__(self, fpass, fstop, gpass, gstop, fs, **kwargs):
        self.fpass, self.fstop = _check_bands(fpass, fstop)
        self.gpass = gpass
        self.gstop = gstop
        self.fs = fs
        self.nyq = fs / 2
        self.width = np.min(np.abs(self.fstop - self.fpass))
        self.coeffs = self._build(**kwargs)

    @property
    def ftype(self):
        return type(self).__name__.lower()

    @property
    def btype(self):
        fp, fs = self.fpass, self.fstop
        if len(fp) < 2:
            return "lowpass" if fp < fs else "highpass"
        if len(fp) == 2:
            return "bandstop" if fp[0] < fs[0] else "bandpass"
        msg = "{} supports only lowpass, highpass, bandpass & bandstop."
        raise ValueError(msg.format(type(self)))

    @property
    def pass_attenuation(self):
        return -20 * np.log10(1 - 10 ** (-self.gpass / 20))

    @property
    def cutoff(self):
        delta = abs(self.fstop - self.fpass) / 2
        return delta + np.min(np.stack((self.fpass, self.fstop)), axis=0)

    @property
    def window_params(self):
        return tuple()

    @property
    @abc.abstractmethod
    def numtaps(self):
        """Number of taps meeting the attenuation criteria."""

    def _build(self, **kwargs):
        window = (self.ftype, *self.window_params)
        return sps.firwin(self.numtaps, cutoff=self.cutoff, width=None,
                          window=window, pass_zero=self.btype, scale=True,
                          fs=self.fs, **kwargs)

    def __call__(self, data, chunksize, axis=-1, mode="same", **kwargs):
        """Applies this filter by overlap-add convolution
        (filtering/bases.py:363-421)."""
        pro = producer(data, chunksize, axis, **kwargs)
        window = self.coeffs
        genfunc = partial(nm.oaconvolve, pro, window, axis, mode)
        shape = nm.convolved_shape(tuple(data.shape), window.shape, mode, axis)
        result = producer(genfunc, chunksize, axis, shape=shape)
        if dev.is_arraylike(data):
            result = result.to_array()
        return result

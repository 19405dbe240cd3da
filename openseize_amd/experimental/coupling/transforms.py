"""Complex transforms of real data for amplitude / phase extraction (SURVEY
section 8f rank 4): consumers of the FIR path.  Same interface as the
reference's ``experimental/coupling/transforms.py`` (Transform :18-107,
Analytic :110-192).  The Hilbert FIR runs through the overlap-add kernel; the
elementwise parts are the device kernels ``osz_complex_join`` (x + i h(x)) and
``osz_magphase`` (|z|, angle) of ``csrc/glue.hip``.
"""

import abc
from functools import partial

from openseize_amd import _device as dev
from openseize_amd.core.producer import producer
from openseize_amd.filtering.special import Hilbert


class Transform(abc.ABC):
    """Holds ``data`` (a producer of the raw arrays) and ``signal`` (a producer
    of its complex transform); concrete classes supply ``estimate``."""

    def __init__(self, data, fs, chunksize=int(10e6), axis=-1, **kwargs):
        self.fs = fs
        self.chunksize = chunksize
        self.axis = axis
        self.data = producer(data, chunksize, axis)
        self.signal = self.estimate(self.data, **kwargs)

    @abc.abstractmethod
    def estimate(self, data, **kwargs):
        """Returns a producer of complex values."""

    def _polar(self, which):
        """|z| (which = 0) or the phase in [0, 2 pi) (which = 1) of every complex
        chunk of ``signal`` with the device kernel ``osz_magphase``."""
        for arr in self.signal:
            layout = dev.Layout(arr.shape, self.axis)
            z2d, host = _complex_rows(arr, layout)
            part = dev.magphase(z2d, want_mag=which == 0, want_phase=which == 1)[which]
            yield layout.from2d(part, host)

    def _envelope(self):
        yield from self._polar(0)

    @property
    def amplitudes(self):
        return producer(self._envelope, self.chunksize, self.axis, shape=self.signal.shape)

    def _phase(self):
        yield from self._polar(1)

    @property
    def phases(self):
        """Phases in [0, 2 pi)."""
        return producer(self._phase, self.chunksize, self.axis, shape=self.signal.shape)


class Analytic(Transform):
    """The analytic signal x + i H(x), H being the type-III Kaiser Hilbert FIR
    of ``filtering.special.Hilbert`` with transition ``width`` around 0 and
    Nyquist.  Real and imaginary parts are produced chunk by chunk from the
    same source and joined into one complex chunk on the memory kind they
    live in."""

    def estimate(self, data, *, width, gpass=0.01, gstop=60, **kwargs):
        source = producer(data, self.chunksize, self.axis)
        quadrature = Hilbert(width, fs=self.fs, gpass=gpass, gstop=gstop)(
            source, self.chunksize, self.axis)
        return producer(partial(_join_complex, source, quadrature), self.chunksize,
                        self.axis, shape=source.shape)


def _complex_rows(arr, layout):
    """Complex chunk -> ((rows, n) complex128 CUDA tensor, came_from_host)."""
    import numpy as np
    import torch
    host = not dev.is_tensor(arr)
    t = torch.from_numpy(np.ascontiguousarray(arr)).cuda() if host else arr
    host = host or not arr.is_cuda
    t = t.cuda().to(torch.complex128).movedim(layout.axis, -1)
    return t.reshape(layout.nch, t.shape[-1]).contiguous(), host


def _join_complex(real_pro, imag_pro):
    """Generator of re + 1j * im over two equally chunked producers, joined on
    the device (``osz_complex_join``)."""
    for re, im in zip(real_pro, imag_pro):
        layout = dev.Layout(re.shape, real_pro.axis)
        re2d, host = layout.to2d(re)
        im2d, _ = layout.to2d(im)
        yield layout.from2d(dev.complex_join(re2d, im2d), host)

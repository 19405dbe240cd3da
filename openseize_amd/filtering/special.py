"""Special-purpose FIR designs.

``Hilbert``: type III (odd tap count, antisymmetric) Kaiser-windowed Hilbert
transformer with the interface of the reference's ``filtering/special.py:16-133``.
The ideal Hilbert impulse response h[n] = (1 - cos(pi n)) / (pi n), h[0] = 0
(Porat, eqn. 9.40) is truncated to the tap count Kaiser's formula gives for the
transition width and tapered with the matching Kaiser window.  Applying it is
the ordinary overlap-add FIR path on the device (``FIR.__call__``), so
``x + 1j * Hilbert(...)(x, chunksize, mode="same")`` is the analytic signal.
"""

import numpy as np
import scipy.signal as sps

from openseize_amd.filtering.fir import Kaiser


class Hilbert(Kaiser):
    def __init__(self, width, fs, gpass=0.01, gstop=60):
        nyq = fs / 2
        super().__init__((0 + width, nyq - width), fstop=(0, nyq), fs=fs,
                         gpass=gpass, gstop=gstop)

    @property
    def numtaps(self):
        ripple = max(self.pass_attenuation, self.gstop)
        ntaps, _ = sps.kaiserord(ripple, self.width / self.nyq)
        return ntaps + 1 if ntaps % 2 == 0 else ntaps      # type III: odd length

    def _build(self, **kwargs):
        ntaps = self.numtaps
        n = np.arange(ntaps) - (ntaps - 1) / 2               # ..., -1, 0, 1, ...
        centre = (ntaps - 1) // 2
        n[centre] = 1.0                                      # avoid 0/0; overwritten below
        ideal = (1 - np.cos(np.pi * n)) / (np.pi * n)
        ideal[centre] = 0.0
        window = sps.get_window(("kaiser", *self.window_params), ntaps)
        return ideal * window

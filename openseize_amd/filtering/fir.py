"""Windowed FIR designs (reference filtering/fir.py:52-480).  Design math is
SciPy's, as in the reference; applying the filter goes through the device
overlap-add path (``FIR.__call__``)."""

import numpy as np
import scipy.signal as sps

from openseize_amd.filtering.bases import FIR


class Kaiser(FIR):
    """Kaiser-window FIR; tap count and beta from the stricter of the pass
    and stop band attenuations (filtering/fir.py:52-137)."""

    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0):
        super().__init__(fpass, fstop, gpass, gstop, fs)

    @property
    def numtaps(self):
        ripple = max(self.pass_attenuation, self.gstop)
        ntaps, _ = sps.kaiserord(ripple, self.width / self.nyq)
        return ntaps + 1 if ntaps % 2 == 0 else ntaps

    @property
    def window_params(self):
        ripple = max(self.pass_attenuation, self.gstop)
        return [sps.kaiser_beta(ripple)]


class _FixedWindow(FIR):
    """Fixed-shape windows (filtering/fir.py:140-480): taps = factor /
    normalised transition width, made odd; gstop is the window's peak
    approximation error and gpass follows from it."""
    _peak_err, _factor = -21, 4

    def __init__(self, fpass, fstop, fs):
        gpass = -20 * np.log10(1 - 10 ** (self._peak_err / 20))
        super().__init__(fpass, fstop, gpass=gpass, gstop=self._peak_err, fs=fs)

    @property
    def numtaps(self):
        ntaps = int(self._factor / (self.width / self.nyq))
        return ntaps + 1 if ntaps % 2 == 0 else ntaps


class Rectangular(_FixedWindow):
    _peak_err, _factor = -21, 4


class Bartlett(_FixedWindow):
    _peak_err, _factor = -25, 8


class Hann(_FixedWindow):
    _peak_err, _factor = -44, 8


class Hamming(_FixedWindow):
    _peak_err, _factor = -53, 8


class Blackman(_FixedWindow):
    _peak_err, _factor = -74, 12

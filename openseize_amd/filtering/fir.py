"""Windowed FIR designs (reference filtering/fir.py:52-480).  Design math is
SciPy's, as in the reference; applying the filter goes through the device
overlap-add path (``FIR.__call__``)."""

import numpy as np
import scipy.signal as sps

from openseize_amd.filtering.bases import FIR


def _make_odd(count):
    """Type I filters (odd length) only: an even count gains one tap."""
    return count | 1


class Kaiser(FIR):
    """Kaiser-window FIR; tap count and beta from the stricter of the pass
    and stop band attenuations (filtering/fir.py:52-137)."""

    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0):
        FIR.__init__(self, fpass, fstop, gpass, gstop, fs)

    @property
    def _design_attenuation(self):
        """The stricter of the two band specifications, in dB."""
        return max(self.pass_attenuation, self.gstop)

    @property
    def numtaps(self):
        count = sps.kaiserord(self._design_attenuation, self.width / self.nyq)[0]
        return _make_odd(count)

    @property
    def window_params(self):
        return [sps.kaiser_beta(self._design_attenuation)]


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
        return _make_odd(int(self._factor / (self.width / self.nyq)))


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

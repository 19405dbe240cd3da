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


class Remez(FIR):
    """Parks-McClellan equiripple FIR (filtering/fir.py:483-662).  ``bands`` lists the
    edges of consecutive bands from 0 to Nyquist, ``desired`` one gain (0 or 1) per band.
    The pass / stop edges the base class keeps are the interior edges of the wanted /
    unwanted bands; every band is weighted by the inverse of the deviation its
    specification allows (1 - 10^(-gpass/20) where the gain is 1, 10^(-gstop/20) where it
    is 0), and the tap count is Bellanger's estimate -- all overridable through the
    keywords of ``scipy.signal.remez`` (``numtaps``, ``weight``, ``maxiter``,
    ``grid_density``)."""

    def __init__(self, bands, desired, fs, gpass=1, gstop=40, **kwargs):
        self.bands = np.array(bands).reshape(-1, 2)
        self.desired = np.array(desired, dtype=bool)
        self.delta_pass = 1 - 10 ** (-gpass / 20)
        self.delta_stop = 10 ** (-gstop / 20)
        self.delta = np.where(self.desired, self.delta_pass, self.delta_stop)

        def interior_edges(rows):
            edges = rows.flatten()
            return edges[(edges > 0) & (edges < fs / 2)]

        FIR.__init__(self, interior_edges(self.bands[self.desired]),
                     interior_edges(self.bands[~self.desired]), gpass, gstop, fs, **kwargs)

    @property
    def btype(self):
        if self.fpass.size > 2:
            return "multiband"
        return super().btype

    @property
    def numtaps(self):
        estimate = -2 / 3 * np.log10(10 * self.delta_pass * self.delta_stop) * self.fs / self.width
        return _make_odd(int(np.ceil(estimate)))

    def _build(self, **kwargs):
        options = {"numtaps": self.numtaps, "weight": 1 / self.delta, "maxiter": 25,
                   "grid_density": 16}
        options.update(kwargs)
        return sps.remez(options.pop("numtaps"), self.bands.flatten(), self.desired,
                         fs=self.fs, **options)

"""IIR designs (reference filtering/iir.py:44-351): minimum order from
SciPy's ``*ord`` helpers, coefficients from ``scipy.signal.iirfilter``
(``IIR._build``)."""

import numpy as np
import scipy.signal as sps

from openseize_amd.filtering.bases import IIR


class Butter(IIR):
    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0, fmt="sos"):
        super().__init__(fpass, fstop, gpass, gstop, fs, fmt)

    @property
    def order(self):
        return sps.buttord(self.fpass, self.fstop, self.gpass, self.gstop,
                           fs=self.fs)


class Cheby1(IIR):
    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0, fmt="sos"):
        super().__init__(fpass, fstop, gpass, gstop, fs, fmt)

    @property
    def order(self):
        return sps.cheb1ord(self.fpass, self.fstop, self.gpass, self.gstop,
                            fs=self.fs)


class Cheby2(IIR):
    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0, fmt="sos"):
        super().__init__(fpass, fstop, gpass, gstop, fs, fmt)

    @property
    def order(self):
        return sps.cheb2ord(self.fpass, self.fstop, self.gpass, self.gstop,
                            fs=self.fs)


class Ellip(IIR):
    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0, fmt="sos"):
        super().__init__(fpass, fstop, gpass, gstop, fs, fmt)

    @property
    def order(self):
        return sps.ellipord(self.fpass, self.fstop, self.gpass, self.gstop,
                            fs=self.fs)


class Notch(IIR):
    """Second-order notch in transfer-function ('ba') format
    (filtering/iir.py:354-404): -3 dB at fstop +- width/2."""

    def __init__(self, fstop, width, fs):
        fpass = np.array([fstop - width / 2, fstop + width / 2])
        fstops = np.array([fstop, fstop])
        self.width = width
        super().__init__(fpass, fstops, gpass=3, gstop=None, fs=fs, fmt="ba")

    @property
    def order(self):
        return len(self.coeffs[0]) - 1, self.fstop[0] - self.width / 2

    def _build(self):
        center = self.fstop[0]
        return sps.iirnotch(center, center / self.width, fs=self.fs)

"""IIR design families (public names and constructor signatures of the
reference's filtering/iir.py:44-404).

The four classical families differ only in the SciPy rule that finds the lowest
order meeting the attenuation specification, so they are generated from one
table; ``IIR`` (bases.py) does the rest.  ``Notch`` is the odd one out: a
second-order section designed directly as a transfer function.
"""

import numpy as np
import scipy.signal as sps

from openseize_amd.filtering.bases import IIR

# class name -> (SciPy minimum-order rule, one-line description)
_FAMILIES = {
    "Butter": (sps.buttord, "Butterworth: maximally flat pass band"),
    "Cheby1": (sps.cheb1ord, "Chebyshev I: ripple gpass in the pass band"),
    "Cheby2": (sps.cheb2ord, "Chebyshev II: ripple gstop in the stop band"),
    "Ellip": (sps.ellipord, "Elliptic: ripple in both bands, steepest roll-off"),
}


def _family(name, rule, summary):
    def __init__(self, fpass, fstop, fs, gpass=1.0, gstop=40.0, fmt="sos"):
        IIR.__init__(self, fpass, fstop, gpass, gstop, fs, fmt)

    doc = (f"{summary}.\n\n    fpass, fstop: pass and stop band edge(s) in the units of fs; "
           "gpass: largest pass-band loss (dB); gstop: smallest stop-band attenuation (dB); "
           "fmt: 'sos' (recommended) or 'ba'.")
    return type(name, (IIR,), {"__init__": __init__, "__doc__": doc, "__module__": __name__,
                               "_order_rule": staticmethod(rule)})


for _name, (_rule, _summary) in _FAMILIES.items():
    globals()[_name] = _family(_name, _rule, _summary)
del _name, _rule, _summary


class Notch(IIR):
    """Second-order notch at ``fstop`` whose -3 dB points lie ``width`` apart,
    in transfer-function ('ba') format (filtering/iir.py:354-404)."""

    def __init__(self, fstop, width, fs):
        self.width = width
        half = width / 2
        IIR.__init__(self, fpass=np.array([fstop - half, fstop + half]),
                     fstop=np.array([fstop, fstop]), gpass=3, gstop=None, fs=fs, fmt="ba")

    def _build(self):
        f0 = self.fstop[0]
        return sps.iirnotch(f0, Q=f0 / self.width, fs=self.fs)

    @property
    def order(self):
        numerator = self.coeffs[0]
        return numerator.size - 1, self.fpass[0]

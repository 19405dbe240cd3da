"""Band metrics of a power spectrum estimate: pure
consumers of the PSD path.  Same signatures and results as the reference's
``spectra/metrics.py`` (power :25-87, power_norm :90-141, confidence_interval
:144-183).  ``power`` integrates the (channels x nfreq) estimate with the
device Simpson kernel (``osz_simpson``) and ``power_norm`` divides by it with
``osz_ew``; a CUDA estimate -- what ``psd`` returns for device input -- never
leaves HBM, an ndarray estimate is uploaded and the result brought back.
``confidence_interval`` scales by two chi-squared quantiles (host scalars).
"""

import numpy as np
from scipy.stats import chi2

from openseize_amd import _device as dev
from openseize_amd import _lib
from openseize_amd.core.arraytools import normalize_axis


def nearest1D(x, x0):
    """Index of the entry of 1-D ``x`` closest to ``x0``
    (reference core/arraytools.py:165-180)."""
    return int(np.argmin(np.abs(np.asarray(x) - x0)))


def _band(freqs, start, stop):
    """(first bin, number of bins, bin width) of [start, stop], ends inclusive
    at the nearest bins (spectra/metrics.py:73-80)."""
    freqs = np.asarray(freqs)
    lo = nearest1D(freqs, freqs[0] if start is None else start)
    hi = nearest1D(freqs, freqs[-1] if stop is None else stop)
    return lo, hi + 1 - lo, float(freqs[1] - freqs[0])


def _band_power(estimate, freqs, start, stop, axis):
    """-> (layout, (rows, nfreq) CUDA estimate, (rows,) CUDA band power, host?)"""
    layout = dev.Layout(estimate.shape, normalize_axis(axis, len(estimate.shape)))
    p2d, host = layout.to2d(estimate)
    first, count, df = _band(freqs, start, stop)
    if count < 1:
        raise ValueError("empty frequency band: stop lies below start")
    return layout, p2d, dev.simpson(p2d, first, count, df), host


def power(psd, freqs, start=None, stop=None, axis=-1):
    """Band power between ``start`` and ``stop`` (nearest bins, inclusive) by
    Simpson's rule with spacing ``freqs[1] - freqs[0]``, one value per signal
    (what ``scipy.integrate.simpson`` returns in the reference, :87)."""
    layout, _, band, host = _band_power(psd, freqs, start, stop, axis)
    out = band.reshape(layout.other)
    return out.cpu().numpy() if host else out


def power_norm(estimate, freqs, start=None, stop=None, axis=-1):
    """The estimate divided by its band power between ``start`` and ``stop``
    (spectra/metrics.py:90-141)."""
    layout, p2d, band, host = _band_power(estimate, freqs, start, stop, axis)
    return layout.from2d(dev.ew(_lib.EW_DIV, p2d, band, None, _lib.BCAST_ROW), host)


def confidence_interval(psd, n_estimates, alpha=0.05):
    """(lower, upper) 1-alpha bounds per signal from the chi-squared
    distribution with ``n_estimates`` degrees of freedom."""
    dof = n_estimates
    lo_q, hi_q = chi2.ppf([alpha / 2, 1 - alpha / 2], dof)
    psd = np.asarray(psd.cpu() if dev.is_tensor(psd) else psd)
    return list(zip(psd * dof / lo_q, psd * dof / hi_q))

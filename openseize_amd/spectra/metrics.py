"""Band metrics of a power spectrum estimate: pure
consumers of the PSD path.  Same signatures and results as the reference's
``spectra/metrics.py`` (power :25-87, power_norm :90-141, confidence_interval
:144-183).  They act on the (channels x nfreq) estimate that ``psd`` already
returns on the host -- a few MB at most -- so they are host arithmetic; no
sample-rate data is touched here.
"""

import numpy as np
from scipy.stats import chi2

from openseize_amd.core.arraytools import slice_along_axis


def nearest1D(x, x0):
    """Index of the entry of 1-D ``x`` closest to ``x0``
    (reference core/arraytools.py:165-180)."""
    return int(np.argmin(np.abs(np.asarray(x) - x0)))


def _simpson(y, dx, axis):
    """Composite Simpson's rule on evenly spaced samples; for an even number
    of samples the last interval uses the three-point correction SciPy >= 1.11
    applies (Cartwright), so results match ``scipy.integrate.simpson``."""
    y = np.moveaxis(np.asarray(y, dtype=float), axis, -1)
    n = y.shape[-1]
    if n == 1:
        return np.zeros(y.shape[:-1])
    if n == 2:
        return 0.5 * dx * (y[..., 0] + y[..., 1])
    m = n if n % 2 else n - 1            # odd count: plain composite rule
    res = dx / 3.0 * (y[..., 0] + y[..., m - 1]
                      + 4.0 * y[..., 1:m - 1:2].sum(-1)
                      + 2.0 * y[..., 2:m - 2:2].sum(-1))
    if n % 2 == 0:
        # last interval [n-2, n-1] from the parabola through the last three points
        res = res + dx * (5.0 * y[..., -1] + 8.0 * y[..., -2] - y[..., -3]) / 12.0
    return res


def power(psd, freqs, start=None, stop=None, axis=-1):
    """Band power between ``start`` and ``stop`` (nearest bins, inclusive) by
    Simpson's rule with spacing ``freqs[1] - freqs[0]``."""
    freqs = np.asarray(freqs)
    start = freqs[0] if start is None else start
    stop = freqs[-1] if stop is None else stop
    a, b = nearest1D(freqs, start), nearest1D(freqs, stop)
    arr = slice_along_axis(np.asarray(psd), start=a, stop=b + 1, axis=axis)
    return _simpson(arr, freqs[1] - freqs[0], axis)


def power_norm(estimate, freqs, start=None, stop=None, axis=-1):
    """The estimate divided by its band power between ``start`` and ``stop``."""
    norm = np.expand_dims(power(estimate, freqs, start, stop, axis), axis=axis)
    return np.asarray(estimate) / norm


def confidence_interval(psd, n_estimates, alpha=0.05):
    """(lower, upper) 1-alpha bounds per signal from the chi-squared
    distribution with ``n_estimates`` degrees of freedom."""
    dof = n_estimates
    lo_q, hi_q = chi2.ppf([alpha / 2, 1 - alpha / 2], dof)
    psd = np.asarray(psd)
    return list(zip(psd * dof / lo_q, psd * dof / hi_q))

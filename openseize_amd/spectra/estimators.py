"""psd / stft: user API over the windowed-DFT path.

Same signatures and return values as reference spectra/estimators.py:59-156
(``psd`` -> (cnt, freqs, mean PSD)) and :160-284 (``stft`` -> (freqs, time,
X)).  ``psd`` keeps the segment average on the device: the ``osz_spec`` handle
accumulates the periodogram sum (K6) and the mean is taken once at the end --
mathematically the running mean of estimators.py:149-152.
"""

import numpy as np

from openseize_amd import _device as dev
from openseize_amd import _lib
from openseize_amd.core import numerical as nm
from openseize_amd.core.producer import producer
from openseize_amd.core.resources import assignable


def psd(data, fs, axis=-1, resolution=0.5, window="hann", overlap=0.5,
        detrend="constant", scaling="density"):
    """Welch power spectrum (density) estimate.  chunksize is forced to
    ``int(fs)`` (estimators.py:141) and nfft = int(fs / resolution) (:144)."""
    pro = producer(data, chunksize=int(fs), axis=axis)
    pro = nm._coarse(pro, int(np.prod(pro.shape)) // max(pro.shape[axis], 1))
    nfft = int(fs / resolution)
    freqs = np.fft.rfftfreq(nfft, 1 / fs)
    noverlap = int(nfft * overlap)
    stride = nfft - noverlap
    coeffs, scale = nm._window_and_scale(window, nfft, fs, scaling)
    axis_n = nm.normalize_axis(axis, len(pro.shape))
    layout = dev.Layout(pro.shape, axis_n)
    spec = dev.SpecStream(nfft, nfft, stride, coeffs, scale, detrend,
                          _lib.SPEC_PSD_MEAN, layout.nch)
    host, pipe = True, None
    try:
        for arr in dev.pull_resident(nm._batched(pro, axis_n, layout.nch), pro):
            if dev.is_tensor(arr):
                x2d, host = layout.to2d(arr)
            else:
                # host-fed: the next piece is staged (pinned ring, H2D stream)
                # while this one's segments are transformed
                pipe = pipe or dev.HostPipe(layout)
                x2d, host = pipe.feed(arr), True
            if x2d.shape[1]:
                spec.push(x2d)
        # a chain of this library's producers over host data hands CUDA tensors to
        # this loop (dev.pull_resident); the estimate still goes back as an ndarray
        host = host or dev.origin_is_host(pro)
        # device input: the average is taken on the device and stays there
        cnt, mean = spec.mean() if host else spec.mean_device()
    finally:
        spec.close()
    if nm._linear_trend_refuses(mean[None], detrend) is not None:
        # (a least-squares trend refuses non-finite data in the reference: core/numerical.py:691)
        raise ValueError(nm._REFUSED)
    if cnt == 0:
        # the reference's loop variable is unbound here (estimators.py:156)
        raise UnboundLocalError(
            "no complete segment: data is shorter than nfft = int(fs/resolution)")
    result = mean.reshape(layout.other + (mean.shape[-1],))
    if host:
        return cnt, freqs, np.moveaxis(result, -1, axis_n)
    return cnt, freqs, result.movedim(-1, axis_n).contiguous()


def stft(data, fs, axis=-1, resolution=0.5, window="hann", overlap=0.5,
         detrend="constant", scaling="density", boundary=True, padded=True,
         asarray=True):
    """Short-time Fourier transform (estimators.py:160-284).  With
    ``asarray`` the per-segment estimates are stacked on a new last axis when
    they fit in memory (:279-282), else a producer is returned."""
    pro = producer(data, chunksize=int(fs), axis=axis)
    pro = nm._coarse(pro, int(np.prod(pro.shape)) // max(pro.shape[axis], 1))
    nfft = int(fs / resolution)
    freqs, time, result = nm.stft(pro, fs, nfft, window, overlap, axis,
                                  detrend, scaling, boundary, padded)
    if asarray:
        if assignable(result.shape):
            result = dev.stack(list(result), axis=-1)
    return freqs, time, result

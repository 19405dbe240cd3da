"""openseize_amd -- MI355X-native implementation of Openseize's chunked DSP
hot path (overlap-add FIR, cascaded SOS IIR, polyphase resampling, Welch/STFT)
behind Openseize's own producer / filter / resample / spectra API.

    from openseize_amd import producer
    from openseize_amd.filtering.iir import Butter
    from openseize_amd.filtering.fir import Kaiser
    from openseize_amd.resampling.resampling import downsample
    from openseize_amd.spectra.estimators import psd, stft

All numerics run in hand-written HIP kernels (``libosz_hip.so``, C ABI in
``include/osz_hip.h``).  There is no CPU fallback.
"""

from openseize_amd.core.producer import producer  # noqa: F401

__version__ = "0.1.0"

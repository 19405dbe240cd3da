#!/usr/bin/env python3
"""What the zero-phase kernel's burst cut costs on inputs riding an offset (with a drift of a
tenth of it): error against SciPy's whole-stream passes, away from the stream's ends, at the cuts
1e-12 (rounds 3-4), 1e-15 (the default since round 5) and 1e-18 -- the cut's share of the error
falls with the tolerance, float64's own rounding on the offset (about 90 eps max|x| through a
4096-point transform there and back) does not.

    python benchmarks/dc_probe.py > profiles/rNN_dc_probe.txt"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import scipy.signal as sps
import torch

from openseize_amd import _device as dev
from openseize_amd import _lib

_lib.load()
import test_gpu_zp as T

h = sps.firwin(1024, 0.2)
C, lens = 4, [150000, 150000, 150000, 90001]
total = sum(lens)
noise = dev.synth_normal(C, total, seed=5)
ramp = torch.linspace(0, 1, total, dtype=torch.float64, device="cuda")[None]
for name, sos in (("bp6", T.BP), ("hp", sps.ellip(4, 0.5, 50, 0.25, "highpass", output="sos")),
                  ("lp5", sps.butter(5, 0.3, output="sos"))):
    for dc in (0.0, 1e2, 1e4, 1e6, 1e7):
        x = noise + dc + ramp * (0.1 * dc)
        ref = T.whole_stream_reference(x.cpu().numpy(), h, sos)
        lo = 20000      # away from the stream's start (the reference's own start transient)
        row = []
        for tol in (1e-12, 0.0, 1e-18):
            fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
            try:
                dev.chain_zp_tolerance(fir, iir, tol)
                lag = dev.chain_zp_lag(fir, iir)
                if lag < 0:
                    row.append((tol, None, None))
                    continue
                iir.set_state_scaled((x[:, :1] * float(h[0])).contiguous(), 0)
                dev.chain_zp_open(fir, iir, 0)
                outs, o = [], 0
                for n in lens:
                    outs.append(dev.chain_zp_step(fir, iir, x[:, o:o + n]))
                    o += n
                g = torch.cat(outs, 1).cpu().numpy()
            finally:
                fir.close()
                iir.close()
            hi = total - lag - 6000
            row.append((tol, lag, float(np.max(np.abs(g[:, lag + lo:lag + hi] - ref[:, lo:hi])))))
        scale = float(np.max(np.abs(ref[:, lo:total - 9000])))
        print(name, f"offset {dc:g}", f"output scale {scale:.3g}",
              "  ".join(f"cut {t or 1e-15:g}: lag {l} abs err {e:.2e} ({e / max(dc, 1.0) / 2.2e-16:.0f} eps max|x|)" if l is not None
                        else f"cut {t or 1e-15:g}: refused" for t, l, e in row), flush=True)

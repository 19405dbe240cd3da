"""Development probe of the zero-phase chain kernel (osz_chain_zp_*): parity of a chunked
stream against SciPy on a few channels, then the steady-state time per 256 x 2^20 chunk."""
import sys
import time

import numpy as np
import scipy.signal as sps
import torch

from openseize_amd import _device as dev
from openseize_amd import _lib


def parity(taps_n, C, lens, sos, seed=3):
    h = sps.firwin(taps_n, 0.2)
    total = sum(lens)
    x = dev.synth_normal(C, total, seed=seed)
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    lag = dev.chain_zp_lag(fir, iir)
    if lag < 0:
        print(f"taps {taps_n}: not eligible")
        return
    first = (x[:, :1] * float(h[0])).contiguous()
    iir.set_state_scaled(first, 0)
    dev.chain_zp_open(fir, iir, 0)
    outs, o = [], 0
    for n in lens:
        outs.append(dev.chain_zp_step(fir, iir, x[:, o:o + n]))
        o += n
    torch.cuda.synchronize()
    got = torch.cat(outs, 1)
    pick = [0, C // 2, C - 1]
    xh = x[pick].cpu().numpy()
    u = sps.oaconvolve(xh, h[None], axes=-1)[:, :total]
    zi = sps.sosfilt_zi(sos)[:, None, :] * u[:, :1][None]
    f, _ = sps.sosfilt(sos, u, axis=-1, zi=zi)
    ref = sps.sosfilt(sos, np.concatenate([f, np.zeros((3, 8192))], 1)[:, ::-1], axis=-1)[:, ::-1][:, :total]
    g = got[pick].cpu().numpy()
    hi = total - lag - 6000
    err = np.max(np.abs(g[:, lag:lag + hi] - ref[:, :hi])) / np.max(np.abs(ref))
    print(f"taps {taps_n} C {C} lens {lens[:3]}..: lag {lag}, max rel err {err:.2e}, finite {np.isfinite(g).all()}")
    fir.close(); iir.close()
    return err


def timing(C=256, cs=1 << 20, steps=20, warm=5):
    h = sps.firwin(1024, 0.2)
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    ring = [dev.synth_normal(C, cs, seed=0, n0=k * cs) for k in range(3)]
    fir, iir = dev.FirStream(h, C), dev.SosStream(sos, C)
    iir.set_state_scaled(ring[0], 0)
    dev.chain_zp_open(fir, iir, 0)
    y = torch.zeros((C, cs), dtype=torch.float64, device="cuda")
    for k in range(warm):
        dev.chain_zp_step(fir, iir, ring[k % 3], out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        dev.chain_zp_step(fir, iir, ring[(warm + k) % 3], out=y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"zp step: {dt * 1e3:.3f} ms per {C} x 2^20 chunk = {C * cs / dt / 1e9:.1f} Gsamples/s")


if __name__ == "__main__":
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
    S = 2816
    parity(1024, 5, [2 * S * 6 + 1024, 2 * S * 4, 2 * S * 3 + S + 17, 2 * S * 2 + 5, 2 * S * 4 - 1, 2 * S * 3 + 300], sos)
    parity(1024, 256, [2 * S * 30 + 777] * 3, sos)
    parity(1024, 64, [1 << 18] * 3, sos)
    parity(300, 7, [150000, 150000, 90001], sos)
    parity(513, 3, [100000] * 3, sps.butter(5, 0.3, output="sos"))
    if len(sys.argv) > 1:
        timing()

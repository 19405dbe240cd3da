"""oaconvolve through the public generator with and without the reference's NaN reach
(OSZ_FIR_REACH / OSZ_CHAIN_REACH = 1 / 0), alternating on one box: resident 256 ch x 24 chunks of 2^20 (1024 taps),
resident 16 ch x 1e6 in chunks of 30 000 (256 taps, cfg-1's geometry), host-fed cfg-1."""
import json
import os
import sys
import time

import numpy as np
import scipy.signal as sps

sys.path.insert(0, ".")


def main():
    import torch
    from openseize_amd import _device as dev
    from openseize_amd import producer
    from openseize_amd.core import numerical as nm
    big = dev.synth_normal(256, 8 << 20, seed=1)
    h1024 = sps.firwin(1024, 0.2)
    small = dev.synth_normal(16, 1_000_000, seed=2)
    small_h = small.cpu().numpy()
    h256 = sps.firwin(256, 0.2)

    def drain(gen):
        n = 0
        for p in gen:
            n += p.shape[-1]
        torch.cuda.synchronize()
        return n

    cases = {
        "resident 256 ch x 8 x 2^20, 1024 taps": lambda: drain(nm.oaconvolve(producer(big, 1 << 20, -1), h1024, -1, "same")),
        "resident 16 ch x 1e6, chunks of 30 000, 256 taps": lambda: drain(nm.oaconvolve(producer(small, 30000, -1), h256, -1, "same")),
        "host-fed 16 ch x 1e6, chunks of 30 000, 256 taps": lambda: drain(nm.oaconvolve(producer(small_h, 30000, -1), h256, -1, "same")),
    }
    from functools import partial
    sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")

    def fused(data, cs, taps):
        src = producer(data, cs, -1)
        fir = producer(partial(nm.oaconvolve, src, taps, -1, "same"), cs, -1, shape=src.shape)
        return drain(nm.sosfilt(fir, sos, -1))

    big_h = None
    cases["FIR -> sosfilt on the fused launch, resident 256 ch x 8 x 2^20, 1024 taps"] = lambda: fused(big, 1 << 20, h1024)
    cases["FIR -> sosfilt on the fused launch, host-fed 16 ch x 2^22 in chunks of 2^18, 256 taps"] = \
        lambda: fused(small_long, 1 << 18, h256)
    small_long = np.random.default_rng(3).standard_normal((16, 1 << 22))
    for name, fn in cases.items():
        row = {"case": name}
        for rep in range(3):
            for reach in ("1", "0"):
                os.environ["OSZ_FIR_REACH"] = os.environ["OSZ_CHAIN_REACH"] = reach
                fn()
                t0 = time.perf_counter()
                fn()
                row.setdefault("reach_ms" if reach == "1" else "plain_ms", []).append(round((time.perf_counter() - t0) * 1e3, 3))
        os.environ.pop("OSZ_FIR_REACH", None)
        os.environ.pop("OSZ_CHAIN_REACH", None)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

"""Where the time of cfg-1 (16 ch x 1e6, firwin(256, 0.2), chunksize 30 000, host ndarray in,
host ndarray out) goes: wall time of repeated runs and a cProfile of one."""
import cProfile
import io
import pstats
import sys
import time

import numpy as np
import scipy.signal as sps

sys.path.insert(0, ".")
from openseize_amd import producer                      # noqa: E402
from openseize_amd.core import numerical as nm          # noqa: E402


def run(x, h, cs):
    return np.concatenate(list(nm.oaconvolve(producer(x, cs, -1), h, -1, "same")), -1)


def main():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((16, 1_000_000))
    h = sps.firwin(256, 0.2)
    for cs in (30000, 1_000_000):
        run(x, h, cs)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            y = run(x, h, cs)
            ts.append(time.perf_counter() - t0)
        print(f"cs={cs}: min {min(ts) * 1e3:.2f} ms  median {sorted(ts)[3] * 1e3:.2f} ms "
              f"-> {x.size / min(ts) / 1e6:.0f} Msamples/s", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    run(x, h, 30000)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
    print(s.getvalue())
    pr = cProfile.Profile()
    pr.enable()
    run(x, h, 30000)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue())


if __name__ == "__main__":
    main()

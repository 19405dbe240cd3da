"""Where does downsample(M=25) on 4 x 18 875 000 resident samples spend its time?"""
import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
import torch
from openseize_amd import producer, _device as dev, _lib
from openseize_amd.resampling.resampling import downsample
lib = _lib.load()
x = dev.synth_normal(4, 18_875_000, seed=1)
def run():
    pro = downsample(producer(x, int(5e6), -1), M=25, fs=5000, chunksize=int(5e6))
    n = 0
    for c in pro:
        n += c.shape[-1]
    torch.cuda.synchronize()
    return n
run()
t0 = time.perf_counter(); n = run(); dt = time.perf_counter() - t0
print("total ms", dt * 1e3, n)
import ctypes
_lib.check(lib.osz_profile_reset()); _lib.check(lib.osz_profile_enable(1))
run()
_lib.check(lib.osz_profile_enable(0))
for kn in ("poly_block", "poly"):
    cnt, ms = ctypes.c_int64(), ctypes.c_double()
    lib.osz_profile_query(kn.encode(), ctypes.byref(cnt), ctypes.byref(ms))
    print(kn, cnt.value, ms.value)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])

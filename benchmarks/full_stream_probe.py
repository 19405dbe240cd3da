"""Where does the literal cfg-3 stream through the PUBLIC API spend its time beyond the kernels?
96 chunks (95 x 2^20 + 385 280) x 256 channels, resident ring: wall time, summed kernel time
(HIP events inside the library), CPU profile of the generator glue."""
import cProfile, ctypes, io, pstats, sys, time
sys.path.insert(0, '.')
from functools import partial
import scipy.signal as sps, torch
from openseize_amd import producer, _device as dev, _lib
from openseize_amd.core import numerical as nm
lib = _lib.load()
C, CHUNK, RAGGED = 256, 1 << 20, 100_000_000 - 95 * (1 << 20)
h = sps.firwin(1024, 0.2); sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
def run(lens):
    n = sum(lens)
    def source():
        for k, m in enumerate(lens):
            yield ring[k % 3][:, :m]
    src = producer(source, CHUNK, -1, shape=(C, n))
    fir = producer(partial(nm.oaconvolve, src, h, -1, "same"), CHUNK, -1, shape=(C, n))
    got = 0
    for out in nm.sosfiltfilt(fir, sos, -1):
        got += out.shape[-1]
    torch.cuda.synchronize()
    assert got == n
run([CHUNK] * 6 + [RAGGED])
lens = [CHUNK] * 95 + [RAGGED]
t0 = time.perf_counter(); run(lens); wall = time.perf_counter() - t0
_lib.check(lib.osz_profile_reset()); _lib.check(lib.osz_profile_enable(1))
t0 = time.perf_counter(); run(lens); wall2 = time.perf_counter() - t0
_lib.check(lib.osz_profile_enable(0))
tot = 0.0
for kn in ("chain_zp", "fir_oa", "fir_seam", "sos_fwd", "sos_bwd", "sos_dual", "sos_fwd_split", "sos_bwd_split", "chain_fwd", "sos_warmup"):
    cnt, ms = ctypes.c_int64(), ctypes.c_double()
    lib.osz_profile_query(kn.encode(), ctypes.byref(cnt), ctypes.byref(ms))
    if cnt.value: print(kn, cnt.value, round(ms.value, 3)); tot += ms.value
print("wall ms", wall * 1e3, "(with kernel timers:", wall2 * 1e3, ") kernels ms", tot)
pr = cProfile.Profile(); pr.enable(); run(lens); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])

import sys, os, time, json
sys.path.insert(0, ".")
import numpy as np, scipy.signal as sps, torch
from openseize_amd import _device as dev, _lib
CH, N = 256, 1 << 20
x = dev.synth_normal(CH, N, seed=3)
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
sizes = [int(v) for v in sys.argv[1:]] or [500, 1000, 2000, 2500, 5000, 6000, 10000, 14000, 20000]
for nf in sizes:
    wn = sps.get_window("hann", nf)
    sc = float(np.sqrt(1 / (float(nf) * np.sum(wn ** 2))))
    for mode, name in ((_lib.SPEC_PSD_MEAN, "psd"), (_lib.SPEC_DFT_SEGMENTS, "stft")):
        xx = x if mode == _lib.SPEC_PSD_MEAN else x[:, : 1 << 18]
        spm = dev.SpecStream(nf, nf, nf // 2, wn, sc, "constant", mode, CH)
        dt = timed(lambda: spm.push(xx), 5)
        print(json.dumps({"nfft": nf, "mode": name, "ms": round(dt * 1e3, 4)}), flush=True)
        spm.close()

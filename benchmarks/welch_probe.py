import sys, time, json
sys.path.insert(0, ".")
import numpy as np, scipy.signal as sps, torch
from openseize_amd import _device as dev, _lib
CH, N = 256, 1 << 20
x = dev.synth_normal(CH, N, seed=3)
nf = 4096
w = sps.get_window("hann", nf)
sc = float(np.sqrt(1 / (float(nf) * np.sum(w ** 2))))
for det in ("constant", "linear"):
    spm = dev.SpecStream(nf, nf, nf // 2, w, sc, det, _lib.SPEC_PSD_MEAN, CH)
    for _ in range(3): spm.push(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): spm.push(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(json.dumps({"detrend": det, "ms": round(dt * 1e3, 4)}), flush=True)
    spm.close()

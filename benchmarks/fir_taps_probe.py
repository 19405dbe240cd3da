import sys, json
sys.path.insert(0, '.')
sys.path.insert(0, 'benchmarks')
import scipy.signal as sps, torch
from openseize_amd import _device as dev
from sweep_chain import timed, CHUNK
C = 256
ring = [dev.synth_normal(C, CHUNK, seed=0, n0=k * CHUNK) for k in range(3)]
out = torch.zeros((C, CHUNK), dtype=torch.float64, device="cuda")
for nt in (1024, 1280, 1536, 1792, 2048):
    fir = dev.FirStream(sps.firwin(nt, 0.2), C)
    dt = timed(lambda k: fir.push(ring[k % 3], 0, out=out), 30, 10)
    fir.close()
    print(json.dumps({"ntaps": nt, "ms": dt * 1e3}), flush=True)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time, scipy.signal as sps, os
from openseize_amd import _device as dev
sos = sps.butter(6, [0.05, 0.3], "bandpass", output="sos")
C=256
x = dev.synth_normal(C, 1<<20, seed=1); y = torch.empty_like(x)
st = dev.SosStream(sos, C)
st.forward(x, out=y); torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): st.forward(x, out=y)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print(os.environ.get("OSZ_SOS_T"), os.environ.get("OSZ_SOS_NW"), os.environ.get("OSZ_SOS_WGS"), "fwd ms", round(dt*1e3,3), "TB/s", round(16*C*(1<<20)/dt/1e12,2), "warm", st.lib.osz_sos_warmup_len(st.h))

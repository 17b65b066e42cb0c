import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import webaudio_modem_amd as wm
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import envopts  # noqa: E402  (FSKHIP_* variables -> fskhip_set_option)
envopts.install()
from webaudio_modem_amd import _lib
cfg = dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200)
st = torch.cuda.Stream(); sh = st.cuda_stream
def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        fn(); st.synchronize(); a.record(st)
        for _ in range(reps): fn()
        b.record(st)
    b.synchronize()
    return a.elapsed_time(b) / reps
# host path (PCIe inclusive)
S, N = 16384, 24000
eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
x = torch.empty((S, N), dtype=torch.float32, device="cuda")
eng.synth_device(x.data_ptr(), N, N, 100, 0xF5C0DE, 400, 0.1, 1.0, sh); st.synchronize()
xh = x.cpu().numpy()
eng.demodulate_data(xh)
t0 = time.perf_counter(); 
for _ in range(3): eng.demodulate_data(xh)
dt = (time.perf_counter() - t0) / 3
print(json.dumps({"row": "demodulate_host (pageable host buffers, H2D + kernel + D2H + python unpack)", "streams": S, "samples": N, "ms": round(dt*1e3,1), "Msamples_per_s": round(S*N/dt/1e6,1), "GBps_in": round(S*N*4/dt/1e9,2)}))
eng.close()
# modulate + synth + awgn device
S = 65536; P = 100
eng = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F32)
L = eng.modulated_length(P) if hasattr(eng, "modulated_length") else int(_lib.lib().fskhip_modulated_length(eng._h, P))
pitch = (L + 63)//64*64
out = torch.empty((S, pitch), dtype=torch.float32, device="cuda")
pl = torch.randint(0, 256, (S, P), dtype=torch.uint8, device="cuda")
lens = torch.full((S,), P, dtype=torch.int32, device="cuda")
olens = torch.zeros((S,), dtype=torch.int32, device="cuda")
eng64 = wm.FSKEngine(S, cfg, precision=wm.PRECISION_F64)
ms = timed(lambda: _lib.check(_lib.lib().fskhip_modulate_device(eng64._h, pl.data_ptr(), lens.data_ptr(), P, out.data_ptr(), pitch, olens.data_ptr(), sh)), 3)
print(json.dumps({"row": "modulate_kernel<exact: V8's Math.sin operation for operation>, 100-byte payloads", "streams": S, "samples_per_stream": L, "ms": round(ms,2), "Msamples_per_s": round(S*L/ms/1e3,1), "written_GBps": round(S*L*4/ms/1e6,1)}))
eng64.close()
ms = timed(lambda: _lib.check(_lib.lib().fskhip_modulate_device(eng._h, pl.data_ptr(), lens.data_ptr(), P, out.data_ptr(), pitch, olens.data_ptr(), sh)), 3)
print(json.dumps({"row": "modulate_kernel<device-library sin> (fsk.ts:377-424), 100-byte payloads", "streams": S, "samples_per_stream": L, "ms": round(ms,2), "Msamples_per_s": round(S*L/ms/1e3,1), "written_GBps": round(S*L*4/ms/1e6,1)}))
ms = timed(lambda: eng.synth_device(out.data_ptr(), L, pitch, P, 0xF5C0DE, 400, 0.1, 1.0, sh), 3)
print(json.dumps({"row": "synth_kernel", "streams": S, "samples_per_stream": L, "ms": round(ms,2), "Msamples_per_s": round(S*L/ms/1e3,1)}))
ms = timed(lambda: eng.add_awgn_device(out.data_ptr(), L, pitch, 10.0, 0xA36, sh), 3)
print(json.dumps({"row": "power_kernel + awgn_kernel", "streams": S, "samples_per_stream": L, "ms": round(ms,2), "Msamples_per_s": round(S*L/ms/1e3,1), "GBps_rw": round(S*L*12/ms/1e6,1)}))

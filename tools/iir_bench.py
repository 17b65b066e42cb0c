import sys, os, time
sys.path.insert(0, "/root/repo")
import webaudio_modem_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-": L.LIB_PATH = sys.argv[1]
import torch, webaudio_modem_amd as wm
for S in (4096, 16384, 65536):
    N = 48000
    for prec, name in ((wm.PRECISION_F32, "f32"), (wm.PRECISION_F64, "f64")):
        co = wm.FilterDesign.butterworthLowpass(1200, 48000); b, a = co["b"], co["a"]
        f = wm.IIRFilterBatch(b, a, S, precision=prec)
        x = torch.randn((S, N), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        f.process_device(x.data_ptr(), N, N, y.data_ptr(), N, st)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5): f.process_device(x.data_ptr(), N, N, y.data_ptr(), N, st)
        torch.cuda.synchronize(); dt = (time.time() - t) / 5
        print(S, name, "%.3f ms %.1f Gsamples/s" % (dt * 1e3, S * N / dt / 1e9))

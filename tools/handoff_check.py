#!/usr/bin/env python3
"""handoff_check.py (GPU box): the bound on the multi-wave kernels' hand-off waits (csrc/fsk_wait.h), shown to work.

A measurement build with -DFSK_SPIN_CAP_LOG2=0 (tools/build_variant.sh cap0 -DFSK_SPIN_CAP_LOG2=0) makes the FIRST failed poll of any wait
run into the bound: every launch of a two-, four- or seven-wave kernel then ends early, flagged, instead of completing -- which is
what a lost counter update would look like with the shipped bound of 2^22 polls.  Checked per kernel, each in a child process under
a timeout: the launch FINISHES (no hung GPU), fskhip_synchronize() returns FSKHIP_E_HANDOFF, so does the next demodulate call
(sticky), and the shipped library on the same input decodes everything with no fault.

  tools/handoff_check.py [libtag]        (default cap0)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import webaudio_modem_amd._lib as L
if sys.argv[1] != "-": L.LIB_PATH = sys.argv[1]
import torch
import webaudio_modem_amd as wm
S, N, kernel = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
eng = wm.FSKEngine(S, dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200), precision=wm.PRECISION_F32)
if kernel != "auto": eng.set_option("kernel", kernel)
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((S, N), dtype=torch.float32, device="cuda")
eng.synth_device(x.data_ptr(), N, N, 100, 0xF5C0DE, 400, 0.1, 1.0, st)
op = eng.max_bytes(N)
out = torch.zeros((S, op), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(S, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
eng.demodulate_device(x.data_ptr(), N, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st)
torch.cuda.synchronize()                                   # the launch has FINISHED, whatever it did
name = eng.last_kernel()
try:
    eng.synchronize()
    first = "ok"
except L.FskHipError as e:
    first = "rc=%%d" %% e.code
try:
    eng.demodulate_device(x.data_ptr(), N, N, out.data_ptr(), op, cnt.data_ptr(), 0, 0, st)
    torch.cuda.synchronize()
    second = "ok"
except L.FskHipError as e:
    second = "rc=%%d" %% e.code
print("RESULT", name.replace(" ", ""), first, second, int(cnt.sum().item()))
''' % ROOT


def run(lib, S, N, kernel):
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, lib, str(S), str(N), kernel], capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return None, "TIMEOUT (a hung launch)"
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        return None, "FAILED " + r.stderr[-300:].replace("\n", " | ")
    return line[0].split()[1:], ""


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "cap0"
    capped = os.path.join(ROOT, "tools", "build", "libfskhip_%s.so" % tag)
    bad = 0
    # (streams, samples, kernel option): four waves, four waves over time slices, seven waves, two waves
    for S, N, kernel in ((65536, 48000, "auto"), (131072, 48000, "auto"), (4096, 48000, "auto"), (16384, 48000, "two-wave")):
        ref, why = run("-", S, N, kernel)
        got, why2 = run(capped, S, N, kernel)
        ok = ref is not None and got is not None and ref[1:3] == ["ok", "ok"] and got[1:3] == ["rc=-8", "rc=-8"] and int(got[3]) < int(ref[3])
        bad += 0 if ok else 1
        print("%-7d x %-6d %-9s shipped: %s   bound 0: %s   %s" % (S, N, kernel, " ".join(ref) if ref else why, " ".join(got) if got else why2, "ok" if ok else "UNEXPECTED"), flush=True)
    print("handoff_check:", "ok" if bad == 0 else "%d UNEXPECTED" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

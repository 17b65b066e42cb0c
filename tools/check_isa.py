#!/usr/bin/env python3
"""Static check of the compiled fast demod kernel (runs anywhere hipcc is installed, no GPU):
its tile prefetch uses inline-asm loads whose results are only valid after a hand-placed s_waitcnt, so
the register allocator must never spill inside that kernel (a spill store of an in-flight load result
would save garbage), and the plain variant must fit 128 VGPRs (4 waves per SIMD)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources():
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_demod.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-c", src, "-o", os.path.join(tmp, "d.o"), "-Rpass-analysis=kernel-resource-usage"]
        out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            res[cur][m.group(1)] = int(m.group(2))
    return res


if __name__ == "__main__":
    r = kernel_resources()
    bad = 0
    for name, v in r.items():
        if "demod_fast_kernel" in name:
            print(name[:60], v)
            if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0):
                bad += 1
    sys.exit(1 if bad else 0)

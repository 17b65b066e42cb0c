"""The fp64 demodulator's own sin/cos/atan2 (webaudio_modem_amd/csrc/fsk_f64math.h) are plain C++: compiled here with g++
(no GPU) and checked against long double libm over their whole argument ranges -- within 2 units of 2^-53 for the NCO
phasor, 2 ulp for atan2, Math.atan2's zero conventions exactly."""
import os
import subprocess

from conftest import ROOT


def test_f64math_against_long_double_libm(tmp_path):
    exe = str(tmp_path / "f64math_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "webaudio_modem_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "f64math_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

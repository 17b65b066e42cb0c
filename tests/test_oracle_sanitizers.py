"""SURVEY section 5 / VERDICT r02 #9: the CPU oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer on the
golden suite (the GPU side cannot run sanitizers on this pool; the oracle is the one piece of C that every parity claim
rests on).  The sanitized build is a separate shared object (oracle/Makefile `asan`), loaded by oracle/pyoracle.py when
FSK_ORACLE_SANITIZE=1; a dlopen'ed ASan library needs the runtime preloaded, so the suite runs in a child interpreter."""
import os
import subprocess
import sys

from conftest import ROOT


def test_oracle_golden_suite_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        import pytest
        pytest.skip("no libasan.so next to this gcc")
    env = dict(os.environ, FSK_ORACLE_SANITIZE="1", LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the golden comparisons of the demodulator, the modulator and the next-row restatements that go through the C library
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert os.path.exists(os.path.join(ROOT, "oracle", "libfsk_oracle_asan.so"))

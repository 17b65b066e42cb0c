import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# The tests pin kernels and launch shapes with FSKHIP_* environment variables (monkeypatch.setenv); the library and the
# package read none -- tools/envopts.py maps them to fskhip_set_option() calls on every engine a test creates.
sys.path.insert(0, os.path.join(ROOT, "tools"))
import envopts  # noqa: E402

envopts.install()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """tests/golden/: vectors captured from the real reference (oracle/refrun/make_golden.py)."""

    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "manifest.json")) as fh:
            self.manifest = json.load(fh)
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "golden.npz"))
        self.cases = {c["name"]: c for c in self.manifest["cases"]}

    def array(self, name):
        return self.arrays[name]

    def case_input(self, case):
        return self.arrays[case["input"]]


class GoldenNext:
    """tests/golden/golden_next.npz: CRC-16 / XModem packets / ChunkedModulator / FSKProcessor quantum loop, captured
    from the real reference classes (oracle/refrun/golden_harness_next.js)."""

    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "manifest_next.json")) as fh:
            self.manifest = json.load(fh)
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "golden_next.npz"))

    def ragged(self, ref):
        data, off = self.arrays[ref["data"]], self.arrays[ref["off"]]
        return [bytes(data[off[i]:off[i + 1]]) for i in range(len(off) - 1)]


class GoldenHostile:
    """tests/golden/golden_hostile.npz (round 6): what the REAL reference does with NaN / Inf / out-of-range / subnormal samples
    (oracle/refrun/golden_harness_hostile.js).  status_vector: the status numbers as doubles (JSON has no NaN): frameStarted,
    globalSampleCounter, receivedBitsLength, byteBufferLength, demodulationCalls, syncDetections, silenceThreshold,
    totalSamplesProcessed, AGC gain."""

    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "manifest_hostile.json")) as fh:
            self.manifest = json.load(fh)
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "golden_hostile.npz"))
        self.cases = {c["name"]: c for c in self.manifest["cases"]}

    def array(self, name):
        return self.arrays[name]

    def case_input(self, case):
        return self.arrays[case["input"]]

    def clean_input(self, case):
        """the same two frames without the bad sample / scaling (h_<cfg>_clean)"""
        return self.arrays[self.cases["_".join(case["name"].split("_")[:2]) + "_clean"]["input"]]


HOSTILE_STATUS_KEYS = ["frameStarted", "globalSampleCounter", "receivedBitsLength", "byteBufferLength", "demodulationCalls",
                       "syncDetections", "silenceThreshold", "totalSamplesProcessed", "agcGain"]


def hostile_status_matches(st, vec, rel):
    """st: an engine's / the oracle's status dict; vec: the reference's status_vector.  Counters exact, the two reals within rel."""
    for i, k in enumerate(HOSTILE_STATUS_KEYS):
        if k == "byteBufferLength":
            continue
        a, b = float(st[k]), float(vec[i])
        if k in ("silenceThreshold", "agcGain"):
            if not ((np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= rel * abs(b)):
                return "%s: %r vs reference %r" % (k, a, b)
        elif a != b:
            return "%s: %r vs reference %r" % (k, a, b)
    return None


_GOLDEN = None
_GOLDEN_NEXT = None
_GOLDEN_HOSTILE = None


def golden_hostile():
    global _GOLDEN_HOSTILE
    if _GOLDEN_HOSTILE is None:
        _GOLDEN_HOSTILE = GoldenHostile()
    return _GOLDEN_HOSTILE



def golden_next():
    global _GOLDEN_NEXT
    if _GOLDEN_NEXT is None:
        _GOLDEN_NEXT = GoldenNext()
    return _GOLDEN_NEXT


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = Golden()
    return _GOLDEN


@pytest.fixture(scope="session")
def gold():
    return golden()


def case_names():
    return [c["name"] for c in golden().manifest["cases"]]


def run_chunked(demod_call, x, chunk):
    """Feed x through demod_call(chunk_array) -> (bytes, eod) in the fixture's chunking.
    Returns (all_bytes, eod_total, nonempty_calls, n_calls) in the manifest's format."""
    chunk = chunk or max(1, x.size)
    out = b""
    eod_total = 0
    nonempty = []
    off = 0
    i = 0
    while True:
        n = min(chunk, x.size - off)
        b, e = demod_call(x[off:off + n])
        if len(b) or e:
            nonempty.append({"i": i, "bytes": list(b), "eod": int(e)})
        out += bytes(b)
        eod_total += int(e)
        off += n
        i += 1
        if off >= x.size:
            break
    return out, eod_total, nonempty, i


STATUS_EXACT_KEYS = ["frameStarted", "globalSampleCounter", "receivedBitsLength", "demodulationCalls",
                     "syncDetections", "totalSamplesProcessed"]

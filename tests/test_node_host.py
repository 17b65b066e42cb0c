"""The reference's host language is TypeScript/Node: napi/fsk-core.js keeps FSKCore's public surface and
calls the HIP engine through the N-API addon.  These tests run the JS suites under `node` (skipped when
node is absent on the box)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
JS = os.path.join(ROOT, "tests", "js", "fsk_core_test.js")
JS_NEXT = os.path.join(ROOT, "tests", "js", "next_rows_test.js")


def _build():
    import __graft_entry__ as ge
    ge.build()
    addon = os.path.join(ROOT, "napi", "fsk_addon.node")
    if not os.path.exists(addon):
        pytest.skip("N-API addon not built (no node headers)")


def _announce(what):
    """make the log SHOW that the N-API path ran (a passing test is a dot): the node version and the addon's own ABI
    version -- read through the addon, i.e. through libfskhip.so -- as a warning, which pytest lists in its summary even with -q."""
    import warnings
    v = subprocess.run([NODE, "--version"], capture_output=True, text=True, timeout=30).stdout.strip()
    abi = subprocess.run([NODE, "-e", "console.log(require(%r).abiVersion)" % os.path.join(ROOT, "napi", "fsk_addon.node")],
                         capture_output=True, text=True, timeout=60).stdout.strip()
    warnings.warn(UserWarning("N-API path ran: %s -- node %s, fsk_addon.node abiVersion %s (libfskhip.so)" % (what, v, abi)))


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_host_cpu_side():
    _build()
    out = subprocess.run([NODE, JS, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "js cpu tests ok" in out.stdout
    _announce("tests/js/fsk_core_test.js cpu side (no compute calls)")


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_host_gpu_roundtrips():
    _build()
    out = subprocess.run([NODE, JS, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "js gpu tests ok" in out.stdout
    _announce("tests/js/fsk_core_test.js gpu: FSKCore / FSKBatch through the addon on the GPU")


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_next_rows_cpu_side():
    _build()
    out = subprocess.run([NODE, JS_NEXT, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "js next cpu tests ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_next_rows_gpu():
    _build()
    out = subprocess.run([NODE, JS_NEXT, "gpu"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "js next gpu tests ok" in out.stdout
    _announce("tests/js/next_rows_test.js gpu: FSKProcessor / XModem / filters through the addon on the GPU")

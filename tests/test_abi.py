"""CPU-side checks of the drop-in boundary: libfskhip.so loads, exports every symbol that
include/fskhip.h declares, its configure-time host functions agree with the reference's golden
vectors, and -- without a GPU -- every compute entry point FAILS LOUDLY (there is no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


def _built():
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="module")
def lib():
    _built()
    from webaudio_modem_amd import _lib
    return _lib


def _declared_symbols():
    src = ""
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if name.endswith(".h"):
            with open(os.path.join(ROOT, "include", name)) as fh:
                src += fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fskhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    L = C.CDLL(lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 55
    for name in declared:
        assert hasattr(L, name), "libfskhip.so does not export %s" % name
    # and the ctypes table binds exactly the header's set
    assert sorted(lib.SYMBOL_NAMES) == declared


def test_abi_version_and_default_config(lib):
    L = lib.lib()
    assert L.fskhip_abi_version() == 8
    c = lib.Config()
    L.fskhip_default_config(C.byref(c))
    # DEFAULT_FSK_CONFIG fsk.ts:19-33
    assert (c.sampleRate, c.baudRate, c.markFrequency, c.spaceFrequency) == (48000, 1200, 1650, 1850)
    assert list(c.preamblePattern)[:c.preambleLen] == [0x55, 0x55] and list(c.sfdPattern)[:c.sfdLen] == [0x7E]
    assert (c.startBits, c.stopBits, c.parity, c.syncThreshold, c.agcEnabled, c.preFilterBandwidth) == (1, 1, 0, 0.85, 1, 800)


def test_error_codes_of_the_header_are_the_bindings(lib):
    """include/fskhip.h's FSKHIP_E_* values and webaudio_modem_amd/_lib.py's E_* constants are one table (ABI 8 added
    FSKHIP_E_HANDOFF: a hand-off wait of a multi-wave kernel ran into its bound, csrc/fsk_wait.h); the kernels that carry the bound
    are the ones with hand-off waits -- no bare s_sleep poll is left in them."""
    import re
    hdr = open(os.path.join(ROOT, "include", "fskhip.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"FSKHIP_(E_[A-Z_]+) = (-\d+)", hdr)}
    assert codes and codes["E_HANDOFF"] == -8
    for name, value in codes.items():
        assert getattr(lib, name) == value, name
    csrc = os.path.join(ROOT, "webaudio_modem_amd", "csrc")
    for f in ("fsk_blk.hip", "fsk_blk6.hip", "fsk_pipe.hip", "fsk_demod.hip", "fsk_mod.hip"):
        src = open(os.path.join(csrc, f)).read()
        assert "FSK_SPIN(" in src or "B6_SPIN(" in src, f
        bare = [l for l in src.splitlines() if "__builtin_amdgcn_s_sleep" in l and "//" not in l.split("__builtin_amdgcn_s_sleep")[0]]
        # (the one left: the persistent launch's wait for another workgroup's time slice, bounded in place -- fsk_blk.hip)
        assert len(bare) <= (1 if f == "fsk_blk.hip" else 0), (f, bare)


def test_config_struct_layout_matches_oracle(lib):
    """fskhip_config and the oracle's fsko_config are the same FSKConfig layout."""
    from oracle import pyoracle as po
    assert C.sizeof(lib.Config) == C.sizeof(po.Config)
    assert [f[0] for f in lib.Config._fields_] == [f[0] for f in po.Config._fields_]
    assert C.sizeof(lib.Status) == C.sizeof(po.Status)


def test_filter_design_host_functions_match_reference(lib):
    from webaudio_modem_amd import FilterDesign
    for fd in golden().manifest["filter_design"]:
        if not fd["fn"].startswith("butterworth"):
            continue
        got = getattr(FilterDesign, fd["fn"])(*fd["args"])
        assert got == fd["out"], fd  # bit-exact doubles


def test_synth_helpers_are_deterministic(lib):
    L = lib.lib()
    a = [L.fskhip_synth_payload_byte(0xF5C0DE, 3, 1, i) for i in range(16)]
    b = [L.fskhip_synth_payload_byte(0xF5C0DE, 3, 1, i) for i in range(16)]
    assert a == b and len(set(a)) > 8
    lead, amp = C.c_uint32(), C.c_double()
    L.fskhip_synth_stream_params(0xF5C0DE, 5, 400, 0.1, 1.0, C.byref(lead), C.byref(amp))
    assert 0 <= lead.value <= 400 and 0.1 <= amp.value <= 1.0


def test_sinc_designs_match_reference(lib):
    from webaudio_modem_amd import FilterDesign
    n = 0
    for fd in golden().manifest["filter_design"]:
        if not fd["fn"].startswith("sinc"):
            continue
        got = getattr(FilterDesign, fd["fn"])(*fd["args"])
        assert len(got) == len(fd["out"])
        np.testing.assert_allclose(got, fd["out"], rtol=0, atol=1e-16)  # libm vs V8 fdlibm last ulp of sin/cos
        n += 1
    assert n >= 7


def test_next_rows_fail_loudly_without_gpu(lib):
    import webaudio_modem_amd as wm
    if lib.lib().fskhip_device_count() > 0:
        pytest.skip("a GPU is present")
    for call in (lambda: wm.CRC16.calculate(b"123456789"), lambda: wm.serialize_batch([1], [b"abc"]),
                 lambda: wm.scan_bursts([b"\x04"], [1]), lambda: wm.FIRFilter([0.5, 0.5])):
        with pytest.raises(wm.FskHipError) as ei:
            call()
        assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)
    # argument checks with the reference's texts come before any device work (packet.ts:22-27)
    with pytest.raises(ValueError, match=r"Invalid sequence: 0\. Must be 1-255\."):
        wm.serialize_batch([0], [b"x"])
    with pytest.raises(ValueError, match=r"Payload too large: 256\. Max 255 bytes\."):
        wm.serialize_batch([1], [bytes(256)])


def test_no_gpu_means_loud_failure(lib):
    """On a box without a HIP device the product must refuse to run, not fall back."""
    import webaudio_modem_amd as wm
    if lib.lib().fskhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wm.FskHipError) as ei:
        wm.FSKEngine(1, {})
    assert ei.value.code == -4  # FSKHIP_E_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)
    core = wm.FSKCore()
    with pytest.raises(RuntimeError, match="not configured"):
        core.demodulateData(np.zeros(4, np.float32))
    with pytest.raises(wm.FskHipError):
        core.configure({})


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under webaudio_modem_amd/ may reference it."""
    pkg = os.path.join(ROOT, "webaudio_modem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc", ".js", ".ts")):
                with open(os.path.join(dirpath, f), encoding="utf-8") as fh:
                    src = fh.read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)


def test_whole_tile_kernels_never_spill():
    """tools/check_isa.py: the one-wave kernel's asm-issued prefetch is only safe without spills; it is built for 3 waves
    per SIMD (<= 168 VGPRs), the two-wave kernel for 4 workgroups = 8 waves per CU, 2 per SIMD (<= 256)."""
    import shutil
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not installed")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    res = check_isa.kernel_resources()
    fused = {k: v for k, v in res.items() if "demod_fused_kernel" in k}
    assert len(fused) == 4  # <write-back, uniform-constants>
    for name, v in fused.items():
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (name, v)
        assert v["VGPRs"] <= 168, (name, v)
    pipe = {k: v for k, v in res.items() if "demod_pipe_kernel" in k}
    assert len(pipe) == 4
    for name, v in pipe.items():
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (name, v)
        assert v["VGPRs"] <= 256, (name, v)
    # the registers the asm prefetch lands in are never touched while a load may still be in flight
    assert check_isa.prefetch_register_hazards() == []
    assert check_isa.pipe_prefetch_hazards() == []
    # the four-wave block kernel (fsk_blk.hip): within 128 VGPRs, no scratch access inside the per-tile loops, the
    # asynchronous hand-off counter read's registers untouched until a wait covers it
    assert check_isa.blk_checks() == []
    # the seven-wave small-batch kernel (fsk_blk6.hip): within 256 VGPRs (two waves per SIMD), no scratch memory at all
    assert check_isa.blk6_resources() == []


def test_time_sliced_launches_hand_every_field_on_with_device_scope_policy():
    """a time slice of fsk_blk.hip's persistent launch hands its state to a slice that may run on another XCD
    (another L2).  That is only correct if EVERY access to what is handed on carries the device-scope cache policy (COH =
    kCohSc1 under SL): one helper instantiated without the template argument would leave one field stale, intermittently.
    The helpers take COH as a template argument (default 0 for the other kernels); here every instantiation of one of them
    in fsk_blk.hip must name it, the PIPE_* state macros expand to `COH` by themselves, and the kernel defines COH from SL."""
    import re
    src = open(os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_blk.hip"), encoding="utf-8").read()
    assert "constexpr int COH = SL ? kCohSc1 : 0;" in src
    src = re.sub(r"//[^\n]*", "", src)          # (code only)
    helpers = ["pipe_free0", "front_load", "back_load", "back_pair", "pipe_store", "ist_load", "ist_store", "back_reset"]
    for h in helpers:
        for m in re.finditer(r"\b%s\s*(<[^>(]*>)?\s*\(" % h, src):
            targs = m.group(1) or ""
            assert "COH" in targs, "fsk_blk.hip: %s instantiated without COH: %r" % (h, src[m.start():m.start() + 60])
    dev = open(os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_pipe_dev.h"), encoding="utf-8").read()
    for macro in ("PIPE_RLOAD", "PIPE_ILOAD", "PIPE_RSTORE", "PIPE_ISTORE"):
        line = [l for l in dev.splitlines() if l.startswith("#define " + macro)][0]
        assert re.search(r", COH\)+$", line.rstrip()), line
    # the helpers that reach state through other helpers pass it down
    for inner in ("ist_store<COH>", "ist_load<COH>", "back_reset<UNI, COH>"):
        assert inner in dev, inner


def test_no_bit_cast_of_a_vector_element_expression():
    """hipcc 7.2 compiles `__builtin_bit_cast(uint32_t, v.y)` on an ext_vector element EXPRESSION to a read of element 0
    (fsk_blk6.hip met it twice: decoded bytes wrong, no diagnostic).  The sources copy the elements into floats first; this keeps it so."""
    import glob
    import re
    pat = re.compile(r"__builtin_bit_cast\(\s*(?:uint32_t|int|int32_t|unsigned)\s*,\s*[A-Za-z_][\w\.\[\]]*\.[xyzw]\s*\)")
    bad = []
    for f in glob.glob(os.path.join(ROOT, "webaudio_modem_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "tools", "*.hip")):
        for i, line in enumerate(open(f), 1):
            if pat.search(line):
                bad.append("%s:%d: %s" % (os.path.relpath(f, ROOT), i, line.strip()))
    assert not bad, "\n".join(bad)

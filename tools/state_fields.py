"""per-stream state field names in fsk_params.h order (tools/*diag*.py, tests/test_gpu_fullsize.py) -- read from the header itself"""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webaudio_modem_amd", "csrc", "fsk_params.h")


def _macro_fields(text, macro):
    m = re.search(r"#define %s\(X\)(.*?)\n(?!\s*(?:X\(|/\*|\\))" % macro, text, re.S)
    body = m.group(1)
    # the definition ends with the first line that does not end in a backslash
    lines = []
    for line in ("#define %s(X)" % macro + body).split("\n"):
        lines.append(line)
        if not line.rstrip().endswith("\\"):
            break
    body = re.sub(r"/\*.*?\*/", "", "\n".join(lines), flags=re.S)
    return re.findall(r"X\((\w+)\)", body)[0:] if "X(" in body else []


def _load():
    text = open(_HDR).read()
    real = sum((_macro_fields(text, m) for m in ("FSK_REAL_FIELDS", "FSK_REAL_FIELDS_PIPE", "FSK_REAL_FIELDS_QUALITY")), [])
    integer = sum((_macro_fields(text, m) for m in ("FSK_INT_FIELDS", "FSK_INT_FIELDS_PIPE", "FSK_INT_FIELDS_QUALITY")), [])
    return real, integer


REAL, INT = _load()

#!/usr/bin/env python3
"""Mechanical TypeScript -> Node-12 JavaScript eraser for the four hot-path files of
the reference.

TEST INFRASTRUCTURE ONLY.  This runs in the build container, where /root/reference is
mounted.  It reads the reference sources *where they lie*, erases TypeScript-only syntax
(annotations, interfaces, generics, access modifiers, `?.`/`??`) with regexes, and writes
the result into a TEMP directory given on the command line.  Nothing it produces is ever
copied into this repository: the stripped JS exists only to run the real reference under
Node and capture golden input/output vectors (oracle/refrun/make_golden.py), which are
data and are committed under tests/golden/.

usage: strip_ts.py <reference_root> <out_dir>
"""
import os
import re
import sys

FILES = ["src/utils.ts", "src/dsp/filters.ts", "src/core.ts", "src/modems/fsk.ts"]
# the "next" rows of SURVEY.md 8(f): CRC-16 / XModem packets, ChunkedModulator
NEXT_FILES = ["src/utils/crc16.ts", "src/transports/xmodem/types.ts", "src/transports/xmodem/packet.ts",
              "src/webaudio/chunked-modulator.ts"]

TYPE_ATOM = (
    r"(?:number\[\]\[\]|number\[\]|number|void|boolean|string|unknown|any"
    r"|Promise<[A-Za-z0-9_]+>"
    r"|\{\s*b:\s*number\[\],\s*a:\s*number\[\]\s*\}"
    r"|\(_event:\s*Event\)\s*=>\s*void"
    r"|[A-Z][A-Za-z0-9_]*(?:<[A-Za-z0-9_<>, ]+>)?(?:\[\])?(?:\s*\|\s*undefined)?)"
)


def drop_blocks(src, start_re):
    """Remove `interface X {...}` style blocks by brace matching."""
    out = []
    i = 0
    pat = re.compile(start_re, re.M)
    while True:
        m = pat.search(src, i)
        if not m:
            out.append(src[i:])
            break
        out.append(src[i:m.start()])
        j = src.index("{", m.start())
        depth = 0
        while True:
            if src[j] == "{":
                depth += 1
            elif src[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        i = j + 1
    return "".join(out)


def strip(src, name):
    if name.endswith("core.ts"):
        # only the Event / EventEmitter / BaseModulator part is needed (core.ts:204-289)
        lines = src.split("\n")
        src = "\n".join(lines[203:289])
    src = re.sub(r"^import .*?;\s*$", "", src, flags=re.M)
    # enum -> frozen object (types.ts:29-34)
    def enum_to_obj(m):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        items = [it.strip() for it in body.split(",") if it.strip()]
        return "const %s = {%s};" % (m.group(1), ", ".join(it.replace("=", ":") for it in items))
    src = re.sub(r"^(?:export )?enum (\w+) \{(.*?)^\}", enum_to_obj, src, flags=re.M | re.S)
    src = src.replace("} as const;", "};")
    src = re.sub(r"<TConfig extends BaseModulatorConfig = BaseModulatorConfig>", "", src)
    src = re.sub(r":\s*IModulator<TConfig>", "", src)
    src = re.sub(r":\s*(?:Float32Array|ChunkResult|DataPacket)(?:\s*\|\s*null)?(?=\s*[=;{])", "", src)
    src = re.sub(r"static (POLYNOMIAL|INITIAL_VALUE|FINAL_XOR) = (0x[0-9A-Fa-f]+);", r"static get \1() { return \2; }", src)
    src = drop_blocks(src, r"^(?:export )?interface [A-Za-z0-9_<>, ]+ (?:extends [A-Za-z0-9_<>, ]+ )?\{")
    src = re.sub(r"^export type [^;]*;\s*$", "", src, flags=re.M)
    src = re.sub(r"^export ", "", src, flags=re.M)
    src = src.replace("abstract class", "class")
    src = re.sub(r"^\s*abstract .*$", "", src, flags=re.M)
    src = re.sub(r"<TConfig extends BaseModulatorConfig>", "", src)
    src = re.sub(r"<T extends TypedArray>", "", src)
    src = src.replace("extends BaseModulator<FSKConfig>", "extends BaseModulator")
    src = re.sub(r"\s+implements [A-Za-z0-9_<>, ]+\s*\{", " {", src)
    src = re.sub(r"new RingBuffer<[A-Za-z0-9_]+>", "new RingBuffer", src)
    src = re.sub(r"new Map<.*>\(\)", "new Map()", src)
    # constructor parameter property (core.ts:206)
    src = src.replace("constructor(public readonly data: unknown = null) {}",
                      "constructor(data = null) { this.data = data; }")
    # casts
    src = re.sub(r"undefined as [^\n]*?(,?)\s*$", r"undefined\1", src, flags=re.M)
    src = re.sub(r"\[\] as [a-z]+\[\]", "[]", src)
    src = re.sub(r"\s+as (?:any|T|FSKConfig|number\[\]|[A-Z][A-Za-z0-9_]*(?:<[^>]*>)?(?:\s*\|\s*undefined)?)", "", src)
    # access modifiers
    src = re.sub(r"\b(?:private|protected|public|readonly)\s+", "", src)
    # definite assignment
    src = re.sub(r"(\w)!:", r"\1:", src)
    src = src.replace("get(eventName)!", "get(eventName)")
    # `x?: T` / `x: T` annotations followed by , ) = ; {
    for _ in range(4):
        src = re.sub(r"(\b[A-Za-z_][A-Za-z0-9_]*|\))\??:\s*" + TYPE_ATOM + r"(?=\s*[,)=;{])", r"\1", src)
    # class field with annotation but no initialiser -> drop to a bare declaration
    # optional chaining / nullish (fsk.ts:184-186, 466, 486)
    src = re.sub(r"(this\.dsp\.iqFilters)\?\.(i|q)\.reset\(\)", r"\1 && \1.\2.reset()", src)
    src = re.sub(r"(this\.dsp\.postFilter)\?\.reset\(\)", r"\1 && \1.reset()", src)
    src = re.sub(r"(this\.frame\.syncSamplesBuffer)\?\.clear\(\)", r"\1 && \1.clear()", src)
    src = re.sub(r"(this\.frame\.syncSamplesBuffer)\?\.length \?\? 0", r"(\1 ? \1.length : 0)", src)
    return src


def strip_transport(ref):
    """BaseTransport (core.ts:292-351) + XModemTransport (transports/xmodem/xmodem.ts), erased to Node-12 JavaScript.
    The receive grammar of SURVEY.md 8(f2) is pinned against THIS class (golden_harness_next.js drives receiveData() through a
    scripted data channel); AbortController / AbortSignal.timeout / AbortSignal.any, which Node 12 lacks, are provided by
    the harness as globals."""
    with open(os.path.join(ref, "src/core.ts"), encoding="utf-8") as fh:
        core = "\n".join(fh.read().split("\n")[298:351])
    with open(os.path.join(ref, "src/transports/xmodem/xmodem.ts"), encoding="utf-8") as fh:
        xm = fh.read()
    src = core + "\n" + xm
    src = re.sub(r"^import .*?;\s*$", "", src, flags=re.M)
    src = drop_blocks(src, r"^(?:export )?interface [A-Za-z0-9_<>, ]+ (?:extends [A-Za-z0-9_<>, ]+ )?\{")

    def state_enum(m):   # auto-numbered enum with reverse mapping (State[this.protocol.state])
        body = re.sub(r"//[^\n]*", "", m.group(2))
        names = [it.strip() for it in body.split(",") if it.strip()]
        return "const %s = {};\n%s.forEach((n, i) => { %s[n] = i; %s[i] = n; });" % (m.group(1), repr(names).replace("'", '"'), m.group(1), m.group(1))
    src = re.sub(r"^(?:export )?enum (\w+) \{(.*?)^\}", state_enum, src, flags=re.M | re.S)
    src = re.sub(r"^export ", "", src, flags=re.M)
    src = re.sub(r"abstract class BaseTransport\s+extends EventEmitter\s+implements ITransport \{", "class BaseTransport extends EventEmitter {", src)
    src = re.sub(r"^\s*abstract .*$", "", src, flags=re.M)
    # method with a type parameter and function-typed arguments
    src = src.replace("private async withRetry<T>(\n    operation: () => Promise<T>,\n    maxRetries: number,\n    onRetry?: (_retryCount: number) => void,\n    externalSignal?: AbortSignal\n  ): Promise<T> {",
                      "async withRetry(operation, maxRetries, onRetry, externalSignal) {")
    # casts
    src = src.replace("[] as Uint8Array[]", "[]").replace("[] as number[]", "[]")
    src = src.replace("undefined as AbortController | undefined", "undefined")
    src = re.sub(r"\s+as (?:StateChangeEvent|[A-Z][A-Za-z0-9_]*)\)", ")", src)
    # access modifiers
    src = re.sub(r"\b(?:private|protected|public|readonly)\s+", "", src)
    # parameter / return annotations
    obj_t = r"\{\s*signal\??:\s*AbortSignal\s*\}"
    atom = r"(?:number\[\]|number|void|boolean|string|unknown|any|" + obj_t + r"|Promise<[A-Za-z0-9_\[\]]+>|Partial<[A-Za-z0-9_]+>|[A-Z][A-Za-z0-9_]*(?:\[\])?(?:\s*\|\s*undefined)?)"
    for _ in range(4):
        src = re.sub(r"(\b[A-Za-z_][A-Za-z0-9_]*|\))\??:\s*" + atom + r"(?=\s*[,)=;{])", r"\1", src)
    # optional chaining (Node 12 has none)
    src = re.sub(r"(\bonRetry)\?\.\(", r"\1 && \1(", src)
    src = re.sub(r"((?:this\.)?[A-Za-z_][A-Za-z0-9_.]*)\?\.([A-Za-z_][A-Za-z0-9_.]*)", r"(\1 ? \1.\2 : undefined)", src)
    return "// ---- src/core.ts:292-351 + src/transports/xmodem/xmodem.ts ----\n" + src


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    parts = []
    for f in FILES + NEXT_FILES:
        with open(os.path.join(ref, f), encoding="utf-8") as fh:
            parts.append("// ---- %s ----\n%s" % (f, strip(fh.read(), f)))
    parts.append(strip_transport(ref))
    parts.append(
        "module.exports = {FSKCore, DEFAULT_FSK_CONFIG, IIRFilter, FIRFilter, "
        "FilterDesign, FilterFactory, RingBuffer, CRC16, XModemPacket, ControlType, PacketConstants, "
        "ChunkedModulator, XModemTransport, Event};\n"
    )
    with open(os.path.join(out, "ref_bundle.js"), "w", encoding="utf-8") as fh:
        fh.write("\n".join(parts))
    leftover = [m for m in ("?.", "??") if m in "\n".join(parts)]
    if leftover:
        print("warning: leftover syntax", leftover, file=sys.stderr)


if __name__ == "__main__":
    main()

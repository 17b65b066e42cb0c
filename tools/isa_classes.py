#!/usr/bin/env python3
"""Per-class instruction histogram of the four role loops of fsk_blk.hip's demod_blk_kernel (VERDICT r03 #1a).

Runs anywhere hipcc is installed (no GPU).  For each of the kernel's per-tile loops -- found by the asynchronous
counter read every role issues at the top of a step (lds_peek4_begin: an asm ds_read_b128) -- walks the HOT path of one
step: from the counter read forward through fall-throughs and unconditional branches, never following a conditional
branch out of line (the rare paths are laid out out of line by __builtin_expect), until the loop's back edge.  Prints,
per role, instructions per tile (16 input samples) by issue class and the class-priced cycles per input sample:

  full   2 cycles   v_fma/mul/add/sub_f32, v_add/sub_u32, v_and/or/xor, v_bitop3, right shifts, v_mov -- VGPR/inline operands only
  half   4 cycles   any vector instruction with an SGPR operand; v_cmp*, v_cndmask, min/max/med3, bfi/bfe/perm/alignbit,
                    left shifts, v_cvt*, v_rndne, v_bcnt, v_pk_*, readlane/readfirstlane, 64-bit integer forms
  trans  8 cycles   v_rcp/rsq/sqrt/sin/cos/exp/log_f32
  salu   2 cycles   s_* except branches, waits, nops
  lds / vmem / branch / wait: counted, priced 0 here (what they cost the issuing wave is in DESIGN.md section 4.1)

(the classes and prices are tools/valu_probe.hip's measurements, profiles/r02_valu_probe_summary.md)

usage: tools/isa_classes.py [--kernel '<false, true, false>'] [--dump] [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize"]

FULL = re.compile(r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|"
                  r"bitop3_b32|lshrrev_b32|ashrrev_i32|mov_b32|not_b32|and_or_b32|or3_b32|xad_u32|add3_u32|xnor_b32|mad_f32|mac_f32|fmaak_f32|fmamk_f32|madak_f32|madmk_f32)")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_"):
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            return "branch"
        if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio")):
            return "wait"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if TRANS.match(op):
            return "trans"
        args = ins[len(op):]
        # an SGPR / VCC / EXEC operand halves the rate of an otherwise full-rate instruction (literal constants too: VOP3
        # cannot encode them, the compiler hoists them into SGPRs -- seen here as s operands)
        sgpr = re.search(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)\b", args) is not None
        if FULL.match(op) and not sgpr:
            return "full"
        return "half"
    return "other"


PRICE = {"full": 2, "half": 4, "trans": 8, "salu": 2, "lds": 0, "vmem": 0, "branch": 0, "wait": 0, "other": 0}


def compile_asm(extra):
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_blk.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "d.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", asm, src],
                       capture_output=True, text=True, check=True)
        return open(asm).read()


FIVE = False     # --five: demod_blk5_kernel (round 6: five waves per group)


def mangled(kernel):
    b = [x.strip() == "true" for x in kernel.strip("<> ").split(",")]
    return ("_ZN3fsk17demod_blk5_kernelILb%dELb%dELb%dEEE" if FIVE else "_ZN3fsk16demod_blk_kernelILb%dELb%dELb%dEEE") % tuple(int(x) for x in b)


def kernel_body(text, kernel):
    m = re.search(r"^(" + re.escape(mangled(kernel)) + r"\w+):[^\n]*\n", text, re.M)
    if not m:
        raise SystemExit("kernel %s not found in the ISA" % kernel)
    return text[m.end():text.index(".Lfunc_end", m.end())].split("\n")


def parse_blocks(body):
    """-> list of blocks {name, lines: [(idx, text)], hdr (loop header name this block lies in, innermost)}"""
    blocks, cur = [], {"name": "entry", "lines": [], "note": ""}
    blocks.append(cur)
    skip_to = None
    for i, l in enumerate(body):
        # a bounded poll (csrc/fsk_wait.h) is ONE asm statement whose tail -- flag the fault word, s_endpgm -- is jumped over by its
        # own s_cbranch_scc1 unless the bound is hit: not part of any path priced here
        if skip_to is not None:
            if l.strip().startswith(skip_to + ":"):
                skip_to = None
            continue
        mj = re.match(r"\s*s_cbranch_scc1\s+(\.Lfsk_spin_ok\d+)", l)
        if mj:
            cur["lines"].append((i, "s_cbranch_scc1 " + mj.group(1)))
            skip_to = mj.group(1)
            continue
        ml = re.match(r"\s*(\.LBB\d+_\d+):(.*)$", l)
        if ml:
            cur = {"name": ml.group(1), "lines": [], "note": ml.group(2)}
            blocks.append(cur)
            continue
        st = l.strip()
        if st.startswith(";") and not st.startswith(";;#") and not cur["lines"]:
            cur["note"] += " " + st
        elif st and not st.startswith(";") and not st.startswith("."):
            cur["lines"].append((i, st.split(";")[0].strip()))
    return blocks


def hot_walk(body, blocks, start_block, start_k):
    """instructions from (block, position) along fall-throughs / unconditional branches until we are back at the start"""
    index = {b["name"]: n for n, b in enumerate(blocks)}
    out, n, k, seen = [], start_block, start_k, 0
    while seen < 4000:
        b = blocks[n]
        jumped = False
        while k < len(b["lines"]):
            i, st = b["lines"][k]
            k += 1
            seen += 1
            if (n, k - 1) == (start_block, start_k) and out:
                return out
            out.append(st)
            mb = re.match(r"s_branch\s+(\.LBB\d+_\d+)", st)
            if mb:
                n, k, jumped = index[mb.group(1)], 0, True
                break
            mc = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", st)
            if mc:
                # a conditional branch back to (or above) the walk's start block that is the last instruction before
                # out-of-line code is the loop's back edge: follow it
                tgt = index[mc.group(1)]
                if _is_back_edge(blocks, n, tgt, start_block):
                    n, k, jumped = tgt, 0, True
                    break
                # a branch over (or back past) a wait loop -- the compiler places the bounded polls' loops in line, behind a test that
                # skips them when the tile is there: the hot path takes it
                ahead = [st2 for bb in [{"lines": b["lines"][k:]}] + blocks[n + 1:(tgt if tgt > n else n + 4)] for _, st2 in bb["lines"]][:30]
                if any("s_sleep" in st2 for st2 in ahead):
                    n, k, jumped = tgt, 0, True
                    break
        if not jumped:
            n, k = n + 1, 0
            if n >= len(blocks):
                return out
        if (n, k) == (start_block, 0) and start_k == 0 and out:
            return out
    return out


def _is_back_edge(blocks, n, tgt, start_block):
    note = blocks[tgt]["note"]
    return "Loop Header" in note and tgt <= start_block <= n


PEEK_REGS = set()


def _is_peek(st):
    return st in PEEK_REGS


def role_of(ins_list):
    text = "\n".join(ins_list)
    if "buffer_load_dwordx4" in text:
        return "wave 4: loads + AGC (five-wave kernel)" if FIVE else "wave 0: loads + AGC + pre-filter + phasors"
    if FIVE and "v_cos_f32" in text:
        return "wave 0: pre-filter + phasors (five-wave kernel)"
    if "v_bcnt_u32_b32" in text:
        return "wave 3: post filter + slicer + correlator + frame FSM (block path)"
    if "v_sqrt_f32" in text:
        return "wave 2: discriminator (atan2 + magnitude)"
    return "wave 1: mixer + I/Q low-pass + pair sums"


def histogram(ins_list):
    h = {}
    for st in ins_list:
        c = classify(st)
        h[c] = h.get(c, 0) + 1
    return h


def main():
    args = sys.argv[1:]
    kernel = "<false, true, false>"
    dump = False
    extra = []
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kernel = args.pop(0)
        elif a == "--five":
            global FIVE
            FIVE = True
            extra.append("-DFSK_BLK_FIVE")     # (the five-wave kernel is a measurement build's: fsk_blk.hip)
        elif a == "--dump":
            dump = True
        else:
            extra.append(a)
    body = kernel_body(compile_asm(extra), kernel)
    blocks = parse_blocks(body)
    sites = []
    for n, b in enumerate(blocks):
        for k, (i, st) in enumerate(b["lines"]):
            if st.startswith("ds_read_b128") and "ASMSTART" in body[i - 1]:
                sites.append((n, k))
                PEEK_REGS.add(st)
    print("# fsk::demod_blk%s_kernel%s: hot path of one step (a tile = 16 input samples) per role loop, by issue class" % ("5" if FIVE else "", kernel))
    print("# classes and prices: see this tool's header; per group-sample = per tile / 16")
    tot = {}
    seen_roles = {}
    for n, k in sites:
        ins = hot_walk(body, blocks, n, k)
        role = role_of(ins)
        h = histogram(ins)
        # a loop unrolled over several tiles (wave 0: three register sets) passes several counter reads per round
        tiles = max(1, sum(1 for st in ins if st.startswith("ds_read_b128") and _is_peek(st)))
        if FIVE and "pre-filter + phasors" in role:
            tiles = max(1, tiles // 2)       # (this part reads both counter quads per tile)
        if tiles > 1:
            h = {c: v / float(tiles) for c, v in h.items()}
        # the compiler clones some loops (the back wave's block loop is peeled): report each site, sum the first per role
        tag = "" if role not in seen_roles else "   (a second copy of this loop: not summed)"
        print("\n== %s%s" % (role, tag))
        print("   block %s, %d instructions per round of %d tile(s)" % (blocks[n]["name"], len(ins), tiles))
        priced = sum(PRICE[c] * v for c, v in h.items())
        for c in ("full", "half", "trans", "salu", "lds", "vmem", "branch", "wait", "other"):
            if h.get(c):
                print("   %-7s %6.1f per tile  %6.2f per sample   x%d = %6.1f cycles per sample" % (c, h[c], h[c] / 16.0, PRICE[c], PRICE[c] * h[c] / 16.0))
        print("   VALU %.1f per tile = %.2f per sample; class-priced %.1f cycles per sample" %
              (h.get("full", 0) + h.get("half", 0) + h.get("trans", 0), (h.get("full", 0) + h.get("half", 0) + h.get("trans", 0)) / 16.0, priced / 16.0))
        halves = {}
        for st in ins:
            if classify(st) == "half":
                op = st.split()[0]
                halves[op] = halves.get(op, 0) + 1.0 / tiles
        print("   half-rate by opcode: " + ", ".join("%s %.4g" % kv for kv in sorted(halves.items(), key=lambda kv: -kv[1])))
        if dump:
            for st in ins:
                print("      %-6s %s" % (classify(st), st))
        if role not in seen_roles:
            seen_roles[role] = True
            for c, v in h.items():
                tot[c] = tot.get(c, 0) + v
    valu = tot.get("full", 0) + tot.get("half", 0) + tot.get("trans", 0)
    print("\n== all four roles, hot paths only (the back wave's per-sample path, set-up and the waits' polls are extra)")
    print("   VALU %.2f per group-sample (full %.2f, half %.2f, trans %.2f); SALU %.2f; LDS %.2f; VMEM %.2f; branches %.2f" %
          (valu / 16.0, tot.get("full", 0) / 16.0, tot.get("half", 0) / 16.0, tot.get("trans", 0) / 16.0, tot.get("salu", 0) / 16.0,
           tot.get("lds", 0) / 16.0, tot.get("vmem", 0) / 16.0, tot.get("branch", 0) / 16.0))
    print("   class-priced: %.1f cycles per group-sample" % (sum(PRICE[c] * v for c, v in tot.items()) / 16.0))


if __name__ == "__main__":
    main()

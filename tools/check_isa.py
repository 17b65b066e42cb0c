#!/usr/bin/env python3
"""Static check of the compiled whole-tile demod kernels of fsk_pipe.hip and fsk_blk.hip (runs anywhere hipcc is installed, no GPU):
the one-wave kernel's tile prefetch uses inline-asm loads whose results are only valid after a hand-placed s_waitcnt, so
the register allocator must never spill inside that kernel (a spill store of an in-flight load result
would save garbage); the one-wave kernel must fit 168 VGPRs (3 waves per SIMD), the two-wave kernel 128."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRA = os.environ.get("FSK_CHECK_ISA_FLAGS", "").split()   # a measurement build's extra hipcc flags (tools/build_variant.sh)


def kernel_resources():
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_pipe.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *EXTRA,
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


def prefetch_register_hazards():
    """The asm-issued tile prefetch lands asynchronously: between its issue and a vmcnt wait that covers it the
    destination VGPRs must not be touched.  Returns a list of violations found in the fast kernels' ISA:
    any instruction after the in-loop prefetch that uses those registers without an `s_waitcnt vmcnt(0)` (the
    epilogue) or the loop-top `s_waitcnt vmcnt(8)` + ds_write (the next iteration) in between."""
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_pipe.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "d.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *EXTRA,
                        "-S", "--cuda-device-only", "-o", asm, src], capture_output=True, text=True, check=True)
        text = open(asm).read()

    def regs_of(line):
        out = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", line):
            out.add(int(m.group(1)))
        return out

    problems = []
    found = list(re.finditer(r"^(_ZN3fsk18demod_fused_kernel\w+):[^\n]*\n", text, re.M))
    if len(found) != 4:
        problems.append(("demod_fused_kernel", "expected 4 kernel bodies in the ISA, found %d" % len(found)))
    for m in found:
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        loads = [i for i, l in enumerate(body) if "buffer_load_dwordx4" in l]
        if len(loads) != 8:
            problems.append((m.group(1), "expected 8 asm prefetch loads, found %d" % len(loads)))
            continue
        dst = set()
        for i in loads[4:]:
            dst |= regs_of(body[i].split(",")[0])
        # loop top: the four ds_write_b128 of the staged tile, preceded by the vmcnt(8) wait
        for i in range(loads[3] + 1, loads[4]):
            if regs_of(body[i]) & dst and "ds_write_b128" in body[i]:
                if not any("s_waitcnt vmcnt(8)" in body[j] for j in range(loads[3], i)):
                    problems.append((m.group(1), "staging before the vmcnt(8) wait: " + body[i].strip()))
                break
        # epilogue = everything after the tile loop's back edge (the last branch to a label above the prefetch)
        labels = {}
        for i, l in enumerate(body):
            ml = re.match(r"\s*(\.LBB\d+_\d+):", l)
            if ml:
                labels[ml.group(1)] = i
        back = None
        for i in range(loads[7] + 1, len(body)):
            mb = re.match(r"\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", body[i])
            if mb and labels.get(mb.group(1), len(body)) < loads[4]:
                back = i
        if back is None:
            problems.append((m.group(1), "tile loop back edge not found"))
            continue
        # inside the loop after the prefetch: nothing may touch the destination registers at all
        for i in range(loads[7] + 1, back):
            line = body[i].strip()
            if line and not line.startswith(";") and regs_of(line) & dst:
                problems.append((m.group(1), "prefetch destination touched inside the loop: " + line))
                break
        waited = False
        for i in range(back + 1, len(body)):
            line = body[i].strip()
            if "s_waitcnt" in line and "vmcnt(0)" in line:
                waited = True
            if line.startswith(";") or not line:
                continue
            if regs_of(line) & dst and not waited:
                problems.append((m.group(1), "prefetch destination touched while the load may be in flight: " + line))
                break
    return problems


def pipe_prefetch_hazards(symbol=r"_ZN3fsk17demod_pipe_kernel"):
    """The two-wave kernel's front wave keeps
    three register sets of asm-issued tile loads in flight (unrolled by three).
    In its tile loop (everything after the prologue's `s_waitcnt vmcnt(0)` up to the epilogue's) a register of a set may
    only be read after the `s_waitcnt vmcnt(N)` placed in front of that set's ds_write_b128 staging, and only by it."""
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_pipe.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "d.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *EXTRA,
                        "-S", "--cuda-device-only", "-o", asm, src], capture_output=True, text=True, check=True)
        text = open(asm).read()

    def regs_of(line):
        out = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", line):
            out.add(int(m.group(1)))
        return out

    problems = []
    found = list(re.finditer(r"^(" + symbol + r"\w+):[^\n]*\n", text, re.M))
    if len(found) != 4:
        problems.append((symbol, "expected 4 kernel bodies in the ISA, found %d" % len(found)))
    for m in found:
        body = [l for l in text[m.end():text.index(".Lfunc_end", m.end())].split("\n")]
        loads = [i for i, l in enumerate(body) if "buffer_load_dwordx4" in l]
        if len(loads) != 24:
            problems.append((m.group(1), "expected 12 prologue + 12 loop prefetch loads, found %d" % len(loads)))
            continue
        loop_loads = loads[12:]
        first, last = loop_loads[0], loop_loads[-1]
        # walk the loop region: from the first staging wait before the first loop load to the last loop load
        start = max(i for i in range(loads[11], first) if "s_waitcnt vmcnt(" in body[i] and "ASMSTART" in body[i - 1])
        pending = {}   # register -> index of the load that targets it
        for i in range(start, len(body)):
            line = body[i].strip()
            if not line or line.startswith(";") or line.startswith("."):
                continue
            if "buffer_load_dwordx4" in line:
                for r in regs_of(line.split(",")[0]):
                    pending[r] = i
                continue
            if line.startswith("s_waitcnt") and "vmcnt(" in line:
                n = int(re.search(r"vmcnt\((\d+)\)", line).group(1))
                # in issue order: all but the n youngest VMEM operations are complete; count loads only (stores of the
                # write-back variant make the real wait stricter, never looser)
                order = sorted(set(pending.values()))
                done = set(order[:max(0, len(order) - (n + 3) // 4 * 4)]) if n else set(order)
                pending = {r: j for r, j in pending.items() if j not in done}
                continue
            if "s_endpgm" in line:
                break
            touched = regs_of(line) & set(pending)
            if touched and i <= last + 400:
                problems.append((m.group(1), "register of an in-flight tile load touched: " + line))
                break
    return problems


def blk_checks():
    """fsk_blk.hip (four waves per group): eight + four kernel bodies (<write-back, uniform, time-sliced>), each within 128
    VGPRs (four workgroups per CU).  The kernel is built with __launch_bounds__(256, 4) and spills in its set-up and in the
    per-sample slow path; the loops that run per tile must not: every block of the innermost loop around each of the four
    parts' asynchronous counter read (lds_peek4_begin: an asm ds_read_b128) has to be free of scratch instructions.  And
    the registers that read lands in must not be touched before a wait that covers it (lgkmcnt(0))."""
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_blk.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "d.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *EXTRA,
                        "-S", "--cuda-device-only", "-o", asm, src], capture_output=True, text=True, check=True)
        text = open(asm).read()

    def regs_of(line):
        out = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", line):
            out.add(int(m.group(1)))
        return out

    problems = []
    # (round 4: plus the four bodies of demod_blk_kernel_r<write-back, time-sliced>, the kernel whose block path takes resets)
    # (round 5: plus the four of demod_blk_kernel_rp<write-back, time-sliced>: the same for per-stream tone pairs)
    # (round 6: a -DFSK_BLK_FIVE measurement build adds the eight of demod_blk5_kernel, five waves per group, compiled for 96 VGPRs; run
    # over such a build this check flags scratch accesses in the rare paths inside its per-tile loops and, in the write-back
    # instantiations, a dead component of the second counter quad re-used before the wait -- one of the reasons it is not shipped)
    found = list(re.finditer(r"^(_ZN3fsk1[6789]demod_blk5?_kernel(?:_rp?)?I\w+):[^\n]*\n", text, re.M))
    if len(found) != 16:     # (24 in a -DFSK_BLK_FIVE measurement build)
        problems.append(("demod_blk_kernel", "expected 8 + 4 + 4 kernel bodies in the ISA, found %d" % len(found)))
    for name, n in re.findall(r"\.name:\s+(_ZN3fsk1[6789]demod_blk5?_kernel(?:_rp?)?I\w+)\s*\n(?:[^\n]*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        if int(n) > (96 if "demod_blk5_kernel" in name else 128):
            problems.append((name, "%s VGPRs: more than four workgroups per CU allow" % n))
    for m in found:
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        # basic blocks with the innermost loop each belongs to (the asm printer's comments on the label)
        blocks, cur = [], None
        for i, l in enumerate(body):
            ml = re.match(r"\s*(\.LBB\d+_\d+):(.*)$", l)
            if ml:
                cur = {"name": ml.group(1), "start": i, "lines": [], "note": ml.group(2)}
                blocks.append(cur)
                continue
            if cur is None:
                continue
            st = l.strip()
            if st.startswith(";") and not cur["lines"]:
                cur["note"] += " " + st
            elif st and not st.startswith(";") and not st.startswith("."):
                cur["lines"].append((i, st))
        for b in blocks:
            mh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", b["note"])
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", b["note"])
            b["loop"] = (b["name"][2:], int(mh.group(1))) if mh else ((mi.group(1), int(mi.group(2))) if mi else None)
        peeks = 0
        for b in blocks:
            for k, (i, st) in enumerate(b["lines"]):
                if not (st.startswith("ds_read_b128") and "ASMSTART" in body[i - 1]):
                    continue
                peeks += 1
                if b["loop"] is None:
                    problems.append((m.group(1), "counter read outside any loop: " + st))
                    continue
                for o in blocks:
                    # (demod_blk_kernel_r is for calls whose tiles mostly leave the fast loop: its guard is blk_medium's
                    # block below; its time-sliced instantiation reloads one register per fast tile)
                    if "demod_blk_kernel_r" in m.group(1):
                        break
                    if o["loop"] == b["loop"] and any(x.startswith("scratch_") for _, x in o["lines"]):
                        problems.append((m.group(1), "scratch access in a per-tile loop, block %s" % o["name"]))
                dst = regs_of(st.split(",")[0])
                for j in range(i + 1, min(len(body), i + 600)):
                    line = body[j].strip()
                    if not line or line.startswith(";") or line.startswith("."):
                        continue
                    if "s_waitcnt" in line and "lgkmcnt(0)" in line:
                        break
                    if regs_of(line) & dst:
                        problems.append((m.group(1), "counter read's registers touched before a wait: " + line))
                        break
        if "demod_blk_kernel_r" in m.group(1):
            # blk_medium: one straight-line basic block of ~1 400 instructions, hand-kept free of scratch traffic (its
            # inputs are read a sample pair ahead, its event captures pinned, the entry state parked in the engine's stash)
            big = [b for b in blocks if len(b["lines"]) > 1000]
            if len(big) != 1:
                problems.append((m.group(1), "expected blk_medium as ONE basic block of > 1000 instructions, found %d" % len(big)))
            # (the time-sliced instantiations carry the queue's bookkeeping too and reload a piece of the lane state at the
            # block's top: three loads and a store, one round trip per block, tolerated -- one load more since the round-5 lag
            # constants, two of which no longer fit an inline operand; the plain ones must have none)
            allowed = 5 if re.search(r"kernel_rILb[01]ELb1E", m.group(1)) else 0
            if "demod_blk_kernel_rp" in m.group(1):
                allowed = 24      # (per-lane NCO phasors and lastPhase values on top of the uniform kernel's block: a few spills, measured)
            for b in big:
                n_scr = sum(1 for _, x in b["lines"] if x.startswith("scratch_"))
                if n_scr > allowed:
                    problems.append((m.group(1), "%d scratch accesses inside blk_medium's block %s (allowed: %d)" % (n_scr, b["name"], allowed)))
        if peeks < (5 if "demod_blk5_kernel" in m.group(1) else 4):   # (the compiler duplicates the back wave's block loop: six sites in the current build)
            problems.append((m.group(1), "expected at least 4 asynchronous counter reads (one per part), found %d" % peeks))
    return problems


def blk6_resources():
    """fsk_blk6.hip (seven waves per group, one workgroup per CU): eight kernel bodies <write-back, group width>, each within the
    256 VGPRs two waves per SIMD leave a wave, none with scratch memory (its frame wave keeps a register copy of the lane state
    where the four-wave kernel parks it in memory)."""
    src = os.path.join(ROOT, "webaudio_modem_amd", "csrc", "fsk_blk6.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *EXTRA,
               "-c", src, "-o", os.path.join(tmp, "d.o"), "-Rpass-analysis=kernel-resource-usage"]
        out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
        if m and cur:
            res[cur][m.group(1)] = int(m.group(2))
    problems = []
    k6 = {k: v for k, v in res.items() if "demod_blk6_kernel" in k}
    if len(k6) != 16:     # <write-back, streams per workgroup, uniform> (round 6: per-stream tone pairs too)
        problems.append(("demod_blk6_kernel", "expected 16 kernel bodies, found %d" % len(k6)))
    for k, v in k6.items():
        if v.get("VGPRs", 999) > 256 or v.get("ScratchSize [bytes/lane]", 1) != 0 or v.get("VGPRs Spill", 1) != 0:
            problems.append((k, "resources %s" % v))
    return problems


if __name__ == "__main__":
    for name, what in prefetch_register_hazards() + pipe_prefetch_hazards() + blk_checks():
        print("HAZARD", name[:50], what)
        sys.exit(2)
    r = kernel_resources()
    bad = 0
    for name, v in r.items():
        if "demod_fused_kernel" in name:
            print(name[:60], v)
            if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0):
                bad += 1
    sys.exit(1 if bad else 0)

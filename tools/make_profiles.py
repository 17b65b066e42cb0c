#!/usr/bin/env python3
"""gpurun_out/final/ (tools/final_run.sh on the GPU box) -> profiles/<round>_*: raw PMC / stats files copied, and the two
JSON summaries bench.py reads (issue: instructions per 64-stream group and input sample + the class-priced cycles of
tools/isa_classes.py; traffic: HBM bytes per input sample with the guide's FETCH_SIZE correction).
usage: tools/make_profiles.py r04"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
KERNEL = "fsk::demod_blk_kernel<false, true, false>"
S, N = 65536, 48000          # tools/pmc.sh passes: --seconds 1 at the default 65 536 streams
groups = S // 64


def counters(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(.*) dispatches (\d+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {"dispatches": int(m.group(2))}
            continue
        m = re.match(r"\s+(\w+)\s+total\s+(\S+)\s+per-dispatch\s+(\S+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(3))
    for k, v in out.items():
        if "demod_blk_kernel<false, true, false>" in k:
            return v
    raise SystemExit("no demod_blk_kernel line in " + path)


for name in ("pmc_insts.txt", "pmc_fetch.txt", "pmc_write.txt", "pmc_clock.txt", "kernel_stats.csv", "deviation.txt"):
    src = os.path.join(F, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_final_%s" % (rnd, name)))
for name in ("bench", "bench_c2", "bench_c4", "bench_c5", "bench_idle", "bench_idle4", "bench_mod", "bench_mod_f64", "bench_c1x", "bench_f64", "bench_8192", "bench_2048", "bench_4096", "bench_16384", "bench_staggered", "bench_c4_4096"):
    src = os.path.join(F, name + ".txt")
    if os.path.exists(src):
        lines = [l for l in open(src) if l.startswith("{")]
        if lines:
            open(os.path.join(P, "%s_%s_line.json" % (rnd, name)), "w").write(lines[-1])

ins = counters(os.path.join(F, "pmc_insts.txt"))
per = lambda c: ins[c] / (groups * N)
valu, salu, lds = per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS")
vmem = per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR")
isa = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_classes.py")], capture_output=True, text=True, check=True).stdout
open(os.path.join(P, "%s_isa_classes.txt" % rnd), "w").write(isa)
m = re.search(r"VALU ([\d.]+) per group-sample \(full ([\d.]+), half ([\d.]+), trans ([\d.]+)\); SALU ([\d.]+)", isa)
hot_valu, full, half, trans, hot_salu = [float(v) for v in m.groups()]
hot_priced_valu = 2 * full + 4 * half + 8 * trans
priced_valu = hot_priced_valu * valu / hot_valu
issue = {
    "kernel": KERNEL,
    "source": "profiles/%s_final_pmc_insts.txt (rocprofv3 --pmc SQ_INSTS_*, %d streams x %d samples per launch: 4 waves per 64-stream group)" % (rnd, S, N),
    "insts_per_group_sample": {"valu": round(valu, 2), "salu": round(salu, 2), "lds": round(lds, 2), "vmem": round(vmem, 2)},
    "hot_path_isa": {"valu": hot_valu, "full_rate": full, "half_rate": half, "transcendental": trans, "salu": hot_salu,
                     "class_priced_valu_cycles": round(hot_priced_valu, 1), "source": "profiles/%s_isa_classes.txt (tools/isa_classes.py)" % rnd},
    "class_priced_cycles_per_group_sample": round(priced_valu, 1),
    "class_priced_source": "profiles/%s_isa_classes.txt: the four role loops' hot paths by issue class (full rate 2, half rate 4, transcendental 8 "
                           "cycles per wave64 instruction) = %.1f cycles for %.2f vector instructions, scaled to the %.2f the PMC pass counts "
                           "(per-sample path, set-up and polls take the same mix); scalar instructions (%.2f per group-sample, 2 cycles each "
                           "if they took vector issue slots) not included" % (rnd, hot_priced_valu, hot_valu, valu, salu),
    "sq_busy_cycles_per_launch": ins.get("SQ_BUSY_CYCLES"),
    "note": "per 64-stream group and input sample, summed over the group's four waves.  Four groups share a CU at 65536 streams, so a SIMD "
            "issues one group's worth of instructions per input sample.",
}
json.dump(issue, open(os.path.join(P, "%s_issue.json" % rnd), "w"), indent=1)
fe, wr = counters(os.path.join(F, "pmc_fetch.txt"))["FETCH_SIZE"], counters(os.path.join(F, "pmc_write.txt"))["WRITE_SIZE"]
rd_b, wr_b = fe * 1024 * 2 / (S * N), wr * 1024 / (S * N)
traffic = {
    "kernel": KERNEL,
    "workload": "BASELINE config #3 shape, %d streams x %d samples per launch (tools/pmc.sh via tools/final_run.sh, separate --pmc passes, no trace domains)" % (S, N),
    "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr,
    "fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE = TCC_EA0_RDREQ x 64 B counts 128-B requests at 64 B)",
    "hbm_read_bytes_per_input_sample": rd_b, "hbm_write_bytes_per_input_sample": wr_b, "hbm_bytes_per_input_sample": rd_b + wr_b,
    "algorithmic_bytes_per_input_sample": 4.0,
    "raw": ["profiles/%s_final_pmc_fetch.txt" % rnd, "profiles/%s_final_pmc_write.txt" % rnd],
    "note": "writes: the reference's syncAmplitudeBuffer (fsk.ts:150), one f32 per decimated sample = 2 B per input sample, plus output bytes / "
            "counts / state; reads: the input (4 B) plus amplitude-column reads at sync time and state",
}
json.dump(traffic, open(os.path.join(P, "%s_traffic.json" % rnd), "w"), indent=1)
print(json.dumps(issue["insts_per_group_sample"]), issue["class_priced_cycles_per_group_sample"], round(rd_b + wr_b, 3))

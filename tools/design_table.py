#!/usr/bin/env python3
"""DESIGN.md section 5's table and section 6's strong-scaling estimate from profiles/<round>_bench_*_line.json (tools/make_profiles.py).
usage: tools/design_table.py r06   -- rewrites the rows between the table header and the "All of one pass" line in place"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
def L(name):
    return json.load(open(os.path.join(ROOT, "profiles", "%s_bench_%s_line.json" % (rnd, name) if name else "%s_bench_line.json" % rnd)))
m = L(None); c = m["config"]
v = lambda d: "{:,}".format(int(round(d))).replace(",", " ")
f = lambda x: "%.1f %%" % (100 * x)
rows = []
row = lambda *a: rows.append("| " + " | ".join(a) + " |")
row("**BASELINE config #3 as written**, 65 536 × 480 000", "`demod_blk_kernel<false,true,false>`", "**%s** (547–551 k on this round's boxes; r05: 554 k — the NaN-proof slicer is one more instruction per decimated sample)" % v(m["value"]), v(m["roofline"]["achieved"]), "**%s**" % f(m["roofline"]["frac"]), "175 / 175")
row("the same, **exact path** (`exact` in the line)", "`demod_kernel<double,…>`", "**%s** (173–180 k over the boxes)" % v(m["exact"]["value"]), v(m["exact"]["value"] * 4 / 1e3), f(m["exact"]["frac"]), "175 / 175")
for k, d in (("staggered", "the same batch, **frames that do not line up** (`config.staggered`: lead-ins up to one frame length)"), ("idle", "the same batch as an **idle bank** (`config.idle`: one frame, then a floor 30 dB under it)")):
    row(d, "`demod_blk_kernel_r<false,false>`", v(c[k]["Msamples_per_s"]), v(c[k]["Msamples_per_s"] * 4 / 1e3), f(c[k]["frac_of_hbm_peak"]), "%d / %d" % (c[k]["streams_byte_identical"], c[k]["streams_checked"]))
for n, desc, k in (("8192", "one GPU's share of config #3 over eight: 8 192", "`demod_blk6_kernel<false,32>`"), ("4096", "4 096 (config #2's size at config #3's parameters)", "`demod_blk6_kernel<false,16>`"),
                   ("2048", "one GPU's share of config #5 over eight: 2 048", "`demod_blk6_kernel<false,8>`"), ("16384", "16 384", "`demod_blk6_kernel<false,64>`"),
                   ("c2", "config #2 (300 baud Bell-103), 4 096 × 480 000", "`demod_blk6_kernel<false,16>`"), ("c5", "config #5 (10 dB, frames from `fskhip_modulate_device`), 16 384", "`demod_blk6_kernel<false,64>`")):
    d = L(n); row(desc, k, "**%s**" % v(d["value"]), v(d["roofline"]["achieved"]), f(d["roofline"]["frac"]), "175 / 175")
d = L("c4"); row("config #4 (per-stream tone pairs), 32 768", "`demod_blk_kernel<false,false,false>`", v(d["value"]), v(d["roofline"]["achieved"]), f(d["roofline"]["frac"]), "175 / 175")
q = c["c4_per_gpu_share"]; row("**one GPU's share of config #4 over eight: 4 096, per-stream pairs, 300 baud** (`config.c4_per_gpu_share`)", "**`demod_blk6_kernel<false,16,false>`** (round 6)", "**%s** (16-byte payloads; 96.8 k on 32-byte ones; four waves: 63–69 k)" % v(q["Msamples_per_s"]), v(q["Msamples_per_s"] * 4 / 1e3), f(q["frac_of_hbm_peak"]), "%d / %d" % (q["streams_byte_identical"], q["streams_checked"]))
d = L("idle4"); row("idle bank, per-stream tone pairs (`--workload idle4`)", "`demod_blk_kernel_rp<false,false>`", v(d["value"]), v(d["roofline"]["achieved"]), f(d["roofline"]["frac"]), "87 / 87")
d = L("c1x"); row("a bank that never syncs (`--workload c1x`)", "`<false,true,true>`", v(d["value"]), v(d["roofline"]["achieved"]), f(d["roofline"]["frac"]), "175 / 175")
q = c["large_batch"]; row("large batch, 262 144 × 24 000", "`<false,true,true>`", v(q["Msamples_per_s"]), v(q["Msamples_per_s"] * 4 / 1e3), f(q["frac_of_hbm_peak"]), "")
d = L("mod"); row("**modulateData**, config #5's TX leg, 16 384 × 11 frames (`--workload mod`)", "`modulate_wide_kernel<false>`", "**%s**" % v(d["value"]), v(d["roofline"]["achieved"]) + " written", f(d["roofline"]["frac"]), "0 / 2.67 M samples differ")
d = L("mod_f64"); row("the same, fp64 engine", "`modulate_wide_kernel<true>`", v(d["value"]), v(d["roofline"]["achieved"]), f(d["roofline"]["frac"]), "bit-identical")
i = c["iir"]; row("**batched `IIRFilter`**, order 2, 65 536 × 48 000 (`config.iir`; 8 B per sample)", "`iir_kernel` (f32 / f64)", "%s / %s" % (v(i["65536_f32"]["Msamples_per_s"]), v(i["65536_f64"]["Msamples_per_s"])), "%s / %s" % (v(i["65536_f32"]["GB_per_s"]), v(i["65536_f64"]["GB_per_s"])), "**%.1f / %.1f %%**" % (100 * i["65536_f32"]["frac_of_hbm_peak"], 100 * i["65536_f64"]["frac_of_hbm_peak"]), "bit-identical to the reference runs (`-m gpu`); 0.62–0.72 over this round's boxes")
row("the same, 16 384 streams (one wave per CU)", "", "%s / %s" % (v(i["16384_f32"]["Msamples_per_s"]), v(i["16384_f64"]["Msamples_per_s"])), "", "%.1f / %.1f %%" % (100 * i["16384_f32"]["frac_of_hbm_peak"], 100 * i["16384_f64"]["frac_of_hbm_peak"]), "")
q = c["pcie_inclusive"]; row("PCIe-inclusive (`fskhip_demodulate_host`, 16 384 × 48 000)", "", v(q["Msamples_per_s"]), "%.1f of input" % q["GB_per_s_of_input"], "", "never the headline")
row("CPU oracle, one core of %d" % m["cpu_baseline"]["host_cpus"], "", "%.1f" % m["cpu_baseline"]["value"], "", "", "")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| shape | kernel | Msamples/s |")
a = s.index("\n", s.index("\n", a) + 1) + 1          # behind the header's separator row
b = s.index("\nAll of one pass on one box")
s = s[:a] + "\n".join(rows) + "\n" + s[b:]
est = ("The strong-scaling estimate from the N = 1 line (`config.per_gpu_share`, `config.c4_per_gpu_share`): 8 × %.1f = %s Gsamples/s for\n"
       "config #3 over eight GPUs (%.1f× one GPU's %s), 8 × %.1f = %s for config #5 (1.3× the %s one GPU gives all 16 384), 8 × %.1f = %s\n"
       "for config #4 (%.1f× the %s one GPU gives all 32 768: the per-stream seven-wave kernel of round 6).") % (
    c["per_gpu_share"]["8192"]["Msamples_per_s"] / 1e3, v(8 * c["per_gpu_share"]["8192"]["Msamples_per_s"] / 1e3), 8 * c["per_gpu_share"]["8192"]["Msamples_per_s"] / m["value"], v(m["value"] / 1e3),
    c["per_gpu_share"]["2048"]["Msamples_per_s"] / 1e3, v(8 * c["per_gpu_share"]["2048"]["Msamples_per_s"] / 1e3), v(L("c5")["value"] / 1e3),
    c["c4_per_gpu_share"]["Msamples_per_s"] / 1e3, v(8 * c["c4_per_gpu_share"]["Msamples_per_s"] / 1e3), 8 * c["c4_per_gpu_share"]["Msamples_per_s"] / L("c4")["value"], v(L("c4")["value"] / 1e3))
s = re.sub(r"The strong-scaling estimate from the N = 1 line.*?kernel of round 6\)\.", lambda _m: est, s, flags=re.S)
open(p, "w").write(s)
print("DESIGN.md: %d rows" % len(rows))

#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s demodulated by the fused HIP FSK demodulator.

One "step" = one pass of the hot path (fskhip_demodulate_device: AGC -> band-pass -> I/Q mix -> low-pass ->
discriminator -> slicer -> sync -> UART framing) over ONE batch of synthetic 48 kHz streams already resident in HBM.

Default workload = BASELINE.json's metric configuration, configs[2] as SURVEY.md section 8 defines it (C3):
65 536 Bell-202 1200-baud streams x 480 000 samples (10 s) per GPU = 126 GB resident.  Streams shard across GPUs with
no collective:
  weak scaling (default)          every rank holds its own --streams streams
  strong scaling (--total-streams T)  T streams split over the ranks with sharding.stream_shard (C3: 8 192 per GPU at 8)

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (driver contract) with the extra objects
  roofline      achieved algorithmic HBM GB/s of the demod kernel (4 B per input sample) from HIP events
                around every launch on the launch stream, against the 8 TB/s peak; `kernel` is what the library says it
                launched (fskhip_last_kernel), not a literal
  cpu_baseline  the CPU oracle (scalar fp64 C port of the reference, oracle/) timed on one host core on a bounded
                sample of the same buffers; doubles as a parity check of that sample (GPU bytes == oracle bytes): a
                mismatch sets parity_ok false and the exit code to 3
and, at N = 1, side measurements in `config` (never the headline): the fp64 parity path on the same shape, a large
batch (262 144 streams), and the PCIe-inclusive rate through fskhip_demodulate_host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)

WORKLOADS = {
    # BASELINE.json configs[2] (the one the metric is quoted on): Bell-202, 1200 baud
    "c3": dict(cfg=dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200), payload=100, snr=None, num="3",
               desc="Bell-202 1200 baud (1200/2200 Hz) @48 kHz"),
    # BASELINE.json configs[1]: 300 baud V.21 tones in the polarity the reference decodes
    "c2": dict(cfg=dict(baudRate=300, markFrequency=1070, spaceFrequency=1270), payload=32, snr=None, num="2",
               desc="V.21 300 baud (1070/1270 Hz) @48 kHz"),
    # BASELINE.json configs[4]: Bell-202 through AWGN at 10 dB (the demodulation half of the round trip)
    "c5": dict(cfg=dict(baudRate=1200, markFrequency=1200, spaceFrequency=2200), payload=100, snr=10.0, num="5",
               desc="Bell-202 1200 baud @48 kHz + AWGN 10 dB"),
    # BASELINE.json configs[3]: per-stream tone pairs (mark_s = 1000 + 10 (s mod 100), space_s = mark_s + 200), 300 baud
    "c4": dict(cfg=dict(baudRate=300, markFrequency=1000, spaceFrequency=1200), payload=16, snr=None, num="4",
               desc="300 baud, per-stream mark/space tone pairs @48 kHz", per_stream=True),
    "default": dict(cfg=dict(), payload=100, snr=None, num="-", desc="default 1650/1850 Hz 1200 baud @48 kHz"),
}


def timed_steps(torch, eng, step, steps):
    """kernel-only time of `steps` calls: HIP events recorded by the library around every launch"""
    torch.cuda.synchronize()
    eng.timing_begin()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return eng.timing_end()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (weak scaling)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong scaling: this many streams in total, sharded over the ranks (0 = weak scaling)")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per stream per step")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (fp64 path, large batch, PCIe)")
    ap.add_argument("--probe-reads", type=int, default=0,
                    help="also launch the read-pattern probe kernel this many times (FETCH_SIZE calibration)")
    ap.add_argument("--pitch-pad", type=int, default=0, help="extra floats of row pitch (experiments)")
    ap.add_argument("--snr-db", type=float, default=None, help="add AWGN at this SNR (overrides the workload's)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import numpy as np
    import torch
    import __graft_entry__ as ge

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    # one rank runs make (a no-op when the shipped libraries are current); the others must not race it in the same tree
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    import webaudio_modem_amd as wm
    from webaudio_modem_amd.sharding import max_over_ranks, stream_shard

    wl = WORKLOADS[args.workload]
    cfg = wl["cfg"]
    snr = args.snr_db if args.snr_db is not None else wl["snr"]
    if args.total_streams:
        first_stream, S = stream_shard(args.total_streams, rank, world)
        total_streams = args.total_streams
        scaling = "strong"
    else:
        S = args.streams
        first_stream = rank * S
        total_streams = S * world
        scaling = "weak"
    sr = 48000
    N = int(round(args.seconds * sr))
    N = (N + 31) // 32 * 32
    pitch = (N + 63) // 64 * 64 + args.pitch_pad  # 256-B row pitch
    prec = wm.PRECISION_F32 if args.precision == "f32" else wm.PRECISION_F64
    spb = sr // int(cfg.get("baudRate", 1200))
    seed = 0xF5C0DE + 0x1000 * rank

    if wl.get("per_stream"):   # config #4: every stream its own tone pair (per-stream constants in the kernels)
        cfgs = [dict(cfg, markFrequency=1000 + 10 * ((first_stream + s) % 100), spaceFrequency=1200 + 10 * ((first_stream + s) % 100))
                for s in range(S)]
    else:
        cfgs = None
    eng = wm.FSKEngine(S, cfgs if cfgs is not None else cfg, device=local_rank, precision=prec)
    stream = torch.cuda.current_stream().cuda_stream
    x = torch.empty((S, pitch), dtype=torch.float32, device="cuda")
    out_pitch = eng.max_bytes(N)
    out = torch.empty((S, out_pitch), dtype=torch.uint8, device="cuda")
    counts = torch.empty(S, dtype=torch.int32, device="cuda")
    eod = torch.empty(S, dtype=torch.int32, device="cuda")
    eng.synth_device(x.data_ptr(), N, pitch, wl["payload"], seed, 10 * spb, 0.1, 1.0, stream)
    if snr is not None:
        eng.add_awgn_device(x.data_ptr(), N, pitch, snr, seed ^ 0xA36, stream)
    torch.cuda.synchronize()

    def step():
        eng.demodulate_device(x.data_ptr(), N, pitch, out.data_ptr(), out_pitch, counts.data_ptr(), eod.data_ptr(),
                              0, stream)

    # ---- first pass doubles as the parity sample: copy its outputs before state moves on -------
    step()
    torch.cuda.synchronize()
    kernel_name = eng.last_kernel()
    first_counts = counts.cpu().numpy().astype(np.int64)
    cpu_obj = None
    parity_ok = True
    if rank == 0 and args.cpu_seconds > 0:
        est_rate = 7.0e6  # oracle samples/s/core
        n_cpu_streams = int(max(1, min(S, args.cpu_seconds * est_rate // N)))
        # a strided sample over the whole batch (first and last groups included)
        rows = np.unique(np.linspace(0, S - 1, n_cpu_streams).astype(np.int64))
        idx = torch.as_tensor(rows, device="cuda")
        xs = x.index_select(0, idx)[:, :N].cpu().numpy()
        gpu_bytes = out.index_select(0, idx).cpu().numpy()
        from oracle import pyoracle as po
        t0 = time.perf_counter()
        mism = 0
        for j, s in enumerate(rows):
            o = po.OracleCore(cfgs[int(s)] if cfgs is not None else cfg)
            ob, _ = o.demodulate(xs[j])
            gb = gpu_bytes[j, :first_counts[s]].tobytes()
            if ob != gb:
                mism += 1
        dt = time.perf_counter() - t0
        parity_ok = mism == 0
        cpu_obj = {
            "value": round(len(rows) * N / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d streams (evenly strided over the rank-0 batch) x %d samples, oracle/fsk_oracle.c (scalar fp64 C "
                      "port of the reference, -O2, 1 thread)" % (len(rows), N),
            "parity_ok": parity_ok, "streams_byte_identical": int(len(rows) - mism), "streams_checked": int(len(rows)),
            "host_cpus": os.cpu_count(),
        }

    if args.probe_reads:
        # counter calibration aid: the same buffer streamed with the kernel's read pattern and nothing else
        for _ in range(args.probe_reads):
            eng.probe_read_device(x.data_ptr(), N, pitch, stream)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = eng.timing_end()
    elapsed = max_over_ranks(elapsed, dist, "cuda")

    # ---- side measurements (rank 0, one GPU): reported in `config`, never as `value` ------------------------------
    side = {}
    if rank == 0 and world == 1 and not args.no_side and args.precision == "f32" and cfgs is None:
        k_side = max(2, min(args.steps, 4))
        # (1) what exactness costs: the fp64 parity path (op for op with the reference) on the same batch, 1/10 of the length
        try:
            n64 = max(1024, (N // 10) // 32 * 32)
            e64 = wm.FSKEngine(S, cfg, device=local_rank, precision=wm.PRECISION_F64)

            def step64():
                e64.demodulate_device(x.data_ptr(), n64, pitch, out.data_ptr(), out_pitch, counts.data_ptr(), eod.data_ptr(), 0, stream)
            step64()
            nl, ms = timed_steps(torch, e64, step64, k_side)
            r = S * n64 * nl / (ms / 1e3) / 1e6
            side["f64_parity_path"] = {"streams": S, "samples_per_stream": n64, "kernel": e64.last_kernel(),
                                       "Msamples_per_s": round(r, 1), "frac_of_hbm_peak": round(r * 4 / 1e3 / HBM_PEAK_GBS, 4)}
            e64.close()
        except Exception as ex:  # a side measurement must not take the headline down
            side["f64_parity_path"] = {"error": str(ex)}
        # (2) a batch large enough to give every SIMD four waves with one wave per 64-stream group
        try:
            S2, n2 = 262144, 24000 // 32 * 32
            p2 = (n2 + 63) // 64 * 64
            e2 = wm.FSKEngine(S2, cfg, device=local_rank, precision=prec)
            x2 = torch.empty((S2, p2), dtype=torch.float32, device="cuda")
            o2 = torch.empty((S2, e2.max_bytes(n2)), dtype=torch.uint8, device="cuda")
            c2 = torch.empty(S2, dtype=torch.int32, device="cuda")
            e2.synth_device(x2.data_ptr(), n2, p2, wl["payload"], seed, 10 * spb, 0.1, 1.0, stream)
            if snr is not None:
                e2.add_awgn_device(x2.data_ptr(), n2, p2, snr, seed ^ 0xA36, stream)

            def step2():
                e2.demodulate_device(x2.data_ptr(), n2, p2, o2.data_ptr(), o2.shape[1], c2.data_ptr(), 0, 0, stream)
            step2()
            nl, ms = timed_steps(torch, e2, step2, k_side)
            r = S2 * n2 * nl / (ms / 1e3) / 1e6
            side["large_batch"] = {"streams": S2, "samples_per_stream": n2, "kernel": e2.last_kernel(),
                                   "Msamples_per_s": round(r, 1), "frac_of_hbm_peak": round(r * 4 / 1e3 / HBM_PEAK_GBS, 4)}
            e2.close()
            del x2, o2, c2
        except Exception as ex:
            side["large_batch"] = {"error": str(ex)}
        # (3) PCIe-inclusive: host buffers through fskhip_demodulate_host (H2D + kernel + D2H; calls longer than ~1.5 time
        # slabs are a two-buffer pipeline, include/fskhip.h), from pageable and from page-locked memory
        try:
            S3, n3 = 16384, 48000
            e3 = wm.FSKEngine(S3, cfg, device=local_rank, precision=prec)
            h = np.ascontiguousarray(x[:S3, :n3].cpu().numpy())
            hp = wm.pinned_empty((S3, n3), np.float32)
            hp[:] = h
            rates = {}
            for name, buf in (("pageable", h), ("pinned", hp)):
                e3.reset()
                e3.demodulate_data(buf)  # first call allocates the staging buffers
                best = 1e9
                for _ in range(2):
                    e3.reset()
                    t1 = time.perf_counter()
                    e3.demodulate_data(buf)
                    best = min(best, time.perf_counter() - t1)
                rates[name] = round(S3 * n3 / best / 1e6, 1)
            side["pcie_inclusive"] = {"streams": S3, "samples_per_stream": n3, "Msamples_per_s": rates["pinned"],
                                      "Msamples_per_s_pageable": rates["pageable"],
                                      "GB_per_s_of_input": round(rates["pinned"] * 4 / 1e3, 1),
                                      "note": "fskhip_demodulate_host: H2D + kernel + D2H pipelined over time slabs, wall clock "
                                              "incl. the Python wrapper; never the headline"}
            e3.close()
            del hp
        except Exception as ex:
            side["pcie_inclusive"] = {"error": str(ex)}

    total_samples = float(total_streams) * N * args.steps if args.total_streams else float(S) * N * args.steps * world
    value = total_samples / elapsed / 1e6
    alg_bytes_per_launch = 4.0 * S * N  # 4 B read per input sample
    avg_kernel_s = kernel_ms / 1e3 / max(1, n_launch)
    achieved = alg_bytes_per_launch / avg_kernel_s / 1e9
    decoded = int(first_counts.sum())
    # HBM traffic per launch: only from a committed PMC pass of THIS round's kernel on this workload (tools/pmc.sh)
    traffic, traffic_src = None, None
    tr_path = os.path.join(ROOT, "profiles", "r02_traffic.json")
    if args.workload == "c3" and args.precision == "f32" and snr is None and os.path.exists(tr_path):
        with open(tr_path) as fh:
            tj = json.load(fh)
        if tj.get("kernel", "") and tj["kernel"].split("<")[0] == kernel_name.split("<")[0]:
            traffic = round(tj["hbm_bytes_per_input_sample"] * S * N / avg_kernel_s / 1e9, 1)
            traffic_src = "profiles/r02_traffic.json (separate rocprofv3 --pmc passes of this kernel; bytes per input sample x this run's rate, not counters of this run)"

    # the ceiling that actually binds (VERDICT r01 #2b): instruction issue.  The instruction mix and its pricing are a
    # committed profile of this kernel (like the traffic figure); the nanoseconds per sample are this run's.
    issue = None
    is_path = os.path.join(ROOT, "profiles", "r02_issue.json")
    if os.path.exists(is_path):
        with open(is_path) as fh:
            ij = json.load(fh)
        if ij.get("kernel", "").split("<")[0] == kernel_name.split("<")[0]:
            ns = avg_kernel_s / N * 1e9
            issue = {"insts_per_group_sample": ij["insts_per_group_sample"],
                     "modelled_cycles_per_group_sample": ij["modelled_cycles_per_group_sample"],
                     "measured_ns_per_sample": round(ns, 1),
                     "measured_cycles_per_sample_at_2.1_to_2.4_GHz": [round(ns * 2.1), round(ns * 2.4)],
                     "source": ij["source"] + "; nanoseconds per sample from this run"}

    if rank == 0:
        line = {
            "metric": "Msamples/s demodulated (fused I/Q demod kernel, inputs resident in HBM)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {
                "workload": "BASELINE config #%s: %d streams%s x %d samples (%.2f s) %s, back-to-back %d-byte "
                            "frames, random lead-in and amplitude" % (
                                wl["num"], total_streams if args.total_streams else S,
                                " in total" if args.total_streams else "/GPU", N, N / sr, wl["desc"], wl["payload"]),
                "streams_per_gpu": S, "total_streams": total_streams, "samples_per_stream": N, "row_pitch_floats": pitch,
                "resident_input_GB_per_gpu": round(S * pitch * 4 / 1e9, 2),
                "parallelism": "streams sharded across %d GPU(s) (%s scaling), no collective" % (world, scaling),
                "decoded_bytes_first_pass_rank0": decoded,
                **side,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kernel_name, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": n_launch,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "issue_ceiling": issue,
                "issue_ceiling_note": "the kernel is instruction-issue bound, not HBM bound: see DESIGN.md section 5 "
                                      "(profiles/r02_valu_probe_summary.md for the per-instruction costs it is priced with)",
            },
            "cpu_baseline": cpu_obj,
        }
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

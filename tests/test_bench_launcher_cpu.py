"""bench.py --gpus N must start its own N ranks when no launcher has: the parent -- which never touches
the GPU -- spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...`, relays rank 0's JSON line and exits
with the child's code.  Run here without a GPU through --dry-engine (gloo, a stand-in engine that only sleeps): what is
tested is the launcher, the rendezvous, the barrier / MAX-over-ranks timing and the aggregation of both scaling shapes."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_self_launch_two_ranks_dry_engine():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-engine", "--streams", "4096", "--seconds", "0.1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak"
    cfg = d["config"]
    assert cfg["ranks_reported"] == 2           # every rank took part in the aggregate
    assert cfg["streams_per_gpu"] == 4096 and cfg["total_streams"] == 8192
    # whole-job value = all ranks' samples over the max-over-ranks time
    n = cfg["samples_per_stream"]
    assert abs(d["value"] - 8192 * n * 3 / (d["ms_per_step"] * 3e-3) / 1e6) / d["value"] < 1e-3
    # the strong-scaling shape of the same job rides in the same line: --streams in total, a contiguous block per rank
    st = cfg["strong_scaling"]
    assert st["total_streams"] == 4096 and st["streams_per_gpu"] == 2048
    assert st["Msamples_per_s"] > 0
    assert "dry-run" in d["data"]
    assert d["roofline"]["binding_bound"] in ("hbm", "valu_issue", "valu_class_priced")   # (the ceiling this run came closest to)


def test_single_rank_dry_engine_has_no_launcher_hop():
    r = _run(["--steps", "2", "--warmup", "0", "--dry-engine", "--streams", "1024", "--seconds", "0.05"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["ranks_reported"] == 1 and "strong_scaling" not in d["config"]


def test_total_streams_is_strong_only():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-engine", "--total-streams", "1001", "--seconds", "0.05"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["total_streams"] == 1001 and d["config"]["streams_per_gpu"] == 501


def test_child_failure_is_the_parents_exit_code():
    r = _run(["--gpus", "2", "--dry-engine", "--steps", "1", "--warmup", "0", "--seconds", "0.05"], {"BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # and no line that could be mistaken for a result

"""world_size-2 `gloo` test of the N > 1 path on CPU: stream sharding (contiguous blocks, no data-path
collective), the MAX-over-ranks time reduction bench.py uses, and the in-order gather of per-stream
outputs.  The per-shard compute is done by the CPU oracle here (the HIP engine needs a GPU); the
sharded result must equal the single-process result stream for stream."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from webaudio_modem_amd import sharding
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = sharding.stream_shard(n_streams, rank, world)
        base = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))["d_default_AB.in"]
        local = []
        for s in range(first, first + count):
            x = np.zeros(base.size + 64, np.float32)
            x[s:s + base.size] = base * np.float32(1.0 / (1 + s % 3))
            o = po.OracleCore({})
            b, _ = o.demodulate(x)
            local.append((s, b))
        dist.barrier()
        elapsed = sharding.max_over_ranks(0.5 + rank, dist)
        merged = sharding.gather_stream_outputs(local, dist)
        if rank == 0:
            q.put((elapsed, merged))
    finally:
        dist.destroy_process_group()


def test_stream_shard_plan():
    from webaudio_modem_amd.sharding import stream_shard, all_shards
    for n, w in [(65536, 8), (10, 3), (7, 8), (1, 1), (16384, 8)]:
        shards = all_shards(n, w)
        assert shards[0][0] == 0
        assert sum(c for _, c in shards) == n
        for (f0, c0), (f1, _) in zip(shards, shards[1:]):
            assert f0 + c0 == f1  # contiguous, in rank order
        assert max(c for _, c in shards) - min(c for _, c in shards) <= 1
    assert stream_shard(65536, 3, 8) == (3 * 8192, 8192)  # BASELINE config #3: 8192 streams per GPU
    with pytest.raises(ValueError):
        stream_shard(8, 8, 8)


def test_two_rank_gloo_sharded_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    n_streams, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 1.5  # MAX over ranks of (0.5, 1.5)
    assert [s for s, _ in merged] == list(range(n_streams))
    base = golden().array("d_default_AB.in")
    for s, b in merged:
        x = np.zeros(base.size + 64, np.float32)
        x[s:s + base.size] = base * np.float32(1.0 / (1 + s % 3))
        ref, _ = po.OracleCore({}).demodulate(x)
        assert b == ref == b"AB", s

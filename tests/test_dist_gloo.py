"""Multi-process orchestration on CPU (gloo, world_size 2): read sharding, the packed-Pg all-gather and the
merge of per-rank results.  The matcher is injected (the oracle stands in for the GPU library, which needs a
device); what is under test is pgrc_amd/dist.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as orc
        from pgrc_amd import dist as pdist
        from util import make_inputs, pack2
        pg, reads = make_inputs(100003, 3001, 100, seed=5, paired=True)

        def packer(sl):
            _, _, sw = pdist.pg_slice(pg.size, rank, world)
            w = np.zeros(sw, dtype=np.uint32)
            if sl.size:
                p = pack2(sl)
                w[: p.size] = p
            return torch.from_numpy(w.view(np.int32))

        def matcher(packed, pg_len, rd):
            # the gathered words must be the packing of the WHOLE text
            full = packed.numpy().view(np.uint32)[: (pg_len + 15) // 16]
            assert np.array_equal(full, pack2(pg)), "all-gather did not rebuild the packed pseudogenome"
            o = orc.oracle_match("c", pg, rd, 38, 2, 0, threads=2)
            return o["pos"], o["rc"], o["mism"], o["hist"]

        lo, hi, pos, rc, mism, hist = pdist.match_sharded(pg, reads, 38, 2, 0, "c", True, packer=packer, matcher=matcher)
        q.put((rank, lo, hi, pos, rc, mism, hist))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from util import make_inputs
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pg, reads = make_inputs(100003, 3001, 100, seed=5, paired=True)
    o = orc.oracle_match("c", pg, reads, 38, 2, 0)
    assert outs[0][1] == 0 and outs[0][2] == outs[1][1] and outs[1][2] == reads.shape[0]
    assert outs[0][2] % 2 == 0  # PE mates stay together
    pos = np.concatenate([t[3] for t in outs])
    rc = np.concatenate([t[4] for t in outs])
    mism = np.concatenate([t[5] for t in outs])
    assert np.array_equal(pos, o["pos"]) and np.array_equal(rc, o["rc"]) and np.array_equal(mism, o["mism"])
    for t in outs:
        assert np.array_equal(t[6], o["hist"])  # every rank holds the global histogram


def test_shard_and_slice_arithmetic():
    from pgrc_amd.dist import pg_slice, shard_range
    for n in (0, 1, 2, 7, 100, 100_000_001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(a[0] % 2 == 0 for a in r if a[1] > a[0])
    for G in (16, 17, 1000, 1_875_000_000):
        for world in (1, 2, 4, 8):
            s = [pg_slice(G, k, world) for k in range(world)]
            assert s[0][0] == 0 and s[-1][1] == G and all(a[0] % 16 == 0 for a in s if a[1] > a[0])
            assert all(a[1] == b[0] for a, b in zip(s, s[1:]))
            assert all((a[1] - a[0] + 15) // 16 <= a[2] for a in s)

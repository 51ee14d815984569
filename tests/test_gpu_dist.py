"""pgrc_amd.dist.match_sharded on its default (HIP) path -- device packer (pgrc_match_pack_pg_slice), the all-gather
of the packed text, pgrc_match_set_pg_packed_device, the HIP matcher, the histogram merge -- as real processes:
world 1, and two gloo ranks sharing device 0 (each a fresh child process).  Result == the unsharded HIP result ==
the oracle; PE mates stay together.  With two or more GPUs visible the same runs over RCCL, one device per rank."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as orc
from util import assert_same_results, gpu_match

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _launch(tmp_path, world, backend, devices):
    port = _free_port()
    procs = []
    for r in range(world):
        out, log = str(tmp_path / f"rank{r}.npz"), open(tmp_path / f"rank{r}.log", "w")   # (a file, not a pipe: no rank can block on a full one)
        procs.append((out, log, subprocess.Popen([sys.executable, os.path.join(HERE, "dist_child.py"), str(r), str(world), port,
                                                  backend, str(devices[r]), out], stdout=log, stderr=subprocess.STDOUT)))
    outs = []
    for out, log, p in procs:
        try:
            p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            for _, _, q in procs:
                q.kill()
            raise
        log.close()
        assert p.returncode == 0, open(log.name).read()[-3000:]
        outs.append(np.load(out))
    return outs


def _check(outs):
    import dist_child
    pg, reads = dist_child.inputs()
    o = orc.oracle_match("c", pg, reads, 38, 3, 0)
    one = gpu_match("c", pg, reads, 38, 3, 0)
    assert_same_results(one, o, "unsharded HIP vs oracle")
    assert int(outs[0]["lo"]) == 0 and int(outs[-1]["hi"]) == reads.shape[0]
    for a, b in zip(outs, outs[1:]):
        assert int(a["hi"]) == int(b["lo"]) and int(a["hi"]) % 2 == 0          # contiguous, PE mates together
    for k in ("pos", "rc", "mism"):
        assert np.array_equal(np.concatenate([t[k] for t in outs]), one[k]), k
    for t in outs:
        assert np.array_equal(t["hist"], one["hist"])                            # every rank holds the global histogram


def test_match_sharded_default_path_world_1(tmp_path):
    _check(_launch(tmp_path, 1, "gloo", [0]))


def test_match_sharded_two_gloo_ranks_sharing_device_0(tmp_path):
    _check(_launch(tmp_path, 2, "gloo", [0, 0]))


def test_match_sharded_rccl_one_device_per_rank(tmp_path):
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one visible device")
    world = min(n, 4)
    _check(_launch(tmp_path, world, "nccl", list(range(world))))

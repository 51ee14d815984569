#!/usr/bin/env python3
"""One rank of pgrc_amd.dist.match_sharded on its DEFAULT path (HIP packer, all-gather, HIP matcher), started as a fresh
process by tests/test_gpu_dist.py so that no rank inherits an initialised GPU runtime.

usage: python tests/dist_child.py RANK WORLD PORT BACKEND DEVICE OUT.npz
BACKEND gloo: ranks may share one device (rehearsal; the collective is staged through the host); nccl: RCCL, one
device per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def inputs():
    from util import make_inputs
    return make_inputs(400_003, 30_001, 150, seed=77, n_with_n=500, paired=True)


def main():
    rank, world, port, backend, device, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(device)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = port
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from pgrc_amd import dist as pdist
    pg, reads = inputs()
    lo, hi, pos, rc, mism, hist = pdist.match_sharded(pg, reads, 38, 3, 0, "c", True)
    np.savez(out, lo=lo, hi=hi, pos=pos, rc=rc, mism=mism, hist=hist)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

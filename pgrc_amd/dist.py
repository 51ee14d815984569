"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

The path shards by READS (each read's result depends only on the read, the pseudogenome and its index --
the reference exploits the same independence with `omp parallel for`, ReadsMatchers.cpp:426-428).  The Pg and its
seed index are replicated: every rank packs 1/world of the host text to 2 bits and ONE all-gather shares the
packed text.  There is no other data-path collective; histograms are summed at the end.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int, align: int = 2) -> Tuple[int, int]:
    """Contiguous read range [lo, hi) of `rank`; boundaries are multiples of `align` so PE mates (2q, 2q+1)
    stay on one rank."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def pg_slice(pg_len: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Symbol range [lo, hi) of the text that `rank` packs, and the common slice size in u32 words.  Slices
    start at multiples of 16 symbols (one packed word) so the gathered words concatenate without shifting."""
    words = (pg_len + 15) // 16
    sw = -(-words // world)
    lo = min(pg_len, rank * sw * 16)
    hi = min(pg_len, (rank + 1) * sw * 16)
    return lo, hi, sw


def all_gather_packed_pg(local_words, world: int, group=None):
    """local_words: torch int32 tensor [slice_words] (device tensor -> RCCL, cpu tensor -> gloo).
    Returns the [world * slice_words] tensor holding the whole packed text (padded at the end)."""
    import torch
    import torch.distributed as dist
    out = torch.empty(world * local_words.numel(), dtype=local_words.dtype, device=local_words.device)
    if world == 1:
        out.copy_(local_words)
    elif local_words.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers (several ranks sharing one GPU): the collective runs on host copies
        host = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host, local_words.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local_words, group=group)
    return out


def merge_histograms(hist: np.ndarray, group=None) -> np.ndarray:
    """Sum of matchedCountPerMismatches over ranks (256 x u64; host-side, outside the data path)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return hist
    t = torch.from_numpy(hist.astype(np.int64))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return t.cpu().numpy().astype(np.uint64)


def match_sharded(pg_ascii: np.ndarray, reads_ascii: np.ndarray, seed_len: int, max_mismatches: int,
                  min_mismatches: int, mode: str = "c", rev_compl_pg: bool = True, group=None,
                  packer: Optional[Callable] = None, matcher: Optional[Callable] = None):
    """Runs the path for this rank's shard of `reads_ascii` (the FULL read set is passed on every rank; only
    the shard is touched).  Returns (lo, hi, pos, rc, mism, global_hist).

    packer(pg_slice_ascii) -> torch int32 tensor of packed words, matcher(packed_pg_tensor, pg_len, reads) ->
    (pos, rc, mism, hist): injectable so the orchestration is testable on CPU with gloo; the defaults use the
    HIP library."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    G = int(pg_ascii.size)
    L = int(reads_ascii.shape[1])
    lo, hi = shard_range(int(reads_ascii.shape[0]), rank, world)
    plo, phi, sw = pg_slice(G, rank, world)

    ctx = None
    if packer is None or matcher is None:
        from .matchers import MatchContext
        ctx = MatchContext(L, seed_len, max_mismatches, min_mismatches, mode, device=torch.cuda.current_device())

    if packer is None:
        def packer(sl):
            t = torch.zeros(sw, dtype=torch.int32, device="cuda")
            if sl.size:
                ctx.pack_pg_slice(sl, t.data_ptr())
            return t
    local = packer(pg_ascii[plo:phi])
    full = all_gather_packed_pg(local, world, group)

    if matcher is None:
        def matcher(packed, pg_len, reads):
            torch.cuda.synchronize()
            ctx.set_pg_packed_device(packed.data_ptr(), pg_len)
            ctx.set_reads_ascii(reads)
            ctx.init_results()
            ctx.run(rev_compl_pg)
            pos, rc, mism, hist, _ = ctx.get_results()
            return pos, rc, mism, hist
    pos, rc, mism, hist = matcher(full, G, reads_ascii[lo:hi])
    return lo, hi, pos, rc, mism, merge_histograms(hist, group)

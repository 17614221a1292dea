"""Utterance-sharded data parallelism for the decode path.

Utterances are independent, so the only communication of a multi-GPU job is ONE all_gather of the
generated token blocks at the end (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_utterances(n_total: int, rank: int, world: int) -> list[int]:
    """Global utterance indices decoded by `rank`: u -> rank u mod world."""
    return list(range(rank, n_total, world))


def gather_token_blocks(gens: list[torch.Tensor], t_max: int, n_slots: int | None = None, K: int = 4,
                        device=None, group=None) -> list[list[torch.Tensor]]:
    """gens: this rank's generated frames, each int64 [K, Tg_i] (Tg_i <= t_max).
    Returns, on every rank, out[r][i] = the i-th utterance of rank r as int64 [K, Tg] (padding removed).
    One collective: an int32 [n_slots, K, t_max+1] block per rank, the length in the extra column
    (n_slots = the largest per-rank utterance count, so every rank sends the same shape; -1 = empty slot)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_local = len(gens)
    n_slots = n_local if n_slots is None else n_slots
    assert n_local <= n_slots
    K = gens[0].shape[0] if n_local else K
    dev = gens[0].device if n_local else torch.device(device or "cpu")
    blk = torch.full((n_slots, K, t_max + 1), -1, dtype=torch.int32, device=dev)
    for i, g in enumerate(gens):
        assert g.shape[0] == K and g.shape[1] <= t_max, (g.shape, K, t_max)
        blk[i, :, : g.shape[1]] = g.to(torch.int32)
        blk[i, :, t_max] = g.shape[1]
    if world == 1:
        blocks = [blk]
    else:
        if dist.get_backend(group) == "gloo" and blk.is_cuda:      # host collectives (ranks sharing one GPU in the single-GPU test
            blk = blk.cpu()                                         # of the N > 1 path): gloo gathers host tensors
        blocks = [torch.empty_like(blk) for _ in range(world)]
        dist.all_gather(blocks, blk, group=group)
        blocks = [b.to(dev) for b in blocks]
    out = []
    for b in blocks:
        out.append([b[i, :, : int(b[i, 0, t_max])].to(torch.int64) for i in range(b.shape[0]) if int(b[i, 0, t_max]) >= 0])
    return out


def merge_in_utterance_order(per_rank: list[list[torch.Tensor]]) -> list[torch.Tensor]:
    """Inverse of shard_utterances: out[u] for u = 0..n_total-1."""
    world = len(per_rank)
    n_total = sum(len(p) for p in per_rank)
    out = [None] * n_total
    for r, lst in enumerate(per_rank):
        for i, t in enumerate(lst):
            out[r + i * world] = t
    return out

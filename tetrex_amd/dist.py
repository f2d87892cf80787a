"""Multi-GPU plumbing for the bin-column sharded index (one process per GPU).

The query path has exactly one exchange: the all-gather of the final per-query masks
(SURVEY.md §8e).  Shards are disjoint column ranges of the full mask, so gathering them in rank
order IS the OR-reduce the reference would do on full-width masks.  `torch.distributed` is used
as plumbing only: backend "nccl" (= RCCL over xGMI) for device tensors, "gloo" for host tensors.
"""
import torch
import torch.distributed as dist


def shard_range(words, rank, world):
    """Mask words [lo, hi) owned by `rank` — must match shard_range() in csrc/txq_api.hip."""
    base, rem = divmod(words, world)
    lo = base * rank + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_final_masks(local, mask_words, group=None):
    """local: int64 tensor [n, shard_words] of this rank -> [n, mask_words] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(mask_words, rank, world)
    assert local.dim() == 2 and local.shape[1] == hi - lo, (tuple(local.shape), lo, hi)
    widest = -(-mask_words // world)
    n = local.shape[0]
    if widest == 0 or n == 0:
        return local.new_zeros((n, mask_words))
    padded = local
    if local.shape[1] != widest:  # uneven split: pad to the widest shard for the collective
        padded = local.new_zeros((n, widest))
        padded[:, : local.shape[1]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=group)
    out = local.new_empty((n, mask_words))
    for r in range(world):
        a, b = shard_range(mask_words, r, world)
        out[:, a:b] = parts[r][:, : b - a]
    return out


def or_reduce_alive(alive, group=None):
    """Bitwise OR of per-shard `alive` bitmaps (a k-mer is dead only if it is dead in every shard).
    RCCL has no bitwise-OR reduction, so the bitmaps are gathered and OR-ed locally."""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(alive) for _ in range(world)]
    dist.all_gather(parts, alive.contiguous(), group=group)
    out = parts[0].clone()
    for p in parts[1:]:
        out |= p
    return out


def or_join_final_masks(local, group=None):
    """Sub-tree shards of a general HIBF (txq_index_upload_subtrees, info.join_or == 1): every rank holds FULL-WIDTH masks
    [n, mask_words] with the user bins of its own sub-trees; a split bin may straddle ranks, so the join is a bitwise OR.
    RCCL has no bitwise-OR reduction (and a SUM would carry where two ranks set the same bit): gathered and ORed locally."""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous(), group=group)
    out = parts[0].clone()
    for p in parts[1:]:
        out |= p
    return out

"""Multi-GPU sharding of the normalisation path: utterance batches are independent, so ranks take whole batches
round-robin with replicated weights and there is no collective in the data path; the only exchange is the final
gather of the (tiny) unit sequences.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm, "gloo"
in the CPU tests)."""
from typing import Any, List, Sequence

import torch.distributed as dist


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def batch_indices(n_items: int, batch_size: int) -> List[range]:
    """The reference's batching: consecutive groups of `batch_size` utterances
    (reference research/TranSpeech/diff_norm_synthesis.py:70-129, batches of 100)."""
    return [range(s, min(s + batch_size, n_items)) for s in range(0, n_items, batch_size)]


def my_batches(n_batches: int, rank: int, world: int) -> List[int]:
    """Batch ids of this rank: round-robin, so every rank keeps the reference's batch composition
    (results are bit-identical to a single-GPU run) and loads differ by at most one batch."""
    return list(range(rank, n_batches, world))


def gather_in_order(local: Sequence[Any], local_ids: Sequence[int], n_total: int, group=None) -> List[Any]:
    """All ranks receive the per-batch results of every rank, ordered by batch id."""
    rank, world = rank_world(group)
    if world == 1:
        out = [None] * n_total
        for i, r in zip(local_ids, local):
            out[i] = r
        return out
    parts = [None] * world
    dist.all_gather_object(parts, (list(local_ids), list(local)), group=group)
    out = [None] * n_total
    for ids, res in parts:
        for i, r in zip(ids, res):
            out[i] = r
    missing = [i for i, r in enumerate(out) if r is None]
    if missing:
        raise RuntimeError(f"batches {missing} were produced by no rank")
    return out

"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed over RCCL/xGMI).

What shards and what does not ("Multi-GPU"):
  - node SoA (4.8 MB at N = 200k) and obstacle tables (8 KB at M = 256) are REPLICATED: with
    288 GB of HBM per GPU, sharding them would only add an exchange of neighbour lists;
  - the sample batch, and therefore the candidate edges derived from it, are SHARDED: rank r owns
    samples [lo, hi) of the global batch; no collective is needed to compute its lists or flags;
  - the only exchange is the per-edge collision bitmask: every rank packs its hit flags
    (1 bit per directed edge, rrtx_pack_hits_dev) into its slice of a global word array and one
    all-reduce makes the whole array visible everywhere.  Slices are disjoint, so SUM == OR
    (RCCL has no bitwise-OR reduction).  The message is (2*cap/8) bytes per rank -- tens of KB,
    latency-bound on xGMI.

The functions below hold only the index arithmetic and the collective, so they run unchanged on
CPU tensors with the gloo backend (tests/test_parallel_gloo.py) and on GPU tensors with nccl.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n items; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def words_for(cap: int) -> int:
    """uint64 words holding the 2*cap hit bits of one rank (hit_out then hit_in)."""
    return (2 * cap + 63) // 64


def exchange_hit_bitmasks(bits: torch.Tensor, rank: int, world: int, words_per_rank: int, group=None,
                          async_op: bool = False):
    """bits: int64[world * words_per_rank]; this rank has filled its own slice
    [rank*wpr, (rank+1)*wpr).  Other slices are zeroed, then one all-reduce(SUM) makes every
    slice visible on every rank (in place).  Returns bits, or with async_op=True the work handle
    (None when world == 1) so the exchange can overlap the next step's compute."""
    assert bits.numel() == world * words_per_rank and bits.dtype == torch.int64
    if world == 1:
        return None if async_op else bits
    lo = rank * words_per_rank
    if lo > 0:
        bits[:lo].zero_()
    if lo + words_per_rank < bits.numel():
        bits[lo + words_per_rank:].zero_()
    work = dist.all_reduce(bits, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else bits


def rank_slice(bits: torch.Tensor, r: int, words_per_rank: int) -> torch.Tensor:
    return bits[r * words_per_rank:(r + 1) * words_per_rank]


def reduce_throughput(local_units: int, local_seconds: float, device=None, group=None) -> Tuple[int, float]:
    """Whole-job figures for the bench line: sum of units over ranks, max of time over ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_units, local_seconds
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    c = torch.tensor([local_units], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return int(c.item()), float(t.item())

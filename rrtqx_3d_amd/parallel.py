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

Second mode, for trees that should not be replicated: the node SoA is partitioned by index
range -- rank g holds nodes [lo_g, hi_g) in a context of its own (local index = global - lo_g; only
rank 0's context holds the root, the others switch the root's <= rule off with RRTX_OPT_ROOT_RULE = 0),
every rank searches ALL queries against its shard, and the per-shard results are merged:
  - range search: all-gather of the per-query counts, then of the (index, distance) payloads padded to
    the largest shard; shard order is index order, so concatenating the shards' lists per query keeps
    every list ascending in node index -- the unsharded CSR, entry for entry;
  - nearest: all-gather of (distance, index) per query and the lexicographic minimum (RCCL has no
    arg-min reduction).
Both are latency-bound exchanges of a few MB at C4 / C5 sizes.

Third mode, north_star's "obstacle set shard" (SURVEY 8e): the OBSTACLE LIST is partitioned -- rank o of an
obstacle group holds list positions [lo_o, hi_o) in its own context and checks ALL the group's candidate edges
against them.  explicitEdgeCheck over the list is an OR with the first hit's list position
(R/DRRT_Q.jl:1802-1826), so the shards combine exactly: per-edge uint8 flags with all_reduce(MAX) (RCCL has no
bitwise OR; MAX of bytes whose bits are independent flags would be wrong, so each flag byte is reduced on its
own bit planes -- see reduce_obstacle_shards), first-hit positions with all_reduce(MIN) after adding the shard's
base (no hit = INT32_MAX).  A 2-D grid composes both splits: rank = e * O + o, edge shard e of E, obstacle shard
o of O; the OR-reduce runs inside each obstacle group {e * O .. e * O + O - 1}, then the E edge shards publish
their packed bitmasks as before (grid_of / obstacle_groups below).

The functions below hold only the index arithmetic and the collectives, so they run unchanged on
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


def grid_of(rank: int, world: int, n_edge_shards: int, n_obstacle_shards: int) -> Tuple[int, int]:
    """(edge shard e, obstacle shard o) of a rank in an E x O grid, rank = e * O + o."""
    assert n_edge_shards * n_obstacle_shards == world and 0 <= rank < world
    return rank // n_obstacle_shards, rank % n_obstacle_shards


def obstacle_groups(world: int, n_edge_shards: int, n_obstacle_shards: int, rank: int, backend=None):
    """The process group of this rank's obstacle group (the O ranks that share edge shard e); every rank creates
    every group, as torch.distributed requires.  None when O == 1 (nothing to reduce) or O == world (the default
    group is the obstacle group)."""
    if n_obstacle_shards == 1 or not dist.is_initialized():
        return None
    if n_obstacle_shards == world:
        return dist.group.WORLD
    mine = None
    for e in range(n_edge_shards):
        ranks = list(range(e * n_obstacle_shards, (e + 1) * n_obstacle_shards))
        g = dist.new_group(ranks=ranks, backend=backend)
        if rank in ranks:
            mine = g
    return mine


_NO_HIT = torch.iinfo(torch.int32).max


def reduce_obstacle_shards(flags: torch.Tensor, first_hit: torch.Tensor = None, obs_lo: int = 0, group=None,
                           flag_bits: int = 1):
    """OR-reduce of per-edge collision flags over the ranks of an obstacle group, in place.
    flags: uint8[...]; with flag_bits == 1 every byte is 0 or 1 and MAX == OR.  The Dubins-with-time preamble
    also sets bit 1 (!validMove, which no obstacle influences: identical on every rank, so MAX keeps it) --
    flag_bits = 2 states that the caller knows bit 1 is rank-invariant; any other mix of bits would need one
    reduction per bit plane and is refused.
    first_hit (optional): int32[...] list positions LOCAL to this rank's obstacle shard, -1 = none; on return the
    smallest GLOBAL list position over the shards (obs_lo = this shard's first list position), -1 = none --
    explicitEdgeCheck returns at the first colliding obstacle in list order, R/DRRT_Q.jl:1808-1822."""
    assert flags.dtype == torch.uint8 and flag_bits in (1, 2)
    if not dist.is_initialized() or (group is None and dist.get_world_size() == 1):
        if first_hit is not None:
            first_hit[first_hit >= 0] += obs_lo
        return flags, first_hit
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    if first_hit is not None:
        assert first_hit.dtype == torch.int32
        f = torch.where(first_hit >= 0, first_hit + obs_lo, torch.full_like(first_hit, _NO_HIT))
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        first_hit.copy_(torch.where(f == _NO_HIT, torch.full_like(f, -1), f))
    return flags, first_hit


def exchange_hit_bitmasks_grouped(bits: torch.Tensor, rank, group=None, async_op: bool = False):
    """The bitmasks of several steps in ONE collective (xGMI rings are latency bound at these sizes: fewer, larger
    all-reduces).  bits: int64[steps, slices, words_per_rank]; this rank has filled bits[:, rank, :].  The other
    slices are zeroed, then one all-reduce(SUM) (disjoint slices: SUM == OR) fills them in place.  rank = None: this
    rank owns no slice (in an E x O grid only obstacle shard 0 of every edge shard publishes; the others add zeros).
    Returns bits, or with async_op=True the work handle (None without a process group)."""
    assert bits.dim() == 3 and bits.dtype == torch.int64 and bits.is_contiguous()
    slices = bits.shape[1]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None if async_op else bits
    if rank is None:
        bits.zero_()
    else:
        if rank > 0:
            bits[:, :rank].zero_()
        if rank + 1 < slices:
            bits[:, rank + 1:].zero_()
    work = dist.all_reduce(bits, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else bits


def exchange_hit_bitmasks_gathered(bits: torch.Tensor, rank: int, group=None, async_op: bool = False):
    """The same exchange as ONE all-gather: bits int64[ranks, steps, words_per_rank], this rank has filled bits[rank]
    (its slices of all the steps, contiguous).  Every rank publishes, so nothing has to be zeroed or summed: a ring
    all-gather moves (n - 1)/n of the buffer over every xGMI link, the all-reduce of exchange_hit_bitmasks_grouped twice
    that -- the choice when every rank owns a slice (edge shards only; in an E x O grid only one rank per obstacle group
    publishes, and the all-reduce over disjoint slices stays).  The input is a copy of the rank's slice (one small
    device copy; no backend has to accept an input that aliases its output).
    Returns bits, or with async_op=True the work handle (None without a process group)."""
    assert bits.dim() == 3 and bits.dtype == torch.int64 and bits.is_contiguous()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None if async_op else bits
    assert bits.shape[0] == dist.get_world_size(group) and 0 <= rank < bits.shape[0]
    own = bits[rank].reshape(-1).clone()
    work = dist.all_gather_into_tensor(bits.view(-1), own, group=group, async_op=async_op)
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


def _all_gather_padded(x: torch.Tensor, n_max: int, world: int, group=None):
    """all_gather of 1-D tensors of different lengths: pad to n_max (the collectives want equal sizes)"""
    buf = torch.zeros(n_max, dtype=x.dtype, device=x.device)
    buf[: x.numel()] = x
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return out


def merge_sharded_radius(offsets: torch.Tensor, idx_global: torch.Tensor, dist_: torch.Tensor, group=None):
    """Node-range-sharded range search: every rank holds the CSR lists of ALL B queries against ITS node
    shard (offsets int64[B+1]; idx_global = shard base + local index, ascending inside each list; dist_ the
    stored keys).  Returns the unsharded CSR (offsets, idx, dist) on every rank: per query the shards' lists
    in rank order, which is ascending node index."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return offsets, idx_global, dist_
    B = offsets.numel() - 1
    counts = (offsets[1:] - offsets[:-1]).contiguous()
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    C = torch.stack(all_counts)                                   # [world, B]
    totals = C.sum(dim=1)
    n_max = max(int(totals.max().item()), 1)
    g_idx = _all_gather_padded(idx_global.to(torch.int64), n_max, world, group)
    g_dist = _all_gather_padded(dist_, n_max, world, group)
    per_q = C.sum(dim=0)
    out_off = torch.zeros(B + 1, dtype=torch.int64, device=offsets.device)
    out_off[1:] = torch.cumsum(per_q, dim=0)
    total = int(out_off[-1].item())
    out_idx = torch.empty(total, dtype=torch.int64, device=offsets.device)
    out_dist = torch.empty(total, dtype=dist_.dtype, device=offsets.device)
    before = torch.zeros(B, dtype=torch.int64, device=offsets.device)     # entries of lower ranks per query
    ar = torch.arange(B, device=offsets.device)
    for r in range(world):
        cr = C[r]
        k_r = int(totals[r].item())
        if k_r:
            off_r = torch.cumsum(cr, dim=0) - cr
            owner = torch.repeat_interleave(ar, cr)
            dest = out_off[:-1][owner] + before[owner] + (torch.arange(k_r, device=offsets.device) - off_r[owner])
            out_idx[dest] = g_idx[r][:k_r]
            out_dist[dest] = g_dist[r][:k_r]
        before = before + cr
    return out_off, out_idx.to(idx_global.dtype), out_dist


def merge_sharded_nearest(idx_global: torch.Tensor, dist_: torch.Tensor, group=None):
    """Node-range-sharded kdFindNearest: every rank's (index, distance) of its shard's nearest node per
    query -> the global nearest: smallest distance, lowest index among equals (a shard with no answer
    reports distance +inf)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return idx_global, dist_
    gi = [torch.empty_like(idx_global) for _ in range(world)]
    gd = [torch.empty_like(dist_) for _ in range(world)]
    dist.all_gather(gi, idx_global.contiguous(), group=group)
    dist.all_gather(gd, dist_.contiguous(), group=group)
    I = torch.stack(gi).to(torch.int64)                           # [world, B]
    D = torch.stack(gd)
    dmin = D.min(dim=0).values
    big = torch.iinfo(torch.int64).max
    cand = torch.where(D == dmin.unsqueeze(0), I, torch.full_like(I, big))
    return cand.min(dim=0).values.to(idx_global.dtype), dmin

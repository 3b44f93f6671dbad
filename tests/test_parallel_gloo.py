"""world_size-2 test of the multi-GPU exchange on CPU (gloo): shard arithmetic and the per-edge
collision bitmask all-reduce of rrtqx_3d_amd/parallel.py (the same code bench.py runs over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rrtqx_3d_amd import parallel


def test_shard_range_partitions():
    for n in (0, 1, 7, 16384, 16385):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _pack(hit_out, hit_in, cap):
    """numpy statement of rrtx_pack_hits_dev's layout: bit e = hit_out[e], bit cap+e = hit_in[e]"""
    flags = np.zeros(parallel.words_for(cap) * 64, dtype=np.uint8)
    flags[:len(hit_out)] = hit_out
    flags[cap:cap + len(hit_in)] = hit_in
    return np.packbits(flags, bitorder="little").view(np.int64)


def _worker(rank, world, port, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        n_valid = cap - 5 * rank                     # ranks own different numbers of candidate edges
        ho = (rng.random(n_valid) < 0.3).astype(np.uint8)
        hi = (rng.random(n_valid) < 0.3).astype(np.uint8)
        wpr = parallel.words_for(cap)
        bits = torch.full((world * wpr,), 0x5555, dtype=torch.int64)      # stale garbage in the other slices
        parallel.rank_slice(bits, rank, wpr).copy_(torch.from_numpy(_pack(ho, hi, cap)))
        parallel.exchange_hit_bitmasks(bits, rank, world, wpr)
        # every rank must now hold every rank's slice
        ok = True
        for r in range(world):
            rr = np.random.default_rng(100 + r)
            nv = cap - 5 * r
            eo = (rr.random(nv) < 0.3).astype(np.uint8)
            ei = (rr.random(nv) < 0.3).astype(np.uint8)
            ok &= np.array_equal(parallel.rank_slice(bits, r, wpr).numpy(), _pack(eo, ei, cap))
        # asynchronous form (what bench.py uses to overlap the exchange with the next step)
        bits2 = torch.zeros(world * wpr, dtype=torch.int64)
        parallel.rank_slice(bits2, rank, wpr).copy_(torch.from_numpy(_pack(ho, hi, cap)))
        work = parallel.exchange_hit_bitmasks(bits2, rank, world, wpr, async_op=True)
        work.wait()
        ok &= torch.equal(bits2, bits)
        units, tmax = parallel.reduce_throughput(1000 + rank, 0.5 + rank)
        ok &= (units == sum(1000 + r for r in range(world))) and (tmax == 0.5 + world - 1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bitmask_exchange_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1000, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]

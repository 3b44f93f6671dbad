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
        # several steps' bitmasks in one collective (bench.py --exchange-every): [steps, world, wpr]
        steps = 3
        grp = torch.full((steps, world, wpr), 0x3333, dtype=torch.int64)
        mine = torch.from_numpy(_pack(ho, hi, cap))
        for k in range(steps):
            grp[k, rank].copy_(mine ^ k)
        w3 = parallel.exchange_hit_bitmasks_grouped(grp, rank, async_op=True)
        w3.wait()
        for k in range(steps):
            for r in range(world):
                ok &= torch.equal(grp[k, r], parallel.rank_slice(bits, r, wpr) ^ k)
        # ... and as one all-gather (every rank publishes: [ranks, steps, words])
        gat = torch.full((world, steps, wpr), 0x5555, dtype=torch.int64)
        for k in range(steps):
            gat[rank, k].copy_(mine ^ k)
        w4 = parallel.exchange_hit_bitmasks_gathered(gat, rank, async_op=True)
        w4.wait()
        for k in range(steps):
            for r in range(world):
                ok &= torch.equal(gat[r, k], parallel.rank_slice(bits, r, wpr) ^ k)
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


# ---- one batch shared by the ranks: node-range sharding and strong-scaling sample shards, oracle-produced ----
def _spawn(target, world, *args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def _node_shard_worker(rank, world, port, q):
    """every rank holds an index range of the tree (its own oracle kd-tree), searches ALL queries against it and
    the merged result must be the unsharded oracle's, entry for entry"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from rrtqx_3d_amd import synth
        n, nq, r = 3001, 97, 9.0                      # ragged shards (3001 = 1001 + 1000 + 1000 at world 3)
        pts, Q = synth.nodes(n, 3), synth.queries(nq, 3)
        Q[5] = pts[0] + [r, 0.0, 0.0]                 # exactly at the range from the ROOT: taken (<=) -- by rank 0 only
        lo, hi = parallel.shard_range(n, rank, world)
        # a shard's local node 0 is the root only on rank 0: the naive scan below applies the reference's rule
        # (root <=, others <) to global indices
        def shard_lists(q):
            d = pts[lo:hi] - q
            dd = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
            keep = dd < r
            if lo == 0:
                keep[0] = dd[0] <= r
            loc = np.nonzero(keep)[0]
            return loc + lo, dd[loc]
        offs, idxs, dists = [0], [], []
        n_idx = np.zeros(nq, dtype=np.int64)
        n_dist = np.zeros(nq)
        for i in range(nq):
            gi, gd = shard_lists(Q[i])
            idxs.append(gi); dists.append(gd); offs.append(offs[-1] + len(gi))
            d = pts[lo:hi] - Q[i]
            s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            k = int(np.argmin(s))
            n_idx[i], n_dist[i] = k + lo, np.sqrt(s[k])
        off_t = torch.tensor(offs, dtype=torch.int64)
        idx_t = torch.from_numpy(np.concatenate(idxs).astype(np.int64))
        dist_t = torch.from_numpy(np.concatenate(dists))
        m_off, m_idx, m_dist = parallel.merge_sharded_radius(off_t, idx_t, dist_t)
        g_idx, g_dist = parallel.merge_sharded_nearest(torch.from_numpy(n_idx), torch.from_numpy(n_dist))
        # the unsharded oracle (the reference's kd-tree)
        tree = O.KDTree(3)
        tree.insert_many(pts)
        ok = True
        for i in range(nq):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            a, b = int(m_off[i]), int(m_off[i + 1])
            ok &= np.array_equal(m_idx[a:b].numpy(), ri[o].astype(np.int64)) and np.array_equal(m_dist[a:b].numpy(), rk[o])
            ni, nd = tree.nearest(Q[i], naive=True)
            ok &= int(g_idx[i]) == ni and float(g_dist[i]) == nd
        ok &= 0 in m_idx[int(m_off[5]):int(m_off[6])].tolist()          # the root at exactly the range
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_node_range_sharded_search_world2_and_3():
    assert _spawn(_node_shard_worker, 2) == [(0, True), (1, True)]
    assert _spawn(_node_shard_worker, 3) == [(0, True), (1, True), (2, True)]


def _strong_worker(rank, world, port, q):
    """strong scaling: ONE global batch, rank r checks the candidate edges of samples shard_range(B, r, world)
    (oracle here, the HIP path in bench.py --scaling strong); after the bitmask exchange every rank holds the
    collision flags of the WHOLE batch, identical to the unsharded oracle's"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from rrtqx_3d_amd import synth
        n, B, r, rr = 4000, 301, 8.0, 0.5
        pts, Q, sph = synth.nodes(n, 3), synth.queries(B, 3), synth.spheres(24)
        tree = O.KDTree(3)
        tree.insert_many(pts)
        osph, m = O.make_spheres(sph)

        def flags(qs):
            ho, hi = [], []
            for qq in qs:
                ri, _ = tree.within_range(r, qq)
                for j in np.sort(ri):
                    ho.append(O.edge_check_spheres(osph, m, qq, pts[j], rr)[0])
                    hi.append(O.edge_check_spheres(osph, m, pts[j], qq, rr)[0])
            return np.array(ho, dtype=np.uint8), np.array(hi, dtype=np.uint8)

        lo, hi_ = parallel.shard_range(B, rank, world)
        my_out, my_in = flags(Q[lo:hi_])
        counts = torch.zeros(world, dtype=torch.int64)
        counts[rank] = len(my_out)
        dist.all_reduce(counts)
        cap = int(counts.max().item())
        wpr = parallel.words_for(cap)
        bits = torch.zeros(world * wpr, dtype=torch.int64)
        parallel.rank_slice(bits, rank, wpr).copy_(torch.from_numpy(_pack(my_out, my_in, cap)))
        parallel.exchange_hit_bitmasks(bits, rank, world, wpr)
        # unpack every rank's slice and compare with the unsharded run over the whole batch
        all_out, all_in = flags(Q)
        got_out, got_in = [], []
        for rk in range(world):
            w = parallel.rank_slice(bits, rk, wpr).numpy().view(np.uint8)
            b = np.unpackbits(w, bitorder="little")
            k = int(counts[rk].item())
            got_out.append(b[:k]); got_in.append(b[cap:cap + k])
        ok = np.array_equal(np.concatenate(got_out), all_out) and np.array_equal(np.concatenate(got_in), all_in)
        q.put((rank, bool(ok and all_out.sum() > 0)))
    finally:
        dist.destroy_process_group()


def test_strong_scaling_batch_shards_world2():
    assert _spawn(_strong_worker, 2) == [(0, True), (1, True)]


def _obstacle_shard_worker(rank, world, port, q, E, O):
    """north_star's obstacle-set shard, alone (E = 1) and composed with sample shards in an E x O grid: rank
    (e, o) checks the candidate edges of sample shard e against list positions [lo_o, hi_o) of the POLYGON list
    (oracle here, the HIP path in bench.py --shard obstacles / --grid ExO); flags are OR-reduced with
    all_reduce(MAX) and first hits with all_reduce(MIN) inside the obstacle group, then the edge shards publish
    packed bitmasks.  Every rank must end up with the unsharded oracle's flags AND first-hit positions."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O_
        from rrtqx_3d_amd import synth
        n, B, r, rr, M = 3000, 120, 9.0, 0.5, 37                # 37 obstacles: ragged obstacle shards
        pts, Q, polys = synth.nodes(n, 3), synth.queries(B, 3), synth.polygons(M)
        active = np.ones(M, dtype=np.uint8); active[[3, 20]] = 0
        tree = O_.KDTree(3)
        tree.insert_many(pts)
        e, o = parallel.grid_of(rank, world, E, O)
        grp = parallel.obstacle_groups(world, E, O, rank)
        s_lo, s_hi = parallel.shard_range(B, e, E)
        o_lo, o_hi = parallel.shard_range(M, o, O)

        def edges_of(qs):
            p0, p1 = [], []
            for qq in qs:
                ri, _ = tree.within_range(r, qq)
                for j in np.sort(ri):
                    p0.append(qq); p1.append(pts[j])
            a, b = np.array(p0), np.array(p1)
            return np.concatenate([a, b]), np.concatenate([b, a])        # out-edges, then in-edges

        p0, p1 = edges_of(Q[s_lo:s_hi])
        sub = O_.PolygonSet(polys[o_lo:o_hi], active=active[o_lo:o_hi])
        hit, first = O_.edges_check_polygons(sub, p0, p1, rr)              # first: position inside the SHARD
        th, tf = torch.from_numpy(hit.copy()), torch.from_numpy(first.copy())
        parallel.reduce_obstacle_shards(th, tf, obs_lo=o_lo, group=grp)
        full = O_.PolygonSet(polys, active=active)
        # list positions of the unsharded oracle count ACTIVE obstacles only?  No: orc positions are list positions
        # of the array it was given, inactive ones included -- the same convention on both sides here.
        rh, rf = O_.edges_check_polygons(full, p0, p1, rr)
        ok = np.array_equal(th.numpy(), rh) and np.array_equal(tf.numpy(), rf) and 0 < rh.sum() < len(rh)
        # edge shards publish: slice e, filled by obstacle shard 0 only
        k = len(p0) // 2
        counts = torch.zeros(E, dtype=torch.int64)
        if o == 0:
            counts[e] = k
        dist.all_reduce(counts)
        cap = int(counts.max().item())
        wpr = parallel.words_for(cap)
        bits = torch.full((1, E, wpr), 0x7777, dtype=torch.int64)
        if o == 0:
            bits[0, e].copy_(torch.from_numpy(_pack(th.numpy()[:k], th.numpy()[k:], cap)))
        parallel.exchange_hit_bitmasks_grouped(bits, e if o == 0 else None)
        a0, a1 = edges_of(Q)
        ah, _ = O_.edges_check_polygons(full, a0, a1, rr)
        ka = len(a0) // 2
        got_out, got_in = [], []
        for ee in range(E):
            b = np.unpackbits(bits[0, ee].numpy().view(np.uint8), bitorder="little")
            kk = int(counts[ee].item())
            got_out.append(b[:kk]); got_in.append(b[cap:cap + kk])
        ok &= np.array_equal(np.concatenate(got_out), ah[:ka]) and np.array_equal(np.concatenate(got_in), ah[ka:])
        # bit 1 of a flag byte (!validMove of the Dubins-with-time preamble) is rank-invariant and survives the MAX
        fb = torch.from_numpy((hit | 2).astype(np.uint8))
        parallel.reduce_obstacle_shards(fb, None, group=grp, flag_bits=2)
        ok &= np.array_equal(fb.numpy(), rh | 2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_obstacle_shards_world2_or_reduce():
    assert _spawn(_obstacle_shard_worker, 2, 1, 2) == [(0, True), (1, True)]


def test_grid_2x2_edge_and_obstacle_shards():
    assert _spawn(_obstacle_shard_worker, 4, 2, 2) == [(r, True) for r in range(4)]

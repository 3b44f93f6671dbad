#!/usr/bin/env python3
"""Benchmark of the RRT^X extend/rewire hot path on MI355X (BASELINE.json metric).

A step = one pass of the fused extend() preamble over one batch of B synthetic
samples with every input already resident in HBM:
    radius-NN (brute force, N nodes)  ->  2*sum(k) directed SimpleEdges: cost +
    collision check against M sphere obstacles  ->  nearest + sample point check.
value = directed edges collision-checked per second (whole job, all ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C4]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (weak scaling): node SoA and obstacle list are replicated (4.8 MB and
8 KB -- trivially small next to 288 GB of HBM), every rank owns its own batch of
B samples, and the per-edge collision bitmask is exchanged with one RCCL
all-reduce so every rank (the planner host of every agent) sees all results.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TOPS = 39.3     # 78.6 TFLOP/s FMA-counted vector fp64 => 39.3 T unfused op/s
ROBOT_RADIUS = 0.5             # R/experimentsForRRTQX.jl:38


def cpu_baseline(cfg, pts, Q, sph, r, budget_s=10.0):
    """Oracle (C restatement of the reference's per-sample loop) on one host core."""
    from oracle import oracle as O
    tree = O.KDTree(cfg.dim)
    tree.insert_many(pts)
    osph, m = O.make_spheres(sph)
    nq = Q.shape[0]
    # calibrate on a slice, then run a bounded sample
    t0 = time.perf_counter()
    e0, _, _, _ = O.extend_batch_spheres(tree, osph, m, Q[:256], r, ROBOT_RADIUS)
    dt = time.perf_counter() - t0
    per_q = dt / 256
    n_sample = int(min(nq, max(256, budget_s / max(per_q, 1e-9))))
    t0 = time.perf_counter()
    edges, neigh, hits, _ = O.extend_batch_spheres(tree, osph, m, Q[:n_sample], r, ROBOT_RADIUS)
    dt = time.perf_counter() - t0
    out = {
        "value": edges / dt, "unit": "edges/s", "cores": 1, "kind": "port",
        "sample": f"first {n_sample} of {nq} samples of the same workload (kd-tree nearest + range + "
                  f"{edges} directed edges x {m} spheres with first-hit early-out), {dt:.1f} s on 1 host core",
        "nn_queries_per_s": n_sample / dt,
    }
    out["all_cores"] = cpu_baseline_all_cores(cfg, pts, Q, sph, r, per_q)
    return out


def cpu_baseline_all_cores(cfg, pts, Q, sph, r, per_q, budget_s=4.0):
    """The same oracle loop run embarrassingly parallel over samples, one private kd-tree per
    thread.  NOT the reference's behaviour (its loop is single-threaded, R/rrtqx.jl:947); reported
    so the GPU/CPU ratio is not read as a core-count artefact."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    n_thr = max(1, min(n_thr, 16))
    nq = Q.shape[0]
    per_thr = int(min(nq // n_thr, max(64, budget_s / max(per_q, 1e-9))))
    osph, m = O.make_spheres(sph)

    def build(_):
        t = O.KDTree(cfg.dim)
        t.insert_many(pts)
        return t

    def run(a):
        tree, k = a
        return O.extend_batch_spheres(tree, osph, m, Q[k * per_thr:(k + 1) * per_thr], r, ROBOT_RADIUS)[0]

    with ThreadPoolExecutor(n_thr) as ex:       # ctypes calls release the GIL
        trees = list(ex.map(build, range(n_thr)))
        t0 = time.perf_counter()
        edges = sum(ex.map(run, [(trees[k], k) for k in range(n_thr)]))
        dt = time.perf_counter() - t0
    return {"value": edges / dt, "unit": "edges/s", "cores": n_thr, "kind": "port",
            "sample": f"{n_thr} threads x {per_thr} samples, private kd-tree each, {dt:.1f} s; "
                      "not the reference's (single-threaded) behaviour"}


def concurrent_agents(k, torch, dev, pts, sph, r, B, cap, N, rounds=20):
    """Side measurement, never `value`: k independent planners (the reference drives 4 agents, each
    with its own tree and obstacle list, R/rrtqx.jl:29-31) share the GPU, one context and one HIP
    stream per agent, every agent stepping its own batch.  Shows how much of the step is launch
    latency that other agents' kernels can fill."""
    from rrtqx_3d_amd import synth
    from rrtqx_3d_amd.context import Context
    agents = []
    for a in range(k):
        c = Context(3, device=dev.index or 0, node_capacity=N)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        c.spheres_set(sph)
        c.nodes_append(pts)
        q = torch.from_numpy(synth.queries(B, 3, seed=synth.SEED + 77 + a)).to(dev)
        buf = dict(off=torch.empty(B + 1, dtype=torch.int64, device=dev), idx=torch.empty(cap, dtype=torch.int32, device=dev),
                   cost=torch.empty(cap, dtype=torch.float64, device=dev), ho=torch.empty(cap, dtype=torch.uint8, device=dev),
                   hi=torch.empty(cap, dtype=torch.uint8, device=dev), need=torch.zeros(1, dtype=torch.int64, device=dev),
                   ni=torch.empty(B, dtype=torch.int32, device=dev), nd=torch.empty(B, dtype=torch.float64, device=dev),
                   un=torch.empty(B, dtype=torch.uint8, device=dev))
        agents.append((c, st, q, buf))

    def step(ag):
        c, st, q, b = ag
        c.extend_candidates_dev(q.data_ptr(), B, r, ROBOT_RADIUS, b["off"].data_ptr(), b["idx"].data_ptr(),
                                b["cost"].data_ptr(), b["ho"].data_ptr(), b["hi"].data_ptr(), cap, b["need"].data_ptr(),
                                b["ni"].data_ptr(), b["nd"].data_ptr(), b["un"].data_ptr())

    for _ in range(3):
        for ag in agents:
            step(ag)
    torch.cuda.synchronize()
    edges = sum(2 * int(ag[3]["need"].item()) for ag in agents)
    t0 = time.perf_counter()
    for _ in range(rounds):
        for ag in agents:
            step(ag)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for ag in agents:
        ag[0].close()
    return {"agents": k, "edges_per_s": edges * rounds / dt, "ms_per_round": 1e3 * dt / rounds,
            "note": "k contexts on k streams, each stepping its own batch of the same config; not `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nn-filter", type=int, default=1, help="1: fp32 prefilter + exact fp64 confirm (default); 0: exact scan")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsal)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: all ranks on cuda:0")
    ap.add_argument("--scan-blocks", type=int, default=0, help="tuning: target workgroups of the range scan")
    ap.add_argument("--scan-items", type=int, default=0, help="tuning: target (tile, segment) work items")
    ap.add_argument("--tile-q", type=int, default=0, help="tuning: query copies per workgroup tile")
    ap.add_argument("--nn-cull", type=int, default=1, help="slab culling of the range scan: 0 off, 1 auto, 2 always")
    ap.add_argument("--tune", type=int, default=0, help="RRTX_OPT_TUNE bit mask (kernel variants under measurement)")
    ap.add_argument("--agents", type=int, default=0, help="side measurement: k independent planners sharing the GPU (0/1: off)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from rrtqx_3d_amd import synth
    from rrtqx_3d_amd.context import Context

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:            # rehearsal of the N>1 code path on a one-GPU box (with --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    cfg = synth.CONFIGS[args.config]
    assert cfg.dim == 3, "the bench line is the SimpleEdge path"
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    r = synth.ball_radius(N, 3)
    pts = synth.nodes(N, 3)
    sph = synth.spheres(M)
    Q = synth.queries(B, 3, seed=synth.SEED + 1 + 1000 * rank)   # every rank its own batch

    ctx = Context(3, device=local_rank, node_capacity=N)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)          # kernels, events and torch share one stream
    ctx.spheres_set(sph)
    from rrtqx_3d_amd import _capi
    ctx.set_option(_capi.RRTX_OPT_NN_FILTER, args.nn_filter)
    if args.scan_blocks:
        ctx.set_option(_capi.RRTX_OPT_SCAN_BLOCKS, args.scan_blocks)
    if args.scan_items:
        ctx.set_option(_capi.RRTX_OPT_SCAN_ITEMS, args.scan_items)
    if args.tile_q:
        ctx.set_option(_capi.RRTX_OPT_SCAN_TILE_Q, args.tile_q)
    ctx.set_option(_capi.RRTX_OPT_NN_CULL, args.nn_cull)
    if args.tune:
        ctx.set_option(_capi.RRTX_OPT_TUNE, args.tune)

    # ---- inputs resident in HBM before the timed region -----------------------
    d_pts = torch.from_numpy(pts).to(dev)
    ctx.nodes_append_dev(d_pts.data_ptr(), N)
    d_q = torch.from_numpy(Q).to(dev)
    cap = 96 * B                                 # expected sum(k) ~ 26 B; generous head room
    d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
    d_cost = torch.empty(cap, dtype=torch.float64, device=dev)
    d_hout = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_hin = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_needed = torch.zeros(1, dtype=torch.int64, device=dev)
    d_nidx = torch.empty(B, dtype=torch.int32, device=dev)
    d_ndist = torch.empty(B, dtype=torch.float64, device=dev)
    d_unsafe = torch.empty(B, dtype=torch.uint8, device=dev)
    from rrtqx_3d_amd import parallel

    def compute():
        ctx.extend_candidates_dev(d_q.data_ptr(), B, r, ROBOT_RADIUS, d_off.data_ptr(), d_idx.data_ptr(),
                                  d_cost.data_ptr(), d_hout.data_ptr(), d_hin.data_ptr(), cap,
                                  d_needed.data_ptr(), d_nidx.data_ptr(), d_ndist.data_ptr(), d_unsafe.data_ptr())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # sizing pass (part of the warm-up): number of candidate entries of this rank's batch
    compute()
    fence()
    k_total = int(d_needed.item())
    if k_total > cap:
        raise SystemExit(f"candidate capacity too small: {k_total} > {cap}")
    edges_per_step = 2 * k_total

    # exchange buffers sized for the largest shard: 2 bits (out, in) per candidate entry and rank
    k_max = k_total
    if world > 1:
        t = torch.tensor([k_total], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        k_max = int(t.item())
    bit_cap = (k_max + 4095) // 4096 * 4096
    words_per_rank = parallel.words_for(bit_cap)
    d_bits = [torch.zeros(world * words_per_rank, dtype=torch.int64, device=dev) for _ in range(2)]
    pending = [None, None]

    def step(i):
        compute()
        if world > 1:
            # per-edge collision bitmask exchange: each rank fills its slice, one RCCL all-reduce
            # (disjoint slices: SUM == OR).  Double-buffered and asynchronous: the exchange of step i
            # runs on RCCL's stream while step i+1 computes.
            b = i & 1
            if pending[b] is not None:
                pending[b].wait()
            ctx.pack_hits_dev(d_hout.data_ptr(), d_hin.data_ptr(), d_off.data_ptr() + 8 * B, bit_cap,
                              d_bits[b].data_ptr() + 8 * rank * words_per_rank)
            pending[b] = parallel.exchange_hit_bitmasks(d_bits[b], rank, world, words_per_rank, async_op=True)

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    fence()

    # HIP events around the dominant kernel only, and only on every 4th step: two event records drain
    # the pipeline for ~10 us, a tenth of the step
    sample_every = 4 if args.steps >= 8 else 1
    ctx.set_option(_capi.RRTX_OPT_PROFILE_EVERY, sample_every)
    ctx.profile(1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    fence()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_option(_capi.RRTX_OPT_PROFILE_EVERY, 1)
    # the other kernel families are timed in a short pass of their own: an event record between two
    # kernels costs ~10 us of pipeline drain, which must not sit inside the timed region
    ctx.profile(2)
    for i in range(5):
        compute()
    fence()
    st_all = ctx.stats()
    # the brute-force form of the search (every tile of 64 copies streams every node, north_star's
    # kernel) is measured beside the default culled form: a short pass with culling switched off
    bf = None
    if args.nn_filter and args.nn_cull and st.last_scan_units > 0:
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
        compute()
        fence()
        ctx.profile(1)
        for i in range(5):
            compute()
        fence()
        st_bf = ctx.stats()
        bf = {"scan_ms": st_bf.ms_nn_scan / max(st_bf.launches_nn_scan, 1), "tile_q": int(st_bf.last_tile_q)}
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, args.nn_cull)
        compute()
        fence()
    ctx.profile(0)

    # ---- aggregate over ranks ----------------------------------------------------
    e_sum, t_max = parallel.reduce_throughput(edges_per_step, dt, device=dev)
    q_sum = B * world

    if rank == 0:
        ms_step = 1e3 * t_max / args.steps
        scan_ms = st.ms_nn_scan / max(st.launches_nn_scan, 1)
        tile_q = int(st.last_tile_q)   # query copies sharing one streamed pass of the node arrays
        n_tiles = (B + tile_q - 1) // tile_q
        units = int(st.last_scan_units)  # slab-culled scan: (tile, 512-node chunk) pairs actually streamed
        node_visits = units * 512 if units > 0 else n_tiles * N
        bytes_streamed = node_visits * 24 + B * 32 + k_total * 16      # node passes + queries + hit records
        achieved = bytes_streamed / (scan_ms * 1e-3) / 1e9
        valu_ops = B * N * 9                                           # 3 sub, 3 mul, 2 add, 1 cmp per (query, node)
        # HBM-side bytes per launch of this kernel from the committed PMC passes (FETCH_SIZE + WRITE_SIZE,
        # rocprofv3 --pmc, scripts_gpu_pmc.sh); PMC counters cannot be read inside this process.
        culled = units > 0
        kernel = ("nn_tile_kernel<3>" if culled else "nn_scan_f32_kernel<3> + nn_confirm_kernel<3>") if args.nn_filter \
            else "nn_scan_kernel<3>"
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if args.config == "C4" and args.nn_filter and os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel") == kernel:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/r01_traffic.json (rocprofv3 PMC passes)"
        roof_bf = None
        if bf:
            nt = (B + bf["tile_q"] - 1) // bf["tile_q"]
            by = nt * N * 24 + B * 32 + k_total * 16
            ach = by / (bf["scan_ms"] * 1e-3) / 1e9
            roof_bf = {"kernel": "nn_scan_f32_kernel<3> + nn_confirm_kernel<3> (RRTX_OPT_NN_CULL=0)", "bound": "hbm",
                       "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": by, "tile_q": bf["tile_q"], "ms": bf["scan_ms"],
                       "pairs_per_s": B * N / (bf["scan_ms"] * 1e-3)}
        out = {
            "metric": "collision-checked edges/sec + radius-NN queries/sec at N=200k nodes, 256 obs",
            "value": e_sum * args.steps / t_max,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg.name, "n_nodes": N, "n_obstacles": M, "batch_per_gpu": B,
                       "radius": r, "edge": "SimpleEdge", "directed_edges_per_step_per_gpu": edges_per_step,
                       "neighbors_per_step_per_gpu": k_total, "sharding": "samples+edges sharded, nodes/obstacles replicated"},
            "nn_queries_per_s": q_sum * args.steps / t_max,
            "kernel_ms": {
                "nn_scan": scan_ms,
                "nn_finish": st_all.ms_nn_finish / 5,
                "edges": st_all.ms_edges / 5,
                "points": st_all.ms_points / 5,
            },
            "roofline": {
                "kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_streamed, "tile_q": tile_q,
                "kernel_ms": scan_ms, "timed_launches": int(st.launches_nn_scan),
                "timing": f"HIP events around every {sample_every}th launch inside the timed region",
                "culled_units": units, "node_visits_per_launch": node_visits,
                "node_visits_unculled": n_tiles * N,
                "pairs_per_s": B * N / (scan_ms * 1e-3),
                "valu_fp64_frac": (valu_ops / (scan_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS) if not args.nn_filter else None,
                "note": ("culled search: bytes = chunks actually streamed x 512 nodes x 24 B (SURVEY 8d figure per node "
                         "visit, T_q = tile_q); the kernel is latency / issue bound and its working set is L2-resident, "
                         "see DESIGN.md 4.1") if culled else
                        "VALU-issue bound, node arrays are L2-resident; see DESIGN.md",
            },
            "roofline_bruteforce": roof_bf,
        }
        if world == 1:
            # the same step through the host-pointer entry point (what a ccall from Julia pays):
            # H2D of the samples, kernels, D2H of lists/costs/flags.  Reported, never `value`.
            ctx.set_stream(None)
            ctx.extend_candidates(Q, r, ROBOT_RADIUS, cap=cap)
            t1 = time.perf_counter()
            for _ in range(5):
                ctx.extend_candidates(Q, r, ROBOT_RADIUS, cap=cap)
            out["host_buffer_path"] = {"edges_per_s": edges_per_step * 5 / (time.perf_counter() - t1),
                                       "note": "PCIe-inclusive: host numpy in/out through rrtx_extend_candidates"}
        if world == 1:
            # BASELINE.json's config text says "polygon obstacles": the reference's 3-D planner checks
            # spheres (explicitEdgeCheck3D) and `value` above is that; this is the same step against 256
            # random polygons in the (x, y) projection (explicitEdgeCheck2D), device-resident, reported only.
            ctx.set_stream(stream.cuda_stream)
            ctx.polygons_set(synth.polygons(M))
            ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
            for _ in range(3):
                compute()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                compute()
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t1) / 20
            out["polygon_obstacles"] = {"edges_per_s": edges_per_step / dtp, "ms_per_step": dtp * 1e3, "n_polygons": M,
                                        "colliding_fraction": float((d_hout[:k_total].sum() + d_hin[:k_total].sum()).item()) / edges_per_step,
                                        "note": "same samples and tree, candidate edges and samples checked against the polygon list"}
            ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 0)
            ctx.set_stream(None)
        if world == 1 and args.agents > 1:
            out["concurrent_agents"] = concurrent_agents(args.agents, torch, dev, pts, sph, r, B, cap, N)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, pts, Q, sph, r)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()

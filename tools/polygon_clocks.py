"""Where the waves of edges_polygons_kernel spend their time (config C4 against 256 polygons): runs the fused extend step on
the measuring build (python -m rrtqx_3d_amd.build --clocks) and prints the share of every stage in the waves' summed
wall-clock ticks."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import _capi, build, synth  # noqa: E402

_capi.LIB_PATH = build.LIB_CLK
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C4"]
ROWS = 65536
pts, Q = synth.nodes(cfg.n_nodes, 3), synth.queries(cfg.batch, 3)
names = ["load + wave's obstacle list", "boxes + box tests", "pair hand-out", "stage A: bounding circle", "stage A: sides",
         "stage B: segment tests", "whole wave", "waves"]
with Context(3, node_capacity=cfg.n_nodes) as ctx:
    ctx.nodes_append(pts)
    ctx.polygons_set(synth.polygons(cfg.n_obstacles))
    ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
    L = _capi.load()
    L.rrtx_debug_polygon_edge_clocks.restype = C.c_int
    L.rrtx_debug_polygon_point_clocks.restype = C.c_int
    pbuf = np.zeros((ROWS, 8), dtype=np.uint64)

    buf = np.zeros((ROWS, 8), dtype=np.uint64)
    r = synth.ball_radius(cfg.n_nodes, 3)
    out = ctx.extend_candidates(Q, r, 0.5, cap=96 * cfg.batch)      # warm
    assert L.rrtx_debug_polygon_edge_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    assert L.rrtx_debug_polygon_point_clocks(pbuf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    ctx.profile(2)
    st0 = ctx.stats()
    out = ctx.extend_candidates(Q, r, 0.5, cap=96 * cfg.batch)
    assert L.rrtx_debug_polygon_edge_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    st1 = ctx.stats()
    print("this build: edges kernel %.4f ms, points kernel %.4f ms" % (st1.ms_edges - st0.ms_edges, st1.ms_points - st0.ms_points))
    live = buf[buf[:, 7] > 0]
    t_end = (live[:, 7] >> np.uint64(8)).astype(np.float64) / 100.0                 # us
    dur = live[:, 6].astype(np.float64) / 2400.0                                    # us at 2.4 GHz (shader clock: an estimate)
    t0 = (t_end - dur).min()
    edges = np.arange(0.0, t_end.max() - t0 + 5.0, 5.0)
    resident = [int(((t_end - dur - t0 < b + 2.5) & (t_end - t0 > b + 2.5)).sum()) for b in edges]
    print("waves resident (whole GPU) every 5 us from the first start:", resident)
    live = live.copy(); live[:, 7] = 1
    rows = live.astype(np.float64)
    wt = rows[:, 6]
    print("wave time (shader clock ticks): median %.0f, 90 %% %.0f, max %.0f" % (np.median(wt), np.percentile(wt, 90), wt.max()))
    buf = rows.sum(axis=0)
    tot = float(buf[6])
    print("waves %d, mean wave %.0f ticks (shader clock)" % (buf[7], tot / max(1, int(buf[7]))))
    for k in range(6):
        print("  %-32s %5.1f %%" % (names[k], 100.0 * float(buf[k]) / tot))
    print("  %-32s %5.1f %%" % ("other (init, results)", 100.0 * (tot - float(buf[:6].sum())) / tot))
    assert L.rrtx_debug_polygon_point_clocks(pbuf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    prow = pbuf[pbuf[:, 5] > 0]
    span = int(prow[:, 7].max()) - int(prow[:, 6].min())
    wt = prow[:, 4].astype(np.float64)
    pbuf = prow.astype(np.float64).sum(axis=0)
    tot = float(pbuf[4])
    print("points_polygons_flag_kernel: waves %d, mean wave %.0f ticks (median %.0f, max %.0f), first start to last end %d ticks" % (pbuf[5], tot / max(1, int(pbuf[5])), np.median(wt), wt.max(), span))
    for k, nm in enumerate(["load + vertex-height table", "list walk", "near polygons", "results"]):
        print("  %-32s %5.1f %%" % (nm, 100.0 * float(pbuf[k]) / tot))

"""Lattice scenes (tools/soak_lattice.py): every coordinate a multiple of 1/4, so that points sit exactly on lines,
distances exactly on thresholds and segments on common lines -- the places where the reference's strict / non-strict
comparisons, its "close to vertical" branches, segmentDistSqrd's answer for coincident lines (R/DRRT.jl:1148-1193)
and pointInPolygon's strict crossing tests (:1009-1056) decide.  Polygon and sphere edge / point checks, range
search, fused extend flags and nearest distances through the C-ABI against the oracle, bit for bit.  Scenes 5 and
124 are the ones that exposed the two shortcuts of round 2 that random scenes never caught (segments dropped by
their boxes alone; far obstacles left out of a flag-only point check)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "soak_lattice", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "soak_lattice.py"))
soak_lattice = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(soak_lattice)


@pytest.mark.parametrize("first", [0, 120])
def test_lattice_scenes_match_the_oracle(oracle, first):
    edges = flips = dubins = diffs = own = 0
    for sc in range(first, first + 8):
        o = soak_lattice.scene(sc)
        edges += o["edges"]; flips += o["dubins_flips"]; dubins += o["dubins_edges"]; diffs += o.get("dubins_cost_diffs", 0)
        own += o.get("dubins_own_flips", 0)
    assert edges == 8 * 3600
    assert own == 0                        # the two-stage check itself: the oracle's answer on the device's own polyline
    # Dubins between lattice poses: the last bit of sin / cos / atan2 decides grazing pieces and zero-length arcs that
    # wrap to a full turn.  Until round 2 (device OCML vs host glibc) 100 scenes differed in 25 booleans and 1 cost of
    # 30 000 edges; device and oracle now share include/rrtx_detmath.h: costs, words, row counts and booleans are equal.
    assert dubins == 8 * 300 and flips == 0 and diffs == 0

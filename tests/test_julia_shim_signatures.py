"""julia/RRTXHip.jl cannot be executed here (no Julia in the image), so its `ccall` sites are checked as
text against include/rrtx.h: every bound symbol exists in the header with the same number of arguments and
compatible argument / return types (Cint <-> int, Int64 <-> int64_t, Cdouble <-> double, pointers <->
pointers), and every compute entry point the reference's per-edge-type contract needs
(R/README.txt:85-99) is bound."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def header_protos():
    txt = open(os.path.join(ROOT, "include", "rrtx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|void \*|const char \*)\s*(rrtx_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        args = [] if args.strip() in ("", "void") else _split_args(" ".join(args.split()))
        protos[name] = (ret, args)
    return protos


def c_class(t):
    t = t.strip()
    if "*" in t:
        return "ptr"
    base = t.split()[:-1] if len(t.split()) > 1 else t.split()
    base = " ".join(w for w in base if w != "const")
    return {"int": "i32", "int64_t": "i64", "double": "f64", "uint8_t": "u8"}.get(base, base)


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Cint": "i32", "Int64": "i64", "Cdouble": "f64", "UInt8": "u8"}.get(t, t)


def julia_ccalls():
    txt = open(os.path.join(ROOT, "julia", "RRTXHip.jl")).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(rrtx_\w+), LIBRRTX\),\s*(\w+),\s*\(", txt):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        types = _split_args(" ".join(txt[m.end():i - 1].split()))
        types = [t for t in types if t]
        # the actual arguments follow up to the ccall's closing parenthesis
        j, depth = i, 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[j], 0)
            j += 1
        actual = _split_args(" ".join(txt[i:j - 1].split()).lstrip(", "))
        calls.append((name, ret, types, actual))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_protos()
    calls = julia_ccalls()
    assert len(calls) >= 25
    for name, ret, types, actual in calls:
        assert name in protos, f"{name} is not declared in include/rrtx.h"
        cret, cargs = protos[name]
        assert len(types) == len(cargs), f"{name}: {len(types)} ccall argument types, header has {len(cargs)}"
        assert len(actual) == len(types), f"{name}: {len(actual)} arguments passed for {len(types)} types"
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            assert jl_class(jt) == c_class(ct), f"{name} argument {k + 1}: Julia {jt} vs C `{ct}`"
        want = {"int": "Cint", "int64_t": "Int64", "const char *": "Cstring", "void *": "Ptr{Cvoid}"}[cret]
        assert ret == want, f"{name}: returns {ret}, header says {cret}"


def test_the_edge_type_contract_is_bound():
    bound = {c[0] for c in julia_ccalls()}
    for sym in ("rrtx_create", "rrtx_destroy", "rrtx_nodes_append", "rrtx_set_wrap", "rrtx_nn_nearest", "rrtx_nn_radius",
                "rrtx_nn_knearest", "rrtx_spheres_set", "rrtx_polygons_set", "rrtx_polygon_paths_set", "rrtx_edges_check",
                "rrtx_points_check", "rrtx_simple_steer", "rrtx_dubins_steer_full", "rrtx_dubins_trajectory",
                "rrtx_dubins_edges_check", "rrtx_set_dubins_velocity", "rrtx_extend_candidates",
                "rrtx_extend_candidates_dubins", "rrtx_graph_edges_append", "rrtx_obstacle_sweep", "rrtx_set_option"):
        assert sym in bound, sym
    txt = open(os.path.join(ROOT, "julia", "RRTXHip.jl")).read()
    for fn in ("kdInsert", "kdFindNearest", "kdFindWithinRange", "kdFindMoreWithinRange", "kdFindKNearest",
               "explicitPointCheck", "explicitNodeCheck", "validMove", "extend_candidates_dubins"):
        assert re.search(rf"\b{fn}\(", txt), fn
    for sig in ("calculateTrajectory(S::TS, edge::SimpleEdge)", "calculateTrajectory(S::TS, edge::DubinsEdge)",
                "explicitEdgeCheck(S::CSpace{T}, edge::DubinsEdge, obstacle::Obstacle)",
                "explicitEdgeCheck(S::CSpace{T}, edge::SimpleEdge, obstacle::SphereObstacle)"):
        assert sig in txt, sig

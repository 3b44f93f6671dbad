# RRTXHip.jl -- Julia-side drop-in for the RRT^X extend/rewire hot path of
# jnetter6/RRTQX_3D, bound to librrtx_hip.so (include/rrtx.h) with ccall.
#
# Usage inside the reference (after its own includes, e.g. at the end of the
# include list of experimentsForRRTQX.jl):
#
#     include("DRRT_data_structures.jl"); include("jlist.jl"); ...   # unchanged reference files
#     include("/path/to/julia/RRTXHip.jl")
#     KD = HipTree{RRTNode{Float64}}(3)                 # instead of KDTree{RRTNode{Float64}}(d, KDdist)
#
# Every method below has the name, arity and return shape of the reference
# function it replaces, so extend()/findBestParent()/addNewObstacle() run
# unchanged on a HipTree.  NOTE: Julia is not installed in the build image, so
# this file has been written against Julia 1.0 semantics but not executed; the
# same C-ABI is exercised through ctypes by tests/ (rrtqx_3d_amd/_capi.py).

const LIBRRTX = get(ENV, "RRTX_HIP_LIB", "librrtx_hip.so")

const RRTX_OK = Cint(0)
const RRTX_E_CAPACITY = Cint(-2)

mutable struct HipTree{T}
  d::Int                       # fields the planner reads (R/rrtqx.jl:382, R/DRRT_Q.jl:2624,2631)
  treeSize::Int
  numWraps::Int
  root::T
  ctx::Ptr{Cvoid}
  nodes::Vector{T}             # device index (0-based) + 1 -> node; the reference has no node ids
  obsSig::UInt64               # signature of the obstacle list last uploaded
  indexOf::IdDict{Any,Int32}   # node -> 0-based device index (for edges given as node pairs)

  function HipTree{T}(d::Int; device::Int = 0, capacity::Int = 1 << 16) where {T}
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:rrtx_create, LIBRRTX), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Int64), ref, d, device, capacity)
    rc == RRTX_OK || error(unsafe_string(ccall((:rrtx_create_error, LIBRRTX), Cstring, ())))
    t = new{T}(d, 0, 0)
    t.ctx = ref[]
    t.nodes = Vector{T}()
    t.obsSig = UInt64(0)
    t.indexOf = IdDict{Any,Int32}()
    finalizer(x -> ccall((:rrtx_destroy, LIBRRTX), Cint, (Ptr{Cvoid},), x.ctx), t)
    return t
  end
end

# KDTree{T}(d, f, wraps, wrapPoints)  (R/kdTree_general.jl:108); wraps are 1-based dimensions
function HipTree{T}(d::Int, f::Function, wraps::Array{Int}, wrapPoints::Array{Float64}) where {T}
  t = HipTree{T}(d)
  for i = 1:length(wraps)
    rrtx_check(t, ccall((:rrtx_set_wrap, LIBRRTX), Cint, (Ptr{Cvoid}, Cint, Cdouble), t.ctx, wraps[i] - 1, wrapPoints[i]))
  end
  t.numWraps = length(wraps)
  return t
end
HipTree{T}(d::Int, f::Function) where {T} = HipTree{T}(d)

# non-zero status -> error(), the reference's only failure idiom (R/rrtqx.jl:70,91)
function rrtx_check(t::HipTree, rc::Cint)
  rc == RRTX_OK && return
  error(unsafe_string(ccall((:rrtx_last_error, LIBRRTX), Cstring, (Ptr{Cvoid},), t.ctx)))
end

# ---------------------------------------------------------------------------
# kdInsert (R/kdTree_general.jl:121-170)
function kdInsert(tree::HipTree{T}, node::T) where {T}
  if node.kdInTree
    return
  end
  node.kdInTree = true
  pos = vec(convert(Array{Float64}, node.position))          # 1 x d row -> d contiguous doubles
  first = Ref{Int64}(0)
  GC.@preserve pos rrtx_check(tree, ccall((:rrtx_nodes_append, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Ref{Int64}), tree.ctx, pos, 1, first))
  push!(tree.nodes, node)
  tree.indexOf[node] = Int32(first[])
  if tree.treeSize == 0
    tree.root = node
  end
  tree.treeSize += 1
end

# kdFindNearest (R/kdTree_general.jl:357-385) -> (node, dist)
function kdFindNearest(tree::HipTree{T}, queryPoint::Array{Float64}) where {T}
  q = vec(queryPoint)
  idx = Ref{Int32}(0); dist = Ref{Float64}(0.0)
  GC.@preserve q rrtx_check(tree, ccall((:rrtx_nn_nearest, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ref{Int32}, Ref{Float64}), tree.ctx, q, 1, idx, dist))
  return (tree.nodes[idx[] + 1], dist[])
end

# kdFindNearestWithGuesstree (R/kdTree_general.jl:503-534): the guess only seeds the reference's descent
kdFindNearestWithGuesstree(tree::HipTree{T}, queryPoint::Array{Float64}, guess::T) where {T} =
  kdFindNearest(tree, queryPoint)

# kdFindKNearest (R/kdTree_general.jl:696-723) -> Array of nodes, node.data = distance.
# max(k, 2) nodes like the reference (its heap starts with root + dummy); ascending distance.
function kdFindKNearest(tree::HipTree{T}, k::Int, queryPoint::Array{Float64}) where {T}
  q = vec(queryPoint)
  w = max(k, 2)
  idx = Array{Int32}(undef, w); dist = Array{Float64}(undef, w); cnt = Ref{Int32}(0)
  GC.@preserve q idx dist rrtx_check(tree, ccall((:rrtx_nn_knearest, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Ptr{Int32}, Ptr{Cdouble}, Ref{Int32}),
      tree.ctx, q, 1, k, idx, dist, cnt))
  ret = Array{T}(undef, cnt[], 1)
  for j = 1:cnt[]
    ret[j, 1] = tree.nodes[idx[j] + 1]
    ret[j, 1].data = dist[j]
  end
  return ret
end

# addToRangeList (R/kdTree_general.jl:765-771)
function addToRangeList(S::Tlist, thisNode::T, key::Float64) where {Tlist, T}
  if thisNode.inHeap
    return
  end
  thisNode.inHeap = true
  JlistPush(S, thisNode, key)
end

# kdFindMoreWithinRange (R/kdTree_general.jl:927-955): neighbours are appended to L with
# key = distance; membership and keys equal the reference's, list order is by node index.
function kdFindMoreWithinRange(tree::HipTree{T}, range::Float64, queryPoint::Array{Float64}, L::TL) where {T, TL}
  q = vec(queryPoint)
  r = [range]
  offsets = Vector{Int64}(undef, 2)
  cap = 256
  while true
    idx = Vector{Int32}(undef, cap); dist = Vector{Float64}(undef, cap)
    needed = Ref{Int64}(0)
    rc = GC.@preserve q r offsets idx dist ccall((:rrtx_nn_radius, LIBRRTX), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}, Int64, Ref{Int64}),
        tree.ctx, q, r, 0, 1, offsets, idx, dist, cap, needed)
    if rc == RRTX_E_CAPACITY            # two-call pattern: retry with the size the library reports
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    for k = Int(needed[]):-1:1           # push in reverse so the list reads in ascending index order
      addToRangeList(L, tree.nodes[idx[k] + 1], dist[k])
    end
    return L
  end
end

# kdFindWithinRange (R/kdTree_general.jl:889-919)
function kdFindWithinRange(tree::HipTree{T}, range::Float64, queryPoint::Array{Float64}) where {T}
  L = JList{T}()
  return kdFindMoreWithinRange(tree, range, queryPoint, L)
end
# popFromRangeList / emptyRangeList (R/kdTree_general.jl:774-787) work on the JList unchanged.

# ---------------------------------------------------------------------------
# file dumps: the reference's writers recurse over kdChildL / kdChildR (R/DRRT_Q.jl:250-364), which a
# HipTree does not have; these walk tree.nodes in insertion order (same rows, different row order).
function saveRRTTree(tree::HipTree{T}, fileName) where {T}
  fptr = open(fileName, "w")
  for node in tree.nodes
    if node.rrtParentUsed
      writedlm(fptr, [node.position node.rrtTreeCost], ',')
      writedlm(fptr, [node.rrtParentEdge.endNode.position node.rrtParentEdge.endNode.rrtTreeCost], ',')
    end
  end
  close(fptr)
end

function saveRRTGraph(tree::HipTree{T}, fileName) where {T}          # R/DRRT_Q.jl:279-306
  fptr = open(fileName, "w")
  for node in tree.nodes
    listItem = node.rrtNeighborsOut.front
    for i = 1:node.rrtNeighborsOut.length
      writedlm(fptr, node.position, ',')
      writedlm(fptr, listItem.data.position, ',')
      listItem = listItem.child
    end
  end
  close(fptr)
end

function saveRRTNodes(tree::HipTree{T}, fileName) where {T}
  fptr = open(fileName, "w")
  for node in tree.nodes
    writedlm(fptr, [node.position node.rrtTreeCost node.rrtLMC], ',')
  end
  close(fptr)
end

function saveRRTNodesCollision(tree::HipTree{T}, fileName) where {T}
  fptr = open(fileName, "w")
  for node in tree.nodes
    writedlm(fptr, [node.position min(node.rrtTreeCost, node.rrtLMC)], ',')
  end
  close(fptr)
end

# ---------------------------------------------------------------------------
# obstacle list upload: CSpace.obstacles in list order (front first, R/list.jl:53-58)
function syncObstacles(tree::HipTree, S::TS) where {TS}
  m = S.obstacles.length
  cxyzr = Array{Float64}(undef, 4, m)       # column-major 4 x m == row-major m x 4 on the C side
  active = Vector{UInt8}(undef, m)
  sig = UInt64(m)
  ptr = S.obstacles.front
  for i = 1:m
    ob = ptr.data
    cxyzr[1:3, i] = ob.position[1:3]
    cxyzr[4, i] = ob.radius
    active[i] = (ob.obstacleUnused || ob.lifeSpan <= 0) ? 0x00 : 0x01   # R/DRRT_Q.jl:1777
    # (position included: the planner moves dynamic spheres in place, R/rrtqx.jl:453-584)
    sig = hash((objectid(ob), ob.position[1], ob.position[2], ob.position[3], ob.radius, active[i]), sig)
    ptr = ptr.child
  end
  # extend_candidates checks against the sphere list (RRTX_OPT_EXTEND_OBSTACLES = 8, value 0), whatever a
  # syncPolygonObstacles call in between has selected
  rrtx_check(tree, ccall((:rrtx_set_option, LIBRRTX), Cint, (Ptr{Cvoid}, Cint, Int64), tree.ctx, 8, 0))
  if sig != tree.obsSig
    GC.@preserve cxyzr active rrtx_check(tree, ccall((:rrtx_spheres_set, LIBRRTX), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{UInt8}, Cint), tree.ctx, cxyzr, active, m))
    tree.obsSig = sig
  end
end

# the same for List{Obstacle} (legacy 2-D / Dubins path): kinds 1, 3 and the moving kinds 6 / 7, whose
# Obstacle.path (rows dx, dy, t) goes up with rrtx_polygon_paths_set.  Call again whenever the host
# changed a path (changeObstacleDirection, R/DRRT.jl:370-443).  Edge / point checks then go through
# rrtx_edges_check / rrtx_points_check with kind = 1 and read time from the third coordinate.
function syncPolygonObstacles(tree::HipTree, S::TS) where {TS}
  # extend_candidates then checks against this list (RRTX_OPT_EXTEND_OBSTACLES = 8, value 1 = polygons)
  rrtx_check(tree, ccall((:rrtx_set_option, LIBRRTX), Cint, (Ptr{Cvoid}, Cint, Int64), tree.ctx, 8, 1))
  m = S.obstacles.length
  vertOff = zeros(Int32, m + 1); pathOff = zeros(Int32, m + 1)
  vxy = Float64[]; pxyt = Float64[]
  cr = Array{Float64}(undef, 3, m)
  kind = Vector{UInt8}(undef, m); active = Vector{UInt8}(undef, m)
  ptr = S.obstacles.front
  for i = 1:m
    ob = ptr.data
    moving = (ob.kind == 6 || ob.kind == 7)
    kind[i] = ob.kind
    active[i] = (ob.obstacleUnused || ob.lifeSpan <= 0) ? 0x00 : 0x01
    cr[1:2, i] = ob.position[1:2]; cr[3, i] = ob.radius
    poly = moving ? ob.originalPolygon : (ob.kind == 1 ? zeros(0, 2) : ob.polygon)
    for v = 1:size(poly, 1)
      push!(vxy, poly[v, 1], poly[v, 2])
    end
    vertOff[i + 1] = vertOff[i] + size(poly, 1)
    if moving
      for v = 1:size(ob.path, 1)
        push!(pxyt, ob.path[v, 1], ob.path[v, 2], ob.path[v, 3])
      end
    end
    pathOff[i + 1] = pathOff[i] + (moving ? size(ob.path, 1) : 0)
    ptr = ptr.child
  end
  GC.@preserve vertOff vxy cr kind active rrtx_check(tree, ccall((:rrtx_polygons_set, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Ptr{UInt8}, Cint),
      tree.ctx, vertOff, vxy, cr, kind, active, m))
  GC.@preserve pathOff pxyt rrtx_check(tree, ccall((:rrtx_polygon_paths_set, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Int32}, Ptr{Cdouble}, Cint), tree.ctx, pathOff, pxyt, m))
end

# explicitEdgeCheck(C, edge) (R/DRRT_Q.jl:1802-1826) for SimpleEdge on sphere obstacles.
# The tree travels in a global per agent because the reference signature has no tree argument.
const HIP_TREE_OF = IdDict{Any, Any}()          # CSpace -> HipTree (set once per agent)
bindTree(S, tree::HipTree) = (HIP_TREE_OF[S] = tree)

function explicitEdgeCheckHip(S::TS, startPos::Array{Float64}, endPos::Array{Float64}, which::Int) where {TS}
  tree = HIP_TREE_OF[S]
  syncObstacles(tree, S)
  p0 = vec(startPos); p1 = vec(endPos)
  hit = Ref{UInt8}(0)
  GC.@preserve p0 p1 rrtx_check(tree, ccall((:rrtx_edges_check, LIBRRTX), Cint,
      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cint, Ref{UInt8}, Ptr{Int32}),
      tree.ctx, 0, p0, p1, 1, S.robotRadius, which, hit, C_NULL))
  return hit[] != 0x00
end

function explicitEdgeCheck(S::CSpace{T}, edge::SimpleEdge, verbose::Bool = false) where {T}
  if S.inWarmupTime                                # R/DRRT_Q.jl:1805-1807
    return false
  end
  return explicitEdgeCheckHip(S, edge.startNode.position, edge.endNode.position, -1)
end

# explicitEdgeCheck(S, edge, obstacle) (R/DRRT_SimpleEdge_functions.jl:210-212)
function explicitEdgeCheck(S::CSpace{T}, edge::SimpleEdge, obstacle::SphereObstacle) where {T}
  which = -1
  ptr = S.obstacles.front
  for i = 1:S.obstacles.length
    if ptr.data === obstacle
      which = i - 1
      break
    end
    ptr = ptr.child
  end
  which >= 0 || error("obstacle is not in CSpace.obstacles")
  return explicitEdgeCheckHip(S, edge.startNode.position, edge.endNode.position, which)
end

# explicitPointCheck (R/DRRT_Q.jl:1520-1556) -> (Bool, Float64)
function explicitPointCheck(S::CSpace{T}, point::Array{Float64}) where {T}
  if S.inWarmupTime
    return (false, Inf)
  end
  tree = HIP_TREE_OF[S]
  syncObstacles(tree, S)
  p = vec(point)
  unsafe = Ref{UInt8}(0); clr = Ref{Float64}(0.0)
  GC.@preserve p rrtx_check(tree, ccall((:rrtx_points_check, LIBRRTX), Cint,
      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Int64, Cdouble, Cint, Ref{UInt8}, Ref{Float64}),
      tree.ctx, 0, p, 1, S.robotRadius, 1, unsafe, clr))
  return (unsafe[] != 0x00, clr[])
end

# explicitPointCheck3D (R/DRRT_Q.jl:1558-1590): the root check of the 3-D driver, no quick pass
function explicitPointCheck3D(S::CSpace{T}, point::Array{Float64}) where {T}
  if S.inWarmupTime
    return (false, Inf)
  end
  tree = HIP_TREE_OF[S]
  syncObstacles(tree, S)
  p = vec(point)
  unsafe = Ref{UInt8}(0); clr = Ref{Float64}(0.0)
  GC.@preserve p rrtx_check(tree, ccall((:rrtx_points_check, LIBRRTX), Cint,
      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Int64, Cdouble, Cint, Ref{UInt8}, Ref{Float64}),
      tree.ctx, 0, p, 1, S.robotRadius, 0, unsafe, clr))
  return (unsafe[] != 0x00, clr[])
end

# explicitNodeCheck / explicitNodeCheck3D (R/DRRT_Q.jl:1594-1595)
explicitNodeCheck(S::CSpace{T}, node::RRTNode{T}) where {T} = explicitPointCheck(S, node.position)
explicitNodeCheck3D(S::CSpace{T}, node::RRTNode{T}) where {T} = explicitPointCheck3D(S, node.position)

# explicitPointCheck for a space whose obstacles are polygon Obstacles (R/DRRT.jl:1434-1470): call
# syncPolygonObstacles(tree, S) after every change of the list, then this
function explicitPointCheckPolygons(S::TS, point::Array{Float64}) where {TS}
  if S.inWarmupTime
    return (false, Inf)
  end
  tree = HIP_TREE_OF[S]
  p = vec(point)
  unsafe = Ref{UInt8}(0); clr = Ref{Float64}(0.0)
  GC.@preserve p rrtx_check(tree, ccall((:rrtx_points_check, LIBRRTX), Cint,
      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Int64, Cdouble, Cint, Ref{UInt8}, Ref{Float64}),
      tree.ctx, 1, p, 1, S.robotRadius, 1, unsafe, clr))
  return (unsafe[] != 0x00, clr[])
end

# calculateTrajectory(S, ::SimpleEdge) (R/DRRT_SimpleEdge_functions.jl:177-181)
function calculateTrajectory(S::TS, edge::SimpleEdge) where {TS}
  tree = HIP_TREE_OF[S]
  s = vec(edge.startNode.position); g = vec(edge.endNode.position)
  d = Ref{Float64}(0.0); w = Ref{Float64}(0.0)
  GC.@preserve s g rrtx_check(tree, ccall((:rrtx_simple_steer, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ref{Float64}, Ref{Float64}), tree.ctx, s, g, 1, d, w))
  edge.dist = d[]
  edge.distOriginal = edge.dist
  edge.Wdist = w[]
end

# ---------------------------------------------------------------------------
# Batched preamble of extend()/findBestParent (R/DRRT_Q.jl:1927-1979, 2546-2642): one call per
# batch of samples returns, per sample, the neighbour list with SimpleEdge costs and both directed
# collision flags, the nearest node and the sample's own point check.  findBestParent/extend then
# only do their list and heap bookkeeping.
struct ExtendCandidates
  offsets::Vector{Int64}      # nq + 1
  idx::Vector{Int32}          # 0-based node indices (tree.nodes[idx + 1])
  cost::Vector{Float64}       # edge.dist for both directions (SimpleEdge)
  hitOut::Vector{UInt8}       # explicitEdgeCheck(newNode -> near)
  hitIn::Vector{UInt8}        # explicitEdgeCheck(near -> newNode)
  nearestIdx::Vector{Int32}
  nearestDist::Vector{Float64}
  sampleUnsafe::Vector{UInt8}
end

function extend_candidates(tree::HipTree, S::TS, positions::Array{Float64,2}, hyberBallRad::Float64) where {TS}
  syncObstacles(tree, S)
  nq = size(positions, 2)                     # d x nq, each sample contiguous
  offsets = Vector{Int64}(undef, nq + 1)
  nidx = Vector{Int32}(undef, nq); ndist = Vector{Float64}(undef, nq); unsafe = Vector{UInt8}(undef, nq)
  cap = 64 * nq
  while true
    idx = Vector{Int32}(undef, cap); cost = Vector{Float64}(undef, cap)
    hout = Vector{UInt8}(undef, cap); hin = Vector{UInt8}(undef, cap)
    needed = Ref{Int64}(0)
    rc = GC.@preserve positions offsets idx cost hout hin nidx ndist unsafe ccall((:rrtx_extend_candidates, LIBRRTX), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cdouble, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}, Ptr{UInt8}, Ptr{UInt8},
         Int64, Ref{Int64}, Ptr{Int32}, Ptr{Cdouble}, Ptr{UInt8}),
        tree.ctx, positions, nq, hyberBallRad, S.robotRadius, offsets, idx, cost, hout, hin, cap, needed, nidx, ndist, unsafe)
    if rc == RRTX_E_CAPACITY
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    n = Int(needed[])
    return ExtendCandidates(offsets, idx[1:n], cost[1:n], hout[1:n], hin[1:n], nidx, ndist, unsafe)
  end
end

# ---------------------------------------------------------------------------
# addNewObstacle's edge loop (R/DRRT_Q.jl:3220-3290) against a device mirror of the planner's
# directed edges.  registerEdges is called where the planner creates edges (makeNeighborOf,
# makeInitialOutNeighborOf, makeParentOf); it returns the id of the first edge, ids are
# consecutive, the caller keeps `edges[id + 1]`.  obstacleSweep returns the 0-based ids of the
# registered edges that start within robotRadius + delta + ob.radius of `ob` and for which
# explicitEdgeCheck(S, edge, ob) is true; the caller sets their dist = Inf and updates its queues.
function registerEdges(tree::HipTree, edges::Vector{TE}) where {TE}
  n = length(edges)
  s = Int32[tree.indexOf[e.startNode] for e in edges]
  g = Int32[tree.indexOf[e.endNode] for e in edges]
  first = Ref{Int64}(0)
  GC.@preserve s g rrtx_check(tree, ccall((:rrtx_graph_edges_append, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Int64, Ref{Int64}), tree.ctx, s, g, n, first))
  return Int(first[])
end

function obstacleSweep(tree::HipTree, S::TS, ob::SphereObstacle) where {TS}
  syncObstacles(tree, S)
  which = -1                                  # list position of ob (0-based)
  ptr = S.obstacles.front
  for i = 1:S.obstacles.length
    if ptr.data === ob
      which = i - 1
      break
    end
    ptr = ptr.child
  end
  which >= 0 || error("obstacle is not in CSpace.obstacles")
  cap = 4096
  while true
    ids = Vector{Int32}(undef, cap)
    needed = Ref{Int64}(0)
    rc = GC.@preserve ids ccall((:rrtx_obstacle_sweep, LIBRRTX), Cint,
        (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Ptr{Int32}, Int64, Ref{Int64}),
        tree.ctx, which, S.robotRadius + S.delta + ob.radius, S.robotRadius, ids, cap, needed)
    if rc == RRTX_E_CAPACITY
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    return ids[1:Int(needed[])]
  end
end

# The same for the POLYGON list (legacy planner, R/DRRT.jl:3048-3290; BASELINE config 5's discoverable / moving
# obstacles): findPointsInConflictWithObstacle(::Obstacle) -- Euclidean query, the Dubins one ([x y 0.0 pi], range +
# pi), one query per path segment for kinds 6 / 7 -- and the edge loop of addNewObstacle (remove = false) or
# removeObstacle (remove = true: blocked edges that collide with ob and with no other obstacle in use) in ONE call;
# the edge type is the tree's (d = 3 SimpleEdge, d = 4 DubinsEdge with S.minTurningRadius).
function obstacleSweep(tree::HipTree, S::TS, ob::Obstacle, remove::Bool = false) where {TS}
  syncPolygonObstacles(tree, S)
  which = -1
  ptr = S.obstacles.front
  for i = 1:S.obstacles.length
    if ptr.data === ob
      which = i - 1
      break
    end
    ptr = ptr.child
  end
  which >= 0 || error("obstacle is not in CSpace.obstacles")
  cap = 4096
  while true
    ids = Vector{Int32}(undef, cap)
    needed = Ref{Int64}(0)
    rc = GC.@preserve ids ccall((:rrtx_obstacle_sweep_polygon, LIBRRTX), Cint,
        (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cint, Ptr{Int32}, Int64, Ref{Int64}),
        tree.ctx, which, S.robotRadius, S.delta, S.minTurningRadius, remove ? 1 : 0, ids, cap, needed)
    if rc == RRTX_E_CAPACITY
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    return ids[1:Int(needed[])]
  end
end

# edge.dist of registered edges first_id, first_id+1, ... (ids from registerEdges); registerEdges itself
# gives every edge the SimpleEdge cost of its two nodes.
function syncEdgeCosts(tree::HipTree, first_id::Int, edges::Vector{TE}) where {TE}
  d = Float64[e.dist for e in edges]
  GC.@preserve d rrtx_check(tree, ccall((:rrtx_graph_edges_set_dist, LIBRRTX), Cint,
      (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Int64), tree.ctx, first_id, d, length(d)))
end

# addNewObstacle's `edge.dist = Inf` (R/DRRT_Q.jl:3249) for the ids obstacleSweep returned
function blockEdges(tree::HipTree, ids::Vector{Int32})
  GC.@preserve ids rrtx_check(tree, ccall((:rrtx_graph_edges_block, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Int32}, Int64), tree.ctx, ids, length(ids)))
end

# The state propogateDescendants + reduceInconsistency (R/DRRT_Q.jl:2703-2817) reach with changeThresh = 0.0
# once the queue is empty: rrtLMC of every node, and the registered id of its parent edge (-1: root / orphan).
# The caller writes them back: node.rrtLMC = node.rrtTreeCost = lmc[i+1]; makeParentOf along edge parent[i+1].
function costToRoot(tree::HipTree, root)
  n = length(tree.nodes)
  lmc = Vector{Float64}(undef, n)
  parent = Vector{Int32}(undef, n)
  passes = Ref{Int32}(0)
  GC.@preserve lmc parent rrtx_check(tree, ccall((:rrtx_graph_cost_to_root, LIBRRTX), Cint,
      (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Int32}, Ref{Int32}), tree.ctx, tree.indexOf[root], lmc, parent, passes))
  return lmc, parent
end


# ---------------------------------------------------------------------------
# Edge = DubinsEdge (R/DRRT_DubinsEdge.jl, R/DRRT_DubinsEdge_functions.jl; README's per-edge-type
# contract, R/README.txt:85-99).  The tree is a HipTree{RRTNode{Float64}}(4, KDdist, [4], [2pi]) and the
# obstacle list a List{Obstacle} (syncPolygonObstacles).  Call syncDubinsSpace(tree, S) once after the
# CSpace is configured (and again when spaceHasTime or the velocity bounds change).
function syncDubinsSpace(tree::HipTree, S::TS) where {TS}
  # RRTX_OPT_SPACE_HAS_TIME = 12: CSpace.spaceHasTime (R/DRRT_data_structures.jl:330)
  rrtx_check(tree, ccall((:rrtx_set_option, LIBRRTX), Cint, (Ptr{Cvoid}, Cint, Int64), tree.ctx, 12, S.spaceHasTime ? 1 : 0))
  rrtx_check(tree, ccall((:rrtx_set_dubins_velocity, LIBRRTX), Cint, (Ptr{Cvoid}, Cdouble, Cdouble),
      tree.ctx, S.dubinsMinVelocity, S.dubinsMaxVelocity))
end

# calculateTrajectory(S, ::DubinsEdge) (R/DRRT_DubinsEdge_functions.jl:329-709): dubinsType, Wdist, dist,
# distOriginal, velocity (space with time) and the discretised trajectory (P x 2, or P x 3 with time)
function calculateTrajectory(S::TS, edge::DubinsEdge) where {TS}
  tree = HIP_TREE_OF[S]
  s = vec(convert(Array{Float64}, edge.startNode.position)); g = vec(convert(Array{Float64}, edge.endNode.position))
  d = Ref{Float64}(0.0); w = Ref{Float64}(0.0); v = Ref{Float64}(0.0)
  word = Vector{UInt8}(undef, 3); ok = Ref{UInt8}(0)
  GC.@preserve s g word rrtx_check(tree, ccall((:rrtx_dubins_steer_full, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ref{Float64}, Ref{Float64}, Ref{Float64}, Ptr{UInt8}, Ref{UInt8}),
      tree.ctx, s, g, 1, S.minTurningRadius, d, w, v, word, ok))
  edge.dubinsType = String(copy(word))
  edge.Wdist = w[]
  edge.dist = d[]
  edge.distOriginal = edge.dist
  if S.spaceHasTime
    edge.velocity = v[]
  end
  if edge.Wdist == Inf                           # no trajectory is built (:661-662)
    return
  end
  # the row width is the CONTEXT's, not S's (the two can drift apart until syncDubinsSpace runs again); the call
  # refuses a width that is not the context's instead of overrunning `rows`
  hasTime = Ref{Int64}(0)
  rrtx_check(tree, ccall((:rrtx_get_option, LIBRRTX), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), tree.ctx, 12, hasTime))
  cols = hasTime[] != 0 ? 3 : 2
  off = Vector{Int64}(undef, 2)
  cap = 256
  while true
    rows = Array{Float64}(undef, cols, cap)      # column-major cols x cap == row-major cap x cols on the C side
    needed = Ref{Int64}(0)
    rc = GC.@preserve s g off rows ccall((:rrtx_dubins_trajectory, LIBRRTX), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Int64}, Ptr{Cdouble}, Cint, Int64, Ref{Int64}),
        tree.ctx, s, g, 1, S.minTurningRadius, off, rows, cols, cap, needed)
    if rc == RRTX_E_CAPACITY
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    edge.trajectory = copy(transpose(rows[:, 1:Int(needed[])]))
    return
  end
end

# validMove(S, ::DubinsEdge) (R/DRRT_DubinsEdge_functions.jl:115-125): host arithmetic on what
# calculateTrajectory stored, exactly the reference's expression
function validMove(S::TS, edge::DubinsEdge) where {TS}
  if S.spaceHasTime
    return ((edge.startNode.position[3] > edge.endNode.position[3]) && (S.dubinsMinVelocity <= edge.velocity <= S.dubinsMaxVelocity))
  end
  return true
end

# explicitEdgeCheck(C, edge) over the whole obstacle list (R/DRRT.jl:1660-1678 with the two-stage test of
# R/DRRT_DubinsEdge_functions.jl:750-774 per obstacle): one call steers and checks
function explicitEdgeCheck(S::CSpace{T}, edge::DubinsEdge, verbose::Bool = false) where {T}
  if S.inWarmupTime
    return false
  end
  tree = HIP_TREE_OF[S]
  s = vec(convert(Array{Float64}, edge.startNode.position)); g = vec(convert(Array{Float64}, edge.endNode.position))
  cost = Ref{Float64}(0.0); hit = Ref{UInt8}(0)
  GC.@preserve s g rrtx_check(tree, ccall((:rrtx_dubins_edges_check, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble, Ref{Float64}, Ptr{UInt8}, Ref{UInt8}, Ptr{Int32}),
      tree.ctx, s, g, 1, S.minTurningRadius, S.robotRadius, cost, C_NULL, hit, C_NULL))
  return hit[] != 0x00
end

# explicitEdgeCheck(S, edge::DubinsEdge, obstacle) (R/DRRT_DubinsEdge_functions.jl:750-774) against ONE
# obstacle of the list: steering, the inflated chord test and every piece of the trajectory in one device call
# (round 2 composed it from two rrtx_edges_check launches per (edge, obstacle) on the host)
function explicitEdgeCheck(S::CSpace{T}, edge::DubinsEdge, obstacle::Obstacle) where {T}
  tree = HIP_TREE_OF[S]
  which = -1
  ptr = S.obstacles.front
  for i = 1:S.obstacles.length
    if ptr.data === obstacle
      which = i - 1
      break
    end
    ptr = ptr.child
  end
  which >= 0 || error("obstacle is not in CSpace.obstacles")
  s = vec(convert(Array{Float64}, edge.startNode.position)); g = vec(convert(Array{Float64}, edge.endNode.position))
  hit = Ref{UInt8}(0)
  GC.@preserve s g rrtx_check(tree, ccall((:rrtx_dubins_edges_check_obstacle, LIBRRTX), Cint,
      (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Cdouble, Cint, Ref{UInt8}),
      tree.ctx, s, g, 1, S.minTurningRadius, S.robotRadius, which, hit))
  return hit[] != 0x00
end

# Batched preamble of extend()/findBestParent for Edge = DubinsEdge (BASELINE configs 3 and 5): per sample
# the wrapped range search, both directed edges steered and checked; in a space with time a flag byte
# also carries bit 1 = !validMove, so `hit != 0` is findBestParent's whole blocking test (R/DRRT_Q.jl:1960).
struct ExtendCandidatesDubins
  offsets::Vector{Int64}
  idx::Vector{Int32}
  key::Vector{Float64}        # KDdist of the range search
  costOut::Vector{Float64}    # edge.dist newNode -> near
  costIn::Vector{Float64}     # edge.dist near -> newNode
  hitOut::Vector{UInt8}
  hitIn::Vector{UInt8}
  nearestIdx::Vector{Int32}
  nearestDist::Vector{Float64}
  sampleUnsafe::Vector{UInt8}
end

function extend_candidates_dubins(tree::HipTree, S::TS, positions::Array{Float64,2}, hyberBallRad::Float64) where {TS}
  nq = size(positions, 2)                     # 4 x nq
  offsets = Vector{Int64}(undef, nq + 1)
  nidx = Vector{Int32}(undef, nq); ndist = Vector{Float64}(undef, nq); unsafe = Vector{UInt8}(undef, nq)
  cap = 2048 * nq
  while true
    idx = Vector{Int32}(undef, cap); key = Vector{Float64}(undef, cap)
    cout = Vector{Float64}(undef, cap); cin = Vector{Float64}(undef, cap)
    hout = Vector{UInt8}(undef, cap); hin = Vector{UInt8}(undef, cap)
    needed = Ref{Int64}(0)
    rc = GC.@preserve positions offsets idx key cout cin hout hin nidx ndist unsafe ccall((:rrtx_extend_candidates_dubins, LIBRRTX), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cdouble, Cdouble, Ptr{Int64}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Ptr{UInt8}, Int64, Ref{Int64}, Ptr{Int32}, Ptr{Cdouble}, Ptr{UInt8}),
        tree.ctx, positions, nq, hyberBallRad, S.robotRadius, S.minTurningRadius, offsets, idx, key, cout, cin,
        C_NULL, C_NULL, hout, hin, cap, needed, nidx, ndist, unsafe)
    if rc == RRTX_E_CAPACITY
      cap = Int(needed[])
      continue
    end
    rrtx_check(tree, rc)
    n = Int(needed[])
    return ExtendCandidatesDubins(offsets, idx[1:n], key[1:n], cout[1:n], cin[1:n], hout[1:n], hin[1:n], nidx, ndist, unsafe)
  end
end

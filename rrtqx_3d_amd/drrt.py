"""Host-side mirror of the reference's function-level interface for the hot path.

The reference is Julia; its seam is a set of free functions chosen by multiple
dispatch (R/README.txt:85-99; R/ = code_RRTQx_3D/).  Julia is not available in
the build image, so this module restates that interface in Python with the same
names, argument order and error behaviour, every call going through the C-ABI of
include/rrtx.h (the boundary julia/RRTXHip.jl binds with ccall).  There is no CPU
implementation behind these names: without the HIP library / a GPU they raise.

    reference                                   here
    ------------------------------------------  -------------------------------
    KDTree{T}(d, f[, wraps, wrapPoints])         KDTree(d, f=None, wraps, wrapPoints)
      (R/kdTree_general.jl:94-112)
    kdInsert / kdFindNearest / kdFindWithinRange same names
    kdFindMoreWithinRange / popFromRangeList /   same names
      emptyRangeList (:121-170, 357-385, 774-955)
    JList + JlistPush/JlistPop... (R/jlist.jl)   JList (front/back/length, keys)
    SphereObstacle, Obstacle, CSpace             same names, the fields the path reads
      (R/DRRT_data_structures.jl:135-397)
    SimpleEdge / DubinsEdge, newEdge,            same names
      calculateTrajectory, validMove
    explicitEdgeCheck(S, edge[, ob]),            same names (+ batched plural forms)
      explicitPointCheck, explicitNodeCheck
      (R/DRRT_Q.jl:1520-1595, 1775-1826)
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .context import Context

Inf = float("inf")


def error(msg: str):
    """Julia's error(): the reference's only failure idiom (R/rrtqx.jl:70,91)."""
    raise RuntimeError(msg)


# ------------------------------------------------------------------ JList ----
class JListNode:
    __slots__ = ("child", "parent", "data", "key")

    def __init__(self):
        self.child = self
        self.parent = self
        self.data = None
        self.key = 0.0


class JList:
    """Doubly linked list with keys (R/jlist.jl:30-197); JlistPush inserts at the front."""

    def __init__(self):
        end = JListNode()
        self.front = end
        self.back = end
        self.bound = end
        self.length = 0

    def __iter__(self):
        n = self.front
        for _ in range(self.length):
            yield n
            n = n.child

    def items(self):
        return [(n.data, n.key) for n in self]


def JlistPush(L: JList, data, key: float = 0.0):
    n = JListNode()
    n.parent = L.front.parent
    n.child = L.front
    if L.length == 0:
        L.back = n
    else:
        L.front.parent = n
    n.data = data
    n.key = key
    L.front = n
    L.length += 1


def JlistPopKey(L: JList):
    if L.length == 0:
        return (None, -1.0)     # R/jlist.jl returns a sentinel on an empty list
    old = L.front
    L.front = old.child
    L.length -= 1
    if L.length == 0:
        L.back = L.bound
        L.front = L.bound
    else:
        L.front.parent = old.parent
    return (old.data, old.key)


def JlistPop(L: JList):
    return JlistPopKey(L)[0]


# ------------------------------------------------------------------ nodes ----
class RRTNode:
    """The fields of RRTNode{T} the hot path touches (R/DRRT_data_structures.jl:22-130)."""

    def __init__(self, position=None):
        self.kdInTree = False
        self.inHeap = False            # "inHeap is a misnomer since this is a list" (R/kdTree_general.jl:769)
        self.position = None if position is None else np.asarray(position, dtype=np.float64).reshape(1, -1)
        self.index = -1                # insertion order == device index (the reference has no ids)
        self.data = 0.0                # key of the k-NN heap (addToKNNHeap, R/kdTree_general.jl:585)
        self.rrtLMC = Inf
        self.rrtTreeCost = Inf
        self.rrtParentUsed = False
        self.rrtParentEdge = None
        self.tempEdge = None


class HipTree:
    """Stands in for KDTree{T}: same fields the planner reads (d, treeSize, root, numWraps)."""

    def __init__(self, d: int, distanceFunction=None, wraps: Sequence[int] = (), wrapPoints: Sequence[float] = (),
                 device: int = 0):
        self.d = d
        self.distanceFunction = distanceFunction
        self.treeSize = 0
        self.numWraps = len(wraps)
        self.wraps = list(wraps)              # 1-based dimension numbers, as in the reference
        self.wrapPoints = list(wrapPoints)
        self.root: Optional[RRTNode] = None
        self.nodes: List[RRTNode] = []
        self.ctx = Context(d, device=device)
        for w, p in zip(self.wraps, self.wrapPoints):
            self.ctx.set_wrap(int(w) - 1, float(p))


def KDTree(d: int, f=None, wraps: Sequence[int] = (), wrapPoints: Sequence[float] = (), device: int = 0) -> HipTree:
    return HipTree(d, f, wraps, wrapPoints, device)


def kdInsert(tree: HipTree, node: RRTNode):
    """R/kdTree_general.jl:121-170"""
    if node.kdInTree:
        return
    node.kdInTree = True
    first = tree.ctx.nodes_append(node.position)
    node.index = first
    tree.nodes.append(node)
    if tree.treeSize == 0:
        tree.root = node
    tree.treeSize += 1


def kdInsertMany(tree: HipTree, nodes: Iterable[RRTNode]):
    nodes = [n for n in nodes if not n.kdInTree]
    if not nodes:
        return
    pos = np.concatenate([n.position for n in nodes], axis=0)
    first = tree.ctx.nodes_append(pos)
    for k, n in enumerate(nodes):
        n.kdInTree = True
        n.index = first + k
        tree.nodes.append(n)
    if tree.treeSize == 0:
        tree.root = nodes[0]
    tree.treeSize += len(nodes)


def _q(tree: HipTree, queryPoint) -> np.ndarray:
    q = np.asarray(queryPoint, dtype=np.float64).reshape(-1)
    if q.shape[0] != tree.d:
        error(f"query point has {q.shape[0]} coordinates, tree has {tree.d}")
    return q.reshape(1, -1)


def kdFindNearest(tree: HipTree, queryPoint) -> Tuple[RRTNode, float]:
    """R/kdTree_general.jl:357-385"""
    idx, dist = tree.ctx.nn_nearest(_q(tree, queryPoint))
    return tree.nodes[int(idx[0])], float(dist[0])


def kdFindNearestWithGuesstree(tree: HipTree, queryPoint, guess: RRTNode) -> Tuple[RRTNode, float]:
    """R/kdTree_general.jl:503-534: the search seeded at `guess` instead of the root.  The seed only
    shortens the reference's descent; the answer is kdFindNearest's (on an exact distance tie the
    reference keeps the guess, this returns the lowest index).  No caller in the reference."""
    return kdFindNearest(tree, queryPoint)


def kdFindKNearest(tree: HipTree, k: int, queryPoint) -> List[RRTNode]:
    """R/kdTree_general.jl:696-723: the nodes of the final heap, each with `.data` = its distance.
    Like the reference this is max(k, 2) nodes (the heap is seeded with root + dummy) and raises
    for a wrapped space; the order here is ascending distance (the reference's is heap order)."""
    idx, dist, count = tree.ctx.nn_knearest(_q(tree, queryPoint), k)
    out = []
    for i, d in zip(idx[0, :int(count[0])], dist[0, :int(count[0])]):
        n = tree.nodes[int(i)]
        n.data = float(d)
        out.append(n)
    return out


def _push_range(tree: HipTree, L: JList, idx: np.ndarray, key: np.ndarray):
    # addToRangeList (R/kdTree_general.jl:765-771): skip nodes already in the list
    for i, k in zip(idx, key):
        n = tree.nodes[int(i)]
        if n.inHeap:
            continue
        n.inHeap = True
        JlistPush(L, n, float(k))


def kdFindWithinRange(tree: HipTree, range_: float, queryPoint) -> JList:
    """R/kdTree_general.jl:889-919.  The list holds (node, key = distance); its ORDER is by node
    index here (the reference's order is reverse kd-discovery order); membership and keys match."""
    L = JList()
    return kdFindMoreWithinRange(tree, range_, queryPoint, L)


def kdFindMoreWithinRange(tree: HipTree, range_: float, queryPoint, L: JList) -> JList:
    """R/kdTree_general.jl:927-955"""
    off, idx, key = tree.ctx.nn_radius(_q(tree, queryPoint), float(range_))
    _push_range(tree, L, idx, key)
    return L


def kdFindWithinRangeBatch(tree: HipTree, range_, queryPoints) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Batched form: CSR (offsets, node indices, keys) for many query points at once."""
    return tree.ctx.nn_radius(np.asarray(queryPoints, dtype=np.float64).reshape(-1, tree.d), range_)


def popFromRangeList(L: JList):
    node, key = JlistPopKey(L)
    node.inHeap = False
    return (node, key)


def emptyRangeList(L: JList):
    while L.length > 0:
        node, _ = JlistPopKey(L)
        node.inHeap = False


# -------------------------------------------------------------- obstacles ----
class SphereObstacle:
    """R/DRRT_data_structures.jl:267-307"""

    def __init__(self, position, radius: Optional[float] = None):
        p = np.asarray(position, dtype=np.float64).reshape(-1)
        if radius is None:               # SphereObstacle(pos::Array) with pos = [x y z r]
            radius = float(p[3])
            p = p[:3]
        self.startTime = 0.0
        self.lifeSpan = Inf
        self.obstacleUnused = False
        self.expired = False
        self.senseableObstacle = False
        self.obstacleUnusedAfterSense = True
        self.position = p.copy()
        self.radius = float(radius)
        self.radiusWithoutAug = float(radius)


class Obstacle:
    """Polygon / ball obstacle of the legacy 2-D path (R/DRRT_data_structures.jl:135-265)."""

    def __init__(self, kind: int, a, b: Optional[float] = None):
        self.kind = int(kind)
        self.startTime = 0.0
        self.lifeSpan = Inf
        self.obstacleUnused = False
        self.senseableObstacle = False
        self.obstacleUnusedAfterSense = True
        if self.kind == 1:
            self.position = np.asarray(a, dtype=np.float64).reshape(-1)[:2].copy()
            self.radius = float(b)
            self.polygon = np.zeros((0, 2))
        elif self.kind in (3, 6, 7):
            self.polygon = np.asarray(a, dtype=np.float64).reshape(-1, 2).copy()
            self.position = None       # centre / radius come from the library's ctor restatement (:229-241)
            self.radius = None
            if self.kind != 3:         # polygon moving in time (:140-143, 181-192); set by the file readers
                self.velocity = 0.0
                self.path = np.zeros((0, 3))           # rows (dx, dy, t) the robot knows about
                self.originalPolygon = self.polygon
                self.unknownPath = np.zeros((0, 3))    # kind 7: what the obstacle will really do
                self.nextDirectionChangeTime = -Inf
        else:
            error("need to impliment this")   # the reference's message for unsupported kinds (R/DRRT.jl:1546)


class List_:
    """R/list.jl: listPush inserts at the front; iteration is front -> back."""

    def __init__(self):
        self._items: list = []

    @property
    def length(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)


def listPush(L: List_, data):
    L._items.insert(0, data)


class CSpace:
    """The fields of CSpace{T} the path reads (R/DRRT_data_structures.jl:314-397)."""

    def __init__(self, D: int, ObsDelta: float, L, U, S, G):
        self.d = D
        self.obstacles = List_()
        self.obsDelta = ObsDelta
        self.lowerBounds = np.asarray(L, dtype=np.float64)
        self.upperBounds = np.asarray(U, dtype=np.float64)
        self.width = self.upperBounds - self.lowerBounds
        self.start = np.asarray(S, dtype=np.float64)
        self.goal = np.asarray(G, dtype=np.float64)
        self.spaceHasTime = False
        self.spaceHasTheta = False
        self.robotRadius = 0.0
        self.delta = 0.0
        self.minTurningRadius = 0.0
        self.inWarmupTime = False
        self.warmupTime = 0.0
        self._sig = None
        self._ctx: Optional[Context] = None    # device context holding this space's obstacle tables

    def bind(self, tree: "HipTree"):
        """Share the tree's device context (one ctx per agent: tree + obstacle lists)."""
        if self._ctx is not tree.ctx:
            self._ctx = tree.ctx
            self._sig = None
        return self

    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = Context(self.d)
        return self._ctx


def addObsToCSpace(S: CSpace, ob):
    listPush(S.obstacles, ob)


def readTimeObstaclesFromfile(S: CSpace, filename: str, obsMult: int = 1):
    """R/DRRT_Q.jl:1022-1061: polygons that move along a known path (kind 6)."""
    from . import envio
    env = envio.read_time_obstacles(filename)
    for poly, speed, path in zip(env.polygons, env.speed, env.paths):
        for _ in range(obsMult):
            ob = Obstacle(6, poly)
            ob.senseableObstacle = False
            ob.obstacleUnusedAfterSense = False
            ob.obstacleUnused = False
            ob.velocity = float(speed)
            ob.path = path.copy()
            addObsToCSpace(S, ob)


def _is_active(ob) -> bool:
    return not (ob.obstacleUnused or ob.lifeSpan <= 0)     # R/DRRT_Q.jl:1777


def _sync_obstacles(S: CSpace) -> int:
    """Mirror CSpace.obstacles (list order) into the ctx; returns kind (0 spheres, 1 polygons)."""
    ctx = S.ctx
    obs = list(S.obstacles)
    poly = any(isinstance(o, Obstacle) for o in obs)
    if poly and any(isinstance(o, SphereObstacle) for o in obs):
        error("CSpace.obstacles mixes SphereObstacle and Obstacle")
    if poly:
        sig = ("p", tuple((id(o), o.kind, _is_active(o), o.radius if o.kind == 1 else o.polygon.tobytes(),
                           np.asarray(o.path, dtype=np.float64).tobytes() if o.kind in (6, 7) else b"") for o in obs))
    else:
        sig = ("s", tuple((id(o), o.radius, _is_active(o), tuple(o.position)) for o in obs))
    key = (id(ctx), sig)
    if S._sig != key:
        if poly:
            polys = [o.polygon if o.kind != 1 else np.zeros((0, 2)) for o in obs]
            cr = None
            if any(o.kind == 1 for o in obs):
                from . import _capi  # noqa: F401  (centre/radius must be given for balls)
                cr = np.zeros((len(obs), 3))
                for i, o in enumerate(obs):
                    if o.kind == 1:
                        cr[i] = [o.position[0], o.position[1], o.radius]
                    else:
                        v = o.polygon
                        px = (v[:, 0].max() + v[:, 0].min()) / 2.0
                        py = (v[:, 1].max() + v[:, 1].min()) / 2.0
                        cr[i] = [px, py, math.sqrt(((v - [px, py]) ** 2).sum(axis=1).max())]
            paths = [o.path if o.kind in (6, 7) else None for o in obs]
            ctx.polygons_set(polys, kinds=[o.kind for o in obs], active=[_is_active(o) for o in obs],
                             centre_radius=cr, paths=paths if any(p is not None for p in paths) else None)
        else:
            c = np.array([[*o.position[:3], o.radius] for o in obs], dtype=np.float64).reshape(-1, 4)
            ctx.spheres_set(c, [_is_active(o) for o in obs])
        S._sig = key
    return 1 if poly else 0


# ------------------------------------------------------------------ edges ----
class SimpleEdge:
    """R/DRRT_SimpleEdge.jl:34-56"""

    def __init__(self):
        self.startNode: Optional[RRTNode] = None
        self.endNode: Optional[RRTNode] = None
        self.dist = 0.0
        self.distOriginal = 0.0
        self.Wdist = 0.0


class DubinsEdge(SimpleEdge):
    """R/DRRT_DubinsEdge.jl:30-64"""

    def __init__(self):
        super().__init__()
        self.dubinsType = "xxx"
        self.velocity = 0.0
        self.trajectory = None


def newEdge(startNode: RRTNode, endNode: RRTNode, Edge=SimpleEdge):
    E = Edge()
    E.startNode = startNode
    E.endNode = endNode
    return E


def validMove(S: CSpace, edge) -> bool:
    """R/DRRT_SimpleEdge_functions.jl:94-104 / R/DRRT_DubinsEdge_functions.jl:115-125 (spaces without time)"""
    if S.spaceHasTime:
        error("spaces with time are outside the hot-path scope (SURVEY.md 8)")
    return True


def calculateTrajectories(S: CSpace, edges: Sequence[SimpleEdge]):
    if not edges:
        return
    ctx = S.ctx
    s = np.concatenate([e.startNode.position for e in edges], axis=0)
    g = np.concatenate([e.endNode.position for e in edges], axis=0)
    if isinstance(edges[0], DubinsEdge):
        cost, word = ctx.dubins_steer(s, g, S.minTurningRadius)
        for e, c, w in zip(edges, cost, word):
            e.dubinsType = w.decode()
            e.Wdist = float(c)
            e.dist = float(c)
            e.distOriginal = e.dist
    else:
        dist, wdist = ctx.simple_steer(s, g)
        for e, d, w in zip(edges, dist, wdist):
            e.dist = float(d)
            e.distOriginal = e.dist
            e.Wdist = float(w)


def calculateTrajectory(S: CSpace, edge):
    """R/DRRT_SimpleEdge_functions.jl:177-181, R/DRRT_DubinsEdge_functions.jl:329-709 (cost + word)"""
    calculateTrajectories(S, [edge])


def explicitEdgeChecks(S: CSpace, edges: Sequence[SimpleEdge], obstacle=None) -> np.ndarray:
    if S.inWarmupTime or not edges:
        return np.zeros(len(edges), dtype=bool)        # R/DRRT_Q.jl:1805-1807
    kind = _sync_obstacles(S)
    ctx = S.ctx
    which = -1
    if obstacle is not None:
        obs = list(S.obstacles)
        which = next((i for i, o in enumerate(obs) if o is obstacle), None)
        if which is None:
            error("obstacle is not in CSpace.obstacles")
    s = np.concatenate([e.startNode.position for e in edges], axis=0)
    g = np.concatenate([e.endNode.position for e in edges], axis=0)
    if isinstance(edges[0], DubinsEdge):
        if which >= 0:      # explicitEdgeCheck(S, edge::DubinsEdge, ob), R/DRRT_DubinsEdge_functions.jl:750-774
            return ctx.dubins_edges_check_obstacle(s, g, S.minTurningRadius, S.robotRadius, which).astype(bool)
        _, _, hit, _ = ctx.dubins_edges_check(s, g, S.minTurningRadius, S.robotRadius)
        return hit.astype(bool)
    hit, _ = ctx.edges_check(s, g, S.robotRadius, kind=kind, obstacle=which, want_first=False)
    return hit.astype(bool)


def explicitEdgeCheck(S: CSpace, edge, obstacle=None) -> bool:
    """explicitEdgeCheck(C, edge) (R/DRRT_Q.jl:1802-1826) / explicitEdgeCheck(S, edge, ob)
    (R/DRRT_SimpleEdge_functions.jl:210-212, R/DRRT_DubinsEdge_functions.jl:750-774)"""
    return bool(explicitEdgeChecks(S, [edge], obstacle)[0])


def explicitPointChecks(S: CSpace, points) -> Tuple[np.ndarray, np.ndarray]:
    p = np.asarray(points, dtype=np.float64).reshape(-1, S.d)
    if S.inWarmupTime:
        return np.zeros(len(p), dtype=bool), np.full(len(p), Inf)   # R/DRRT_Q.jl:1523-1525
    kind = _sync_obstacles(S)
    unsafe, clr = S.ctx.points_check(p, S.robotRadius, kind=kind, quick=True)
    return unsafe.astype(bool), clr


def explicitPointCheck(S: CSpace, point) -> Tuple[bool, float]:
    """R/DRRT_Q.jl:1520-1556"""
    u, c = explicitPointChecks(S, np.asarray(point, dtype=np.float64).reshape(1, -1))
    return bool(u[0]), float(c[0])


def explicitNodeCheck(S: CSpace, node: RRTNode) -> Tuple[bool, float]:
    """R/DRRT_Q.jl:1594"""
    return explicitPointCheck(S, node.position)


def hyberBallRad(treeSize: int, d: int, delta: float, ballConstant: float) -> float:
    """R/rrtqx.jl:382"""
    return min(delta, ballConstant * ((math.log(1 + treeSize) / treeSize) ** (1.0 / d)))


def extend_candidates(tree: HipTree, S: CSpace, newPositions, hyberBallRad_: float):
    """Batched preamble of extend()/findBestParent (R/DRRT_Q.jl:1927-1979, 2546-2642) for many
    samples at once: neighbour lists, SimpleEdge costs, both directed collision flags, nearest node
    and the sample's own point check.  The planner keeps the list/heap bookkeeping."""
    if tree.d != 3:
        error("extend_candidates is the SimpleEdge (3-D) path")
    S.bind(tree)
    kind = _sync_obstacles(S)          # 0: List{SphereObstacle}, 1: List{Obstacle} (polygons, (x, y) projection)
    from . import _capi
    tree.ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, kind)
    q = np.asarray(newPositions, dtype=np.float64).reshape(-1, 3)
    out = tree.ctx.extend_candidates(q, float(hyberBallRad_), S.robotRadius)
    if S.inWarmupTime:
        out["hit_out"][:] = 0
        out["hit_in"][:] = 0
        out["sample_unsafe"][:] = 0
    return out


# ------------------------------------------------------------ file dumps ----
# The reference's dump writers walk the kd-tree's child pointers (R/DRRT_Q.jl:250-364); with the device
# tree there are none, so these walk HipTree.nodes in insertion order instead.  Same rows, same number
# format (Julia's writedlm), different row order -- the MATLAB viewers read them as unordered sets
# (R/make_videoUAMExample.m:88-285).
def saveRRTTree(tree: HipTree, fileName: str):
    """R/DRRT_Q.jl:250-275: each node with a parent, followed by that parent (edge form)."""
    from . import envio
    with open(fileName, "w") as f:
        for n in tree.nodes:
            if n.rrtParentUsed:
                envio.writedlm_row(f, [*n.position.reshape(-1), n.rrtTreeCost])
                par = n.rrtParentEdge.endNode
                envio.writedlm_row(f, [*par.position.reshape(-1), par.rrtTreeCost])


def saveRRTNodes(tree: HipTree, fileName: str):
    """R/DRRT_Q.jl:310-333: position, rrtTreeCost, rrtLMC of every node."""
    from . import envio
    with open(fileName, "w") as f:
        for n in tree.nodes:
            envio.writedlm_row(f, [*n.position.reshape(-1), n.rrtTreeCost, n.rrtLMC])


def saveRRTNodesCollision(tree: HipTree, fileName: str):
    """R/DRRT_Q.jl:336-362: position and min(rrtTreeCost, rrtLMC)."""
    from . import envio
    with open(fileName, "w") as f:
        for n in tree.nodes:
            envio.writedlm_row(f, [*n.position.reshape(-1), min(n.rrtTreeCost, n.rrtLMC)])


# ------------------------------------------------------- obstacle sweeps ----
def findPointsInConflictWithObstacle(S: CSpace, KD: HipTree, ob, root=None) -> JList:
    """R/DRRT_Q.jl:3195-3215: nodes within robotRadius + delta + ob.radius of the obstacle."""
    if S.spaceHasTime:
        error("this type of obstacle not coded for this type of space")
    if not S.spaceHasTheta:
        searchRange = S.robotRadius + S.delta + ob.radius
        return kdFindWithinRange(KD, searchRange, ob.position)
    searchRange = S.robotRadius + S.delta + ob.radius + math.pi        # Dubins [x y 0 theta]
    centre = np.array([ob.position[0], ob.position[1], 0.0, math.pi])
    return kdFindWithinRange(KD, searchRange, centre)


def _list_position(S: CSpace, ob) -> int:
    for i, o in enumerate(S.obstacles):
        if o is ob:
            return i
    error("obstacle is not in CSpace.obstacles")


def obstacleSweepEdgeChecks(S: CSpace, KD: HipTree, edges: Sequence[SimpleEdge], ob=None, others_of=None,
                            timeElapsed: Optional[float] = None) -> np.ndarray:
    """Batched body of the edge loops in addNewObstacle / removeObstacle (R/DRRT_Q.jl:3220-3362).

      ob given        -> explicitEdgeCheck(S, edge, ob) for every edge (addNewObstacle :3248, :3257;
                         first test of removeObstacle :3321)
      others_of given -> "conflictsWithOtherObs": any obstacle other than `others_of` that is not
                         unused and, when timeElapsed is given, inside its time window
                         startTime <= timeElapsed <= startTime + lifeSpan (:3326-3337)
    Edges are graph edges between nodes already in the tree; only their node indices travel."""
    if not edges:
        return np.zeros(0, dtype=bool)
    S.bind(KD)
    _sync_obstacles(S)
    s = [e.startNode.index for e in edges]
    g = [e.endNode.index for e in edges]
    if ob is not None:
        hit, _ = S.ctx.edges_check_idx(s, g, S.robotRadius, obstacle=_list_position(S, ob), want_first=False)
        return hit.astype(bool)
    mask = []
    for o in S.obstacles:
        ok = (o is not others_of) and not o.obstacleUnused
        if ok and timeElapsed is not None:
            ok = o.startTime <= timeElapsed <= (o.startTime + o.lifeSpan)
        mask.append(1 if ok else 0)
    hit, _ = S.ctx.edges_check_idx(s, g, S.robotRadius, obstacle=-1, obstacle_mask=mask, want_first=False)
    return hit.astype(bool)


# ------------------------------------- device mirror of the planner's edges ----
def registerEdges(KD: HipTree, edges: Sequence[SimpleEdge]) -> int:
    """Mirror graph edges (what RRTNodeNeighborIterator walks: out-neighbour edges and parent
    edges) on the device; returns the id of the first one, ids are consecutive."""
    if not edges:
        return KD.ctx.n_graph_edges
    return KD.ctx.graph_edges_append([e.startNode.index for e in edges], [e.endNode.index for e in edges])


def obstacleSweep(S: CSpace, KD: HipTree, ob, remove: bool = False) -> np.ndarray:
    """The edge loop of addNewObstacle in one call: ids of the registered edges that start at a node of
    findPointsInConflictWithObstacle(S, KD, ob) and for which explicitEdgeCheck(S, edge, ob) is true; the caller
    sets those edges' dist = Inf and updates its queues.
      SphereObstacle (R/DRRT_Q.jl:3195-3290): range robotRadius + delta + ob.radius, SimpleEdge in 3-D;
      Obstacle (polygon list, R/DRRT.jl:3048-3200): the Euclidean query or the Dubins one ([x y 0.0 pi], range + pi),
        one query per path segment for obstacles that move in time; the edge type is the space's (SimpleEdge in a
        d = 3 tree, DubinsEdge with S.minTurningRadius in a d = 4 one).
    remove=True (polygon list only) is removeObstacle's loop (R/DRRT.jl:3202-3290): the registered edges that are
    blocked, collide with ob and with no other obstacle in use -- the caller resets them to distOriginal."""
    S.bind(KD)
    _sync_obstacles(S)
    if isinstance(ob, Obstacle):
        return KD.ctx.obstacle_sweep_polygon(_list_position(S, ob), S.robotRadius, S.delta,
                                             r_min=float(getattr(S, "minTurningRadius", 0.0) or 0.0), remove=remove)
    if remove:
        error("removeObstacle frees no edge for sphere obstacles (the reference marks ob unused before its edge loop, "
              "R/DRRT_Q.jl:3302); use obstacleSweepEdgeChecks for a corrected caller")
    if S.spaceHasTime or S.spaceHasTheta:
        error("this type of obstacle not coded for this type of space")
    return KD.ctx.obstacle_sweep(_list_position(S, ob), S.robotRadius + S.delta + ob.radius, S.robotRadius)


def syncEdgeCosts(KD: HipTree, first_id: int, edges: Sequence[SimpleEdge]):
    """edge.dist of registered edges first_id, first_id + 1, ... (registerEdges gives every edge the
    SimpleEdge cost of its two nodes; Dubins costs and costs in a space with time are sent with this)."""
    KD.ctx.graph_edges_set_dist(first_id, [e.dist for e in edges])


def blockEdges(KD: HipTree, ids):
    """addNewObstacle's `edge.dist = Inf` (R/DRRT_Q.jl:3249) for the ids obstacleSweep returned."""
    KD.ctx.graph_edges_block(ids)


def costToRoot(KD: HipTree, root: RRTNode) -> Tuple[np.ndarray, np.ndarray]:
    """The state propogateDescendants + reduceInconsistency (R/DRRT_Q.jl:2703-2817) reach with
    changeThresh = 0.0 once rrtXQueue is empty, computed over the registered edges: (rrtLMC per node in
    insertion order, registered id of each node's rrtParentEdge or -1 for the root and for orphans)."""
    lmc, parent, _ = KD.ctx.graph_cost_to_root(root.index)
    return lmc, parent

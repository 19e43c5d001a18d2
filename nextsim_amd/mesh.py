"""Synthetic meshes and their domain decomposition, in the reference's per-rank layout.

The reference reads Gmsh MSH-2.2 meshes that are not in its repository (mesh/README.md) and
partitions them with the Gmsh-embedded METIS (core/src/gmshmeshseq.cpp:470-540); both are out
of scope.  What the hot path needs is the *result*: per-rank arrays that obey
the ordering contract of core/src/gmshmesh.cpp:1165-1169 (owned nodes first, then ghosts, each
sorted by global id), :1379-1417 (owned triangles first, then ghost triangles), :1289-1301
(ghostNodes flags), FE.cpp:150-271 (boundary masks; M_mask_dirichlet is false on ghosts,
M_neumann_flags includes ghosts and is sorted) and FE.cpp:14003-14088 (halo lists ordered by
ascending global id per neighbour).

Meshes: seeded, jittered-structured triangulations
  toy   -- ~2k triangles, 100 km box, left/right coast (Dirichlet), top/bottom open (Neumann)
           (stands in for square_with_point.msh of config-files/nextsim.toy.cfg)
  10km  -- pan-Arctic-like disc of radius 2500 km with an irregular coast, ~10 km edge, ~60k tri
  2km   -- the same at ~2 km edge, ~1.5M triangles
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SEED = 20240501


@dataclass
class GlobalMesh:
    x: np.ndarray            # [Nn] float64, polar-stereographic metres
    y: np.ndarray
    tri: np.ndarray          # [Ne,3] int32, 0-based, counter-clockwise
    dirichlet: np.ndarray    # [Nn] bool: closed (coast) boundary nodes
    neumann: np.ndarray      # [Nn] bool: open boundary nodes
    lat: np.ndarray          # [Nn] degrees north
    name: str = ""

    @property
    def num_nodes(self) -> int:
        return int(self.x.size)

    @property
    def num_elements(self) -> int:
        return int(self.tri.shape[0])

    def resolution(self) -> float:
        """FiniteElement::resolution (FE.cpp:1845-1861): sqrt of the mean element area."""
        x, y, t = self.x, self.y, self.tri
        jac = (x[t[:, 1]] - x[t[:, 0]]) * (y[t[:, 2]] - y[t[:, 0]]) - (x[t[:, 2]] - x[t[:, 0]]) * (y[t[:, 1]] - y[t[:, 0]])
        return float(np.sqrt(np.mean(0.5 * np.abs(jac))))


@dataclass
class LocalMesh:
    """One rank's view, exactly the arrays of include/nxs_dyn.h::nxs_dyn_mesh + nxs_dyn_halo."""
    rank: int
    nranks: int
    num_nodes: int
    num_elements: int
    local_ndof: int
    local_nelements: int
    indices: np.ndarray         # [3*Ne] int32 1-based local
    ghost_nodes: np.ndarray     # [3*Ne] uint8
    coord_x: np.ndarray
    coord_y: np.ndarray
    lat: np.ndarray
    mask_dirichlet: np.ndarray  # [Nn] uint8
    neumann_flags: np.ndarray   # sorted int32 0-based local
    node_gid: np.ndarray        # [Nn] global node id of each local node
    elem_gid: np.ndarray        # [Ne] global element id of each local element
    send_procs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    send_offsets: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    send_index: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    recv_procs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    recv_offsets: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    recv_index: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))


# ---- projection ------------------------------------------------------------------------------

def polar_stereographic_lat(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Latitude [deg] of polar-stereographic (x, y) [m] for the ellipsoid / true-scale latitude
    of mesh/NpsNextsim.mpp (a = 6378.273 km, e = 0.081816153, lat1 = 60N) -- what
    GmshMesh::lat() (core/src/gmshmesh.cpp:1800-1824) gets from mapx's inverse_mapx.  Snyder's
    series-free iteration; only `lat` feeds the path (Coriolis, sign of the turning angle)."""
    a = 6378.273e3
    e = 0.081816153
    phi_c = np.deg2rad(60.0)
    t_c = np.tan(np.pi / 4 - phi_c / 2) / ((1 - e * np.sin(phi_c)) / (1 + e * np.sin(phi_c))) ** (e / 2)
    m_c = np.cos(phi_c) / np.sqrt(1 - e * e * np.sin(phi_c) ** 2)
    rho = np.hypot(x, y)
    t = rho * t_c / (a * m_c)
    phi = np.pi / 2 - 2 * np.arctan(t)
    for _ in range(8):
        es = e * np.sin(phi)
        phi = np.pi / 2 - 2 * np.arctan(t * ((1 - es) / (1 + es)) ** (e / 2))
    return np.rad2deg(phi)


# ---- generators ------------------------------------------------------------------------------

def _morton_key(ix: np.ndarray, iy: np.ndarray) -> np.ndarray:
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v
    return spread(ix) | (spread(iy) << np.uint64(1))


def _hilbert_key(ix: np.ndarray, iy: np.ndarray, order: int = 16) -> np.ndarray:
    """Hilbert-curve index of integer points (vectorised xy2d): chunks of consecutive keys are compact
    blobs, which is what the library's node patches want (smaller halos than Z-order chunks)."""
    x = ix.astype(np.int64).copy(); y = iy.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (order - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x); y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def tile_numbering(x: np.ndarray, y: np.ndarray, tile: int, key: np.ndarray) -> np.ndarray:
    """A node numbering made of compact TILES of exactly `tile` nodes (the last one fewer): the node set is bisected recursively, by count, across the longer
    side of its bounding box, into parts that are whole multiples of `tile`; a part of one tile is numbered along `key` (a space-filling-curve key).  Returns
    order with order[new] = old.  Runs of `tile` consecutive numbers are then near-square blocks -- what a patch kernel with two rings of halo wants: rings of
    x 1.20 / x 1.42 nodes instead of the x 1.25 / x 1.51 of runs along a Hilbert curve."""
    n = x.size
    out = np.empty(n, np.int64)
    stack = [(np.arange(n, dtype=np.int64), 0)]
    while stack:
        idx, start = stack.pop()
        k = -(-idx.size // tile)
        if k <= 1:
            out[start:start + idx.size] = idx[np.argsort(key[idx], kind="stable")]
            continue
        xs, ys = x[idx], y[idx]
        c = xs if (xs.max() - xs.min()) >= (ys.max() - ys.min()) else ys
        o = np.argsort(c, kind="stable")
        n1 = (k // 2) * tile
        # which half comes first: the one that holds the smaller curve key, so that consecutive tiles stay neighbours in space as far as a bisection tree allows
        left, right = idx[o[:n1]], idx[o[n1:]]
        if key[left].min() > key[right].min():
            # (the right part may hold the short last tile: it must stay LAST in its range, so only equal-multiple parts swap)
            if right.size % tile == 0:
                left, right = right, left
        stack.append((right, start + left.size))
        stack.append((left, start))
    return out


def _lattice_mesh(nx: int, ny: int, h: float, x0: float, y0: float, inside, jitter: float, seed: int,
                  open_boundary, name: str, reorder: bool = True) -> GlobalMesh:
    """Triangulate the (nx+1) x (ny+1) lattice of spacing h (rows offset by h/2: near-equilateral
    triangles), keep triangles whose centroid satisfies inside(cx, cy), jitter interior nodes."""
    rng = np.random.default_rng(seed)
    jj, ii = np.meshgrid(np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    px = x0 + (ii + 0.5 * (jj % 2)) * h
    py = y0 + jj * (h * np.sqrt(3.0) / 2.0)
    nid = (jj * (nx + 1) + ii)
    # quads (i,j)-(i+1,j)-(i+1,j+1)-(i,j+1); the diagonal follows the row offset
    a = nid[:-1, :-1].ravel(); b = nid[:-1, 1:].ravel(); c = nid[1:, 1:].ravel(); d = nid[1:, :-1].ravel()
    even = (jj[:-1, :-1].ravel() % 2) == 0
    t1 = np.where(even[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
    t2 = np.where(even[:, None], np.stack([b, c, d], 1), np.stack([a, c, d], 1))
    tri = np.concatenate([t1, t2], 0)
    px = px.ravel(); py = py.ravel()
    cx = px[tri].mean(1); cy = py[tri].mean(1)
    tri = tri[inside(cx, cy)]
    # drop triangles that hang on by a single node ("ears" with two boundary edges are fine)
    used = np.zeros(px.size, bool); used[tri.ravel()] = True
    new_id = np.cumsum(used) - 1
    tri = new_id[tri]
    px = px[used]; py = py[used]
    lat_i = (ii.ravel())[used]; lat_j = (jj.ravel())[used]
    nn = px.size
    # boundary edges: edges that belong to exactly one triangle
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], 0)
    es = np.sort(e, 1)
    key = es[:, 0].astype(np.int64) * nn + es[:, 1]
    uk, cnt = np.unique(key, return_counts=True)
    bkey = uk[cnt == 1]
    bnd = np.zeros(nn, bool)
    bnd[(bkey // nn).astype(np.int64)] = True
    bnd[(bkey % nn).astype(np.int64)] = True
    # jitter interior nodes only (keeps the boundary shape and the triangles valid for jitter<0.25)
    dx = rng.uniform(-jitter, jitter, nn) * h
    dy = rng.uniform(-jitter, jitter, nn) * h
    px = np.where(bnd, px, px + dx); py = np.where(bnd, py, py + dy)
    neumann = bnd & open_boundary(px, py)
    dirichlet = bnd & ~neumann
    if reorder:
        # space-filling-curve numbering of nodes and elements: neighbours in space are neighbours in memory
        import os as _os
        curve = _os.environ.get("NXS_MESH_CURVE", "hilbert")
        key = _hilbert_key(lat_i, lat_j) if curve in ("hilbert", "tiles") else _morton_key(lat_i, lat_j)
        order = np.argsort(key, kind="stable")
        if curve == "tiles":   # compact tiles of NXS_MESH_TILE nodes (recursive bisection by node count), tiles and the nodes inside a tile along the Hilbert curve
            order = tile_numbering(px, py, int(_os.environ.get("NXS_MESH_TILE", "428")), key)
        inv = np.empty(nn, np.int64); inv[order] = np.arange(nn)
        px, py, dirichlet, neumann = px[order], py[order], dirichlet[order], neumann[order]
        tri = inv[tri]
        ekey = tri.min(1)
        tri = tri[np.argsort(ekey, kind="stable")]
    tri = np.ascontiguousarray(tri.astype(np.int32))
    # make every triangle counter-clockwise
    jac = (px[tri[:, 1]] - px[tri[:, 0]]) * (py[tri[:, 2]] - py[tri[:, 0]]) - (px[tri[:, 2]] - px[tri[:, 0]]) * (py[tri[:, 1]] - py[tri[:, 0]])
    flip = jac < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return GlobalMesh(x=np.ascontiguousarray(px), y=np.ascontiguousarray(py), tri=tri,
                      dirichlet=dirichlet, neumann=neumann,
                      lat=polar_stereographic_lat(px, py), name=name)


def make_toy_mesh(n: int = 32, seed: int = SEED) -> GlobalMesh:
    """~2 n^2 triangles on a 100 km box placed near (0, -1000 km) of the stereographic plane.
    Left/right edges closed (Dirichlet), top/bottom open (Neumann) -- nextsim.toy.cfg stand-in."""
    L = 100e3
    h = L / n
    ny = int(round(L / (h * np.sqrt(3) / 2)))
    x0, y0 = -L / 2, -1000e3
    ymax = y0 + ny * h * np.sqrt(3) / 2
    def inside(cx, cy):
        return np.ones_like(cx, bool)
    def open_b(px, py):
        return (py <= y0 + 1e-6) | (py >= ymax - 1e-6)
    m = _lattice_mesh(n, ny, h, x0, y0, inside, 0.2, seed, open_b, "toy")
    # corners belong to the coast
    corner = m.neumann & ((m.x <= m.x.min() + 0.75 * h) | (m.x >= m.x.max() - 0.75 * h))
    m.neumann &= ~corner
    m.dirichlet |= corner
    return m


def make_disc_mesh(h: float, radius: float = 2500e3, seed: int = SEED, name: str = "disc") -> GlobalMesh:
    """Pan-Arctic-like basin: a disc with an irregular coast r < R(theta), quasi-uniform edge h.
    The sector |theta - 20deg| < 12deg is an open (Neumann) boundary ("Fram Strait"); the rest is
    coast (Dirichlet)."""
    n = int(np.ceil(2.2 * radius / h))
    x0 = -n * h / 2
    y0 = -n * (h * np.sqrt(3) / 2) / 2
    ny = n
    def rmax(theta):
        return radius * (1.0 + 0.08 * np.sin(3 * theta + 0.5) + 0.05 * np.sin(7 * theta + 1.3) + 0.03 * np.sin(13 * theta))
    def inside(cx, cy):
        r = np.hypot(cx, cy)
        return r < rmax(np.arctan2(cy, cx))
    def open_b(px, py):
        th = np.arctan2(py, px)
        return np.abs(th - np.deg2rad(20.0)) < np.deg2rad(12.0)
    return _lattice_mesh(n, ny, h, x0, y0, inside, 0.2, seed, open_b, name)


def make_mesh(kind: str, seed: int = SEED) -> GlobalMesh:
    if kind == "toy":
        return make_toy_mesh(32, seed)
    if kind == "tiny":
        return make_toy_mesh(6, seed)
    if kind == "small":
        return make_disc_mesh(125e3, seed=seed, name="small")     # ~2.9k triangles
    if kind == "40km":
        return make_disc_mesh(46e3, seed=seed, name="40km")       # ~21k triangles
    if kind == "10km":
        return make_disc_mesh(27.6e3, seed=seed, name="10km")     # ~60k triangles (config-10km)
    if kind == "2km":
        return make_disc_mesh(5.52e3, seed=seed, name="2km")      # ~1.5M triangles (config-2km)
    if kind.startswith("h") and kind[1:].replace(".", "", 1).isdigit():
        return make_disc_mesh(float(kind[1:]), seed=seed, name=kind)  # "h15600": edge length in metres (sweeps between the named sizes)
    raise ValueError(f"unknown mesh kind {kind!r}")


# ---- partitioning ----------------------------------------------------------------------------

def partition_elements(mesh: GlobalMesh, nparts: int) -> np.ndarray:
    """Recursive coordinate bisection of the element centroids -> part id per element.
    (Stands in for Gmsh's METIS k-way partition, core/src/gmshmeshseq.cpp:492-499.)"""
    cx = mesh.x[mesh.tri].mean(1)
    cy = mesh.y[mesh.tri].mean(1)
    part = np.zeros(mesh.num_elements, np.int32)

    def split(idx, lo, n):
        if n == 1:
            part[idx] = lo
            return
        nl = n // 2
        xs, ys = cx[idx], cy[idx]
        coord = xs if (xs.max() - xs.min()) >= (ys.max() - ys.min()) else ys
        k = int(round(idx.size * nl / n))
        order = np.argsort(coord, kind="stable")
        split(idx[order[:k]], lo, nl)
        split(idx[order[k:]], lo + nl, n - nl)

    split(np.arange(mesh.num_elements), 0, nparts)
    return part


def localize(mesh: GlobalMesh, nparts: int = 1, elem_part: np.ndarray | None = None, symmetrise: bool = False) -> list[LocalMesh]:
    """Per-rank meshes in the reference layout (see module docstring).  The halo lists are what initUpdateGhosts() (FE.cpp:14003-14088) leaves: a ragged
    partition may send a node to a rank it receives nothing from -- nxs_dyn_set_halo adds the missing direction itself.  symmetrise = True does it here
    instead (every partner in both lists, the direction without nodes as an empty segment, partners in ascending order): what a caller of the low-level
    nxs_dyn_ipc_connect has to hand over."""
    Nn, Ne = mesh.num_nodes, mesh.num_elements
    tri = mesh.tri.astype(np.int64)
    if elem_part is None:
        elem_part = partition_elements(mesh, nparts) if nparts > 1 else np.zeros(Ne, np.int32)
    # node owner = lowest rank among the elements holding it (gmshmesh.cpp:1076-1090: lower rank wins)
    node_owner = np.full(Nn, nparts, np.int64)
    for k in range(3):
        np.minimum.at(node_owner, tri[:, k], elem_part)
    out: list[LocalMesh] = []
    g2l_all = []
    for r in range(nparts):
        owned_nodes = np.flatnonzero(node_owner == r)                      # ascending global id
        owned_mask = np.zeros(Nn, bool); owned_mask[owned_nodes] = True
        mine = elem_part == r
        touch = owned_mask[tri].any(1)
        owned_el = np.flatnonzero(mine)
        ghost_el = np.flatnonzero(touch & ~mine)
        elems = np.concatenate([owned_el, ghost_el])                      # owned first, each ascending
        ltri = tri[elems]
        used = np.zeros(Nn, bool); used[ltri.ravel()] = True
        ghost_nodes_g = np.flatnonzero(used & ~owned_mask)                # ascending global id
        nodes = np.concatenate([owned_nodes, ghost_nodes_g])
        g2l = np.full(Nn, -1, np.int64); g2l[nodes] = np.arange(nodes.size)
        lidx = g2l[ltri]
        no = owned_nodes.size
        lm = LocalMesh(
            rank=r, nranks=nparts,
            num_nodes=int(nodes.size), num_elements=int(elems.size),
            local_ndof=int(no), local_nelements=int(owned_el.size),
            indices=np.ascontiguousarray((lidx + 1).astype(np.int32).ravel()),
            ghost_nodes=np.ascontiguousarray((lidx >= no).astype(np.uint8).ravel()),
            coord_x=np.ascontiguousarray(mesh.x[nodes]), coord_y=np.ascontiguousarray(mesh.y[nodes]),
            lat=np.ascontiguousarray(mesh.lat[nodes]),
            mask_dirichlet=np.ascontiguousarray((mesh.dirichlet[nodes] & (np.arange(nodes.size) < no)).astype(np.uint8)),
            neumann_flags=np.ascontiguousarray(np.flatnonzero(mesh.neumann[nodes]).astype(np.int32)),
            node_gid=nodes.astype(np.int64), elem_gid=elems.astype(np.int64),
        )
        out.append(lm)
        g2l_all.append(g2l)
    # halo lists (FE.cpp:14003-14088): per neighbour, ascending global id
    if nparts > 1:
        for r, lm in enumerate(out):
            gh_l = np.arange(lm.local_ndof, lm.num_nodes)
            gh_g = lm.node_gid[gh_l]
            own = node_owner[gh_g]
            rp, ro, ri = [], [0], []
            for q in np.unique(own):
                sel = own == q
                rp.append(int(q)); ri.append(gh_l[sel]); ro.append(ro[-1] + int(sel.sum()))
            lm.recv_procs = np.array(rp, np.int32)
            lm.recv_offsets = np.array(ro, np.int32)
            lm.recv_index = np.ascontiguousarray(np.concatenate(ri).astype(np.int32)) if ri else np.zeros(0, np.int32)
        for r, lm in enumerate(out):
            sp, so, si = [], [0], []
            for q, other in enumerate(out):
                if q == r:
                    continue
                w = np.flatnonzero(other.recv_procs == r)
                if w.size == 0:
                    continue
                k = int(w[0])
                gl = other.node_gid[other.recv_index[other.recv_offsets[k]:other.recv_offsets[k + 1]]]
                sp.append(q); si.append(g2l_all[r][gl]); so.append(so[-1] + gl.size)
            lm.send_procs = np.array(sp, np.int32)
            lm.send_offsets = np.array(so, np.int32)
            lm.send_index = np.ascontiguousarray(np.concatenate(si).astype(np.int32)) if si else np.zeros(0, np.int32)
        # symmetrise: every exchange partner is both a sender and a receiver: where a ragged partition sends to a rank it receives nothing from (a node of mine
        # touches an element of yours, none of yours touches one of mine), the missing direction becomes an EMPTY segment.  The device-direct mailboxes need that
        # hand-shake (two buffers per link); nxs_dyn_set_halo adds it by itself, this is the same thing done on the caller's side.
        for lm in (out if symmetrise else []):
            partners = sorted(set(lm.send_procs.tolist()) | set(lm.recv_procs.tolist()))
            for side in ("send", "recv"):
                procs, offs = getattr(lm, side + "_procs").tolist(), getattr(lm, side + "_offsets").tolist()
                if procs == partners:
                    continue
                new_offs, pos = [0], 0
                for q in partners:
                    if pos < len(procs) and procs[pos] == q:
                        new_offs.append(new_offs[-1] + offs[pos + 1] - offs[pos]); pos += 1
                    else:
                        new_offs.append(new_offs[-1])
                assert pos == len(procs) and new_offs[-1] == offs[-1]
                setattr(lm, side + "_procs", np.array(partners, np.int32)); setattr(lm, side + "_offsets", np.array(new_offs, np.int32))
    return out

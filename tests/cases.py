"""Shared builders for the tests: a (mesh, params, fields) case on one or several ranks."""
from __future__ import annotations

import numpy as np

from nextsim_amd import _abi, forcing as F, mesh as M

_cache = {}


def global_mesh(kind):
    if kind not in _cache:
        _cache[kind] = M.make_mesh(kind)
    return _cache[kind]


def ragged_partition(gm, nparts, seed):
    """An element partition nobody would choose: a coarse random mosaic (blocks of ~40 elements thrown at random ranks), so
    that every rank has every other rank as a neighbour, several disconnected pieces, ghost elements without any owned node
    and boundary nodes shared by three or more ranks -- the corner cases of the halo lists."""
    rng = np.random.default_rng(seed)
    cx = gm.x[gm.tri].mean(1); cy = gm.y[gm.tri].mean(1)
    nb = max(4, int(np.sqrt(gm.num_elements / 40.)))
    ix = np.minimum(((cx - cx.min()) / (np.ptp(cx) + 1e-9) * nb).astype(int), nb - 1)
    iy = np.minimum(((cy - cy.min()) / (np.ptp(cy) + 1e-9) * nb).astype(int), nb - 1)
    owner = rng.integers(0, nparts, (nb, nb))
    owner.flat[:nparts] = np.arange(nparts)          # nobody is empty
    return owner[ix, iy].astype(np.int32)


def make_case(kind="small", forcing_kind=None, nparts=1, alea=0.33, ragged_seed=None, **param_over):
    gm = global_mesh(kind)
    forcing_kind = forcing_kind or ("toy" if kind in ("toy", "tiny") else "arctic")
    p = F.default_params(**param_over)
    p, C_fix, C_alea = F.scale_params_to_mesh(p, gm, alea_factor=alea)
    g = F.global_fields(gm, p, forcing_kind, C_fix, C_alea)
    lms = M.localize(gm, nparts, elem_part=ragged_partition(gm, nparts, ragged_seed) if ragged_seed is not None else None)
    fields = [F.localize_fields(g, lm, gm.num_nodes) for lm in lms]
    return gm, p, g, lms, fields


def mesh_with_holes(kind="small", holes=((0.25, -0.1, 0.12), (-0.3, 0.3, 0.07))):
    """The mesh `kind` with islands cut out of it: every triangle with a vertex inside one of the circles (centre and radius in
    units of the mesh radius) is removed, unused nodes are dropped.  Returns x, y, tri (0-based): a domain with inner boundaries,
    as a pan-Arctic mesh has around its islands."""
    gm = global_mesh(kind)
    R = np.hypot(gm.x, gm.y).max()
    kill = np.zeros(gm.num_nodes, bool)
    for cx, cy, r in holes:
        kill |= np.hypot(gm.x - cx * R, gm.y - cy * R) < r * R
    tri = gm.tri[~kill[gm.tri].any(1)]
    used = np.unique(tri)
    new = np.full(gm.num_nodes, -1, np.int64); new[used] = np.arange(used.size)
    return gm.x[used].copy(), gm.y[used].copy(), np.ascontiguousarray(new[tri], np.int32)


def awkward_meshes():
    """Meshes whose boundary is not "one outer loop + holes": name -> (x, y, tri 0-based).  Several components (a disc and the toy box; three
    of them; a lake inside an island of another mesh) and boundaries that PINCH (two triangular holes meeting at an interior vertex; a hole
    that touches the coast at one vertex) -- real coastlines do all of it."""
    sm, toy = global_mesh("small"), global_mesh("toy")

    def join(parts):
        xs, ys, ts, off = [], [], [], 0
        for x, y, t in parts:
            xs.append(x); ys.append(y); ts.append(t + off); off += x.size
        return np.concatenate(xs), np.concatenate(ys), np.ascontiguousarray(np.concatenate(ts), np.int32)
    out = {"two_components": join([(sm.x, sm.y, sm.tri), (toy.x * 8 + 4.0e6, toy.y * 8 - 1e6, toy.tri)])}
    xh, yh, th = mesh_with_holes("small")
    out["three_components"] = join([(sm.x, sm.y, sm.tri), (toy.x * 8 + 4.0e6, toy.y * 8 - 1e6, toy.tri), (xh * 0.7 - 1e6, yh * 0.7 + 6.5e6, th)])
    xi, yi, ti = mesh_with_holes("40km", holes=((0.0, 0.0, 0.45),))
    out["lake_in_island"] = join([(xi, yi, ti), (sm.x * 0.25, sm.y * 0.25, sm.tri)])
    tri = sm.tri
    edges = {}
    for t in tri.tolist():
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges[(min(a, b), max(a, b))] = edges.get((min(a, b), max(a, b)), 0) + 1
    bndv = {v for e, c in edges.items() if c == 1 for v in e}
    deg = np.bincount(tri.ravel(), minlength=sm.num_nodes)
    for v in range(sm.num_nodes):            # two opposite triangles of an interior vertex's fan of six removed: two holes that meet at that vertex
        if deg[v] != 6 or v in bndv:
            continue
        fan = [i for i, t in enumerate(tri.tolist()) if v in t]
        rest = {i: set(tri[i].tolist()) - {v} for i in fan}
        if any(u in bndv for i in fan for u in rest[i]):
            continue
        order = [fan[0]]
        while len(order) < 6:
            nxt = [i for i in fan if i not in order and len(rest[i] & rest[order[-1]]) == 1]
            if not nxt:
                break
            order.append(nxt[0])
        if len(order) == 6:
            keep = np.ones(tri.shape[0], bool); keep[order[0]] = False; keep[order[3]] = False
            out["pinch_inside"] = (sm.x, sm.y, np.ascontiguousarray(tri[keep]))
            break
    for i, t in enumerate(tri.tolist()):     # a triangle with exactly one coast vertex removed: a hole that touches the coast at that vertex
        if sum(u in bndv for u in t) == 1:
            keep = np.ones(tri.shape[0], bool); keep[i] = False
            out["pinch_at_the_coast"] = (sm.x, sm.y, np.ascontiguousarray(tri[keep]))
            break
    return out


def rel_err(a, b):
    """max |a-b| / max|b| (fields here have a natural scale; pointwise relative error is meaningless
    near zero crossings)."""
    scale = max(float(np.abs(b).max()), 1e-300)
    return float(np.abs(a - b).max()) / scale


# ---- regrid pairs for the conservative remapping (N1) ----------------------------------------------------

def _delaunay(x, y):
    from scipy.spatial import Delaunay
    t = Delaunay(np.column_stack([x, y])).simplices.astype(np.int64)
    jac = (x[t[:, 1]] - x[t[:, 0]]) * (y[t[:, 2]] - y[t[:, 0]]) - (x[t[:, 2]] - x[t[:, 0]]) * (y[t[:, 1]] - y[t[:, 0]])
    t = t[np.abs(jac) > 1e-3]
    jac = (x[t[:, 1]] - x[t[:, 0]]) * (y[t[:, 2]] - y[t[:, 0]]) - (x[t[:, 2]] - x[t[:, 0]]) * (y[t[:, 1]] - y[t[:, 0]])
    t[jac < 0] = t[jac < 0][:, [0, 2, 1]]    # counter-clockwise, as the meshes of the model
    return np.ascontiguousarray(t, np.int32)


def rect_mesh(n, seed, L=400e3, H=300e3, x0=-150e3, y0=-900e3):
    """A Delaunay mesh of a rectangle: boundary vertices first (they play bamg's geometric vertices: same
    numbers and positions in every mesh of the same rectangle), then jittered interior vertices."""
    rng = np.random.default_rng(seed)
    nb = 24
    s = np.arange(nb) / nb
    bx = np.concatenate([x0 + L * s, np.full(nb, x0 + L), x0 + L * (1 - s), np.full(nb, x0)])
    by = np.concatenate([np.full(nb, y0), y0 + H * s, np.full(nb, y0 + H), y0 + H * (1 - s)])
    gx, gy = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n)
    ix = x0 + L * (gx.ravel() + rng.uniform(-0.3, 0.3, n * n) / n)
    iy = y0 + H * (gy.ravel() + rng.uniform(-0.3, 0.3, n * n) / n)
    x = np.concatenate([bx, ix]); y = np.concatenate([by, iy])
    return x, y, _delaunay(x, y), 4 * nb


def adapted_mesh(x, y, tri, n_geom, seed, frac_touched=0.15):
    """What a bamg adaptation leaves behind, built by hand: most triangles survive, some edges are flipped,
    some triangles are split by a new vertex, some interior vertices move; vertices (beyond the geometric
    ones) and triangles are renumbered.  Returns x, y, tri (0-based), previous_numbering (1-based, 0 = new)."""
    rng = np.random.default_rng(seed)
    tri = tri.copy()
    ne = tri.shape[0]
    # edge -> triangles
    emap = {}
    for e in range(ne):
        for j in range(3):
            p, q = int(tri[e, (j + 1) % 3]), int(tri[e, (j + 2) % 3])
            emap.setdefault((min(p, q), max(p, q)), []).append((e, j))
    touched = np.zeros(ne, bool)

    def ccw(a, b, c):
        return (x[b] - x[a]) * (y[c] - y[a]) - (x[c] - x[a]) * (y[b] - y[a])

    edges = [k for k, v in emap.items() if len(v) == 2]
    rng.shuffle(edges)
    nflip = 0
    for k in edges:
        (e1, j1), (e2, j2) = emap[k]
        if touched[e1] or touched[e2]:
            continue
        a = int(tri[e1, j1]); c = int(tri[e2, j2])           # the two apexes
        p, q = int(tri[e1, (j1 + 1) % 3]), int(tri[e1, (j1 + 2) % 3])
        if ccw(a, p, c) > 1e3 and ccw(a, c, q) > 1e3:        # convex quad: flip p-q into a-c
            tri[e1] = (a, p, c); tri[e2] = (a, c, q)
            touched[e1] = touched[e2] = True
            nflip += 1
            if nflip >= frac_touched * ne / 4:
                break
    free = np.flatnonzero(~touched)
    split = rng.choice(free, size=max(1, int(frac_touched * ne / 6)), replace=False)
    nx, ny, new_tri = list(x), list(y), []
    for e in split:
        a, b, c = (int(v) for v in tri[e])
        m = len(nx)
        w = rng.dirichlet([2., 2., 2.])
        nx.append(w[0] * x[a] + w[1] * x[b] + w[2] * x[c]); ny.append(w[0] * y[a] + w[1] * y[b] + w[2] * y[c])
        tri[e] = (a, b, m)
        new_tri += [(b, c, m), (c, a, m)]
        touched[e] = True
    tri = np.vstack([tri, np.array(new_tri, np.int32)])
    xn, yn = np.array(nx), np.array(ny)
    prev = np.concatenate([np.arange(1, x.size + 1), np.zeros(xn.size - x.size, np.int64)]).astype(np.float64)
    # renumber the non-geometric vertices and all the triangles
    perm = np.concatenate([np.arange(n_geom), n_geom + rng.permutation(xn.size - n_geom)])   # new id -> old id
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
    xn, yn, prev = xn[perm], yn[perm], prev[perm]
    tri = inv[tri].astype(np.int32)
    tri = tri[rng.permutation(tri.shape[0])]
    return xn, yn, np.ascontiguousarray(tri), prev


def mesh_operator(gm, nonsymmetric=False, shift=0.05):
    """A node-numbered sparse operator on the mesh for the Krylov tests (scipy CSR, sorted columns), its right-hand side
    for a known solution, and that solution: the P1 stiffness matrix (element matrix of research/laplacian.cpp:163-224)
    plus `shift` times the lumped mass -- symmetric positive definite; nonsymmetric=True adds a skew edge term of
    the size of half the stiffness entries (what a Coriolis / advection term does to a momentum matrix)."""
    import scipy.sparse as sp
    L = max(np.ptp(gm.x), np.ptp(gm.y))
    x, y = gm.x / L, gm.y / L
    t = gm.tri
    xs, ys = x[t], y[t]
    area = 0.5 * np.abs((xs[:, 1] - xs[:, 0]) * (ys[:, 2] - ys[:, 0]) - (xs[:, 2] - xs[:, 0]) * (ys[:, 1] - ys[:, 0]))
    rows, cols, vals = [], [], []
    for j in range(3):
        jp1, jp2 = (j + 1) % 3, (j + 2) % 3
        for k in range(3):
            kp1, kp2 = (k + 1) % 3, (k + 2) % 3
            m = ((ys[:, jp1] - ys[:, jp2]) * (ys[:, kp1] - ys[:, kp2]) + (xs[:, jp1] - xs[:, jp2]) * (xs[:, kp1] - xs[:, kp2])) / (4.0 * area)
            if nonsymmetric and j != k:
                m = m + (0.5 if (k - j) % 3 == 1 else -0.5) * np.abs(m)      # skew: +w on (j,k), -w on (k,j)
            rows.append(t[:, j]); cols.append(t[:, k]); vals.append(m)
        rows.append(t[:, j]); cols.append(t[:, j]); vals.append(shift * area / 3.0 * gm.num_nodes)
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(gm.num_nodes, gm.num_nodes))
    A.sum_duplicates(); A.sort_indices()
    sol = np.sin(3.0 * x) * np.cos(2.0 * y) + 0.3
    return A, A @ sol, sol

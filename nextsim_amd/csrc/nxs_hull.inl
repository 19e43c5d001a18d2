// nxs_hull.inl -- host code: the convex completion of a triangle mesh as contrib/bamg builds it (textually included by
// nxs_interp.hip).
//
// InterpFromMeshToMesh2dx does not locate its target points in the data mesh but in bamg's RECONSTRUCTED mesh
// (Mesh::Mesh(index, x, y, ...) -> ReconstructExistingMesh, contrib/bamg/src/Mesh.cpp:3135-3440): the given triangles, plus
// triangles that fill every hole and every concave part of the boundary up to the convex hull, plus one "infinite" triangle
// behind every hull edge.  With isdefault == false (the regrid call, FE.cpp:3131-3139) a point outside the data mesh is
// therefore interpolated INSIDE A FILL TRIANGLE (between boundary vertices that may be far apart), or, beyond the hull,
// projected on a hull edge (CloseBoundaryEdge, Mesh.cpp:4590-4627).
//
// bamg builds the fill triangles as an incremental Delaunay triangulation of the boundary vertices in its integer
// coordinates, with the boundary edges forced (Mesh.cpp:3262-3328) and everything inside the domain removed: away from
// cocircular vertex sets that is THE constrained Delaunay triangulation of each pocket between the hull and the boundary
// and of each hole, which is what this file computes -- ear clipping, then Lawson flips of the interior diagonals, all with
// exact integer predicates (64-bit orientation, 128-bit in-circle).  tests/test_interp.py compares the result with the
// real bamg: same triangles, same hull.  What cannot be reproduced is the ORDER of a fill triangle's three vertices inside
// bamg's Triangle object (it depends on the history of bamg's point insertions and edge swaps): the three products of
// the P1 sum are the same numbers, added in a possibly rotated order -- 1 ulp at most.
//
// Meshes this does not cover are reported (ok == false: several outer boundary loops, a boundary vertex with two outgoing
// boundary edges, a pocket that is not a simple polygon): the caller then keeps the nearest-boundary-edge rule for exterior
// points and says so (num_approx).

namespace nxs_hull {

struct HullEdge {
    int a, b;     // hull edge a -> b, counter-clockwise (the domain on its left)
    int tri, k;   // the triangle inside it (>= nels: fill triangle nels + i) and the local edge index there (opposite vertex k)
};

struct Completion {
    bool ok = false;
    std::string why;
    std::vector<int> fill;        // 3 per fill triangle, 0-based vertices, counter-clockwise
    std::vector<HullEdge> hull;   // in counter-clockwise order around the hull
};

typedef long long i64;
typedef __int128 i128;

// bamg's integer plane (SetIntCoor, Mesh.cpp:3441-3468; R2ToI2): bounding box + 5 %, 2^30 - 1 units along its longer side, truncation.
// false: "coefIcoor should be positive" (a degenerate or non-finite geometry).
inline bool int_plane(const double *x, const double *y, int nods, std::vector<int> &ix, std::vector<int> &iy, double &coef, double &pminx, double &pminy) {
    if (nods < 1) return false;
    double pmaxx = x[0], pmaxy = y[0];
    pminx = x[0]; pminy = y[0];
    for (int i = 0; i < nods; ++i) {
        pminx = std::min(pminx, x[i]); pminy = std::min(pminy, y[i]);
        pmaxx = std::max(pmaxx, x[i]); pmaxy = std::max(pmaxy, y[i]);
    }
    const double DDx = (pmaxx - pminx) * 0.05, DDy = (pmaxy - pminy) * 0.05;
    pminx = pminx - DDx; pminy = pminy - DDy;
    pmaxx = pmaxx + DDx; pmaxy = pmaxy + DDy;
    coef = 1073741823. / std::max(pmaxx - pminx, pmaxy - pminy);
    if (!(coef > 0.) || !(coef < 1e300)) return false;
    ix.resize(nods); iy.resize(nods);
    for (int i = 0; i < nods; ++i) {
        const double fx = coef * (x[i] - pminx), fy = coef * (y[i] - pminy);
        if (!(fx >= 0. && fx < 1073741824. && fy >= 0. && fy < 1073741824.)) return false;  // a NaN coordinate: its conversion would be undefined
        ix[i] = (int)fx; iy[i] = (int)fy;
    }
    return true;
}

struct Pts {
    const int *ix, *iy;
    i64 orient(int a, int b, int c) const {  // include/det.h:8-12; |coordinates| < 2^30, so the result fits
        return ((i64)ix[b] - ix[a]) * ((i64)iy[c] - iy[a]) - ((i64)iy[b] - iy[a]) * ((i64)ix[c] - ix[a]);
    }
    // > 0: d strictly inside the circle through the counter-clockwise triangle a, b, c
    bool in_circle(int a, int b, int c, int d) const {
        const i64 ax = (i64)ix[a] - ix[d], ay = (i64)iy[a] - iy[d], bx = (i64)ix[b] - ix[d], by = (i64)iy[b] - iy[d],
                  cx = (i64)ix[c] - ix[d], cy = (i64)iy[c] - iy[d];
        const i128 A = (i128)(ax * ax + ay * ay), B = (i128)(bx * bx + by * by), C = (i128)(cx * cx + cy * cy);
        const i128 det = A * (i128)(bx * cy - by * cx) - B * (i128)(ax * cy - ay * cx) + C * (i128)(ax * by - ay * bx);
        return det > 0;
    }
};

// Ear clipping of a simple counter-clockwise polygon on a doubly linked list: after an ear is cut only its two neighbours
// change (their reflex flags are redone), the search goes on from there, and only reflex vertices can lie inside a candidate
// ear -- O(n r) for the coast pockets met here instead of O(n^2 r).
inline bool ear_clip(const Pts &P, const std::vector<int> &poly, std::vector<int> &tris) {
    const int n = (int)poly.size();
    if (n < 3) return false;
    std::vector<int> prv(n), nxt(n);
    std::vector<char> reflex(n), alive(n, 1);
    for (int i = 0; i < n; ++i) { prv[i] = (i + n - 1) % n; nxt[i] = (i + 1) % n; }
    for (int i = 0; i < n; ++i) reflex[i] = P.orient(poly[prv[i]], poly[i], poly[nxt[i]]) <= 0;
    int left = n, i = 0, since = 0;
    while (left > 3) {
        if (since > left) return false;  // a full round without an ear: not a simple polygon (or fully degenerate)
        const int a = poly[prv[i]], b = poly[i], c = poly[nxt[i]];
        bool ear = !reflex[i];
        if (ear)
            for (int j = nxt[nxt[i]]; j != prv[i] && ear; j = nxt[j]) {
                if (!reflex[j]) continue;
                const int q = poly[j];
                if (q == a || q == b || q == c) continue;
                if (P.orient(a, b, q) >= 0 && P.orient(b, c, q) >= 0 && P.orient(c, a, q) >= 0) ear = false;
            }
        if (!ear) { i = nxt[i]; ++since; continue; }
        tris.push_back(a); tris.push_back(b); tris.push_back(c);
        const int p = prv[i], q = nxt[i];
        alive[i] = 0; nxt[p] = q; prv[q] = p; --left; since = 0;
        reflex[p] = P.orient(poly[prv[p]], poly[p], poly[nxt[p]]) <= 0;
        reflex[q] = P.orient(poly[prv[q]], poly[q], poly[nxt[q]]) <= 0;
        i = p;
    }
    const int a = poly[prv[i]], b = poly[i], c = poly[nxt[i]];
    if (P.orient(a, b, c) <= 0) return false;
    tris.push_back(a); tris.push_back(b); tris.push_back(c);
    return true;
}

// Lawson flips of the interior diagonals of one polygon's triangulation until every one is locally Delaunay (a work list of
// edges; a flip puts the four edges around it back on the list)
inline void make_delaunay(const Pts &P, std::vector<int> &t /* 3 per triangle */) {
    const int nt = (int)t.size() / 3;
    if (nt < 2) return;
    std::map<std::pair<int, int>, int> half;  // directed edge -> 3*triangle + position of its first vertex
    for (int i = 0; i < nt; ++i)
        for (int k = 0; k < 3; ++k) half[{t[3 * i + k], t[3 * i + (k + 1) % 3]}] = 3 * i + k;
    std::vector<std::pair<int, int>> work;
    for (const auto &h : half) if (h.first.first < h.first.second && half.count({h.first.second, h.first.first})) work.push_back(h.first);
    long long guard = 64ll * nt * nt + 1024;
    while (!work.empty() && guard-- > 0) {
        const std::pair<int, int> e = work.back();
        work.pop_back();
        const auto h = half.find(e), o = half.find({e.second, e.first});
        if (h == half.end() || o == half.end()) continue;  // flipped away meanwhile, or a polygon edge
        const int a = e.first, b = e.second;
        const int i = h->second / 3, ki = h->second % 3, j = o->second / 3, kj = o->second % 3;
        const int c = t[3 * i + (ki + 2) % 3], d = t[3 * j + (kj + 2) % 3];  // a, b, c and b, a, d are the two triangles
        if (!P.in_circle(a, b, c, d)) continue;
        if (P.orient(c, a, d) <= 0 || P.orient(d, b, c) <= 0) continue;  // the quadrilateral is not strictly convex
        half.erase({a, b}); half.erase({b, a});
        t[3 * i] = c; t[3 * i + 1] = a; t[3 * i + 2] = d;
        t[3 * j] = d; t[3 * j + 1] = b; t[3 * j + 2] = c;
        half[{c, a}] = 3 * i; half[{a, d}] = 3 * i + 1; half[{d, c}] = 3 * i + 2;
        half[{d, b}] = 3 * j; half[{b, c}] = 3 * j + 1; half[{c, d}] = 3 * j + 2;
        for (const std::pair<int, int> &q : {std::pair<int, int>{c, a}, {a, d}, {d, b}, {b, c}}) work.push_back(q.first < q.second ? q : std::pair<int, int>{q.second, q.first});
    }
}

// Boundary edges of a triangle mesh (index 1-based): 3*e + k for every local edge k of triangle e (vertices VOTE[k][0] ->
// VOTE[k][1]) whose reverse belongs to no triangle, in ascending order.  Directed edges are bucketed by their tail (counting
// sort), then every edge looks for its reverse among the few edges leaving its head: linear time, no comparison sort.
// Returns false when a directed edge occurs twice (an edge shared by more than two triangles / inconsistent orientation).
inline bool find_boundary_edges(const int32_t *index, int nods, int nels, std::vector<int> &out) {
    static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    std::vector<int> start((size_t)nods + 1, 0);
    for (int e = 0; e < nels; ++e) for (int k = 0; k < 3; ++k) start[(size_t)index[3 * e + k]]++;  // each vertex is the tail of one edge per triangle it is in
    for (int v = 0; v < nods; ++v) start[(size_t)v + 1] += start[v];
    std::vector<int> head((size_t)3 * nels), fill(start.begin(), start.end() - 1);
    for (int e = 0; e < nels; ++e)
        for (int k = 0; k < 3; ++k) {
            const int p = index[3 * e + VOTE[k][0]] - 1, q = index[3 * e + VOTE[k][1]] - 1;
            head[(size_t)fill[p]++] = q;
        }
    out.clear();
    bool ok = true;
    for (int e = 0; e < nels; ++e)
        for (int k = 0; k < 3; ++k) {
            const int p = index[3 * e + VOTE[k][0]] - 1, q = index[3 * e + VOTE[k][1]] - 1;
            bool twin = false;
            for (int i = start[q]; i < start[(size_t)q + 1] && !twin; ++i) twin = head[i] == p;
            int same = 0;
            for (int i = start[p]; i < start[(size_t)p + 1]; ++i) same += head[i] == q;
            if (same > 1) ok = false;
            if (!twin) out.push_back(3 * e + k);
        }
    return ok;
}

// index: 1-based triangles; ix, iy: bamg's integer coordinates of the vertices (SetIntCoor)
inline Completion complete(const int32_t *index, const int *ix, const int *iy, int nods, int nels) {
    Completion out;
    const Pts P{ix, iy};
    static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    // boundary edges, oriented as in their triangle (the domain on the left)
    std::vector<int> bnd;
    if (!find_boundary_edges(index, nods, nels, bnd)) { out.why = "an edge belongs to more than two triangles"; return out; }
    std::vector<int> nxt(nods, -1), etri(nods, -1);  // boundary edge leaving each vertex: its head, and 3*triangle + k
    for (int be : bnd) {
        const int e = be / 3, k = be % 3;
        const int p = index[3 * e + VOTE[k][0]] - 1, q = index[3 * e + VOTE[k][1]] - 1;
        if (nxt[p] >= 0) { out.why = "a boundary vertex has two outgoing boundary edges (the domain pinches there)"; return out; }
        nxt[p] = q; etri[p] = be;
    }
    if (bnd.size() < 3) { out.why = "no boundary"; return out; }
    // loops; the outer one has positive area
    std::vector<std::vector<int>> loops;
    std::vector<char> seen(nods, 0);
    for (int v = 0; v < nods; ++v) {
        if (nxt[v] < 0 || seen[v]) continue;
        std::vector<int> L;
        int c = v;
        while (!seen[c]) { seen[c] = 1; L.push_back(c); c = nxt[c]; if (c < 0) { out.why = "open boundary chain"; return out; } }
        if (c != v) { out.why = "boundary chain does not close"; return out; }
        loops.push_back(std::move(L));
    }
    int outer = -1;
    for (size_t l = 0; l < loops.size(); ++l) {
        i128 a2 = 0;
        const auto &L = loops[l];
        for (size_t i = 0; i < L.size(); ++i) {
            const int p = L[i], q = L[(i + 1) % L.size()];
            a2 += (i128)ix[p] * iy[q] - (i128)ix[q] * iy[p];
        }
        if (a2 > 0) {
            if (outer >= 0) { out.why = "several outer boundary loops (the mesh has several components)"; return out; }
            outer = (int)l;
        }
    }
    if (outer < 0) { out.why = "no counter-clockwise boundary loop"; return out; }
    const std::vector<int> &O = loops[outer];
    const int m = (int)O.size();
    // convex hull of the outer loop, collinear vertices kept (a boundary vertex ON a hull edge is a hull vertex for bamg too:
    // every boundary edge is forced, so a straight coast on the hull is a chain of hull edges)
    std::vector<int> pts(O);
    std::sort(pts.begin(), pts.end(), [&](int p, int q) { return ix[p] != ix[q] ? ix[p] < ix[q] : iy[p] < iy[q]; });
    for (size_t i = 1; i < pts.size(); ++i)
        if (ix[pts[i]] == ix[pts[i - 1]] && iy[pts[i]] == iy[pts[i - 1]]) { out.why = "two boundary vertices share one integer point"; return out; }
    std::vector<char> on_hull(nods, 0);
    {
        // strict hull first, then every loop vertex lying exactly on one of its edges
        std::vector<int> H;
        auto half = [&](const std::vector<int> &S) {
            std::vector<int> h;
            for (int p : S) {
                while (h.size() >= 2 && P.orient(h[h.size() - 2], h[h.size() - 1], p) <= 0) h.pop_back();
                h.push_back(p);
            }
            return h;
        };
        std::vector<int> lower = half(pts), rev(pts.rbegin(), pts.rend()), upper = half(rev);
        H.insert(H.end(), lower.begin(), lower.end() - 1);
        H.insert(H.end(), upper.begin(), upper.end() - 1);
        if (H.size() < 3) { out.why = "degenerate hull"; return out; }
        for (int p : H) on_hull[p] = 1;
        // collinear boundary vertices on a hull edge: walk the loop between consecutive strict hull vertices
        std::vector<int> pos(nods, -1);
        for (int i = 0; i < m; ++i) pos[O[i]] = i;
        std::sort(H.begin(), H.end(), [&](int p, int q) { return pos[p] < pos[q]; });
        for (size_t i = 0; i < H.size(); ++i) {
            const int a = H[i], b = H[(i + 1) % H.size()];
            for (int q = (pos[a] + 1) % m; q != pos[b]; q = (q + 1) % m) {
                const int v = O[q];
                if (P.orient(a, b, v) == 0 && ((i64)ix[v] - ix[a]) * ((i64)ix[b] - ix[v]) + ((i64)iy[v] - iy[a]) * ((i64)iy[b] - iy[v]) >= 0) on_hull[v] = 1;
            }
        }
    }
    std::vector<int> hs;  // hull vertices in loop order = counter-clockwise around the hull
    for (int i = 0; i < m; ++i) if (on_hull[O[i]]) hs.push_back(i);
    for (size_t i = 0; i < hs.size(); ++i) {
        const int a = O[hs[i]], b = O[hs[(i + 1) % hs.size()]], c = O[hs[(i + 2) % hs.size()]];
        if (P.orient(a, b, c) < 0) { out.why = "the hull vertices do not follow the boundary loop (self-intersecting boundary?)"; return out; }
    }
    // pockets between the hull and the outer loop, and the holes: counter-clockwise polygons, triangulated one by one
    std::map<std::pair<int, int>, int> closing;  // hull edge a -> b that closes a pocket -> index of the fill triangle holding it
    auto add_polygon = [&](const std::vector<int> &poly) -> bool {
        std::vector<int> t;
        if (getenv("NXS_DEBUG_HULL")) fprintf(stderr, "[hull] polygon of %zu vertices\n", poly.size());
        if (!ear_clip(P, poly, t)) return false;
        make_delaunay(P, t);
        out.fill.insert(out.fill.end(), t.begin(), t.end());
        return true;
    };
    for (size_t i = 0; i < hs.size(); ++i) {
        const int ia = hs[i], ib = hs[(i + 1) % hs.size()];
        const int len = ((ib - ia) % m + m) % m;
        if (len <= 1) continue;  // a boundary edge on the hull
        std::vector<int> poly;
        for (int k = len; k >= 0; --k) poly.push_back(O[(ia + k) % m]);  // b, ..., a: counter-clockwise with the closing edge a -> b
        if (!add_polygon(poly)) { out.why = "a pocket between the hull and the boundary is not a simple polygon"; out.fill.clear(); return out; }
    }
    for (size_t l = 0; l < loops.size(); ++l) {
        if ((int)l == outer) continue;
        std::vector<int> poly(loops[l].rbegin(), loops[l].rend());
        if (!add_polygon(poly)) { out.why = "a hole of the mesh is not a simple polygon"; out.fill.clear(); return out; }
    }
    const int nfill = (int)out.fill.size() / 3;
    for (int i = 0; i < nfill; ++i)
        for (int k = 0; k < 3; ++k) closing[{out.fill[3 * i + VOTE[k][0]], out.fill[3 * i + VOTE[k][1]]}] = 3 * i + k;
    // hull edges with the triangle inside each
    for (size_t i = 0; i < hs.size(); ++i) {
        const int a = O[hs[i]], b = O[hs[(i + 1) % hs.size()]];
        HullEdge h{a, b, -1, -1};
        if (nxt[a] == b) { h.tri = etri[a] / 3; h.k = etri[a] % 3; }
        else {
            const auto f = closing.find({a, b});
            if (f == closing.end()) { out.why = "a hull edge has no triangle behind it"; out.fill.clear(); out.hull.clear(); return out; }
            h.tri = nels + f->second / 3; h.k = f->second % 3;
        }
        out.hull.push_back(h);
    }
    out.ok = true;
    return out;
}

}  // namespace nxs_hull

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
// bnd_given: the boundary edges (3 * triangle + k, ascending) when the caller has them already -- the regrid context takes them from the
// ElementConnectivity it builds on the device, which spares the host the pass over every triangle (25-30 ms at 1.5 M triangles)
inline Completion complete(const int32_t *index, const int *ix, const int *iy, int nods, int nels, const std::vector<int> *bnd_given = nullptr) {
    Completion out;
    const Pts P{ix, iy};
    static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    // boundary edges, oriented as in their triangle (the domain on the left)
    std::vector<int> bnd;
    if (bnd_given) bnd = *bnd_given;
    else if (!find_boundary_edges(index, nods, nels, bnd)) { out.why = "an edge belongs to more than two triangles"; return out; }
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

// ------------------------------------------------------------------------------------------------
// The general case: a mesh with several components, or whose boundary pinches (a vertex with more than one outgoing boundary edge), has no
// "outer loop with pockets".  What bamg builds there is still the same object -- the Delaunay triangulation of the boundary vertices with every
// boundary edge forced, minus the triangles inside the domain (Mesh.cpp:3262-3328) -- so it is built that way: incremental Delaunay in
// lexicographic order (every new vertex lies outside the hull of the earlier ones: each hull edge it sees gets a triangle, Lawson flips
// restore the empty-circle property), the boundary edges that are missing recovered by flipping the edges they cross (Sloan), the free edges
// made Delaunay again, and a flood fill from the boundary edges (the domain is on the left of each) that tells the fill triangles from the
// domain's.  Exact integer predicates throughout.  On the meshes the pocket construction above covers both give the same triangles (tested).
struct CDT {
    typedef std::pair<int, int> E;
    const Pts &P;
    std::vector<int> t;        // 3 per triangle, counter-clockwise
    std::map<E, int> half;     // directed edge -> 3 * triangle + position of its first vertex
    std::map<E, char> fixed;   // forced edges, as (min, max)
    explicit CDT(const Pts &p) : P(p) {}
    static E und(int a, int b) { return a < b ? E{a, b} : E{b, a}; }
    int add(int a, int b, int c) {
        const int i = (int)t.size() / 3;
        t.push_back(a); t.push_back(b); t.push_back(c);
        half[{a, b}] = 3 * i; half[{b, c}] = 3 * i + 1; half[{c, a}] = 3 * i + 2;
        return i;
    }
    // flips the edge {a, b} when both triangles exist and their quadrilateral is strictly convex; the new diagonal is (c, d)
    bool flip(int a, int b, int &c, int &d) {
        const auto h = half.find({a, b}), o = half.find({b, a});
        if (h == half.end() || o == half.end()) return false;
        const int i = h->second / 3, ki = h->second % 3, j = o->second / 3, kj = o->second % 3;
        c = t[3 * i + (ki + 2) % 3]; d = t[3 * j + (kj + 2) % 3];  // a, b, c and b, a, d
        if (P.orient(c, a, d) <= 0 || P.orient(d, b, c) <= 0) return false;
        half.erase({a, b}); half.erase({b, a});
        t[3 * i] = c; t[3 * i + 1] = a; t[3 * i + 2] = d;
        t[3 * j] = d; t[3 * j + 1] = b; t[3 * j + 2] = c;
        half[{c, a}] = 3 * i; half[{a, d}] = 3 * i + 1; half[{d, c}] = 3 * i + 2;
        half[{d, b}] = 3 * j; half[{b, c}] = 3 * j + 1; half[{c, d}] = 3 * j + 2;
        return true;
    }
    // Lawson: flips the listed free edges (and what the flips disturb) until each is locally Delaunay
    bool lawson(std::vector<E> &work) {
        long long guard = 64ll * (long long)(t.size() / 3 + 16) * 64;
        while (!work.empty()) {
            if (guard-- <= 0) return false;
            const E e = work.back();
            work.pop_back();
            if (fixed.count(und(e.first, e.second))) continue;
            const auto h = half.find({e.first, e.second}), o = half.find({e.second, e.first});
            if (h == half.end() || o == half.end()) continue;
            const int a = e.first, b = e.second;
            const int c = t[3 * (h->second / 3) + (h->second % 3 + 2) % 3], d = t[3 * (o->second / 3) + (o->second % 3 + 2) % 3];
            if (!P.in_circle(a, b, c, d)) continue;
            int c2, d2;
            if (!flip(a, b, c2, d2)) continue;
            work.push_back({c, a}); work.push_back({a, d}); work.push_back({d, b}); work.push_back({b, c});
        }
        return true;
    }
};

// index: 1-based triangles; the general construction (see above).  Same output as complete().
inline Completion complete_general(const int32_t *index, const int *ix, const int *iy, int nods, int nels, const std::vector<int> *bnd_given = nullptr) {
    Completion out;
    const Pts P{ix, iy};
    typedef CDT::E E;
    static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    std::vector<int> bnd;
    if (bnd_given) bnd = *bnd_given;
    else if (!find_boundary_edges(index, nods, nels, bnd)) { out.why = "an edge belongs to more than two triangles"; return out; }
    if (bnd.size() < 3) { out.why = "no boundary"; return out; }
    std::map<E, int> etri;  // directed boundary edge p -> q (domain on its left) -> 3 * triangle + k
    std::vector<int> pts;
    {
        std::vector<char> isb(nods, 0);
        for (int be : bnd) {
            const int e = be / 3, k = be % 3;
            const int p = index[3 * e + VOTE[k][0]] - 1, q = index[3 * e + VOTE[k][1]] - 1;
            etri[{p, q}] = be;
            isb[p] = isb[q] = 1;
        }
        for (int i = 0; i < nods; ++i) if (isb[i]) pts.push_back(i);
    }
    std::sort(pts.begin(), pts.end(), [&](int p, int q) { return ix[p] != ix[q] ? ix[p] < ix[q] : iy[p] < iy[q]; });
    for (size_t i = 1; i < pts.size(); ++i)
        if (ix[pts[i]] == ix[pts[i - 1]] && iy[pts[i]] == iy[pts[i - 1]]) { out.why = "two boundary vertices share one integer point"; return out; }
    const int n = (int)pts.size();
    // ---- incremental Delaunay of the boundary vertices in lexicographic order
    CDT T(P);
    int apex = 2;  // the first vertex not collinear with pts[0], pts[1]
    while (apex < n && P.orient(pts[0], pts[1], pts[apex]) == 0) ++apex;
    if (apex >= n) { out.why = "every boundary vertex lies on one line"; return out; }
    std::vector<int> hnext(n, -1), hprev(n, -1);  // the hull, counter-clockwise, as a circular list over the positions in pts
    {
        // the collinear run pts[0 .. apex-1] (ascending along its line) and the apex: a fan of triangles
        const bool left = P.orient(pts[0], pts[1], pts[apex]) > 0;
        for (int i = 0; i + 1 < apex; ++i) { if (left) T.add(pts[i], pts[i + 1], pts[apex]); else T.add(pts[i + 1], pts[i], pts[apex]); }
        std::vector<int> cyc;
        if (left) { for (int i = 0; i < apex; ++i) cyc.push_back(i); }
        else { for (int i = apex - 1; i >= 0; --i) cyc.push_back(i); }
        cyc.push_back(apex);
        const int m = (int)cyc.size();
        for (int i = 0; i < m; ++i) { hnext[cyc[i]] = cyc[(i + 1) % m]; hprev[cyc[(i + 1) % m]] = cyc[i]; }
    }
    std::vector<E> work;
    int last = apex;  // the newest vertex: always on the hull, and a new vertex always sees an edge next to it or beyond
    for (int i = apex + 1; i < n; ++i) {
        const int p = pts[i];
        auto visible = [&](int h) { return P.orient(pts[h], pts[hnext[h]], p) < 0; };
        int e0 = -1, steps = 0;
        for (int h = hprev[last]; steps <= n + 1; h = hnext[h], ++steps) if (visible(h)) { e0 = h; break; }
        if (e0 < 0) { out.why = "internal: no hull edge visible from a new boundary vertex"; return out; }
        int lo = e0, hi = e0;  // the strictly visible edges are one chain lo .. hi
        while (hprev[lo] != hi && visible(hprev[lo])) lo = hprev[lo];
        while (hnext[hi] != lo && visible(hnext[hi])) hi = hnext[hi];
        const int end = hnext[hi];
        for (int h = lo; h != end;) {
            const int nx = hnext[h];
            T.add(pts[nx], pts[h], p);   // (b, a, p) for the hull edge a -> b: counter-clockwise
            work.push_back({pts[h], pts[nx]});
            if (h != lo) { hnext[h] = -1; hprev[h] = -1; }
            h = nx;
        }
        hnext[lo] = i; hprev[i] = lo; hnext[i] = end; hprev[end] = i;
        last = i;
        if (!T.lawson(work)) { out.why = "internal: Delaunay flips do not terminate"; return out; }
    }
    // ---- force the boundary edges (Sloan: flip what crosses a missing edge until it appears)
    std::vector<E> loosened;
    for (const auto &kv : etri) {
        const int p = kv.first.first, q = kv.first.second;
        long long guard = 8ll * n + 64;
        while (!T.half.count({p, q}) && !T.half.count({q, p})) {
            if (guard-- <= 0) { out.why = "a boundary edge could not be recovered in the triangulation of the boundary vertices (crossing boundary edges?)"; return out; }
            // the triangle at p the segment p -> q leaves through: (p, x, y) with x strictly left of p -> q ... no: x right, y left of it
            int x = -1, y = -1;
            for (auto it = T.half.lower_bound({p, -1}); it != T.half.end() && it->first.first == p; ++it) {
                const int tri = it->second / 3, k = it->second % 3;
                const int a = T.t[3 * tri + (k + 1) % 3], b = T.t[3 * tri + (k + 2) % 3];  // triangle (p, a, b), counter-clockwise
                const i64 oa = P.orient(p, q, a), ob = P.orient(p, q, b);
                if (oa == 0 && ((i64)ix[a] - ix[p]) * ((i64)ix[q] - ix[p]) + ((i64)iy[a] - iy[p]) * ((i64)iy[q] - iy[p]) > 0) { out.why = "a boundary vertex lies on another boundary edge"; return out; }
                if (oa < 0 && ob > 0) { x = a; y = b; break; }  // a on the right, b on the left: the segment crosses a - b
            }
            if (x < 0) { out.why = "internal: a missing boundary edge leaves its vertex through no triangle"; return out; }
            // walk along p -> q flipping the first flippable crossed edge
            bool flipped = false;
            int r = x, l = y;  // the crossed edge: r on the right, l on the left; it is the directed edge r -> l of the triangle on p's side
            for (long long w = 0; w < 8ll * n + 64 && !flipped; ++w) {
                int c, d;
                if (!T.fixed.count(CDT::und(r, l)) && T.flip(r, l, c, d)) {
                    flipped = true;
                    // (the new diagonal may still cross p -> q: the outer loop looks again)
                    if (!((c == p && d == q) || (c == q && d == p))) loosened.push_back(CDT::und(c, d));
                    break;
                }
                if (T.fixed.count(CDT::und(r, l))) { out.why = "two boundary edges cross"; return out; }
                // not convex here: go on to the next crossed edge, across r -> l
                const auto o = T.half.find({l, r});
                if (o == T.half.end()) { out.why = "internal: a missing boundary edge leaves the hull"; return out; }
                const int z = T.t[3 * (o->second / 3) + (o->second % 3 + 2) % 3];
                if (z == q) break;
                const i64 oz = P.orient(p, q, z);
                if (oz == 0) { out.why = "a boundary vertex lies on another boundary edge"; return out; }
                if (oz > 0) l = z; else r = z;
            }
            if (!flipped) {
                // every crossed edge sits in a non-convex quadrilateral right now: flip them from the far end (the last one is always convex towards q)
                // -- rare; fall back to trying every crossed edge in reverse order
                std::vector<E> crossed;
                int rr = x, ll = y;
                for (long long w = 0; w < 8ll * n + 64; ++w) {
                    crossed.push_back({rr, ll});
                    const auto o = T.half.find({ll, rr});
                    if (o == T.half.end()) break;
                    const int z = T.t[3 * (o->second / 3) + (o->second % 3 + 2) % 3];
                    if (z == q) break;
                    if (P.orient(p, q, z) > 0) ll = z; else rr = z;
                }
                for (size_t k = crossed.size(); k-- > 0 && !flipped;) {
                    int c, d;
                    if (!T.fixed.count(CDT::und(crossed[k].first, crossed[k].second)) && T.flip(crossed[k].first, crossed[k].second, c, d)) {
                        flipped = true;
                        if (!((c == p && d == q) || (c == q && d == p))) loosened.push_back(CDT::und(c, d));
                    }
                }
                if (!flipped) { out.why = "a boundary edge could not be recovered (no crossed edge can be flipped)"; return out; }
            }
        }
        T.fixed[CDT::und(p, q)] = 1;
    }
    // ---- the free edges Delaunay again (all of them: the recovery flips were not Delaunay flips)
    {
        std::vector<E> all;
        for (const auto &kv : T.half) if (kv.first.first < kv.first.second) all.push_back(kv.first);
        if (!T.lawson(all)) { out.why = "internal: constrained Delaunay flips do not terminate"; return out; }
    }
    // ---- which triangles are the domain's: flood fill from the boundary edges, never across one
    const int nt = (int)T.t.size() / 3;
    std::vector<signed char> side(nt, 0);  // +1 inside the domain, -1 outside (fill)
    std::vector<int> stack;
    for (const auto &kv : etri) {
        const auto in = T.half.find({kv.first.first, kv.first.second});
        if (in == T.half.end()) { out.why = "internal: a forced boundary edge has no triangle on the domain's side"; return out; }
        if (side[in->second / 3] == -1) { out.why = "the boundary edges do not bound a consistent domain"; return out; }
        if (side[in->second / 3] == 0) { side[in->second / 3] = 1; stack.push_back(in->second / 3); }
        const auto ot = T.half.find({kv.first.second, kv.first.first});
        if (ot != T.half.end() && !etri.count({kv.first.second, kv.first.first})) {
            if (side[ot->second / 3] == 1) { out.why = "the boundary edges do not bound a consistent domain"; return out; }
            if (side[ot->second / 3] == 0) { side[ot->second / 3] = -1; stack.push_back(ot->second / 3); }
        }
    }
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        for (int k = 0; k < 3; ++k) {
            const int a = T.t[3 * i + k], b = T.t[3 * i + (k + 1) % 3];
            if (T.fixed.count(CDT::und(a, b))) continue;
            const auto o = T.half.find({b, a});
            if (o == T.half.end()) continue;
            const int j = o->second / 3;
            if (side[j] == 0) { side[j] = side[i]; stack.push_back(j); }
            else if (side[j] != side[i]) { out.why = "the boundary edges do not bound a consistent domain"; return out; }
        }
    }
    std::vector<int> fill_no(nt, -1);
    for (int i = 0; i < nt; ++i) {
        if (side[i] > 0) continue;  // (0: a region no boundary edge touches cannot exist inside the hull of the boundary vertices; taken as outside)
        fill_no[i] = (int)out.fill.size() / 3;
        out.fill.push_back(T.t[3 * i]); out.fill.push_back(T.t[3 * i + 1]); out.fill.push_back(T.t[3 * i + 2]);
    }
    // ---- hull edges, counter-clockwise, with the triangle inside each
    {
        int start = last, steps = 0;
        for (int h = start; steps <= n; ++steps) {
            const int a = pts[h], b = pts[hnext[h]];
            HullEdge he{a, b, -1, -1};
            const auto in = T.half.find({a, b});
            if (in == T.half.end()) { out.why = "internal: a hull edge has no triangle behind it"; out.fill.clear(); out.hull.clear(); return out; }
            const int i = in->second / 3;
            if (side[i] > 0) {
                const auto m = etri.find({a, b});
                if (m == etri.end()) { out.why = "internal: a domain triangle on the hull without a boundary edge"; out.fill.clear(); out.hull.clear(); return out; }
                he.tri = m->second / 3; he.k = m->second % 3;
            } else {
                const int pos = in->second % 3;         // a is the vertex at `pos`, b the next: they are VOTE[k] = ((k+1)%3, (k+2)%3) for k = (pos + 2) % 3
                he.tri = nels + fill_no[i]; he.k = (pos + 2) % 3;
            }
            out.hull.push_back(he);
            h = hnext[h];
            if (h == start) break;
        }
    }
    out.ok = true;
    return out;
}

// complete(), and where it does not apply (several components, a pinching boundary) the general construction
inline Completion complete_any(const int32_t *index, const int *ix, const int *iy, int nods, int nels, int mode = 0, const std::vector<int> *bnd_given = nullptr) {
    if (mode == 1) return complete_general(index, ix, iy, nods, nels, bnd_given);
    Completion c = complete(index, ix, iy, nods, nels, bnd_given);
    if (c.ok || mode == 2) return c;
    Completion g = complete_general(index, ix, iy, nods, nels, bnd_given);
    if (!g.ok) g.why = c.why + "; general construction: " + g.why;
    return g;
}

}  // namespace nxs_hull

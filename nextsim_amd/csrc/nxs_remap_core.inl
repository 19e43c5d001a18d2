// nxs_remap_core.inl -- the per-target-triangle work of the conservative remapping at regrid
// (contrib/bamg/src/ConservativeRemapping.cpp:176-328 ConservativeRemappingMeshToMesh, called at FE.cpp:3108),
// written once and compiled twice: as __device__ code into libnxsdyn.so (nxs_interp.hip, one thread per
// new triangle) and as plain host code into the TEST-ONLY oracle/libremap_host.so, so that the very
// functions the kernel runs can be checked against the real contrib/bamg in a container without a GPU.
//
// The reference recurses (checkTriangle, ConservativeRemapping.cpp:330-449) and keeps std::vectors per
// call; a GPU thread has neither.  Two observations remove both:
//   1. the polygon whose area is the weight of an old triangle depends only on (old triangle, new cell),
//      never on the path the recursion took to reach it, so it is computed once on entry;
//   2. the recursion itself only decides WHICH old triangles are visited and in WHAT order (the order of
//      the floating-point sum), so it is replayed with an explicit stack of (triangle, stage, i, j) frames.
// Every floating-point expression below keeps the reference's operand order; std::sort on <= 16 points is
// libstdc++'s insertion sort (bits/stl_algo.h __insertion_sort / __unguarded_linear_insert), replayed here.
#ifndef NXS_REMAP_CORE_INL
#define NXS_REMAP_CORE_INL

#ifndef NXS_HD
#define NXS_HD
#endif

namespace nxs_remap {

constexpr int kMaxVisit = 96;   // old triangles one new triangle may overlap, fast path (lists in per-thread scratch);
constexpr int kMaxVisitBig = 4096;  // second pass for the few that exceed it (lists in global memory); reference: unbounded
constexpr int kMaxPoints = 16;  // 3 nodes + 3 corners + 6 intersections <= 12

struct OldMesh {
    int nels, nods;
    const int *tri;       // [3*nels] 0-based vertices, bamgmesh_old->Triangles
    const double *x, *y;  // bamgmesh_old->Vertices
    const int *nec;       // [nods*nec_w] 0-based element fan per vertex in bamg's row order, -1 = NaN pad
    int nec_w;
    const int *ec;        // [3*nels] 0-based neighbour across local edge j, -1 = NaN (boundary)
};

struct Pt {
    double x, y;
};

// Order of two points (relative to the centroid) on the clockwise tour the reference sorts its polygon along (its comparator:
// ConservativeRemapping.cpp:614-645): quadrants in the order (+,+), (-,+), (-,-), (+,-) -- a coordinate that is exactly zero counts as
// negative -- and inside a quadrant by the sign of the cross product, evaluated as b.x*a.y < b.y*a.x.
NXS_HD inline int tour_quadrant(const Pt &p) {
    return (p.y > 0) ? ((p.x > 0) ? 0 : 1) : ((p.x > 0) ? 3 : 2);
}
NXS_HD inline bool cw_less(Pt a, Pt b) {
    const int ka = tour_quadrant(a), kb = tour_quadrant(b);
    return (ka != kb) ? (ka < kb) : (b.x * a.y < b.y * a.x);
}

// Area of the polygon spanned by n <= 16 unordered points: they are put on the tour above around their mean (which is subtracted
// before the sort and added back after it, as the reference does -- the roundings of both passes are part of the result), then the
// shoelace sum is taken in that order (ConservativeRemapping.cpp:556-611).  The array is left sorted.
NXS_HD inline double polygon_area(Pt *p, int n) {
    if (n < 3) return 0.;
    Pt mean = {0., 0.};
    for (int k = 0; k < n; ++k) { mean.x += p[k].x; mean.y += p[k].y; }
    const double inv_n = 1. / double(n);
    mean.x *= inv_n; mean.y *= inv_n;
    for (int k = 0; k < n; ++k) { p[k].x -= mean.x; p[k].y -= mean.y; }
    // std::sort on fewer than 17 elements is libstdc++'s insertion sort: a new minimum is rotated to the front, anything else sinks
    // from the back without a bound check (bits/stl_algo.h) -- the same comparisons in the same order, hence the same permutation
    // where the comparator is not a strict weak order (coincident points)
    for (int k = 1; k < n; ++k) {
        const Pt moving = p[k];
        int hole = k;
        if (cw_less(moving, p[0])) {
            for (; hole > 0; --hole) p[hole] = p[hole - 1];
        } else {
            for (; cw_less(moving, p[hole - 1]); --hole) p[hole] = p[hole - 1];
        }
        p[hole] = moving;
    }
    for (int k = 0; k < n; ++k) { p[k].x += mean.x; p[k].y += mean.y; }
    double twice = 0.;
    for (int k = 0, before = n - 1; k < n; before = k++) twice += (p[before].x + p[k].x) * (p[before].y - p[k].y);
    return fabs(twice) * 0.5;
}

// Is (qx, qy) inside the triangle (tx, ty)?  The reference's test (ConservativeRemapping.cpp:463-514), with its tolerances: a point
// within 1 mm (in both coordinates) of a corner, or with an edge cross product inside the band of +-1e-8, is "on the boundary" and
// gets the answer the caller asks for (`boundary_counts`); cross products of both signs mean outside.
NXS_HD inline bool inside3(const double *tx, const double *ty, double qx, double qy, bool boundary_counts) {
    const double corner_tol = 1.e-3, band = 1e-8;
    for (int k = 0; k < 3; ++k)
        if (fabs(qx - tx[k]) < corner_tol && fabs(qy - ty[k]) < corner_tol) return boundary_counts;
    int left = 0, right = 0;
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3;
        const double turn = (tx[k1] - tx[k]) * (qy - ty[k]) - (ty[k1] - ty[k]) * (qx - tx[k]);
        if (turn > band) ++left;
        else if (turn < -band) ++right;
    }
    if (left > 0 && right > 0) return false;
    return (left + right < 3) ? boundary_counts : true;
}

// The proper crossings of the segment (x0, y0) -> (x1, y1) with the three edges of the cell, appended to pts in edge order (edge k
// runs from corner k-1 to corner k).  Parametric form of ConservativeRemapping.cpp:518-553: with d the segment and e the edge,
// D = -e.x*d.y + d.x*e.y (edges with |D| < 1e-6 are taken as parallel), both parameters strictly inside (0, 1), the point taken on
// the segment as start + t*d.  Every product and sum in the reference's operand order.
NXS_HD inline bool intersect3(double x1, double y1, double x0, double y0, const double *cx, const double *cy, Pt *pts, int &npts) {
    bool crossed = false;
    const double dx = x1 - x0, dy = y1 - y0;
    for (int k = 0, from = 2; k < 3; from = k++) {
        const double ex = cx[k] - cx[from], ey = cy[k] - cy[from];
        const double D = -ex * dy + dx * ey;
        if (fabs(D) < 1e-6) continue;
        const double inv = 1. / D;
        const double on_edge = (-dy * (x0 - cx[from]) + dx * (y0 - cy[from])) * inv;
        const double on_segment = (ex * (y0 - cy[from]) - ey * (x0 - cx[from])) * inv;
        if (on_edge > 0. && on_edge < 1. && on_segment > 0. && on_segment < 1.) {
            if (npts < kMaxPoints) { pts[npts].x = x0 + (on_segment * dx); pts[npts].y = y0 + (on_segment * dy); }
            ++npts;
            crossed = true;
        }
    }
    return crossed;
}

// What checkTriangle computes for ONE old triangle against the cell, independent of the recursion:
// bits 0-2 inCell[i]; bit 3 "returned early" (all nodes in the cell, or the whole cell in the triangle);
// bits 4-6 edge (i, prev) intersects the cell and is followed.  *weight = the area it stores.
NXS_HD inline int classify(const OldMesh &m, int t, const double *cx, const double *cy, double *weight, int *overflow) {
    Pt pts[kMaxPoints];
    int np = 0, bits = 0;
    double X[3], Y[3];
    for (int i = 0; i < 3; ++i) {
        const int node = m.tri[3 * t + i];
        X[i] = m.x[node];
        Y[i] = m.y[node];
        if (inside3(cx, cy, X[i], Y[i], false)) {
            bits |= 1 << i;
            pts[np].x = X[i]; pts[np].y = Y[i]; ++np;
        }
    }
    if ((bits & 7) == 7) { *weight = polygon_area(pts, np); return bits | 8; }
    int counter = 0;
    for (int c = 0; c < 3; ++c)
        if (inside3(X, Y, cx[c], cy[c], true)) { pts[np].x = cx[c]; pts[np].y = cy[c]; ++np; ++counter; }
    if (counter == 3) { *weight = polygon_area(pts, np); return bits | 8; }
    int prev = 2;
    for (int i = 0; i < 3; prev = i++) {
        if ((bits >> i & 1) && (bits >> prev & 1)) continue;
        if (intersect3(X[i], Y[i], X[prev], Y[prev], cx, cy, pts, np)) bits |= 16 << i;
    }
    if (np > kMaxPoints) { *overflow = 1; np = kMaxPoints; }
    *weight = polygon_area(pts, np);
    return bits;
}

struct Frame {
    int t;
    unsigned char bits, stage, i, j;
};

// One new triangle: corners (cx, cy), the old triangle `seed` that holds its barycentre, `same` = the three
// vertices are those of the seed (ConservativeRemapping.cpp:263-289).  Fills tris/w in the reference's
// push order and returns their number, or -1 on capacity overflow (cap entries in tris / w, cap + 1 frames).
NXS_HD inline int collect(const OldMesh &m, const double *cx, const double *cy, int seed, bool same, int *tris, double *w, Frame *stack,
                          int cap = kMaxVisit) {
    if (same) {
        Pt p[3] = {{cx[0], cy[0]}, {cx[1], cy[1]}, {cx[2], cy[2]}};
        tris[0] = seed;
        w[0] = polygon_area(p, 3);
        return 1;
    }
    int n = 0, sp = 0, overflow = 0;
    auto visited = [&](int e) {
        for (int k = 0; k < n; ++k)
            if (tris[k] == e) return true;
        return false;
    };
    auto enter = [&](int t) {  // checkTriangle's prologue: plant the flag, reserve the weight slot
        tris[n] = t;
        const int bits = classify(m, t, cx, cy, &w[n], &overflow);
        ++n;
        stack[sp].t = t; stack[sp].bits = (unsigned char)bits; stack[sp].stage = 1; stack[sp].i = 0; stack[sp].j = 0;
        ++sp;
    };
    enter(seed);
    while (sp > 0) {
        if (overflow) return -1;
        Frame &f = stack[sp - 1];
        int child = -1;
        if (f.stage == 1) {  // nodes inside the cell: visit their whole element fan
            while (f.i < 3 && child < 0) {
                if (f.bits >> f.i & 1) {
                    const int node = m.tri[3 * f.t + f.i];
                    while (f.j < m.nec_w) {
                        const int e = m.nec[(long long)node * m.nec_w + f.j];
                        if (e < 0) { f.j = (unsigned char)m.nec_w; break; }
                        ++f.j;
                        if (!visited(e)) { child = e; break; }
                    }
                    if (child >= 0) break;
                }
                ++f.i; f.j = 0;
            }
            if (child < 0) {
                if (f.bits & 8) { --sp; continue; }  // the two early returns
                f.stage = 2; f.i = 0; f.j = 0;
            }
        }
        if (child < 0 && f.stage == 2) {  // edges that cut the cell: cross into the neighbour
            while (f.i < 3 && child < 0) {
                if (f.bits >> (4 + f.i) & 1) {
                    const int prev = (f.i + 2) % 3;
                    const int a = m.tri[3 * f.t + f.i], b = m.tri[3 * f.t + prev];
                    while (f.j < 3) {
                        const int e = m.ec[3 * f.t + f.j];
                        if (e < 0) { f.j = 3; break; }
                        ++f.j;
                        int shared = 0;
                        for (int k = 0; k < 3; ++k) {
                            const int id = m.tri[3 * e + k];
                            if (id == a || id == b) ++shared;
                        }
                        if (shared == 2 && !visited(e)) { child = e; break; }
                    }
                    if (child >= 0) break;
                }
                ++f.i; f.j = 0;
            }
            if (child < 0) { --sp; continue; }
        }
        if (child >= 0) {
            if (n >= cap) return -1;
            enter(child);
        }
    }
    return overflow ? -1 : n;
}

// ConservativeRemappingMeshToGrid with num_corners == 3 (ConservativeRemapping.cpp:97-131), one cell
NXS_HD inline void apply(const double *in, int nb_var, const double *cx, const double *cy, const int *tris, const double *w, int n, double *out) {
    Pt p[3] = {{cx[0], cy[0]}, {cx[1], cy[1]}, {cx[2], cy[2]}};
    const double r_cell_area = 1. / polygon_area(p, 3);
    for (int var = 0; var < nb_var; ++var) {
        double v = 0.;
        for (int k = 0; k < n; ++k) v += in[(long long)tris[k] * nb_var + var] * w[k];
        v *= r_cell_area;
        out[var] = v;
    }
}

// nodes_old == nodes_new after sorting (ConservativeRemapping.cpp:263-289).  previous_numbering may be
// NULL (no vertex survived); the ">" (not ">=") against the number of geometric vertices is the reference's.
NXS_HD inline bool same_triangle(const OldMesh &m, int seed, const int *new_tri, const double *previous_numbering, int n_geom) {
    int a[3], b[3];
    for (int i = 0; i < 3; ++i) {
        const int id = new_tri[i];
        if (id > n_geom) a[i] = previous_numbering ? (int)previous_numbering[id] - 1 : -1;
        else a[i] = id;
        b[i] = m.tri[3 * seed + i];
    }
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 2 - i; ++k) {
            if (a[k] > a[k + 1]) { const int s = a[k]; a[k] = a[k + 1]; a[k + 1] = s; }
            if (b[k] > b[k + 1]) { const int s = b[k]; b[k] = b[k + 1]; b[k + 1] = s; }
        }
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
}

}  // namespace nxs_remap
#endif

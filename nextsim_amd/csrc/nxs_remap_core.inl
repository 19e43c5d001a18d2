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

// CWSort, ConservativeRemapping.cpp:614-645
NXS_HD inline bool cw_less(Pt p1, Pt p2) {
    const int dax = (p1.x > 0) ? 1 : 0, day = (p1.y > 0) ? 1 : 0;
    const int qa = (1 - dax) + (1 - day) + ((dax & (1 - day)) << 1);
    const int dbx = (p2.x > 0) ? 1 : 0, dby = (p2.y > 0) ? 1 : 0;
    const int qb = (1 - dbx) + (1 - dby) + ((dbx & (1 - dby)) << 1);
    if (qa == qb) return p2.x * p1.y < p2.y * p1.x;
    return qa < qb;
}

// area() + sortClockwise(), ConservativeRemapping.cpp:556-611.  p is modified as the reference modifies it.
NXS_HD inline double polygon_area(Pt *p, int n) {
    if (n < 3) return 0.;
    double cx = 0., cy = 0.;
    for (int i = 0; i < n; ++i) { cx += p[i].x; cy += p[i].y; }
    const double rn = 1. / double(n);
    cx *= rn; cy *= rn;
    for (int i = 0; i < n; ++i) { p[i].x -= cx; p[i].y -= cy; }
    for (int i = 1; i < n; ++i) {  // std::sort, n <= 16
        const Pt val = p[i];
        if (cw_less(val, p[0])) {
            for (int k = i; k > 0; --k) p[k] = p[k - 1];
            p[0] = val;
        } else {
            int last = i;
            while (cw_less(val, p[last - 1])) { p[last] = p[last - 1]; --last; }
            p[last] = val;
        }
    }
    for (int i = 0; i < n; ++i) { p[i].x += cx; p[i].y += cy; }
    double area = 0.;
    int j = n - 1;
    for (int i = 0; i < n; j = i++) area += (p[j].x + p[i].x) * (p[j].y - p[i].y);
    return fabs(area) * 0.5;
}

// checkIfInside, ConservativeRemapping.cpp:463-514 (always a 3-gon here)
NXS_HD inline bool inside3(const double *vx, const double *vy, double tx, double ty, bool inclusive) {
    const double eps = 1.e-3;
    for (int i = 0; i < 3; ++i)
        if (fabs(tx - vx[i]) < eps && fabs(ty - vy[i]) < eps) return inclusive;
    bool hasPos = false, hasNeg = false, hasMaybe = false;
    const double epsx = 1e-8;
    for (int i = 0; i < 3; ++i) {
        const double ax = vx[i], ay = vy[i];
        const int n = (i + 1) % 3;
        const double bx = vx[n], by = vy[n];
        const double cp = (bx - ax) * (ty - ay) - (by - ay) * (tx - ax);
        if (cp > epsx) hasPos = true;
        else if (cp < -epsx) hasNeg = true;
        else hasMaybe = true;
        if (hasPos && hasNeg) return false;
    }
    if (hasMaybe) return inclusive;
    return true;
}

// checkIfIntersecting, ConservativeRemapping.cpp:518-553
NXS_HD inline bool intersect3(double X, double Y, double Xp, double Yp, const double *cx, const double *cy, Pt *pts, int &npts) {
    bool ret = false;
    const double s1_x = X - Xp, s1_y = Y - Yp;
    int prev = 2;
    for (int i = 0; i < 3; prev = i++) {
        const double s2_x = cx[i] - cx[prev], s2_y = cy[i] - cy[prev];
        const double det = -s2_x * s1_y + s1_x * s2_y;
        if (fabs(det) < 1e-6) continue;
        const double rdet = 1. / det;
        const double s = (-s1_y * (Xp - cx[prev]) + s1_x * (Yp - cy[prev])) * rdet;
        const double t = (s2_x * (Yp - cy[prev]) - s2_y * (Xp - cx[prev])) * rdet;
        if (s > 0. && s < 1. && t > 0. && t < 1.) {
            if (npts < kMaxPoints) { pts[npts].x = Xp + (t * s1_x); pts[npts].y = Yp + (t * s1_y); }
            ++npts;
            ret = true;
        }
    }
    return ret;
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

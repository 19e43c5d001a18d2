// nxs_interp.hip -- mesh-to-mesh interpolation at regrid as a HIP gather kernel (include/nxs_interp.h).
//
// Reference: contrib/bamg/src/InterpFromMeshToMesh2dx.cpp:17-179 (called at FE.cpp:3131-3139), which
// locates every target point by walking bamg's triangulation with EXACT integer predicates:
//   integer coordinates  Mesh::SetIntCoor  contrib/bamg/src/Mesh.cpp:3441-3468  (bbox grown by 5 %,
//                        coefIcoor = (2^30-1)/max extent, truncation)   and R2ToI2 Mesh.cpp:3688-3690
//   determinants         include/det.h:8-12 (64-bit)
//   area coordinates     det3[k]/det, InterpFromMeshToMesh2dx.cpp:113-116
// The same integers are used here, so a point inside the data mesh gets the reference's weights bit for
// bit whatever triangle search is used; the search itself is a uniform bucket grid over the integer plane
// (one thread per target point, candidates tested in ascending triangle number).
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <future>
#include <map>
#include <memory>
#include <string>
#include <system_error>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_guard.hpp"
#include "nxs_interp.h"
#include "nxs_hull.inl"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// handler of every extern "C" function-try-block of this file (nxs_guard.hpp): status code + text, never an exception across the ABI
int entry_caught(const char *entry) noexcept {
    return nxs_guard::caught(entry, [](int code, const char *text) { (void)fail(code, "%s", text); });
}

// Where the time of the last regrid call of this thread went (nxs_interp_last_timing): [0] connectivity tables of the old mesh, [1] integer
// plane + bucket grid, [2] convex completion, [3] host -> device copies, [4] kernels (HIP events), [5] device -> host copies, [6] whole call; ms.
thread_local double g_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
struct Tick {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit Tick(int s) : slot(s), t0(std::chrono::steady_clock::now()) {}
    ~Tick() { g_ms[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// Every device operation of this file runs on a stream of its own (one per host thread, created on first use), never on the
// legacy default stream: a host that drives several GPUs from one process may be capturing a graph of the dynamics step on
// another thread, and legacy-stream work is refused while a capture is open.
hipStream_t S() {
    thread_local hipStream_t st = nullptr;
    thread_local int st_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!st || st_dev != dev) {
        st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; (void)hipGetLastError(); }
        st_dev = dev;
    }
    return st;
}
hipError_t copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, S());
    return e != hipSuccess ? e : hipStreamSynchronize(S());
}
hipError_t memset_sync(void *dst, int value, size_t bytes) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, S());
    return e != hipSuccess ? e : hipStreamSynchronize(S());
}

struct BEdge {  // a boundary edge as seen from its (inside) triangle
    int tri, k;  // triangle number, local edge index (vertices VOTE[k][0] -> VOTE[k][1])
};

struct HullDev { int a, b, tri, k; };  // nxs_hull::HullEdge

struct InterpDev {
    int nods, nels, N_data, N_interp, nodal;
    int nels_all;                  // nels + the fill triangles of bamg's convex completion (isdefault == 0 only), numbered behind the mesh's
    int nhull;                     // hull edges, counter-clockwise; 0 = no completion (then the nearest boundary edge stands in)
    const HullDev *hull;
    const int *t0, *t1, *t2;       // 0-based vertices
    const int *ix, *iy;            // integer coordinates of the data vertices
    int G, shift;                  // bucket grid: G x G cells of 2^shift integer units
    const int *cell_off, *cell_tri;
    int nbe;
    const BEdge *bedges;
    double coef, pminx, pminy;     // R2ToI2
    double xmin, xmax, ymin, ymax; // unexpanded bbox (isdefault test, InterpFromMeshToMesh2dx.cpp:93)
    int isdefault;
    double defaultvalue;
};

__device__ __forceinline__ long long det3(long long ax, long long ay, long long bx, long long by, long long cx, long long cy) {
    // include/det.h:8-12
    const long long bax = bx - ax, bay = by - ay, cax = cx - ax, cay = cy - ay;
    return bax * cay - bay * cax;
}

// Exact point location: R2ToI2 (Mesh.cpp:3688-3690: truncation of coefIcoor*(P - pmin)), then the candidates of
// the bucket in ascending triangle number.  Returns the triangle (dd = its three integer area coordinates) or
// -1 when the point is in no triangle of the data mesh; Bx/By = the integer point.
__device__ __forceinline__ int locate(const InterpDev &d, double x, double y, long long dd[3], long long &Bx, long long &By) {
    const double fx = d.coef * (x - d.pminx), fy = d.coef * (y - d.pminy);
    const bool in_plane = fx > -1. && fx < 1073741824. && fy > -1. && fy < 1073741824.;
    // (Icoor1) of a double: truncation; beyond the range of an int the x86 conversion the reference runs on yields INT_MIN.  A
    // point outside the integer plane (more than 5 % of the extent beyond the data mesh's bounding box) keeps its true integer
    // coordinates -- the hull projection below needs them
    const bool fits_x = fx > -2147483649. && fx < 2147483648., fits_y = fy > -2147483649. && fy < 2147483648.;
    Bx = fits_x ? (long long)(int)fx : -2147483648ll;
    By = fits_y ? (long long)(int)fy : -2147483648ll;
    if (in_plane && Bx >= 0 && By >= 0) {
        const int cx = (int)(Bx >> d.shift), cy = (int)(By >> d.shift);
        if (cx < d.G && cy < d.G) {
            const int c = cy * d.G + cx;
            for (int q = d.cell_off[c]; q < d.cell_off[c + 1]; ++q) {
                const int t = d.cell_tri[q];
                const int v0 = d.t0[t], v1 = d.t1[t], v2 = d.t2[t];
                const long long x0 = d.ix[v0], y0 = d.iy[v0], x1 = d.ix[v1], y1 = d.iy[v1], x2 = d.ix[v2], y2 = d.iy[v2];
                const long long e0 = det3(x1, y1, x2, y2, Bx, By);  // area coordinate of vertex 0
                const long long e1 = det3(x2, y2, x0, y0, Bx, By);
                const long long e2 = det3(x0, y0, x1, y1, Bx, By);
                if (e0 >= 0 && e1 >= 0 && e2 >= 0 && (e0 + e1 + e2) > 0) { dd[0] = e0; dd[1] = e1; dd[2] = e2; return t; }
            }
        }
    }
    return -1;
}

// counters of one call: [0] points in no triangle of the data mesh, [1] of those, interpolated inside a fill triangle,
// [2] projected on the hull (CloseBoundaryEdge), [3] handled by the nearest-boundary-edge stand-in
__global__ void __launch_bounds__(256) k_interp(InterpDev d, const double *__restrict__ data, const double *__restrict__ xi,
                                                const double *__restrict__ yi, double *__restrict__ out, int *num_exterior) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.N_interp) return;
    const double x = xi[i], y = yi[i];
    double *o = out + (size_t)i * d.N_data;
    if (d.isdefault && (x < d.xmin || x > d.xmax || y < d.ymin || y > d.ymax)) {
        for (int j = 0; j < d.N_data; ++j) o[j] = d.defaultvalue;
        return;
    }
    long long dd[3] = {0, 0, 0}, Bx, By;
    int it = locate(d, x, y, dd, Bx, By);
    double a[3];
    if (it >= d.nels) {  // inside the hull, outside the mesh: a fill triangle of bamg's reconstructed mesh (Mesh.cpp:3135-3440)
        atomicAdd(num_exterior, 1);
        if (d.isdefault) {   // InterpFromMeshToMesh2dx.cpp:124-128 (reft < 0)
            for (int j = 0; j < d.N_data; ++j) o[j] = d.defaultvalue;
            return;
        }
        if (d.nodal) atomicAdd(num_exterior + 1, 1);
        else it = -2;        // element data: the reference stops with "Triangle number ... not in [0 nels]"; nearest boundary edge here
    }
    if (it >= 0) {
        const long long det = dd[0] + dd[1] + dd[2];  // == det(v0,v1,v2) exactly
        a[0] = (double)dd[0] / det;                   // InterpFromMeshToMesh2dx.cpp:113-116
        a[1] = (double)dd[1] / det;
        a[2] = (double)dd[2] / det;
    } else {
        if (d.isdefault) {
            for (int j = 0; j < d.N_data; ++j) o[j] = d.defaultvalue;
            return;
        }
        if (it == -1) atomicAdd(num_exterior, 1);
        const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
        // Beyond the hull: CloseBoundaryEdge (Mesh.cpp:4590-4627) slides along the hull edges until the point projects inside one
        // (a = IJ.AJ / IJ2 for I, b = IJ.IA / IJ2 for J, placed in the slots of the triangle behind that edge,
        // InterpFromMeshToMesh2dx.cpp:137-147) or changes direction at a vertex (weight 1 there).  On a convex hull the outward
        // strips of the edges and the cones of the vertices tile the outside, so the edge or vertex it stops at does not depend on
        // where it starts: the strip of a VISIBLE edge that holds the point, else the nearest hull vertex.
        if (it == -1 && d.nhull > 0 && d.nodal) {
            int hit = -1;
            double ha = 0., hb = 0.;
            long long bestd = 0x7fffffffffffffffll;
            int bestv = -1;
            for (int e = 0; e < d.nhull; ++e) {
                const HullDev h = d.hull[e];
                const long long ax = d.ix[h.a], ay = d.iy[h.a], bx = d.ix[h.b], by = d.iy[h.b];
                if (det3(ax, ay, bx, by, Bx, By) >= 0) continue;  // not visible from the point
                // seen from outside the edge runs I = b -> J = a
                const long long IJx = ax - bx, IJy = ay - by;
                const long long IJ_IA = IJx * (Bx - bx) + IJy * (By - by), IJ_AJ = IJx * (ax - Bx) + IJy * (ay - By);
                if (IJ_IA >= 0 && IJ_AJ >= 0) {
                    const double IJ2 = (double)(IJ_IA + IJ_AJ);
                    hit = e; ha = IJ_AJ / IJ2; hb = IJ_IA / IJ2;
                    break;
                }
                const long long da = (Bx - ax) * (Bx - ax) + (By - ay) * (By - ay), db = (Bx - bx) * (Bx - bx) + (By - by) * (By - by);
                if (da < bestd) { bestd = da; bestv = h.a; }
                if (db < bestd) { bestd = db; bestv = h.b; }
            }
            if (hit >= 0) {
                atomicAdd(num_exterior + 2, 1);
                const HullDev h = d.hull[hit];
                const int tv[3] = {d.t0[h.tri], d.t1[h.tri], d.t2[h.tri]};
                // in the triangle behind the edge: VOTE[k][1] is I = b, VOTE[k][0] is J = a
                a[VOTE[h.k][1]] = ha;
                a[VOTE[h.k][0]] = hb;
                a[h.k] = 1 - ha - hb;
                for (int j = 0; j < d.N_data; ++j)
                    o[j] = a[0] * data[(size_t)d.N_data * tv[0] + j] + a[1] * data[(size_t)d.N_data * tv[1] + j] + a[2] * data[(size_t)d.N_data * tv[2] + j];
                return;
            }
            if (bestv >= 0) {  // a = 1, b = 0 at a hull vertex: 1 * data + 0 * data + 0 * data
                atomicAdd(num_exterior + 2, 1);
                for (int j = 0; j < d.N_data; ++j) o[j] = data[(size_t)d.N_data * bestv + j];
                return;
            }
        }
        // stand-in (no completion for this mesh, or element data): nearest boundary edge, a/b as CloseBoundaryEdge
        atomicAdd(num_exterior + 3, 1);
        double best = INFINITY;
        int bk = -1;
        double ba = 0., bb = 0.;
        for (int e = 0; e < d.nbe; ++e) {
            const BEdge be = d.bedges[e];
            const int tv[3] = {d.t0[be.tri], d.t1[be.tri], d.t2[be.tri]};
            // seen from OUTSIDE the edge runs from the triangle's VOTE[k][1] to VOTE[k][0]
            const int vI = tv[VOTE[be.k][1]], vJ = tv[VOTE[be.k][0]];
            const long long Ix = d.ix[vI], Iy = d.iy[vI], Jx = d.ix[vJ], Jy = d.iy[vJ];
            const long long IJx = Jx - Ix, IJy = Jy - Iy;
            const long long IJ_IA = IJx * (Bx - Ix) + IJy * (By - Iy);
            const long long IJ_AJ = IJx * (Jx - Bx) + IJy * (Jy - By);
            double aa, b2, dist;
            if (IJ_IA < 0) { aa = 1.; b2 = 0.; dist = hypot((double)(Bx - Ix), (double)(By - Iy)); }
            else if (IJ_AJ < 0) { aa = 0.; b2 = 1.; dist = hypot((double)(Bx - Jx), (double)(By - Jy)); }
            else {
                const double IJ2 = (double)(IJ_IA + IJ_AJ);
                aa = IJ_AJ / IJ2;
                b2 = IJ_IA / IJ2;
                const double cr = (double)(IJx * (By - Iy) - IJy * (Bx - Ix));
                dist = fabs(cr) / sqrt(IJ2);
            }
            if (dist < best) { best = dist; bk = e; ba = aa; bb = b2; }
        }
        if (bk < 0) { for (int j = 0; j < d.N_data; ++j) o[j] = d.defaultvalue; return; }
        const BEdge be = d.bedges[bk];
        it = be.tri;
        // InterpFromMeshToMesh2dx.cpp:141-143, expressed in the inside triangle's local numbering
        a[VOTE[be.k][1]] = ba;
        a[VOTE[be.k][0]] = bb;
        a[be.k] = 1 - ba - bb;
    }
    if (d.nodal) {
        const int i0 = d.t0[it], i1 = d.t1[it], i2 = d.t2[it];
        for (int j = 0; j < d.N_data; ++j)  // InterpFromMeshToMesh2dx.cpp:151-156
            o[j] = a[0] * data[(size_t)d.N_data * i0 + j] + a[1] * data[(size_t)d.N_data * i1 + j] + a[2] * data[(size_t)d.N_data * i2 + j];
    } else {
        for (int j = 0; j < d.N_data; ++j) o[j] = data[(size_t)d.N_data * it + j];
    }
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {   // (a buffer that is allocated again lets go of what it held)
        if (p) { (void)hipFree(p); p = nullptr; }
        return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? 0 : -1;
    }
    int upload(const T *src, size_t n) {
        Tick tk(3);
        if (alloc(n)) return -1;
        return (n == 0 || copy_sync(p, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess) ? 0 : -1;
    }
};

}  // namespace

#include "nxs_regrid_tables.inl"

namespace {

struct GridDev {
    int nods, nels, N_data, nrows, ncols, nodal;
    const int *t0, *t1, *t2;
    const double *x, *y;
    const double *xg, *yg;  // grid coordinates as the reference builds them (InterpFromMeshToGridx.cpp:73-92)
    int G; double bx0, by0, bdx, bdy;  // bucket grid over the mesh bbox
    const int *cell_off, *cell_tri;
    double default_value;
};

// one thread per grid point; candidates in ascending element number, the LAST hit wins (the reference's
// element loop overwrites, InterpFromMeshToGridx.cpp:96-181)
__global__ void __launch_bounds__(256) k_mesh_to_grid(GridDev d, const double *__restrict__ data, double *__restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)d.nrows * d.ncols) return;
    const int i = (int)(gid / d.ncols), j = (int)(gid % d.ncols);
    const double xg = d.xg[i], yg = d.yg[j];
    double *o = out + (size_t)d.N_data * gid;
    for (int k = 0; k < d.N_data; ++k) o[k] = d.default_value;
    const int cx = (int)floor((xg - d.bx0) / d.bdx), cy = (int)floor((yg - d.by0) / d.bdy);
    if (cx < 0 || cy < 0 || cx >= d.G || cy >= d.G) return;
    const int c = cy * d.G + cx;
    int hit = -1;
    double h1 = 0., h2 = 0., h3 = 0.;
    for (int q = d.cell_off[c]; q < d.cell_off[c + 1]; ++q) {
        const int n = d.cell_tri[q];
        const int a = d.t0[n], b = d.t1[n], c3 = d.t2[n];
        const double x1 = d.x[a], y1 = d.y[a], x2 = d.x[b], y2 = d.y[b], x3 = d.x[c3], y3 = d.y[c3];
        double xmin = x1, xmax = x1, ymin = y1, ymax = y1;
        if (x2 < xmin) xmin = x2; if (x2 > xmax) xmax = x2; if (y2 < ymin) ymin = y2; if (y2 > ymax) ymax = y2;
        if (x3 < xmin) xmin = x3; if (x3 > xmax) xmax = x3; if (y3 < ymin) ymin = y3; if (y3 > ymax) ymax = y3;
        if ((xg > xmax) || (xg < xmin) || (yg > ymax) || (yg < ymin)) continue;  // :143, :148
        const double area = x2 * y3 - y2 * x3 + x1 * y2 - y1 * x2 + x3 * y1 - y3 * x1;  // :135-137
        const double area_1 = ((xg - x3) * (y2 - y3) - (yg - y3) * (x2 - x3)) / area;     // :152-153
        const double area_2 = ((x1 - x3) * (yg - y3) - (y1 - y3) * (xg - x3)) / area;     // :155-156
        const double area_3 = 1 - area_1 - area_2;
        if (area_1 > -10e-12 && area_2 > -10e-12 && area_3 > -10e-12) { hit = n; h1 = area_1; h2 = area_2; h3 = area_3; }
    }
    if (hit < 0) return;
    for (int k = 0; k < d.N_data; ++k) {
        double v;
        if (d.nodal) {
            v = h1 * data[(size_t)d.N_data * d.t0[hit] + k];
            v += h2 * data[(size_t)d.N_data * d.t1[hit] + k];
            v += h3 * data[(size_t)d.N_data * d.t2[hit] + k];
        } else {
            v = data[(size_t)d.N_data * hit + k];
        }
        if (isnan(v)) v = d.default_value;
        o[k] = v;
    }
}

}  // namespace

namespace {

// [n][3] triangles (1-based when `one_based`) -> three 0-based arrays
__global__ void __launch_bounds__(256) k_split_triangles(int n, const int *__restrict__ tri3, int one_based, int *__restrict__ t0, int *__restrict__ t1, int *__restrict__ t2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    t0[e] = tri3[3 * e] - one_based; t1[e] = tri3[3 * e + 1] - one_based; t2[e] = tri3[3 * e + 2] - one_based;
}

__global__ void __launch_bounds__(256) k_add_int(int n, int *__restrict__ a, int c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += c;
}

// The data mesh in bamg's integer plane + the bucket grid used by locate(), resident on the device.
thread_local int g_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // of the last mesh-to-mesh call of this thread (nxs_interp_last_info)
thread_local std::string g_completion_note;

struct Locator {
    InterpDev d{};
    DevBuf<int> dtri3, dt0, dt1, dt2, dix, diy, doff, dtri;
    DevBuf<BEdge> dbe;
    DevBuf<HullDev> dhull;
    std::vector<int> ix, iy;      // host copies of the integer plane (the completion is host code)
    nxs_hull::Completion comp;
    bool with_completion = false;
    int nods = 0, nels_mesh = 0;
    double coef = 0., pminx = 0., pminy = 0.;

    // the data mesh in bamg's integer plane + the bucket grid of locate(), the grid built on the device (nxs_regrid_tables.inl)
    int build(const int32_t *index_data, const double *x_data, const double *y_data, int32_t nods_, int32_t nels, bool need_boundary_edges,
              const nxs_hull::Completion *ready = nullptr) {
        nods = nods_; nels_mesh = nels;
        {   // ---- SetIntCoor (Mesh.cpp:3441-3468)
            Tick tk(1);
            double bx0 = x_data[0], by0 = y_data[0], bx1 = x_data[0], by1 = y_data[0];
            for (int i = 0; i < nods; ++i) {
                bx0 = std::min(bx0, x_data[i]); by0 = std::min(by0, y_data[i]);
                bx1 = std::max(bx1, x_data[i]); by1 = std::max(by1, y_data[i]);
            }
            d.xmin = bx0; d.xmax = bx1; d.ymin = by0; d.ymax = by1;
            if (!nxs_hull::int_plane(x_data, y_data, nods, ix, iy, coef, pminx, pminy))
                return fail(NXS_ERR_INVALID, "coefIcoor should be positive, a problem in the geometry is likely");
        }
        // ---- bamg's convex completion (isdefault == 0 only): fill triangles numbered behind the mesh's, hull edges
        std::vector<HullDev> hull;
        std::vector<int> fill3;  // AoS, 0-based
        if (need_boundary_edges) {
            Tick tk(2);
            comp = ready ? *ready : nxs_hull::complete_any(index_data, ix.data(), iy.data(), nods, nels);
            with_completion = true;
            if (comp.ok) {
                fill3 = comp.fill;
                for (const auto &h : comp.hull) hull.push_back(HullDev{h.a, h.b, h.tri, h.k});
            }
        }
        const int nfill = (int)fill3.size() / 3, nall = nels + nfill;
        // ---- boundary edges (edges held by exactly one triangle): the stand-in for exterior points when no completion exists
        std::vector<BEdge> bedges;
        if (need_boundary_edges && !comp.ok) {
            Tick tk(2);
            std::vector<int> bnd;
            (void)nxs_hull::find_boundary_edges(index_data, nods, nels, bnd);
            for (int be : bnd) bedges.push_back(BEdge{be / 3, be % 3});
        }
        // ---- triangles to the device (1-based AoS as the caller has them), SoA 0-based there; the fill triangles behind them
        if (dtri3.alloc(3 * (size_t)nall) || dt0.alloc(nall) || dt1.alloc(nall) || dt2.alloc(nall))
            return fail(NXS_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        {
            Tick tk(3);
            if (copy_sync(dtri3.p, index_data, 3 * (size_t)nels * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(NXS_ERR_HIP, "upload failed");
        }
        hipLaunchKernelGGL(k_split_triangles, dim3((nels + 255) / 256), dim3(256), 0, S(), nels, dtri3.p, 1, dt0.p, dt1.p, dt2.p);
        if (nfill > 0) {
            Tick tk(3);
            if (copy_sync(dtri3.p + 3 * (size_t)nels, fill3.data(), fill3.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(NXS_ERR_HIP, "upload failed");
            hipLaunchKernelGGL(k_split_triangles, dim3((nfill + 255) / 256), dim3(256), 0, S(), nfill, dtri3.p + 3 * (size_t)nels, 0, dt0.p + nels, dt1.p + nels, dt2.p + nels);
        }
        if (dix.upload(ix.data(), nods) || diy.upload(iy.data(), nods) || dbe.upload(bedges.data(), bedges.size()) || dhull.upload(hull.data(), hull.size()))
            return fail(NXS_ERR_HIP, "device allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
        // ---- bucket grid over the integer plane, on the device: count -> scan -> fill -> sort every cell's list ascending
        int G = 1;
        while ((long long)G * G * 2 < nall && G < 4096) G <<= 1;
        int shift = 30;
        for (int g = G; g > 1; g >>= 1) --shift;  // cell = 2^shift units, G cells cover [0, 2^30)
        const size_t ncell = (size_t)G * G;
        DevBuf<int> cursor, wide, nwide;
        if (doff.alloc(ncell + 1) || cursor.alloc(ncell) || wide.alloc(nall) || nwide.alloc(1)) return fail(NXS_ERR_HIP, "device allocation failed");
        int total = 0, n_wide = 0;
        {
            Tick tk(1);
            if (hipMemsetAsync(doff.p, 0, (ncell + 1) * sizeof(int), S()) != hipSuccess || hipMemsetAsync(cursor.p, 0, ncell * sizeof(int), S()) != hipSuccess ||
                hipMemsetAsync(nwide.p, 0, sizeof(int), S()) != hipSuccess)
                return fail(NXS_ERR_HIP, "hipMemset failed");
            hipLaunchKernelGGL(regrid_tables::k_grid_count, dim3((nall + 255) / 256), dim3(256), 0, S(), nall, (const int *)dt0.p, (const int *)dt1.p, (const int *)dt2.p,
                               (const int *)dix.p, (const int *)diy.p, shift, G, doff.p, wide.p, nwide.p);
            if (copy_sync(&n_wide, nwide.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
            if (n_wide > 0)  // triangles that span many cells (fill triangles across a bay): a workgroup each
                hipLaunchKernelGGL(regrid_tables::k_grid_count_wide, dim3(n_wide), dim3(256), 0, S(), (const int *)wide.p, (const int *)dt0.p, (const int *)dt1.p, (const int *)dt2.p,
                                   (const int *)dix.p, (const int *)diy.p, shift, G, doff.p);
            if (regrid_tables::exclusive_scan(doff.p, (int)(ncell + 1), S()) != hipSuccess) return fail(NXS_ERR_HIP, "scan of the bucket grid failed");
            if (copy_sync(&total, doff.p + ncell, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
            if (dtri.alloc((size_t)std::max(total, 1))) return fail(NXS_ERR_HIP, "device allocation failed");
            hipLaunchKernelGGL(regrid_tables::k_grid_fill, dim3((nall + 255) / 256), dim3(256), 0, S(), nall, (const int *)dt0.p, (const int *)dt1.p, (const int *)dt2.p,
                               (const int *)dix.p, (const int *)diy.p, shift, G, (const int *)doff.p, cursor.p, dtri.p);
            if (n_wide > 0)
                hipLaunchKernelGGL(regrid_tables::k_grid_fill_wide, dim3(n_wide), dim3(256), 0, S(), (const int *)wide.p, (const int *)dt0.p, (const int *)dt1.p, (const int *)dt2.p,
                                   (const int *)dix.p, (const int *)diy.p, shift, G, (const int *)doff.p, cursor.p, dtri.p);
            hipLaunchKernelGGL(regrid_tables::k_rows_sort, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, S(), (int)ncell, (const int *)doff.p, dtri.p, 0);
            if (hipStreamSynchronize(S()) != hipSuccess) return fail(NXS_ERR_HIP, "bucket grid kernels failed: %s", hipGetErrorString(hipGetLastError()));
        }
        d.nods = nods; d.nels = nels; d.nels_all = nall;
        d.nhull = (int)hull.size(); d.hull = dhull.p;
        d.t0 = dt0.p; d.t1 = dt1.p; d.t2 = dt2.p; d.ix = dix.p; d.iy = diy.p;
        d.G = G; d.shift = shift; d.cell_off = doff.p; d.cell_tri = dtri.p;
        d.nbe = (int)bedges.size(); d.bedges = dbe.p;
        d.coef = coef; d.pminx = pminx; d.pminy = pminy;
        return 0;
    }
};

}  // namespace

namespace {
int mesh_to_grid(double *griddata, const int32_t *index_mesh, const double *x_mesh, const double *y_mesh,
                 int32_t nods, int32_t nels, const double *data_mesh, bool data_on_device, int32_t data_length, int32_t N_data,
                 double xmin, double ymax, double xposting, double yposting, int32_t nrows, int32_t ncols,
                 double default_value, int32_t device, double *kernel_ms) {
    if (!griddata || !index_mesh || !x_mesh || !y_mesh || !data_mesh) return fail(NXS_ERR_INVALID, "NULL argument");
    if (nels < 1 || nods < 3 || ncols < 1 || nrows < 1 || xposting == 0 || yposting == 0 || N_data < 1)  // :34-36
        return fail(NXS_ERR_INVALID, "nothing to be done according to the mesh given in input");
    if (data_length != nods && data_length != nels)
        return fail(NXS_ERR_INVALID, "length of vector data not supported yet. It should be of length (number of nodes) or (number of elements)!");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the interpolation has no CPU path");
    if (device < 0 || device >= ndev) return fail(NXS_ERR_INVALID, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_HIP, "hipSetDevice failed");
    for (int64_t i = 0; i < 3ll * nels; ++i)
        if (index_mesh[i] < 1 || index_mesh[i] > nods) return fail(NXS_ERR_INVALID, "index_mesh[%lld] out of range", (long long)i);

    // grid coordinates exactly as the reference builds them (flips included), :66-92
    std::vector<double> xg(nrows), yg(ncols);
    if (xposting < 0) { for (int i = 0; i < nrows; ++i) xg[nrows - 1 - i] = xmin - xposting * i; }
    else { for (int i = 0; i < nrows; ++i) xg[i] = xmin + xposting * i; }
    if (yposting < 0) { for (int i = 0; i < ncols; ++i) yg[i] = ymax + yposting * i; }
    else { for (int i = 0; i < ncols; ++i) yg[ncols - 1 - i] = ymax - yposting * i; }

    std::vector<int> t0(nels), t1(nels), t2(nels);
    double bx0 = x_mesh[0], bx1 = x_mesh[0], by0 = y_mesh[0], by1 = y_mesh[0];
    for (int i = 0; i < nods; ++i) { bx0 = std::min(bx0, x_mesh[i]); bx1 = std::max(bx1, x_mesh[i]); by0 = std::min(by0, y_mesh[i]); by1 = std::max(by1, y_mesh[i]); }
    for (int e = 0; e < nels; ++e) { t0[e] = index_mesh[3 * e] - 1; t1[e] = index_mesh[3 * e + 1] - 1; t2[e] = index_mesh[3 * e + 2] - 1; }
    int G = 1;
    while ((long long)G * G * 2 < nels && G < 4096) G <<= 1;
    const double bdx = (bx1 - bx0) / G * (1. + 1e-12) + 1e-300, bdy = (by1 - by0) / G * (1. + 1e-12) + 1e-300;
    auto cellr = [&](int e, int &a, int &b, int &c, int &dd2) {
        const double xs[3] = {x_mesh[t0[e]], x_mesh[t1[e]], x_mesh[t2[e]]}, ys[3] = {y_mesh[t0[e]], y_mesh[t1[e]], y_mesh[t2[e]]};
        a = (int)std::floor((std::min({xs[0], xs[1], xs[2]}) - bx0) / bdx) - 1; b = (int)std::floor((std::max({xs[0], xs[1], xs[2]}) - bx0) / bdx) + 1;
        c = (int)std::floor((std::min({ys[0], ys[1], ys[2]}) - by0) / bdy) - 1; dd2 = (int)std::floor((std::max({ys[0], ys[1], ys[2]}) - by0) / bdy) + 1;
        a = std::max(a, 0); c = std::max(c, 0); b = std::min(b, G - 1); dd2 = std::min(dd2, G - 1);
    };
    std::vector<int> cnt((size_t)G * G + 1, 0);
    for (int e = 0; e < nels; ++e) { int a, b, c, dd2; cellr(e, a, b, c, dd2); for (int cy = c; cy <= dd2; ++cy) for (int cx = a; cx <= b; ++cx) cnt[(size_t)cy * G + cx + 1]++; }
    for (size_t c = 0; c < (size_t)G * G; ++c) cnt[c + 1] += cnt[c];
    std::vector<int> cell_tri(cnt[(size_t)G * G]), fill(cnt.begin(), cnt.end() - 1);
    for (int e = 0; e < nels; ++e) { int a, b, c, dd2; cellr(e, a, b, c, dd2); for (int cy = c; cy <= dd2; ++cy) for (int cx = a; cx <= b; ++cx) cell_tri[fill[(size_t)cy * G + cx]++] = e; }

    DevBuf<int> dt0, dt1, dt2, doff, dtri;
    DevBuf<double> dx, dy, dxg, dyg, ddata, dout;
    const size_t npts = (size_t)nrows * ncols;
    if (dt0.upload(t0.data(), nels) || dt1.upload(t1.data(), nels) || dt2.upload(t2.data(), nels) || doff.upload(cnt.data(), cnt.size()) ||
        dtri.upload(cell_tri.data(), cell_tri.size()) || dx.upload(x_mesh, nods) || dy.upload(y_mesh, nods) || dxg.upload(xg.data(), nrows) ||
        dyg.upload(yg.data(), ncols) || (!data_on_device && ddata.upload(data_mesh, (size_t)data_length * N_data)) || dout.alloc(npts * N_data))
        return fail(NXS_ERR_HIP, "device allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
    GridDev d{};
    d.nods = nods; d.nels = nels; d.N_data = N_data; d.nrows = nrows; d.ncols = ncols; d.nodal = (data_length == nods);
    d.t0 = dt0.p; d.t1 = dt1.p; d.t2 = dt2.p; d.x = dx.p; d.y = dy.p; d.xg = dxg.p; d.yg = dyg.p;
    d.G = G; d.bx0 = bx0; d.by0 = by0; d.bdx = bdx; d.bdy = bdy; d.cell_off = doff.p; d.cell_tri = dtri.p;
    d.default_value = default_value;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, S());
    hipLaunchKernelGGL(k_mesh_to_grid, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, S(), d, data_on_device ? data_mesh : (const double *)ddata.p, dout.p);
    (void)hipEventRecord(e1, S());
    hipError_t err = hipStreamSynchronize(S());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "mesh-to-grid kernel failed: %s", hipGetErrorString(err));
    if (kernel_ms) *kernel_ms = ms;
    if (copy_sync(griddata, dout.p, npts * N_data * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    return NXS_OK;
}
}  // namespace

extern "C" int nxs_interp_mesh_to_grid(double *griddata, const int32_t *index_mesh, const double *x_mesh, const double *y_mesh,
                                       int32_t nods, int32_t nels, const double *data_mesh, int32_t data_length, int32_t N_data,
                                       double xmin, double ymax, double xposting, double yposting, int32_t nrows, int32_t ncols,
                                       double default_value, int32_t device, double *kernel_ms) try {
    return mesh_to_grid(griddata, index_mesh, x_mesh, y_mesh, nods, nels, data_mesh, false, data_length, N_data, xmin, ymax, xposting, yposting, nrows, ncols,
                        default_value, device, kernel_ms);
} catch (...) { return entry_caught("nxs_interp_mesh_to_grid"); }

// The same sampling with the mesh data ALREADY ON THE DEVICE (e.g. the rows of nxs_dyn_ice_diagnostics, or any [data_length][N_data] array of
// the caller's on `device`): a Moorings record (gridoutput.cpp:496) then needs no round trip of the element state through the host.  The caller
// makes sure the producer of data_mesh_device has finished (nxs_dyn_ice_diagnostics returns synchronised).
extern "C" int nxs_interp_mesh_to_grid_device(double *griddata, const int32_t *index_mesh, const double *x_mesh, const double *y_mesh,
                                              int32_t nods, int32_t nels, const double *data_mesh_device, int32_t data_length, int32_t N_data,
                                              double xmin, double ymax, double xposting, double yposting, int32_t nrows, int32_t ncols,
                                              double default_value, int32_t device, double *kernel_ms) try {
    return mesh_to_grid(griddata, index_mesh, x_mesh, y_mesh, nods, nels, data_mesh_device, true, data_length, N_data, xmin, ymax, xposting, yposting, nrows, ncols,
                        default_value, device, kernel_ms);
} catch (...) { return entry_caught("nxs_interp_mesh_to_grid_device"); }

extern "C" const char *nxs_interp_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------------------------------------
// A regrid's context: everything about the OLD mesh that both interpolation calls of FiniteElement::interpFields (FE.cpp:3071-3154: the
// conservative remapping of the element variables, then InterpFromMeshToMesh2dx of the nodal ones) and any later call on the same mesh need --
// the integer plane, the bucket grid, bamg's convex completion, the two connectivity tables -- built once, on the device where that is
// possible (nxs_regrid_tables.inl), and kept.  The one-shot entry points below make a context, use it and drop it.
// ---------------------------------------------------------------------------------------------------------
struct nxs_regrid {
    std::future<nxs_hull::Completion> completion;   // bamg's convex completion, started on a host thread when the context is made (it is pure
                                                    // host work on the context's own copies): ready by the time the nodal interpolation asks for it
    int device = 0;
    int32_t nods = 0, nels = 0;
    std::vector<int32_t> index;   // host copies (the completion is host code; the caller's arrays may go away)
    std::vector<double> x, y;
    Locator plain, full;          // without / with bamg's convex completion (isdefault != 0 / == 0)
    bool have_plain = false, have_full = false;
    DevBuf<int> dtrio, ddeg, dnec, dec;   // old triangles (AoS, 0-based), NodalElementConnectivity, ElementConnectivity as ints
    DevBuf<double> dxo, dyo;
    int nec_width = 0;
    bool have_conn = false;
};

namespace {

int regrid_locator(nxs_regrid *r, bool with_completion, Locator **out) {
    Locator &L = with_completion ? r->full : r->plain;
    bool &have = with_completion ? r->have_full : r->have_plain;
    if (!have) {
        nxs_hull::Completion ready;
        const bool use_ready = with_completion && r->completion.valid();
        if (use_ready) { Tick tk(2); ready = r->completion.get(); }   // (what is left to wait for, if anything)
        if (int rc = L.build(r->index.data(), r->x.data(), r->y.data(), r->nods, r->nels, with_completion, use_ready ? &ready : nullptr)) return rc;
        have = true;
    }
    *out = &L;
    return 0;
}

// the two tables checkTriangle walks, as ints (-1 = bamg's NaN): built on the device, or converted from the caller's bamg tables
int regrid_connectivity(nxs_regrid *r, const double *nec_old, int32_t nec_width, const double *ec_old) {
    if (r->have_conn && !nec_old && !ec_old) return 0;
    Tick tk(0);
    const int nods = r->nods, nels = r->nels;
    if (r->dtrio.p == nullptr) {
        if (r->dtrio.alloc(3 * (size_t)nels) || r->dxo.upload(r->x.data(), nods) || r->dyo.upload(r->y.data(), nods)) return fail(NXS_ERR_HIP, "device allocation / upload failed");
        if (copy_sync(r->dtrio.p, r->index.data(), 3 * (size_t)nels * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(NXS_ERR_HIP, "upload failed");
        hipLaunchKernelGGL(k_add_int, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), 3 * nels, r->dtrio.p, -1);
    }
    auto to_index = [&](double v, int hi) -> int {  // "(int)(v - 1)" of the reference; NaN and junk become "no more entries"
        if (!(v >= 1.) || !(v <= (double)hi)) return -1;
        return (int)v - 1;
    };
    if (nec_old) {  // bamgmesh_previous->NodalElementConnectivity as bamg left it
        if (nec_width < 1 || nec_width > 255) return fail(NXS_ERR_INVALID, "NodalElementConnectivity width %d (1 .. 255 expected)", nec_width);
        std::vector<int> neci((size_t)nods * nec_width);
        for (size_t i = 0; i < neci.size(); ++i) neci[i] = to_index(nec_old[i], nels);
        DevBuf<int> nb;
        if (nb.upload(neci.data(), neci.size())) return fail(NXS_ERR_HIP, "upload failed");
        std::swap(r->dnec.p, nb.p);
        r->nec_width = nec_width;
    } else if (!r->have_conn) {
        DevBuf<int> cursor, maxdeg;
        if (r->ddeg.alloc(nods) || cursor.alloc(nods) || maxdeg.alloc(1)) return fail(NXS_ERR_HIP, "device allocation failed");
        if (hipMemsetAsync(r->ddeg.p, 0, (size_t)nods * sizeof(int), S()) != hipSuccess || hipMemsetAsync(cursor.p, 0, (size_t)nods * sizeof(int), S()) != hipSuccess ||
            hipMemsetAsync(maxdeg.p, 0, sizeof(int), S()) != hipSuccess) return fail(NXS_ERR_HIP, "hipMemset failed");
        hipLaunchKernelGGL(regrid_tables::k_fan_count, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), nels, (const int *)r->dtrio.p, r->ddeg.p, maxdeg.p);
        int w1 = 0;
        if (copy_sync(&w1, maxdeg.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
        if (w1 < 1 || w1 > 255) return fail(NXS_ERR_INVALID, "NodalElementConnectivity wider than 255 (%d)", w1);
        if (r->dnec.alloc((size_t)nods * w1)) return fail(NXS_ERR_HIP, "device allocation failed");
        hipLaunchKernelGGL(regrid_tables::k_fan_fill, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), nels, (const int *)r->dtrio.p, cursor.p, r->dnec.p, w1);
        hipLaunchKernelGGL(regrid_tables::k_fan_sort_desc, dim3((nods + 255) / 256), dim3(256), 0, S(), nods, (const int *)r->ddeg.p, r->dnec.p, w1);
        if (hipStreamSynchronize(S()) != hipSuccess) return fail(NXS_ERR_HIP, "connectivity kernels failed: %s", hipGetErrorString(hipGetLastError()));
        r->nec_width = w1;
    }
    if (ec_old) {
        std::vector<int> eci(3 * (size_t)nels);
        for (size_t i = 0; i < eci.size(); ++i) eci[i] = to_index(ec_old[i], nels);
        DevBuf<int> eb;
        if (eb.upload(eci.data(), eci.size())) return fail(NXS_ERR_HIP, "upload failed");
        std::swap(r->dec.p, eb.p);
    } else if (!r->have_conn || nec_old) {
        // (from the ascending-independent fans: with the caller's NodalElementConnectivity the degrees are not known here, so the fans are rebuilt)
        DevBuf<int> deg2, cur2, nec2, maxdeg, bad;
        const int *deg = r->ddeg.p; const int *nec = r->dnec.p; int w = r->nec_width;
        if (nec_old) {
            if (deg2.alloc(nods) || cur2.alloc(nods) || maxdeg.alloc(1)) return fail(NXS_ERR_HIP, "device allocation failed");
            if (hipMemsetAsync(deg2.p, 0, (size_t)nods * sizeof(int), S()) != hipSuccess || hipMemsetAsync(cur2.p, 0, (size_t)nods * sizeof(int), S()) != hipSuccess ||
                hipMemsetAsync(maxdeg.p, 0, sizeof(int), S()) != hipSuccess) return fail(NXS_ERR_HIP, "hipMemset failed");
            hipLaunchKernelGGL(regrid_tables::k_fan_count, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), nels, (const int *)r->dtrio.p, deg2.p, maxdeg.p);
            int w1 = 0;
            if (copy_sync(&w1, maxdeg.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
            if (nec2.alloc((size_t)nods * std::max(w1, 1))) return fail(NXS_ERR_HIP, "device allocation failed");
            hipLaunchKernelGGL(regrid_tables::k_fan_fill, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), nels, (const int *)r->dtrio.p, cur2.p, nec2.p, w1);
            deg = deg2.p; nec = nec2.p; w = w1;
        }
        if (bad.alloc(1) || hipMemsetAsync(bad.p, 0, sizeof(int), S()) != hipSuccess) return fail(NXS_ERR_HIP, "device allocation failed");
        if (r->dec.p == nullptr && r->dec.alloc(3 * (size_t)nels)) return fail(NXS_ERR_HIP, "device allocation failed");
        hipLaunchKernelGGL(regrid_tables::k_elem_conn, dim3((3 * nels + 255) / 256), dim3(256), 0, S(), nels, (const int *)r->dtrio.p, deg, nec, w, r->dec.p, bad.p);
        int nbad = 0;
        if (copy_sync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
        if (nbad > 0) return fail(NXS_ERR_INVALID, "an edge of the old mesh is shared by more than two triangles");
    }
    r->have_conn = !nec_old && !ec_old;   // (tables converted from the caller's are not kept for a later call without them)
    return 0;
}

}  // namespace

namespace {
// want_completion == false: the context of a one-shot call that will never look outside the mesh (the conservative remapping, the nodal interpolation
// with a default value): no completion thread is started -- nxs_regrid_destroy would only wait for work nobody asked for
int regrid_create(const int32_t *index_old, const double *x_old, const double *y_old, int32_t nods_old, int32_t nels_old, int32_t device,
                  nxs_regrid **out, bool want_completion) {
    if (!out) return fail(NXS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!index_old || !x_old || !y_old) return fail(NXS_ERR_INVALID, "NULL argument");
    if (nods_old < 3 || nels_old < 1) return fail(NXS_ERR_INVALID, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the regrid interpolation has no CPU path");
    if (device < 0 || device >= ndev) return fail(NXS_ERR_INVALID, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_HIP, "hipSetDevice failed");
    for (int64_t i = 0; i < 3ll * nels_old; ++i)
        if (index_old[i] < 1 || index_old[i] > nods_old) return fail(NXS_ERR_INVALID, "index_old[%lld] out of range", (long long)i);
    std::unique_ptr<nxs_regrid> r(new nxs_regrid());
    r->device = device; r->nods = nods_old; r->nels = nels_old;
    r->index.assign(index_old, index_old + 3 * (size_t)nels_old);
    r->x.assign(x_old, x_old + nods_old); r->y.assign(y_old, y_old + nods_old);
    // The two connectivity tables are built on the device right away (2 ms at 1.5 M triangles): the conservative remapping walks them, and the
    // boundary edges bamg's convex completion starts from are the entries of ElementConnectivity without a neighbour -- a few thousand numbers to
    // copy back instead of a host pass over every triangle.  A mesh the device tables cannot describe (an edge in three triangles, inconsistent
    // orientation) is left to the host code, which says what is wrong with it when the completion is asked for.
    std::vector<int> bnd;
    bool have_bnd = false;
    if (regrid_connectivity(r.get(), nullptr, 0, nullptr) == 0) {
        DevBuf<int> cnt, list;
        const int cap = 3 * nels_old;
        if (!cnt.alloc(2) && !list.alloc((size_t)cap) && hipMemsetAsync(cnt.p, 0, 2 * sizeof(int), S()) == hipSuccess) {
            hipLaunchKernelGGL(regrid_tables::k_boundary_list, dim3((3 * nels_old + 255) / 256), dim3(256), 0, S(), nels_old, (const int *)r->dtrio.p, (const int *)r->dec.p,
                               cnt.p, list.p, cap, cnt.p + 1);
            int c2[2] = {0, 0};
            if (copy_sync(c2, cnt.p, sizeof c2, hipMemcpyDeviceToHost) == hipSuccess && c2[1] == 0 && c2[0] >= 0 && c2[0] <= cap) {
                bnd.resize((size_t)c2[0]);
                if (c2[0] == 0 || copy_sync(bnd.data(), list.p, bnd.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
                    std::sort(bnd.begin(), bnd.end());
                    have_bnd = true;
                }
            }
        }
    } else {
        (void)hipGetLastError();
        g_err.clear();   // (not this call's failure: the remapping reports it if it is asked for)
    }
    if (want_completion) try {   // bamg's convex completion on a host thread, from now on (pure host work on this context's own copies); without a thread it is made when asked for
        nxs_regrid *q = r.get();
        r->completion = std::async(std::launch::async, [q, have_bnd, bnd = std::move(bnd)]() {
            std::vector<int> ix, iy;
            double coef = 0., px = 0., py = 0.;
            if (!nxs_hull::int_plane(q->x.data(), q->y.data(), q->nods, ix, iy, coef, px, py)) { nxs_hull::Completion c; c.why = "coefIcoor should be positive"; return c; }
            return nxs_hull::complete_any(q->index.data(), ix.data(), iy.data(), q->nods, q->nels, 0, have_bnd ? &bnd : nullptr);
        });
    } catch (const std::system_error &) { }
    *out = r.release();
    return NXS_OK;
}
}  // namespace

extern "C" int nxs_regrid_create(const int32_t *index_old, const double *x_old, const double *y_old, int32_t nods_old, int32_t nels_old, int32_t device,
                                 nxs_regrid **out) try {
    return regrid_create(index_old, x_old, y_old, nods_old, nels_old, device, out, true);
} catch (...) { return entry_caught("nxs_regrid_create"); }

extern "C" int nxs_regrid_destroy(nxs_regrid *r) try {
    if (!r) return NXS_OK;
    (void)hipSetDevice(r->device);
    if (r->completion.valid()) { try { (void)r->completion.get(); } catch (...) { } }   // the thread reads this context: it ends before the context does
    delete r;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_regrid_destroy"); }

// InterpFromMeshToMesh2dx on the context's mesh.  flags: NXS_REGRID_IN_DEVICE = `data` is a device pointer, NXS_REGRID_OUT_DEVICE = `data_interp` is.
extern "C" int nxs_regrid_interp_nodes(nxs_regrid *r, double *data_interp, const double *data, int32_t M_data, int32_t N_data, const double *x_interp,
                                       const double *y_interp, int32_t N_interp, int32_t isdefault, double defaultvalue, int32_t flags,
                                       int32_t *num_exterior, double *kernel_ms) try {
    if (!r || !data_interp || !data || !x_interp || !y_interp) return fail(NXS_ERR_INVALID, "NULL argument");
    if (N_data <= 0 || N_interp < 0) return fail(NXS_ERR_INVALID, "bad sizes");
    const int nods = r->nods, nels = r->nels;
    if (M_data != nods && M_data != nels)  // InterpFromMeshToMesh2dx.cpp:39-42
        return fail(NXS_ERR_INVALID, "data provided should have either %d or %d lines (not %d)", nods, nels, M_data);
    if (hipSetDevice(r->device) != hipSuccess) return fail(NXS_ERR_HIP, "hipSetDevice failed");
    for (double &v : g_ms) v = 0.;
    Tick whole(6);
    Locator *locp = nullptr;
    if (int rc = regrid_locator(r, !isdefault, &locp)) return rc;
    Locator &loc = *locp;
    InterpDev d = loc.d;
    DevBuf<int> dnext;
    DevBuf<double> ddata, dxi, dyi, dout;
    const bool in_dev = flags & NXS_REGRID_IN_DEVICE, out_dev = flags & NXS_REGRID_OUT_DEVICE;
    if ((!in_dev && ddata.upload(data, (size_t)M_data * N_data)) || dxi.upload(x_interp, N_interp) || dyi.upload(y_interp, N_interp) ||
        (!out_dev && dout.alloc((size_t)N_interp * N_data)) || dnext.alloc(4))
        return fail(NXS_ERR_HIP, "device allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
    if (memset_sync(dnext.p, 0, 4 * sizeof(int)) != hipSuccess) return fail(NXS_ERR_HIP, "hipMemset failed");
    d.N_data = N_data; d.N_interp = N_interp; d.nodal = (M_data == nods);
    d.isdefault = isdefault != 0; d.defaultvalue = defaultvalue;
    double *outp = out_dev ? data_interp : dout.p;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, S());
    if (N_interp > 0)
        hipLaunchKernelGGL(k_interp, dim3((N_interp + 255) / 256), dim3(256), 0, S(), d, in_dev ? data : (const double *)ddata.p, (const double *)dxi.p,
                           (const double *)dyi.p, outp, dnext.p);
    (void)hipEventRecord(e1, S());
    hipError_t err = hipStreamSynchronize(S());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "interpolation kernel failed: %s", hipGetErrorString(err));
    g_ms[4] += ms;
    if (kernel_ms) *kernel_ms = ms;
    {
        Tick tk(5);
        if (!out_dev && N_interp > 0 && copy_sync(data_interp, dout.p, (size_t)N_interp * N_data * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(NXS_ERR_HIP, "copy back failed");
    }
    int next[4] = {0, 0, 0, 0};
    (void)copy_sync(next, dnext.p, sizeof next, hipMemcpyDeviceToHost);
    if (num_exterior) *num_exterior = next[0];
    g_info[0] = (int)loc.comp.fill.size() / 3; g_info[1] = (int)loc.comp.hull.size();
    g_info[2] = next[0]; g_info[3] = next[1]; g_info[4] = next[2]; g_info[5] = next[3];
    g_info[6] = (!isdefault && !loc.comp.ok) ? 1 : 0;
    g_completion_note = loc.comp.ok ? "" : loc.comp.why;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_regrid_interp_nodes"); }

extern "C" int nxs_interp_mesh_to_mesh_2d(double *data_interp, const int32_t *index_data, const double *x_data,
                                          const double *y_data, int32_t nods, int32_t nels, const double *data, int32_t M_data,
                                          int32_t N_data, const double *x_interp, const double *y_interp, int32_t N_interp,
                                          int32_t isdefault, double defaultvalue, int32_t device, int32_t *num_exterior,
                                          double *kernel_ms) try {
    if (!data_interp || !index_data || !x_data || !y_data || !data || !x_interp || !y_interp) return fail(NXS_ERR_INVALID, "NULL argument");
    if (nods <= 0 || nels <= 0 || N_data <= 0 || N_interp < 0) return fail(NXS_ERR_INVALID, "bad sizes");
    if (M_data != nods && M_data != nels)  // InterpFromMeshToMesh2dx.cpp:39-42
        return fail(NXS_ERR_INVALID, "data provided should have either %d or %d lines (not %d)", nods, nels, M_data);
    nxs_regrid *r = nullptr;
    if (int rc = regrid_create(index_data, x_data, y_data, nods, nels, device, &r, isdefault == 0)) return rc;
    const int rc = nxs_regrid_interp_nodes(r, data_interp, data, M_data, N_data, x_interp, y_interp, N_interp, isdefault, defaultvalue, 0, num_exterior, kernel_ms);
    (void)nxs_regrid_destroy(r);
    return rc;
} catch (...) { return entry_caught("nxs_interp_mesh_to_mesh_2d"); }

extern "C" int nxs_interp_last_timing(double *ms8) try {
    if (!ms8) return fail(NXS_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 8; ++i) ms8[i] = g_ms[i];
    return NXS_OK;
} catch (...) { return entry_caught("nxs_interp_last_timing"); }

extern "C" int nxs_interp_last_info(int32_t *num_fill_triangles, int32_t *num_hull_edges, int32_t *num_exterior, int32_t *num_in_fill,
                                    int32_t *num_on_hull, int32_t *num_stand_in, const char **completion_refused) try {
    if (num_fill_triangles) *num_fill_triangles = g_info[0];
    if (num_hull_edges) *num_hull_edges = g_info[1];
    if (num_exterior) *num_exterior = g_info[2];
    if (num_in_fill) *num_in_fill = g_info[3];
    if (num_on_hull) *num_on_hull = g_info[4];
    if (num_stand_in) *num_stand_in = g_info[5];
    if (completion_refused) *completion_refused = g_info[6] ? g_completion_note.c_str() : nullptr;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_interp_last_info"); }

// Host only: bamg's convex completion of a mesh (see nxs_hull.inl) -- what tests compare with the real bamg.
// mode 0: the pocket construction, and the general one (constrained Delaunay of the boundary vertices) where that does not apply; 1: the general one; 2: pockets only
extern "C" int nxs_mesh_convex_completion_mode(const int32_t *index, const double *x, const double *y, int32_t nods, int32_t nels,
                                          int32_t *num_fill, int32_t *fill_tri, int32_t cap_fill, int32_t *num_hull, int32_t *hull_edges,
                                          int32_t cap_hull, int32_t mode) try {
    if (!index || !x || !y || nods < 3 || nels < 1 || !num_fill || !num_hull) return fail(NXS_ERR_INVALID, "bad arguments");
    for (int64_t i = 0; i < 3ll * nels; ++i)
        if (index[i] < 1 || index[i] > nods) return fail(NXS_ERR_INVALID, "index[%lld] out of range", (long long)i);
    // SetIntCoor, Mesh.cpp:3441-3468 (as Locator::build)
    double coef = 0., pminx = 0., pminy = 0.;
    std::vector<int> ix, iy;
    if (!nxs_hull::int_plane(x, y, nods, ix, iy, coef, pminx, pminy)) return fail(NXS_ERR_INVALID, "coefIcoor should be positive");
    const nxs_hull::Completion c = nxs_hull::complete_any(index, ix.data(), iy.data(), nods, nels, mode);
    if (!c.ok) return fail(NXS_ERR_INVALID, "no convex completion: %s", c.why.c_str());
    *num_fill = (int)c.fill.size() / 3; *num_hull = (int)c.hull.size();
    if (fill_tri) { if (cap_fill < *num_fill) return fail(NXS_ERR_INVALID, "fill_tri too small"); for (size_t i = 0; i < c.fill.size(); ++i) fill_tri[i] = c.fill[i] + 1; }
    if (hull_edges) {
        if (cap_hull < *num_hull) return fail(NXS_ERR_INVALID, "hull_edges too small");
        for (size_t i = 0; i < c.hull.size(); ++i) { hull_edges[2 * i] = c.hull[i].a + 1; hull_edges[2 * i + 1] = c.hull[i].b + 1; }
    }
    return NXS_OK;
} catch (...) { return entry_caught("nxs_mesh_convex_completion_mode"); }

extern "C" int nxs_mesh_convex_completion(const int32_t *index, const double *x, const double *y, int32_t nods, int32_t nels,
                                          int32_t *num_fill, int32_t *fill_tri, int32_t cap_fill, int32_t *num_hull, int32_t *hull_edges,
                                          int32_t cap_hull) try {
    return nxs_mesh_convex_completion_mode(index, x, y, nods, nels, num_fill, fill_tri, cap_fill, num_hull, hull_edges, cap_hull, 0);
} catch (...) { return entry_caught("nxs_mesh_convex_completion"); }

// ---------------------------------------------------------------------------------------------------------
// Conservative remapping of the element variables at regrid (ConservativeRemappingMeshToMesh, FE.cpp:3108)
// ---------------------------------------------------------------------------------------------------------
#undef NXS_HD
#define NXS_HD __device__
#include "nxs_remap_core.inl"

namespace {

struct RemapDev {
    InterpDev loc;           // old mesh in bamg's integer plane (seed search = InterpFromMeshToMesh2dx with isdefault)
    nxs_remap::OldMesh m;
    int nels_new, nb_var, n_geom;
    const int *tri_new;      // [3*nels_new] 0-based
    const double *xn, *yn;
    const double *prev;      // bamgmesh_new->PreviousNumbering or NULL
};

// one thread per new triangle: seed, identity test, replay of checkTriangle's recursion, weighted sum
__global__ void __launch_bounds__(128) k_remap(RemapDev r, const double *__restrict__ in, double *__restrict__ out, int *failed, int *visits,
                                               int *failed_list, int failed_cap) {
    const int t = blockIdx.x * 128 + threadIdx.x;
    if (t >= r.nels_new) return;
    double cx[3], cy[3];
    int nt[3];
    double gx = 0., gy = 0.;  // barycentre, ConservativeRemapping.cpp:217-231
    for (int i = 0; i < 3; ++i) {
        nt[i] = r.tri_new[3 * t + i];
        cx[i] = r.xn[nt[i]];
        cy[i] = r.yn[nt[i]];
        gx += cx[i];
        gy += cy[i];
    }
    gx /= 3.;
    gy /= 3.;
    double *o = out + (size_t)t * r.nb_var;
    int seed = -1;
    if (!(gx < r.loc.xmin || gx > r.loc.xmax || gy < r.loc.ymin || gy > r.loc.ymax)) {
        long long dd[3], Bx, By;
        seed = locate(r.loc, gx, gy, dd, Bx, By);
    }
    int n = -1;
    int tris[nxs_remap::kMaxVisit];
    double w[nxs_remap::kMaxVisit];
    nxs_remap::Frame stack[nxs_remap::kMaxVisit + 1];
    if (seed >= 0) {
        const bool same = nxs_remap::same_triangle(r.m, seed, nt, r.prev, r.n_geom);
        n = nxs_remap::collect(r.m, cx, cy, seed, same, tris, w, stack);
    }
    if (visits) visits[t] = n;
    if (n < 0) {  // barycentre outside the old mesh (the reference asserts) or capacity exceeded: flagged, never silent
        const int slot = atomicAdd(failed, 1);
        if (failed_list && slot < failed_cap) failed_list[slot] = seed >= 0 ? t : ~t;  // ~t: no seed, nothing a second pass can do
        for (int v = 0; v < r.nb_var; ++v) o[v] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    nxs_remap::apply(in, r.nb_var, cx, cy, tris, w, n, o);
}

// second pass for the new triangles that overlap more than kMaxVisit old ones (a much coarser new mesh): the same
// walk with lists of kMaxVisitBig entries in global memory, one thread per such triangle
__global__ void __launch_bounds__(64) k_remap_big(RemapDev r, const double *__restrict__ in, double *__restrict__ out, const int *__restrict__ list, int nlist,
                                                  int *tris_all, double *w_all, nxs_remap::Frame *stack_all, int *still_failed, int *visits) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nlist) return;
    const int t = list[i];
    double cx[3], cy[3], gx = 0., gy = 0.;
    int nt[3];
    for (int k = 0; k < 3; ++k) { nt[k] = r.tri_new[3 * t + k]; cx[k] = r.xn[nt[k]]; cy[k] = r.yn[nt[k]]; gx += cx[k]; gy += cy[k]; }
    gx /= 3.; gy /= 3.;
    long long dd[3], Bx, By;
    const int seed = locate(r.loc, gx, gy, dd, Bx, By);
    int *tris = tris_all + (size_t)i * nxs_remap::kMaxVisitBig;
    double *w = w_all + (size_t)i * nxs_remap::kMaxVisitBig;
    nxs_remap::Frame *stack = stack_all + (size_t)i * (nxs_remap::kMaxVisitBig + 1);
    const int n = seed >= 0 ? nxs_remap::collect(r.m, cx, cy, seed, false, tris, w, stack, nxs_remap::kMaxVisitBig) : -1;
    if (visits) visits[t] = n;
    if (n < 0) { atomicAdd(still_failed, 1); return; }
    nxs_remap::apply(in, r.nb_var, cx, cy, tris, w, n, out + (size_t)t * r.nb_var);
}

}  // namespace

// ConservativeRemappingMeshToMesh on the context's (old) mesh.  flags: NXS_REGRID_IN_DEVICE = interp_in is a device pointer ([nels_old][nb_var]),
// NXS_REGRID_OUT_DEVICE = interp_out is ([nels_new][nb_var]) -- the element state of nxs_dyn lives on the device already.
// nec_old / ec_old (may be NULL): bamg's own tables instead of the ones built here.
extern "C" int nxs_regrid_remap_elements(nxs_regrid *rg, double *interp_out, const double *interp_in, int32_t nb_var, const double *nec_old, int32_t nec_width,
                                         const double *ec_old, const int32_t *index_new, const double *x_new, const double *y_new, int32_t nods_new,
                                         int32_t nels_new, const double *previous_numbering, int32_t n_geom_vertices, int32_t flags,
                                         int32_t *num_failed, int32_t *visits, double *kernel_ms) try {
    if (!rg || !interp_out || !interp_in || !index_new || !x_new || !y_new) return fail(NXS_ERR_INVALID, "NULL argument");
    if (nb_var < 1 || nods_new < 3 || nels_new < 1) return fail(NXS_ERR_INVALID, "bad sizes");
    if (nec_old && nec_width < 1) return fail(NXS_ERR_INVALID, "nec_width must be given with the NodalElementConnectivity table");
    if (hipSetDevice(rg->device) != hipSuccess) return fail(NXS_ERR_HIP, "hipSetDevice failed");
    const int nods_old = rg->nods, nels_old = rg->nels;
    for (int64_t i = 0; i < 3ll * nels_new; ++i)
        if (index_new[i] < 1 || index_new[i] > nods_new) return fail(NXS_ERR_INVALID, "index_new[%lld] out of range", (long long)i);
    if (previous_numbering)
        for (int i = 0; i < nods_new; ++i)
            if (!(previous_numbering[i] >= 0.) || previous_numbering[i] > (double)nods_old) return fail(NXS_ERR_INVALID, "previous_numbering[%d] out of range", i);
    for (double &v : g_ms) v = 0.;
    Tick whole(6);
    if (int rc = regrid_connectivity(rg, nec_old, nec_width, ec_old)) return rc;
    Locator *locp = nullptr;
    if (int rc = regrid_locator(rg, false, &locp)) return rc;
    Locator &loc = *locp;
    const bool in_dev = flags & NXS_REGRID_IN_DEVICE, out_dev = flags & NXS_REGRID_OUT_DEVICE;
    DevBuf<int> dtrin, dfail, dvis;
    DevBuf<double> dxn, dyn, dprev, din, dout;
    if (dtrin.upload(index_new, 3 * (size_t)nels_new) || dxn.upload(x_new, nods_new) ||
        dyn.upload(y_new, nods_new) || (previous_numbering && dprev.upload(previous_numbering, nods_new)) ||
        (!in_dev && din.upload(interp_in, (size_t)nels_old * nb_var)) || (!out_dev && dout.alloc((size_t)nels_new * nb_var)) || dfail.alloc(1) || (visits && dvis.alloc(nels_new)))
        return fail(NXS_ERR_HIP, "device allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(k_add_int, dim3((3 * nels_new + 255) / 256), dim3(256), 0, S(), 3 * nels_new, dtrin.p, -1);
    if (memset_sync(dfail.p, 0, sizeof(int)) != hipSuccess) return fail(NXS_ERR_HIP, "hipMemset failed");
    const double *inp = in_dev ? interp_in : (const double *)din.p;
    double *outp = out_dev ? interp_out : dout.p;

    RemapDev r{};
    r.loc = loc.d;
    r.loc.isdefault = 1;
    r.m = nxs_remap::OldMesh{nels_old, nods_old, rg->dtrio.p, rg->dxo.p, rg->dyo.p, rg->dnec.p, rg->nec_width, rg->dec.p};
    r.nels_new = nels_new; r.nb_var = nb_var; r.n_geom = n_geom_vertices;
    r.tri_new = dtrin.p; r.xn = dxn.p; r.yn = dyn.p; r.prev = previous_numbering ? dprev.p : nullptr;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, S());
    const int failed_cap = 1 << 16;
    DevBuf<int> dflist, dstill, dbt;
    DevBuf<double> dbw;
    DevBuf<nxs_remap::Frame> dbs;
    if (dflist.alloc(failed_cap) || dstill.alloc(1) || memset_sync(dstill.p, 0, sizeof(int)) != hipSuccess) return fail(NXS_ERR_HIP, "device allocation failed");
    hipLaunchKernelGGL(k_remap, dim3((nels_new + 127) / 128), dim3(128), 0, S(), r, inp, outp, dfail.p, visits ? dvis.p : nullptr,
                       dflist.p, failed_cap);
    int nf1 = 0;
    if (copy_sync(&nf1, dfail.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    int unrecoverable = 0, nbig = 0;
    if (nf1 > 0) {  // the few that exceeded the fast path's capacity: second pass with large lists in global memory
        std::vector<int> fl(std::min(nf1, failed_cap));
        if (copy_sync(fl.data(), dflist.p, fl.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
        std::vector<int> big;
        for (int v : fl) { if (v >= 0) big.push_back(v); else ++unrecoverable; }
        unrecoverable += nf1 - (int)fl.size();  // more failures than the list holds: left as failures
        std::sort(big.begin(), big.end());
        nbig = (int)big.size();
        const size_t per = (size_t)nxs_remap::kMaxVisitBig;
        if (nbig > 0) {
            if (dflist.p && copy_sync(dflist.p, big.data(), big.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail(NXS_ERR_HIP, "upload failed");
            if (dbt.alloc(nbig * per) || dbw.alloc(nbig * per) || dbs.alloc(nbig * (per + 1)))
                return fail(NXS_ERR_HIP, "second remapping pass: %d triangles need %zu MB of lists", nbig, nbig * per * 28 >> 20);
            hipLaunchKernelGGL(k_remap_big, dim3((nbig + 63) / 64), dim3(64), 0, S(), r, inp, outp, (const int *)dflist.p, nbig, dbt.p, dbw.p,
                               dbs.p, dstill.p, visits ? dvis.p : nullptr);
        }
    }
    (void)hipEventRecord(e1, S());
    hipError_t err = hipStreamSynchronize(S());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "remapping kernel failed: %s", hipGetErrorString(err));
    g_ms[4] += ms;
    if (kernel_ms) *kernel_ms = ms;
    {
        Tick tk(5);
        if (!out_dev && copy_sync(interp_out, dout.p, (size_t)nels_new * nb_var * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
        if (visits && copy_sync(visits, dvis.p, (size_t)nels_new * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    }
    int still = 0;
    (void)copy_sync(&still, dstill.p, sizeof(int), hipMemcpyDeviceToHost);
    if (num_failed) *num_failed = unrecoverable + still;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_regrid_remap_elements"); }

extern "C" int nxs_interp_conservative_remap(double *interp_out, const double *interp_in, int32_t nb_var, const int32_t *index_old,
                                             const double *x_old, const double *y_old, int32_t nods_old, int32_t nels_old,
                                             const double *nec_old, int32_t nec_width, const double *ec_old, const int32_t *index_new,
                                             const double *x_new, const double *y_new, int32_t nods_new, int32_t nels_new,
                                             const double *previous_numbering, int32_t n_geom_vertices, int32_t device,
                                             int32_t *num_failed, int32_t *visits, double *kernel_ms) try {
    if (!interp_out || !interp_in || !index_old || !x_old || !y_old || !index_new || !x_new || !y_new) return fail(NXS_ERR_INVALID, "NULL argument");
    if (nb_var < 1 || nods_old < 3 || nels_old < 1 || nods_new < 3 || nels_new < 1) return fail(NXS_ERR_INVALID, "bad sizes");
    nxs_regrid *r = nullptr;
    if (int rc = regrid_create(index_old, x_old, y_old, nods_old, nels_old, device, &r, false)) return rc;
    const int rc = nxs_regrid_remap_elements(r, interp_out, interp_in, nb_var, nec_old, nec_width, ec_old, index_new, x_new, y_new, nods_new, nels_new,
                                             previous_numbering, n_geom_vertices, 0, num_failed, visits, kernel_ms);
    (void)nxs_regrid_destroy(r);
    return rc;
} catch (...) { return entry_caught("nxs_interp_conservative_remap"); }

// test door: the device-built tables of a context (the bucket grid's offsets and lists; NodalElementConnectivity and ElementConnectivity as ints)
extern "C" int nxs_regrid_debug_tables(nxs_regrid *rg, int32_t which, int32_t *out, int64_t cap, int64_t *count) try {
    if (!rg || !count) return fail(NXS_ERR_INVALID, "NULL argument");
    if (hipSetDevice(rg->device) != hipSuccess) return fail(NXS_ERR_HIP, "hipSetDevice failed");
    Locator *loc = nullptr;
    if (which <= 1) { if (int rc = regrid_locator(rg, false, &loc)) return rc; }
    else if (int rc = regrid_connectivity(rg, nullptr, 0, nullptr)) return rc;
    const int *src = nullptr; int64_t n = 0;
    if (which == 0) { src = loc->d.cell_off; n = (int64_t)loc->d.G * loc->d.G + 1; }
    else if (which == 1) { src = loc->d.cell_tri; int tot = 0; (void)copy_sync(&tot, loc->d.cell_off + (size_t)loc->d.G * loc->d.G, sizeof(int), hipMemcpyDeviceToHost); n = tot; }
    else if (which == 2) { src = rg->dnec.p; n = (int64_t)rg->nods * rg->nec_width; }
    else if (which == 3) { src = rg->dec.p; n = 3ll * rg->nels; }
    else return fail(NXS_ERR_INVALID, "which must be 0 .. 3");
    *count = n;
    if (out) {
        if (cap < n) return fail(NXS_ERR_INVALID, "buffer too small (%lld entries)", (long long)n);
        if (n > 0 && copy_sync(out, src, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    }
    return NXS_OK;
} catch (...) { return entry_caught("nxs_regrid_debug_tables"); }

// ---------------------------------------------------------------------------------------------------------
// Structured grid -> mesh nodes: the forcing ingest (InterpFromGridToMeshx, called at externaldata.cpp:1436)
// ---------------------------------------------------------------------------------------------------------
namespace {

struct G2M {
    const double *x, *y;  // pixel centres, x_rows / y_rows entries
    int x_rows, y_rows, M, N, N_data, nods, interp, row_major, mono_x, mono_y;
    double default_value;
};

// findindices (InterpFromGridToMeshx.cpp:361-395): the FIRST interval that brackets v, either orientation; the last
// coordinate itself belongs to the last interval.  On a strictly monotone axis that interval is unique: bisection.
__device__ __forceinline__ bool find_interval(const double *a, int rows, int mono, double v, int &idx) {
    bool found = false;
    idx = -1;
    if (mono != 0) {
        const bool asc = mono > 0;
        int lo = 0, hi = rows - 1;  // invariant: the bracketing interval, if any, lies in [lo, hi]
        if (asc ? (v >= a[0] && v < a[rows - 1]) : (v <= a[0] && v > a[rows - 1])) {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (asc ? (a[mid] <= v) : (a[mid] >= v)) lo = mid; else hi = mid;
            }
            idx = lo;
            found = true;
        }
    } else {
        for (int i = 0; i < rows - 1; ++i)
            if (((a[i] <= v) && (v < a[i + 1])) || ((a[i] >= v) && (v > a[i + 1]))) { idx = i; found = true; break; }
    }
    if (v == a[rows - 1]) { idx = rows - 2; found = true; }
    return found;
}

__global__ void __launch_bounds__(256) k_grid_to_mesh(G2M g, const double *__restrict__ data, const double *__restrict__ xm, const double *__restrict__ ym,
                                                      double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.nods) return;
    const double xg = xm[i], yg = ym[i];
    double *o = out + (size_t)i * g.N_data;
    int n, m;
    const bool fx = find_interval(g.x, g.x_rows, g.mono_x, xg, n), fy = find_interval(g.y, g.y_rows, g.mono_y, yg, m);
    if (!(fx && fy)) {
        for (int j = 0; j < g.N_data; ++j) o[j] = g.default_value;
        return;
    }
    int n_min, n_max, m_min, m_max;
    if (g.x[n] < g.x[n + 1]) { n_min = n; n_max = n + 1; } else { n_min = n + 1; n_max = n; }
    if (g.y[m] < g.y[m + 1]) { m_min = m; m_max = m + 1; } else { m_min = m + 1; m_max = m; }
    const double x1 = g.x[n_min], x2 = g.x[n_max], y1 = g.y[m_min], y2 = g.y[m_max];
    const size_t ND = (size_t)g.N_data;
    for (int j = 0; j < g.N_data; ++j) {
        double Q11, Q12, Q21, Q22;
        if (g.row_major) {
            Q11 = data[ND * ((size_t)n_min * g.M + m_min) + j]; Q12 = data[ND * ((size_t)n_min * g.M + m_max) + j];
            Q21 = data[ND * ((size_t)n_max * g.M + m_min) + j]; Q22 = data[ND * ((size_t)n_max * g.M + m_max) + j];
        } else {
            Q11 = data[ND * ((size_t)m_min * g.N + n_min) + j]; Q12 = data[ND * ((size_t)m_max * g.N + n_min) + j];
            Q21 = data[ND * ((size_t)m_min * g.N + n_max) + j]; Q22 = data[ND * ((size_t)m_max * g.N + n_max) + j];
        }
        double v;
        if (g.interp == NXS_INTERP_TRIANGLE) {  // triangleinterp, :397-430
            const double area = (x2 - x1) * (y2 - y1);
            if ((xg - x1) / (x2 - x1) < (yg - y1) / (y2 - y1)) {
                const double area_1 = ((y2 - yg) * (x2 - x1)) / area, area_2 = ((xg - x1) * (y2 - y1)) / area, area_3 = 1 - area_1 - area_2;
                v = area_1 * Q11 + area_2 * Q22 + area_3 * Q12;
            } else {
                const double area_1 = ((yg - y1) * (x2 - x1)) / area, area_2 = ((x2 - xg) * (y2 - y1)) / area, area_3 = 1 - area_1 - area_2;
                v = area_1 * Q22 + area_2 * Q11 + area_3 * Q21;
            }
        } else if (g.interp == NXS_INTERP_BILINEAR) {  // bilinearinterp, :432-455
            v = +Q11 * (x2 - xg) * (y2 - yg) / ((x2 - x1) * (y2 - y1)) + Q21 * (xg - x1) * (y2 - yg) / ((x2 - x1) * (y2 - y1)) +
                Q12 * (x2 - xg) * (yg - y1) / ((x2 - x1) * (y2 - y1)) + Q22 * (xg - x1) * (yg - y1) / ((x2 - x1) * (y2 - y1));
        } else {  // nearestinterp, :457-485 -- xm, ym are HALF EXTENTS compared with absolute coordinates, as the reference does
            const double xmid = (x2 - x1) / 2, ymid = (y2 - y1) / 2;
            if (xg <= xmid && yg <= ymid) v = Q11;
            else if (xg <= xmid && yg > ymid) v = Q12;
            else if (xg > xmid && yg <= ymid) v = Q21;
            else v = Q22;
        }
        if (isnan(v)) v = g.default_value;
        o[j] = v;
    }
}

int monotone(const double *a, int n) {
    bool asc = true, desc = true;
    for (int i = 0; i + 1 < n; ++i) { asc = asc && a[i] < a[i + 1]; desc = desc && a[i] > a[i + 1]; }
    return asc ? 1 : desc ? -1 : 0;
}

}  // namespace

extern "C" int nxs_interp_grid_to_mesh(double *data_mesh, const double *x_in, int32_t x_rows, const double *y_in, int32_t y_rows, const double *data,
                                       int32_t M, int32_t N, int32_t N_data, const double *x_mesh, const double *y_mesh, int32_t nods,
                                       double default_value, int32_t interp, int32_t row_major, int32_t device, double *kernel_ms) try {
    if (!data_mesh || !x_in || !y_in || !data || !x_mesh || !y_mesh) return fail(NXS_ERR_INVALID, "NULL argument");
    if ((M < 2) || (N < 2) || (nods <= 0) || N_data < 1) return fail(NXS_ERR_INVALID, "nothing to be done according to the dimensions of input matrices and vectors.");
    if (interp != NXS_INTERP_TRIANGLE && interp != NXS_INTERP_BILINEAR && interp != NXS_INTERP_NEAREST) return fail(NXS_ERR_INVALID, "Interpolation %d not supported yet", interp);
    std::vector<double> x(N), y(M);
    if (N == (x_rows - 1) && M == (y_rows - 1)) {  // contours of the pixels given: take the centres (:44-53)
        for (int i = 0; i < N; ++i) x[i] = (x_in[i] + x_in[i + 1]) / 2.;
        for (int i = 0; i < M; ++i) y[i] = (y_in[i] + y_in[i + 1]) / 2.;
    } else if (N == x_rows && M == y_rows) {
        for (int i = 0; i < N; ++i) x[i] = x_in[i];
        for (int i = 0; i < M; ++i) y[i] = y_in[i];
    } else {
        return fail(NXS_ERR_INVALID, "x and y vectors length should be 1 or 0 more than data number of rows.");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the interpolation has no CPU path");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_INVALID, "bad device %d", device);
    DevBuf<double> dx, dy, dd, dxm, dym, dout;
    if (dx.upload(x.data(), N) || dy.upload(y.data(), M) || dd.upload(data, (size_t)M * N * N_data) || dxm.upload(x_mesh, nods) || dym.upload(y_mesh, nods) ||
        dout.alloc((size_t)nods * N_data))
        return fail(NXS_ERR_HIP, "device allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
    G2M g{dx.p, dy.p, N, M, M, N, N_data, nods, interp, row_major != 0, monotone(x.data(), N), monotone(y.data(), M), default_value};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, S());
    hipLaunchKernelGGL(k_grid_to_mesh, dim3((nods + 255) / 256), dim3(256), 0, S(), g, (const double *)dd.p, (const double *)dxm.p, (const double *)dym.p, dout.p);
    (void)hipEventRecord(e1, S());
    hipError_t err = hipStreamSynchronize(S());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "grid-to-mesh kernel failed: %s", hipGetErrorString(err));
    if (kernel_ms) *kernel_ms = ms;
    if (copy_sync(data_mesh, dout.p, (size_t)nods * N_data * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    return NXS_OK;
} catch (...) { return entry_caught("nxs_interp_grid_to_mesh"); }

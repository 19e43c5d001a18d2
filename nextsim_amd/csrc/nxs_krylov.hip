// nxs_krylov.hip -- EXTENSION (see include/nxs_krylov.h): coloured CSR assembly of the
// P1 stiffness matrix and a Jacobi-preconditioned CG built from SpMV / axpy / dot kernels.  No live
// counterpart in the reference (research/laplacian.cpp is its only, un-buildable, trace): PARITY UNPINNED.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_krylov.h"

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

constexpr int BS = 256;

// element matrices of one colour scattered with plain adds: elements of a colour share no node, hence no
// CSR entry -- the "graph-coloured scatter".  research/laplacian.cpp:163-224.
__global__ void __launch_bounds__(BS) k_assemble_colour(int n, const int *__restrict__ elems, const int *__restrict__ t0, const int *__restrict__ t1,
                                                        const int *__restrict__ t2, const double *__restrict__ x, const double *__restrict__ y,
                                                        const int *__restrict__ pos /*[9*Ne]*/, const double *__restrict__ f_elem,
                                                        double *__restrict__ val, double *__restrict__ rhs) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const int e = elems[i];
    const int nd[3] = {t0[e], t1[e], t2[e]};
    const double xs[3] = {x[nd[0]], x[nd[1]], x[nd[2]]}, ys[3] = {y[nd[0]], y[nd[1]], y[nd[2]]};
    double area = (xs[1] - xs[0]) * (ys[2] - ys[0]);
    area -= (xs[2] - xs[0]) * (ys[1] - ys[0]);
    area = (1. / 2) * fabs(area);
    const double fj = f_elem[e] * area / 3.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int jp1 = (j + 1) % 3, jp2 = (j + 2) % 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
            double m_jk = (ys[jp1] - ys[jp2]) * (ys[kp1] - ys[kp2]) + (xs[jp1] - xs[jp2]) * (xs[kp1] - xs[kp2]);
            m_jk = m_jk / (4.0 * area);
            val[pos[9 * e + 3 * j + k]] += m_jk;
        }
        rhs[nd[j]] += fj;
    }
}

// homogeneous Dirichlet: row and column zeroed, unit diagonal, rhs 0 (MatrixPetsc::on in the demo)
__global__ void __launch_bounds__(BS) k_apply_dirichlet(int Nn, const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                        const unsigned char *__restrict__ dir, double *__restrict__ val, double *__restrict__ rhs) {
    const int r = blockIdx.x * BS + threadIdx.x;
    if (r >= Nn) return;
    const bool dr = dir[r];
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
        const int c = colidx[q];
        if (dr || dir[c]) val[q] = (c == r && dr) ? 1. : (dr || dir[c]) ? 0. : val[q];
    }
    if (dr) rhs[r] = 0.;
}

__global__ void __launch_bounds__(BS) k_spmv(int Nn, const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
                                             const double *__restrict__ v, double *__restrict__ out) {
    const int r = blockIdx.x * BS + threadIdx.x;
    if (r >= Nn) return;
    double s = 0.;
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) s += val[q] * v[colidx[q]];
    out[r] = s;
}

__device__ __forceinline__ double block_sum(double v) {
    __shared__ double sh[BS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x == 0) for (int i = 0; i < BS / 64; ++i) t += sh[i];
    return t;  // valid on thread 0
}

// partial[b] = sum over the block of a[i]*b[i]  (stage 1 of the deterministic dot)
__global__ void __launch_bounds__(BS) k_dot_partial(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial) {
    double s = 0.;
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) s += a[i] * b[i];
    s = block_sum(s);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// stage 2: one block adds the partials in index order; op selects what the scalar is used for
__global__ void __launch_bounds__(BS) k_dot_final(int nb, const double *__restrict__ partial, double *__restrict__ scal, int slot) {
    double s = 0.;
    for (int i = threadIdx.x; i < nb; i += BS) s += partial[i];
    s = block_sum(s);
    if (threadIdx.x == 0) scal[slot] = s;
}

// scal: [0] rz, [1] pAp, [2] rz_new, [3] rr
__global__ void __launch_bounds__(BS) k_update_xr(int n, const double *__restrict__ scal, const double *__restrict__ p, const double *__restrict__ Ap,
                                                  const double *__restrict__ dinv, double *__restrict__ x, double *__restrict__ r, double *__restrict__ z) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const double alpha = scal[0] / scal[1];
    x[i] += alpha * p[i];                 // axpy
    const double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    z[i] = dinv[i] * ri;                  // Jacobi preconditioner
}
__global__ void __launch_bounds__(BS) k_update_p(int n, double *__restrict__ scal, const double *__restrict__ z, double *__restrict__ p) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const double beta = scal[2] / scal[0];
    p[i] = z[i] + beta * p[i];
}
__global__ void k_shift_rz(double *scal) { scal[0] = scal[2]; }
__global__ void __launch_bounds__(BS) k_diag_inv(int Nn, const int *__restrict__ rowptr, const int *__restrict__ colidx, const double *__restrict__ val,
                                                 double *__restrict__ dinv) {
    const int r = blockIdx.x * BS + threadIdx.x;
    if (r >= Nn) return;
    double d = 1.;
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) if (colidx[q] == r) d = val[q];
    dinv[r] = 1. / d;
}
__global__ void __launch_bounds__(BS) k_init_cg(int n, const double *__restrict__ b, const double *__restrict__ dinv, double *__restrict__ x,
                                                double *__restrict__ r, double *__restrict__ z, double *__restrict__ p) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    x[i] = 0.; r[i] = b[i]; z[i] = dinv[i] * b[i]; p[i] = z[i];
}


// ---- BiCGStab (right Jacobi preconditioner); scalars stay on the device: scal[0] rho, [1] alpha, [2] omega,
//      [3] rho_new, [4] (rhat,v), [5] (t,s), [6] (t,t), [7] (r,r)
__global__ void __launch_bounds__(BS) k_bicg_p(int n, const double *__restrict__ scal, const double *__restrict__ r, const double *__restrict__ v,
                                               const double *__restrict__ dinv, double *__restrict__ p, double *__restrict__ y) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const double beta = (scal[3] / scal[0]) * (scal[1] / scal[2]);
    const double pi = r[i] + beta * (p[i] - scal[2] * v[i]);
    p[i] = pi;
    y[i] = dinv[i] * pi;
}
__global__ void __launch_bounds__(BS) k_bicg_s(int n, const double *__restrict__ scal, const double *__restrict__ r, const double *__restrict__ v,
                                               const double *__restrict__ dinv, double *__restrict__ sv, double *__restrict__ z) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const double alpha = scal[3] / scal[4];
    const double si = r[i] - alpha * v[i];
    sv[i] = si;
    z[i] = dinv[i] * si;
}
__global__ void __launch_bounds__(BS) k_bicg_x(int n, const double *__restrict__ scal, const double *__restrict__ y, const double *__restrict__ z,
                                               const double *__restrict__ sv, const double *__restrict__ t, double *__restrict__ x, double *__restrict__ r) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const double alpha = scal[3] / scal[4];
    const double omega = scal[6] != 0. ? scal[5] / scal[6] : 0.;
    x[i] += alpha * y[i] + omega * z[i];
    r[i] = sv[i] - omega * t[i];
}
__global__ void k_bicg_shift(double *scal) {  // after an iteration: rho <- rho_new, alpha, omega kept for the next beta
    scal[1] = scal[3] / scal[4];
    scal[2] = scal[6] != 0. ? scal[5] / scal[6] : 0.;
    scal[0] = scal[3];
}
__global__ void __launch_bounds__(BS) k_copy2(int n, const double *__restrict__ b, double *__restrict__ r, double *__restrict__ rhat) {
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i < n) { r[i] = b[i]; rhat[i] = b[i]; }
}
__global__ void k_set_ones(double *scal) { scal[0] = scal[1] = scal[2] = 1.; }

template <typename T>
struct DBuf {
    T *p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess; }
    bool up(const T *s, size_t n) { return alloc(n) && (n == 0 || hipMemcpy(p, s, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess); }
    bool zero(size_t n) { return alloc(n) && hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess; }
};

int check_mesh(const int32_t *indices, int32_t Nn, int32_t Ne) {
    if (!indices || Nn <= 0 || Ne <= 0) return fail(NXS_ERR_INVALID, "bad mesh sizes");
    for (int64_t i = 0; i < 3ll * Ne; ++i) if (indices[i] < 1 || indices[i] > Nn) return fail(NXS_ERR_INVALID, "indices[%lld] out of range", (long long)i);
    return NXS_OK;
}

void build_pattern(const int32_t *indices, int32_t Nn, int32_t Ne, std::vector<int> &rowptr, std::vector<int> &colidx) {
    std::vector<std::vector<int>> adj(Nn);
    for (int e = 0; e < Ne; ++e)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) adj[indices[3 * e + j] - 1].push_back(indices[3 * e + k] - 1);
    rowptr.assign(Nn + 1, 0);
    for (int r = 0; r < Nn; ++r) {
        auto &a = adj[r];
        if (a.empty()) a.push_back(r);
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
        rowptr[r + 1] = rowptr[r] + (int)a.size();
    }
    colidx.resize(rowptr[Nn]);
    for (int r = 0; r < Nn; ++r) std::copy(adj[r].begin(), adj[r].end(), colidx.begin() + rowptr[r]);
}

void colour_elements(const int32_t *indices, int32_t Nn, int32_t Ne, std::vector<int> &colour, int &ncol) {
    // greedy: smallest colour not used by any element sharing a node (64 colours are plenty for a planar mesh)
    std::vector<unsigned long long> used(Nn, 0ull);
    colour.assign(Ne, 0);
    ncol = 0;
    for (int e = 0; e < Ne; ++e) {
        const unsigned long long m = used[indices[3 * e] - 1] | used[indices[3 * e + 1] - 1] | used[indices[3 * e + 2] - 1];
        int c = 0;
        while (c < 63 && ((m >> c) & 1ull)) ++c;
        colour[e] = c;
        for (int k = 0; k < 3; ++k) used[indices[3 * e + k] - 1] |= (1ull << c);
        ncol = std::max(ncol, c + 1);
    }
}

}  // namespace

extern "C" {

const char *nxs_krylov_last_error(void) { return g_err.c_str(); }

int nxs_fem_csr_pattern(const int32_t *indices, int32_t Nn, int32_t Ne, int32_t *rowptr, int32_t *colidx, int64_t *nnz) {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!rowptr || !nnz) return fail(NXS_ERR_INVALID, "NULL argument");
    std::vector<int> rp, ci;
    build_pattern(indices, Nn, Ne, rp, ci);
    std::copy(rp.begin(), rp.end(), rowptr);
    *nnz = (int64_t)ci.size();
    if (colidx) std::copy(ci.begin(), ci.end(), colidx);
    return NXS_OK;
}

int nxs_fem_colour_elements(const int32_t *indices, int32_t Nn, int32_t Ne, int32_t *colour, int32_t *ncolours) {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!colour || !ncolours) return fail(NXS_ERR_INVALID, "NULL argument");
    std::vector<int> c; int n = 0;
    colour_elements(indices, Nn, Ne, c, n);
    std::copy(c.begin(), c.end(), colour);
    *ncolours = n;
    return NXS_OK;
}

int nxs_fem_poisson_solve(const int32_t *indices, const double *x, const double *y, int32_t Nn, int32_t Ne, const uint8_t *dirichlet,
                          const double *f_elem, double *u, double rtol, int32_t max_iter, int32_t device, int32_t *iterations,
                          double *rel_residual, double *ms_assembly, double *ms_solve) {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!x || !y || !dirichlet || !f_elem || !u) return fail(NXS_ERR_INVALID, "NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the solver has no CPU path");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_INVALID, "bad device %d", device);

    std::vector<int> rp, ci, colour; int ncol = 0;
    build_pattern(indices, Nn, Ne, rp, ci);
    colour_elements(indices, Nn, Ne, colour, ncol);
    std::vector<int> t0(Ne), t1(Ne), t2(Ne), pos(9 * (size_t)Ne);
    for (int e = 0; e < Ne; ++e) {
        t0[e] = indices[3 * e] - 1; t1[e] = indices[3 * e + 1] - 1; t2[e] = indices[3 * e + 2] - 1;
        const int nd[3] = {t0[e], t1[e], t2[e]};
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const int *b = ci.data() + rp[nd[j]], *en = ci.data() + rp[nd[j] + 1];
                pos[9 * (size_t)e + 3 * j + k] = (int)(std::lower_bound(b, en, nd[k]) - ci.data());
            }
    }
    std::vector<std::vector<int>> by_col(ncol);
    for (int e = 0; e < Ne; ++e) by_col[colour[e]].push_back(e);
    std::vector<int> elems; std::vector<int> coff(ncol + 1, 0);
    for (int c = 0; c < ncol; ++c) { elems.insert(elems.end(), by_col[c].begin(), by_col[c].end()); coff[c + 1] = (int)elems.size(); }

    DBuf<int> dt0, dt1, dt2, dpos, drp, dci, delems;
    DBuf<unsigned char> ddir;
    DBuf<double> dx, dy, df, dval, drhs, dxv, dr, dz, dp, dAp, ddinv, dpart, dscal;
    const int nparts = 512;
    const size_t nnz = ci.size();
    if (!dt0.up(t0.data(), Ne) || !dt1.up(t1.data(), Ne) || !dt2.up(t2.data(), Ne) || !dpos.up(pos.data(), pos.size()) || !drp.up(rp.data(), rp.size()) ||
        !dci.up(ci.data(), nnz) || !delems.up(elems.data(), elems.size()) || !ddir.up(dirichlet, Nn) || !dx.up(x, Nn) || !dy.up(y, Nn) ||
        !df.up(f_elem, Ne) || !dval.zero(nnz) || !drhs.zero(Nn) || !dxv.zero(Nn) || !dr.zero(Nn) || !dz.zero(Nn) || !dp.zero(Nn) || !dAp.zero(Nn) ||
        !ddinv.zero(Nn) || !dpart.zero(nparts) || !dscal.zero(8))
        return fail(NXS_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));

    hipEvent_t e0, e1, e2;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    const dim3 gN((Nn + BS - 1) / BS), blk(BS);
    (void)hipEventRecord(e0, nullptr);
    for (int c = 0; c < ncol; ++c) {
        const int n = coff[c + 1] - coff[c];
        if (n > 0)
            hipLaunchKernelGGL(k_assemble_colour, dim3((n + BS - 1) / BS), blk, 0, nullptr, n, (const int *)(delems.p + coff[c]), (const int *)dt0.p, (const int *)dt1.p,
                               (const int *)dt2.p, (const double *)dx.p, (const double *)dy.p, (const int *)dpos.p, (const double *)df.p, dval.p, drhs.p);
    }
    hipLaunchKernelGGL(k_apply_dirichlet, gN, blk, 0, nullptr, Nn, (const int *)drp.p, (const int *)dci.p, (const unsigned char *)ddir.p, dval.p, drhs.p);
    hipLaunchKernelGGL(k_diag_inv, gN, blk, 0, nullptr, Nn, (const int *)drp.p, (const int *)dci.p, (const double *)dval.p, ddinv.p);
    (void)hipEventRecord(e1, nullptr);

    auto dot = [&](const double *a, const double *b, int slot) {
        hipLaunchKernelGGL(k_dot_partial, dim3(nparts), blk, 0, nullptr, Nn, a, b, dpart.p);
        hipLaunchKernelGGL(k_dot_final, dim3(1), blk, 0, nullptr, nparts, (const double *)dpart.p, dscal.p, slot);
    };
    hipLaunchKernelGGL(k_init_cg, gN, blk, 0, nullptr, Nn, (const double *)drhs.p, (const double *)ddinv.p, dxv.p, dr.p, dz.p, dp.p);
    dot(dr.p, dz.p, 0);
    dot(drhs.p, drhs.p, 4);  // ||b||^2
    double h_scal[8] = {0};
    (void)hipMemcpy(h_scal, dscal.p, sizeof h_scal, hipMemcpyDeviceToHost);
    const double bb = h_scal[4] > 0. ? h_scal[4] : 1.;
    int it = 0;
    double rr = bb;
    const int check_every = 20;
    while (it < max_iter) {
        hipLaunchKernelGGL(k_spmv, gN, blk, 0, nullptr, Nn, (const int *)drp.p, (const int *)dci.p, (const double *)dval.p, (const double *)dp.p, dAp.p);
        dot(dp.p, dAp.p, 1);
        hipLaunchKernelGGL(k_update_xr, gN, blk, 0, nullptr, Nn, (const double *)dscal.p, (const double *)dp.p, (const double *)dAp.p, (const double *)ddinv.p, dxv.p, dr.p, dz.p);
        dot(dr.p, dz.p, 2);
        hipLaunchKernelGGL(k_update_p, gN, blk, 0, nullptr, Nn, dscal.p, (const double *)dz.p, dp.p);
        hipLaunchKernelGGL(k_shift_rz, dim3(1), dim3(1), 0, nullptr, dscal.p);
        ++it;
        if (it % check_every == 0 || it == max_iter) {
            dot(dr.p, dr.p, 3);
            (void)hipMemcpy(h_scal, dscal.p, sizeof h_scal, hipMemcpyDeviceToHost);
            rr = h_scal[3];
            if (!(rr == rr)) break;  // NaN
            if (std::sqrt(rr / bb) <= rtol) break;
        }
    }
    (void)hipEventRecord(e2, nullptr);
    hipError_t err = hipDeviceSynchronize();
    float msa = 0.f, mss = 0.f;
    (void)hipEventElapsedTime(&msa, e0, e1); (void)hipEventElapsedTime(&mss, e1, e2);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "solver kernels failed: %s", hipGetErrorString(err));
    if (hipMemcpy(u, dxv.p, (size_t)Nn * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    if (iterations) *iterations = it;
    if (rel_residual) *rel_residual = std::sqrt(rr / bb);
    if (ms_assembly) *ms_assembly = msa;
    if (ms_solve) *ms_solve = mss;
    return NXS_OK;
}

int nxs_krylov_solve(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *val, const double *b, double *x, int32_t method,
                     double rtol, int32_t max_iter, int32_t device, int32_t *iterations, double *rel_residual, double *ms_solve) {
    if (n < 1 || !rowptr || !colidx || !val || !b || !x) return fail(NXS_ERR_INVALID, "NULL argument / empty system");
    if (method != NXS_KRYLOV_CG && method != NXS_KRYLOV_BICGSTAB) return fail(NXS_ERR_INVALID, "method must be NXS_KRYLOV_CG or NXS_KRYLOV_BICGSTAB");
    if (rowptr[0] != 0) return fail(NXS_ERR_INVALID, "rowptr[0] must be 0");
    for (int i = 0; i < n; ++i) {
        if (rowptr[i + 1] < rowptr[i]) return fail(NXS_ERR_INVALID, "rowptr not monotone at row %d", i);
        bool diag = false;
        for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            if (colidx[q] < 0 || colidx[q] >= n) return fail(NXS_ERR_INVALID, "column index out of range in row %d", i);
            diag = diag || (colidx[q] == i && val[q] != 0.);
        }
        if (!diag) return fail(NXS_ERR_INVALID, "row %d has no diagonal entry (Jacobi preconditioner)", i);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the solver has no CPU path");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_INVALID, "bad device %d", device);
    const size_t nnz = (size_t)rowptr[n];
    DBuf<int> drp, dci;
    DBuf<double> dval, db, dx, dr, drh, dp, dv, dy, ds, dz, dt, ddinv, dpart, dscal;
    const int nparts = 512;
    if (!drp.up(rowptr, (size_t)n + 1) || !dci.up(colidx, nnz) || !dval.up(val, nnz) || !db.up(b, n) || !dx.zero(n) || !dr.zero(n) || !drh.zero(n) ||
        !dp.zero(n) || !dv.zero(n) || !dy.zero(n) || !ds.zero(n) || !dz.zero(n) || !dt.zero(n) || !ddinv.zero(n) || !dpart.zero(nparts) || !dscal.zero(8))
        return fail(NXS_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
    const dim3 gN((n + BS - 1) / BS), blk(BS);
    auto dot = [&](const double *a, const double *c, int slot) {
        hipLaunchKernelGGL(k_dot_partial, dim3(nparts), blk, 0, nullptr, n, a, c, dpart.p);
        hipLaunchKernelGGL(k_dot_final, dim3(1), blk, 0, nullptr, nparts, (const double *)dpart.p, dscal.p, slot);
    };
    auto spmv = [&](const double *in, double *out) {
        hipLaunchKernelGGL(k_spmv, gN, blk, 0, nullptr, n, (const int *)drp.p, (const int *)dci.p, (const double *)dval.p, in, out);
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_diag_inv, gN, blk, 0, nullptr, n, (const int *)drp.p, (const int *)dci.p, (const double *)dval.p, ddinv.p);
    double h[8] = {0};
    int it = 0;
    double rr = 0., bb = 1.;
    const int check_every = 10;
    if (method == NXS_KRYLOV_CG) {
        // r = b, z = M^-1 r, p = z (x0 = 0); buffers: dz = z, dv = A p
        hipLaunchKernelGGL(k_init_cg, gN, blk, 0, nullptr, n, (const double *)db.p, (const double *)ddinv.p, dx.p, dr.p, dz.p, dp.p);
        dot(dr.p, dz.p, 0);
        dot(db.p, db.p, 4);
        (void)hipMemcpy(h, dscal.p, sizeof h, hipMemcpyDeviceToHost);
        bb = h[4] > 0. ? h[4] : 1.;
        rr = bb;
        while (it < max_iter && h[4] > 0.) {
            spmv(dp.p, dv.p);
            dot(dp.p, dv.p, 1);
            hipLaunchKernelGGL(k_update_xr, gN, blk, 0, nullptr, n, (const double *)dscal.p, (const double *)dp.p, (const double *)dv.p, (const double *)ddinv.p, dx.p, dr.p, dz.p);
            dot(dr.p, dz.p, 2);
            hipLaunchKernelGGL(k_update_p, gN, blk, 0, nullptr, n, dscal.p, (const double *)dz.p, dp.p);
            hipLaunchKernelGGL(k_shift_rz, dim3(1), dim3(1), 0, nullptr, dscal.p);
            ++it;
            if (it % check_every == 0 || it == max_iter) {
                dot(dr.p, dr.p, 3);
                (void)hipMemcpy(h, dscal.p, sizeof h, hipMemcpyDeviceToHost);
                rr = h[3];
                if (!(rr == rr) || std::sqrt(rr / bb) <= rtol) break;
            }
        }
    } else {
        hipLaunchKernelGGL(k_copy2, gN, blk, 0, nullptr, n, (const double *)db.p, dr.p, drh.p);
        hipLaunchKernelGGL(k_set_ones, dim3(1), dim3(1), 0, nullptr, dscal.p);
        dot(db.p, db.p, 7);
        (void)hipMemcpy(h, dscal.p, sizeof h, hipMemcpyDeviceToHost);
        bb = h[7] > 0. ? h[7] : 1.;
        rr = bb;
        while (it < max_iter && h[7] > 0.) {
            dot(drh.p, dr.p, 3);                                                  // rho_new
            hipLaunchKernelGGL(k_bicg_p, gN, blk, 0, nullptr, n, (const double *)dscal.p, (const double *)dr.p, (const double *)dv.p, (const double *)ddinv.p, dp.p, dy.p);
            spmv(dy.p, dv.p);                                                     // v = A M^-1 p
            dot(drh.p, dv.p, 4);
            hipLaunchKernelGGL(k_bicg_s, gN, blk, 0, nullptr, n, (const double *)dscal.p, (const double *)dr.p, (const double *)dv.p, (const double *)ddinv.p, ds.p, dz.p);
            spmv(dz.p, dt.p);                                                     // t = A M^-1 s
            dot(dt.p, ds.p, 5);
            dot(dt.p, dt.p, 6);
            hipLaunchKernelGGL(k_bicg_x, gN, blk, 0, nullptr, n, (const double *)dscal.p, (const double *)dy.p, (const double *)dz.p, (const double *)ds.p, (const double *)dt.p, dx.p, dr.p);
            hipLaunchKernelGGL(k_bicg_shift, dim3(1), dim3(1), 0, nullptr, dscal.p);
            ++it;
            if (it % check_every == 0 || it == max_iter) {
                dot(dr.p, dr.p, 7);
                (void)hipMemcpy(h, dscal.p, sizeof h, hipMemcpyDeviceToHost);
                rr = h[7];
                if (!(rr == rr) || std::sqrt(rr / bb) <= rtol) break;
                if (h[0] == 0.) break;  // breakdown (rho = 0)
            }
        }
    }
    (void)hipEventRecord(e1, nullptr);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "solver kernels failed: %s", hipGetErrorString(err));
    if (hipMemcpy(x, dx.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(NXS_ERR_HIP, "copy back failed");
    if (iterations) *iterations = it;
    if (rel_residual) *rel_residual = std::sqrt(rr / bb);
    if (ms_solve) *ms_solve = ms;
    return NXS_OK;
}

}  // extern "C"

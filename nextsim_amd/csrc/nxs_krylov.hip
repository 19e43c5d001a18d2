// nxs_krylov.hip -- EXTENSION (see include/nxs_krylov.h): coloured assembly of the P1
// stiffness matrix and Jacobi-preconditioned CG / BiCGStab built from SpMV / fused vector-update / dot kernels,
// single GPU or row-distributed over ranks (halo exchange of the SpMV operand + all-reduce of the dots).
// No live counterpart in the reference (research/laplacian.cpp is its only, un-buildable, trace): PARITY UNPINNED.
//
// Matrix layout in HBM: sliced ELLPACK.  A slice = 64 consecutive rows = one wavefront; inside a slice entry j of
// every row is stored side by side (val[off + 64 j + lane], col likewise), the slice width is its longest row, short
// rows are padded with (col = row, val = 0).  A P1 mesh row has 5..9 entries, so the padding is a few per cent and
// every load of the SpMV is a full 512 B / 256 B line per wavefront -- the CSR-scalar kernel it replaces read 7
// scattered segments per wavefront load.  Entries keep their CSR order inside a row: the row sums have the bits of
// the CSR loop.
// Dots: per-block partial sums, the last block to finish (a ticket counter) adds them in index order -- no atomics
// on data, bit-reproducible run to run, and no separate reduction launch.  All scalars stay on the device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_guard.hpp"
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

// handler of every extern "C" function-try-block of this file (nxs_guard.hpp): status code + text, never an exception across the ABI
int entry_caught(const char *entry) noexcept {
    return nxs_guard::caught(entry, [](int code, const char *text) { (void)fail(code, "%s", text); });
}
#define KCHK(call)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(NXS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));          \
    } while (0)

// (never the legacy default stream: another handle of this process may be capturing a graph on its own thread)
inline hipError_t copy_on(hipStream_t st, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
inline hipError_t memset_on(hipStream_t st, void *dst, int value, size_t bytes) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

constexpr int BS = 256;     // threads per block = 4 slices
constexpr int SL = 64;      // rows per slice
constexpr int MAXG = 2048;  // most blocks of a reducing kernel (= partial sums per value)
constexpr unsigned int NGRP = 16, CSTRIDE = 32;  // ticket counters: [0] global, [CSTRIDE (g+1)] group g (128 B apart)

// scalar slots (device array of 16 doubles).  A pair that is all-reduced together is adjacent.
//   CG:       [0] rz (even iterations) [1] rr   [2] rz (odd) [3] rr   [4] pAp   [5] bb
//   BiCGStab: [0] rho (even)           [1] rr   [2] rho (odd) [3] rr  [4] (rhat,v)  [5] bb  [6] (t,s) [7] (t,t)
enum { S_PAP = 4, S_BB = 5, S_TS = 6 };

// ---- reductions ----------------------------------------------------------------------------------
// Every thread brings NV running sums; afterwards scal[slot + v] holds the grid's total of value v.
// The partial sums travel as agent-scope atomic stores / loads (they bypass the per-XCD L2s) and the storing lane drains
// them before it takes its ticket: no __threadfence, which on this chip writes the XCD's whole dirty L2 back -- the result
// vector the block has just written -- once per block (measured: 45-65 us per reducing kernel, 3x the kernel itself).
template <int NV>
__device__ __forceinline__ void grid_sum(double (&acc)[NV], double *__restrict__ partial, unsigned int *__restrict__ counter,
                                         double *__restrict__ scal, int slot) {
    __shared__ double sh[NV][BS / 64];
    __shared__ bool last;
    const int t = threadIdx.x;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double a = acc[v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((t & 63) == 0) sh[v][t >> 6] = a;
    }
    __syncthreads();
    if (t == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double s = 0.;
            for (int i = 0; i < BS / 64; ++i) s += sh[v][i];
            __hip_atomic_store(partial + v * MAXG + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two-level ticket: atomics on ONE address are served one after the other (~10 ns each at agent scope, 20 us for
        // 2048 blocks); 16 group counters on separate lines, then one more ticket for the last block of each group
        const unsigned int g = blockIdx.x % NGRP, members = (gridDim.x - g + NGRP - 1) / NGRP;
        last = false;
        if (__hip_atomic_fetch_add(counter + CSTRIDE * (g + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            const unsigned int groups = gridDim.x < NGRP ? gridDim.x : NGRP;
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
        }
    }
    __syncthreads();
    if (!last) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double a = 0.;
        for (int i = t; i < (int)gridDim.x; i += BS) a += __hip_atomic_load(partial + v * MAXG + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        __syncthreads();
        if ((t & 63) == 0) sh[v][t >> 6] = a;
        __syncthreads();
        if (t == 0) {
            double s = 0.;
            for (int i = 0; i < BS / 64; ++i) s += sh[v][i];
            scal[slot + v] = s;
        }
    }
    if (t <= NGRP) __hip_atomic_store(counter + CSTRIDE * t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- SpMV ---------------------------------------------------------------------------------------
// out = A in for the n owned rows; ND dots on the way: ND >= 1: sum w0[r] out[r], ND == 2: also sum out[r]^2.
template <int ND>
__global__ void __launch_bounds__(BS) k_spmv_sell(int n, int nslices, const int *__restrict__ off, const int *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ in, double *__restrict__ out,
                                                  const double *__restrict__ w0, double *__restrict__ partial, unsigned int *__restrict__ counter,
                                                  double *__restrict__ scal, int slot) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[ND > 0 ? ND : 1] = {0.};
    for (int s = blockIdx.x * (BS / SL) + wv; s < nslices; s += gridDim.x * (BS / SL)) {
        const int o = off[s], w = (off[s + 1] - o) >> 6;
        const int *c = col + o + lane;
        const double *v = val + o + lane;
        double sum = 0.;
        int j = 0;
        for (; j + 4 <= w; j += 4) {  // four independent gathers in flight
            const int c0 = c[j * SL], c1 = c[(j + 1) * SL], c2 = c[(j + 2) * SL], c3 = c[(j + 3) * SL];
            const double v0 = v[j * SL], v1 = v[(j + 1) * SL], v2 = v[(j + 2) * SL], v3 = v[(j + 3) * SL];
            const double x0 = in[c0], x1 = in[c1], x2 = in[c2], x3 = in[c3];
            sum += v0 * x0; sum += v1 * x1; sum += v2 * x2; sum += v3 * x3;
        }
        for (; j < w; ++j) sum += v[j * SL] * in[c[j * SL]];
        const int r = s * SL + lane;
        if (r < n) {
            out[r] = sum;
            if (ND >= 1) acc[0] += w0[r] * sum;
            if (ND == 2) acc[1] += sum * sum;
        }
    }
    if (ND > 0) grid_sum<(ND > 0 ? ND : 1)>(acc, partial, counter, scal, slot);
}

// first entry of the row that sits on the diagonal (padding repeats col = row behind the real entries)
__global__ void __launch_bounds__(BS) k_diag_inv(int n, int nslices, const int *__restrict__ off, const int *__restrict__ col,
                                                 const double *__restrict__ val, double *__restrict__ dinv, int *__restrict__ bad) {
    const int r = blockIdx.x * BS + threadIdx.x;
    if (r >= n) return;
    const int s = r >> 6, lane = r & 63, o = off[s], w = (off[s + 1] - o) >> 6;
    double d = 0.;
    for (int j = 0; j < w; ++j)
        if (col[o + j * SL + lane] == r) { d = val[o + j * SL + lane]; break; }
    if (d == 0.) atomicExch(bad, r + 1);
    dinv[r] = 1. / d;
}

// ---- CG -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BS) k_cg_init(int n, const double *__restrict__ b, const double *__restrict__ dinv, double *__restrict__ x,
                                                double *__restrict__ r, double *__restrict__ z, double *__restrict__ p, double *__restrict__ partial,
                                                unsigned int *__restrict__ counter, double *__restrict__ scal) {
    double acc[2] = {0., 0.};
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const double bi = b[i], zi = dinv[i] * bi;
        x[i] = 0.; r[i] = bi; z[i] = zi; p[i] = zi;
        acc[0] += bi * zi; acc[1] += bi * bi;
    }
    grid_sum<2>(acc, partial, counter, scal, 0);  // rz -> [0], rr = bb -> [1]
}
// x += alpha p, r -= alpha Ap, z = M^-1 r, and the dots (r,z), (r,r) of the new residual
__global__ void __launch_bounds__(BS) k_cg_xr(int n, int par, const double *__restrict__ p, const double *__restrict__ Ap, const double *__restrict__ dinv,
                                              double *__restrict__ x, double *__restrict__ r, double *__restrict__ z, double *__restrict__ partial,
                                              unsigned int *__restrict__ counter, double *__restrict__ scal) {
    const double alpha = scal[2 * par] / scal[S_PAP];
    double acc[2] = {0., 0.};
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i];
        const double zi = dinv[i] * ri;
        r[i] = ri; z[i] = zi;
        acc[0] += ri * zi; acc[1] += ri * ri;
    }
    grid_sum<2>(acc, partial, counter, scal, 2 * (par ^ 1));
}
__global__ void __launch_bounds__(BS) k_cg_p(int n, int par, const double *__restrict__ scal, const double *__restrict__ z, double *__restrict__ p) {
    const double beta = scal[2 * (par ^ 1)] / scal[2 * par];
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) p[i] = z[i] + beta * p[i];
}

// ---- BiCGStab (right Jacobi preconditioner) ---------------------------------------------------------
__global__ void __launch_bounds__(BS) k_bicg_init(int n, const double *__restrict__ b, double *__restrict__ x, double *__restrict__ r, double *__restrict__ rhat,
                                                  double *__restrict__ p, double *__restrict__ v, double *__restrict__ partial,
                                                  unsigned int *__restrict__ counter, double *__restrict__ scal) {
    double acc[2] = {0., 0.};
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const double bi = b[i];
        x[i] = 0.; r[i] = bi; rhat[i] = bi; p[i] = 0.; v[i] = 0.;
        acc[0] += bi * bi; acc[1] += bi * bi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal[2] = 1.; scal[S_PAP] = 1.; scal[S_TS] = 1.; scal[S_TS + 1] = 1.; }  // rho_old, (rhat,v), (t,s), (t,t) of "iteration -1"
    grid_sum<2>(acc, partial, counter, scal, 0);  // rho -> [0], rr = bb -> [1]
}
__global__ void __launch_bounds__(BS) k_bicg_p(int n, int par, const double *__restrict__ scal, const double *__restrict__ r, const double *__restrict__ v,
                                               const double *__restrict__ dinv, double *__restrict__ p, double *__restrict__ y) {
    const double rho = scal[2 * par], rho_old = scal[2 * (par ^ 1)];
    const double alpha_prev = rho_old / scal[S_PAP], omega_prev = scal[S_TS + 1] != 0. ? scal[S_TS] / scal[S_TS + 1] : 0.;
    const double beta = (rho / rho_old) * (alpha_prev / omega_prev);
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const double pi = r[i] + beta * (p[i] - omega_prev * v[i]);
        p[i] = pi;
        y[i] = dinv[i] * pi;
    }
}
__global__ void __launch_bounds__(BS) k_bicg_s(int n, int par, const double *__restrict__ scal, const double *__restrict__ r, const double *__restrict__ v,
                                               const double *__restrict__ dinv, double *__restrict__ sv, double *__restrict__ z) {
    const double alpha = scal[2 * par] / scal[S_PAP];
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        const double si = r[i] - alpha * v[i];
        sv[i] = si;
        z[i] = dinv[i] * si;
    }
}
__global__ void __launch_bounds__(BS) k_bicg_x(int n, int par, const double *__restrict__ y, const double *__restrict__ z, const double *__restrict__ sv,
                                               const double *__restrict__ t, const double *__restrict__ rhat, double *__restrict__ x, double *__restrict__ r,
                                               double *__restrict__ partial, unsigned int *__restrict__ counter, double *__restrict__ scal) {
    const double alpha = scal[2 * par] / scal[S_PAP];
    const double omega = scal[S_TS + 1] != 0. ? scal[S_TS] / scal[S_TS + 1] : 0.;
    double acc[2] = {0., 0.};
    for (int i = blockIdx.x * BS + threadIdx.x; i < n; i += gridDim.x * BS) {
        x[i] += alpha * y[i] + omega * z[i];
        const double ri = sv[i] - omega * t[i];
        r[i] = ri;
        acc[0] += rhat[i] * ri; acc[1] += ri * ri;
    }
    grid_sum<2>(acc, partial, counter, scal, 2 * (par ^ 1));  // next rho, rr
}

// ---- halo of the SpMV operand -------------------------------------------------------------------
__global__ void __launch_bounds__(BS) k_pack(int total, const int *__restrict__ index, const double *__restrict__ vec, double *__restrict__ buf) {
    const int j = blockIdx.x * BS + threadIdx.x;
    if (j < total) buf[j] = vec[index[j]];
}
__global__ void __launch_bounds__(BS) k_unpack(int total, const int *__restrict__ index, const double *__restrict__ buf, double *__restrict__ vec) {
    const int j = blockIdx.x * BS + threadIdx.x;
    if (j < total) vec[index[j]] = buf[j];
}

// ---- coloured assembly of the P1 Laplacian (research/laplacian.cpp:163-224) -------------------------
// Graph-coloured scatter, no atomics, ONE launch and one pass over the matrix.
// Colour-by-colour scatter over the whole mesh (one launch per colour) re-reads and re-writes nearly every line of the
// matrix once per colour: 11 passes, measured 1.0 TB/s of algorithmic bytes; colouring chunks of elements against each
// other (6 launches) is bound by the ~11 us dependent chain of one workgroup per launch: 2.9 TB/s.  Here a workgroup
// owns a PATCH of 128 consecutive rows = two slices of the matrix = one contiguous piece of val[], which it builds
// in LDS: every element touching one of its rows is computed by a thread (elements on a patch border are computed by
// two or three patches), the element colours take turns -- a barrier in between, elements of one colour share no node --
// adding the rows the patch owns into the LDS copy, and the finished piece is written to HBM with plain
// contiguous stores: no read-modify-write on HBM, no zero fill.  Colours in ascending order: every entry is summed in a
// fixed order, run to run bit-identical.
constexpr int PROWS = 2 * SL;   // rows per patch
constexpr int PT = 512;         // threads per patch (a patch has ~330 elements)
__global__ void __launch_bounds__(PT) k_assemble_patches(int Nn, int nslices, int ncol, const int *__restrict__ off, const int *__restrict__ pel_off,
                                                         const int *__restrict__ ptri /*[3][tot]*/, const double *__restrict__ pf, const unsigned char *__restrict__ pcol,
                                                         const unsigned short *__restrict__ lpos /*[9][tot]*/, const unsigned char *__restrict__ lrow /*[3][tot]*/,
                                                         size_t tot, const double *__restrict__ x, const double *__restrict__ y, int Lmax,
                                                         double *__restrict__ val, double *__restrict__ rhs) {
    extern __shared__ double lacc[];  // [Lmax] the patch's piece of val[], then its PROWS rhs entries
    const int p = blockIdx.x, t = threadIdx.x;
    const int i0 = pel_off[p], nE = pel_off[p + 1] - i0;
    const int base = off[2 * p], len = off[min(2 * p + 2, nslices)] - base;
    for (int i = t; i < len; i += PT) lacc[i] = 0.;
    if (t < PROWS) lacc[Lmax + t] = 0.;
    for (int b = 0; b < nE || b == 0; b += PT) {
        const bool active = b + t < nE;
        const size_t i = (size_t)i0 + b + t;
        double m[9], fj = 0.;
        int lp[9], lr[3], mycol = -1;
        if (active) {
            const int nd[3] = {ptri[i], ptri[tot + i], ptri[2 * tot + i]};
#pragma unroll
            for (int k = 0; k < 9; ++k) lp[k] = lpos[(size_t)k * tot + i];
#pragma unroll
            for (int k = 0; k < 3; ++k) lr[k] = lrow[(size_t)k * tot + i];
            mycol = pcol[i];
            const double xs[3] = {x[nd[0]], x[nd[1]], x[nd[2]]}, ys[3] = {y[nd[0]], y[nd[1]], y[nd[2]]};
            double area = (xs[1] - xs[0]) * (ys[2] - ys[0]);
            area -= (xs[2] - xs[0]) * (ys[1] - ys[0]);
            area = (1. / 2) * fabs(area);
            fj = pf[i] * area / 3.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int jp1 = (j + 1) % 3, jp2 = (j + 2) % 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                    const double m_jk = (ys[jp1] - ys[jp2]) * (ys[kp1] - ys[kp2]) + (xs[jp1] - xs[jp2]) * (xs[kp1] - xs[kp2]);
                    m[3 * j + k] = m_jk / (4.0 * area);
                }
            }
        }
        __syncthreads();
        for (int col = 0; col < ncol; ++col) {  // the graph-coloured scatter
            if (mycol == col) {
#pragma unroll
                for (int k = 0; k < 9; ++k) if (lp[k] != 0xFFFF) lacc[lp[k]] += m[k];   // 0xFFFF: a row of another patch
#pragma unroll
                for (int k = 0; k < 3; ++k) if (lr[k] != 0xFF) lacc[Lmax + lr[k]] += fj;
            }
            __syncthreads();
        }
    }
    for (int i = t; i < len; i += PT) val[base + i] = lacc[i];
    if (t < PROWS && p * PROWS + t < Nn) rhs[p * PROWS + t] = lacc[Lmax + t];
}

// homogeneous Dirichlet: row and column zeroed, unit diagonal, rhs 0 (MatrixPetsc::on in the demo)
__global__ void __launch_bounds__(BS) k_apply_dirichlet(int n, const int *__restrict__ off, const int *__restrict__ col, const unsigned char *__restrict__ dir,
                                                        double *__restrict__ val, double *__restrict__ rhs) {
    const int r = blockIdx.x * BS + threadIdx.x;
    if (r >= n) return;
    const int s = r >> 6, lane = r & 63, o = off[s], w = (off[s + 1] - o) >> 6;
    const bool dr = dir[r];
    bool seen = false;  // the diagonal comes before the padding, which repeats its column
    for (int j = 0; j < w; ++j) {
        const int q = o + j * SL + lane, c = col[q];
        if (c == r) {
            if (seen) continue;
            seen = true;
            if (dr) val[q] = 1.;
        } else if (dr || dir[c]) {
            val[q] = 0.;
        }
    }
    if (dr) rhs[r] = 0.;
}

// ---- host side ----------------------------------------------------------------------------------
struct Rccl {  // resolved at comm_init, no link-time dependency (as in nxs_dyn.hip)
    void *lib = nullptr;
    void *CommInitRank = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct NcclId { char internal[128]; };
typedef int (*nccl_comm_init_rank_t)(void **, int, NcclId, int);

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

// CSR -> sliced ELLPACK.  where[q] = position of CSR entry q.  Returns false if the matrix is too large for int offsets.
bool csr_to_sell(int n, const int *rowptr, const int *colidx, const double *val, std::vector<int> &off, std::vector<int> &col, std::vector<double> &sval,
                 std::vector<int> *where) {
    const int ns = (n + SL - 1) / SL;
    off.assign(ns + 1, 0);
    long long tot = 0;
    for (int s = 0; s < ns; ++s) {
        int w = 0;
        for (int r = s * SL; r < std::min(n, (s + 1) * SL); ++r) w = std::max(w, rowptr[r + 1] - rowptr[r]);
        tot += (long long)w * SL;
        if (tot > 0x7fffffffll) return false;
        off[s + 1] = (int)tot;
    }
    col.assign((size_t)tot, 0);
    sval.assign((size_t)tot, 0.);
    if (where) where->assign((size_t)rowptr[n], 0);
    for (int s = 0; s < ns; ++s) {
        const int w = (off[s + 1] - off[s]) / SL;
        for (int lane = 0; lane < SL; ++lane) {
            const int r = s * SL + lane;
            const int cnt = r < n ? rowptr[r + 1] - rowptr[r] : 0;
            for (int j = 0; j < w; ++j) {
                const size_t q = (size_t)off[s] + (size_t)j * SL + lane;
                if (j < cnt) {
                    col[q] = colidx[rowptr[r] + j];
                    if (val) sval[q] = val[rowptr[r] + j];
                    if (where) (*where)[rowptr[r] + j] = (int)q;
                } else {
                    col[q] = r < n ? r : 0;
                }
            }
        }
    }
    return true;
}

}  // namespace

struct nxs_krylov_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    // matrix
    int n = 0, n_local = 0, nslices = 0;
    size_t entries = 0, nnz = 0;
    int *d_off = nullptr, *d_col = nullptr, *d_bad = nullptr;
    double *d_val = nullptr, *d_dinv = nullptr;
    // work vectors (n_local each), partial sums, scalars
    double *vec[9] = {nullptr};
    double *d_partial = nullptr, *d_scal = nullptr;
    unsigned int *d_counter = nullptr;
    // halo of the SpMV operand
    int rank = 0, nranks = 1;
    std::vector<int> send_procs, send_offsets, recv_procs, recv_offsets;
    int *d_send_index = nullptr, *d_recv_index = nullptr;
    double *d_send_buf = nullptr, *d_recv_buf = nullptr, *h_send = nullptr, *h_recv = nullptr, *h_scal = nullptr;
    bool have_halo = false;
    Rccl rccl;
    void *comm = nullptr;
    nxs_krylov_exchange_fn exchange_fn = nullptr;
    nxs_krylov_allreduce_fn allreduce_fn = nullptr;
    void *user = nullptr;
    long long n_rccl_allreduce = 0, n_rccl_exchange = 0;  // RCCL calls issued since create (nxs_krylov_comm_stats)
};

namespace {

template <typename T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

void free_matrix(nxs_krylov_handle *h) {
    dfree(h->d_off); dfree(h->d_col); dfree(h->d_val); dfree(h->d_dinv);
    for (auto &v : h->vec) dfree(v);
    h->n = h->n_local = h->nslices = 0;
}
void free_halo(nxs_krylov_handle *h) {
    dfree(h->d_send_index); dfree(h->d_recv_index); dfree(h->d_send_buf); dfree(h->d_recv_buf);
    if (h->h_send) { (void)hipHostFree(h->h_send); h->h_send = nullptr; }
    if (h->h_recv) { (void)hipHostFree(h->h_recv); h->h_recv = nullptr; }
    h->have_halo = false;
}

int grid_for(int n) { return std::max(1, std::min(MAXG, (n + BS - 1) / BS)); }
int spmv_grid(const nxs_krylov_handle *h) { return std::max(1, std::min(MAXG, (h->nslices + BS / SL - 1) / (BS / SL))); }

// device matrix from a host SELL pattern (+ values, or zeros to be assembled into)
int upload_matrix(nxs_krylov_handle *h, int n, int n_local, const std::vector<int> &off, const std::vector<int> &col, const std::vector<double> &val, size_t nnz) {
    free_matrix(h);
    h->n = n; h->n_local = n_local; h->nslices = (int)off.size() - 1; h->entries = col.size(); h->nnz = nnz;
    KCHK(hipMalloc((void **)&h->d_off, off.size() * sizeof(int)));
    KCHK(hipMalloc((void **)&h->d_col, std::max<size_t>(col.size(), 1) * sizeof(int)));
    KCHK(hipMalloc((void **)&h->d_val, std::max<size_t>(val.size(), 1) * sizeof(double)));
    KCHK(hipMalloc((void **)&h->d_dinv, (size_t)n * sizeof(double)));
    KCHK(copy_on(h->stream, h->d_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    KCHK(copy_on(h->stream, h->d_col, col.data(), col.size() * sizeof(int), hipMemcpyHostToDevice));
    KCHK(copy_on(h->stream, h->d_val, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
    for (auto &v : h->vec) {
        KCHK(hipMalloc((void **)&v, (size_t)n_local * sizeof(double)));
        KCHK(memset_on(h->stream, v, 0, (size_t)n_local * sizeof(double)));
    }
    return NXS_OK;
}

int finish_matrix(nxs_krylov_handle *h) {  // Jacobi preconditioner; refuses a row without a diagonal
    KCHK(hipMemsetAsync(h->d_bad, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_diag_inv, dim3((h->n + BS - 1) / BS), dim3(BS), 0, h->stream, h->n, h->nslices, (const int *)h->d_off, (const int *)h->d_col,
                       (const double *)h->d_val, h->d_dinv, h->d_bad);
    int bad = 0;
    KCHK(hipMemcpyAsync(&bad, h->d_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    KCHK(hipStreamSynchronize(h->stream));
    if (bad) return fail(NXS_ERR_INVALID, "row %d has no diagonal entry (Jacobi preconditioner)", bad - 1);
    return NXS_OK;
}

bool distributed(const nxs_krylov_handle *h) { return h->nranks > 1; }

// ghost entries of vec <- their owners' values
int exchange(nxs_krylov_handle *h, double *vec) {
    if (!distributed(h)) return NXS_OK;
    if (!h->have_halo) return fail(NXS_ERR_STATE, "distributed solve without nxs_krylov_set_halo");
    if (!h->comm && !h->exchange_fn) return fail(NXS_ERR_STATE, "distributed solve without a communicator (nxs_krylov_comm_init / nxs_krylov_set_comm_fns)");
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    if (ts > 0) hipLaunchKernelGGL(k_pack, dim3((ts + BS - 1) / BS), dim3(BS), 0, h->stream, ts, (const int *)h->d_send_index, (const double *)vec, h->d_send_buf);
    if (h->exchange_fn) {  // the caller's communicator, host staged
        if (ts > 0) KCHK(hipMemcpyAsync(h->h_send, h->d_send_buf, (size_t)ts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        KCHK(hipStreamSynchronize(h->stream));
        const int rc = h->exchange_fn(h->user, h->h_send, h->h_recv);
        if (rc != 0) return fail(NXS_ERR_COMM, "exchange callback returned %d", rc);
        if (tr > 0) KCHK(hipMemcpyAsync(h->d_recv_buf, h->h_recv, (size_t)tr * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
        const int ncclDouble = 8;
        int e = h->rccl.GroupStart();
        for (int k = 0; k < ns && e == 0; ++k)
            e = h->rccl.Send(h->d_send_buf + h->send_offsets[k], (size_t)(h->send_offsets[k + 1] - h->send_offsets[k]), ncclDouble, h->send_procs[k], h->comm, h->stream);
        for (int k = 0; k < nr && e == 0; ++k)
            e = h->rccl.Recv(h->d_recv_buf + h->recv_offsets[k], (size_t)(h->recv_offsets[k + 1] - h->recv_offsets[k]), ncclDouble, h->recv_procs[k], h->comm, h->stream);
        const int e2 = h->rccl.GroupEnd();
        if (e == 0) e = e2;
        if (e != 0) return fail(NXS_ERR_COMM, "halo send/recv: %s", h->rccl.GetErrorString(e));
        h->n_rccl_exchange++;
    }
    if (tr > 0) hipLaunchKernelGGL(k_unpack, dim3((tr + BS - 1) / BS), dim3(BS), 0, h->stream, tr, (const int *)h->d_recv_index, (const double *)h->d_recv_buf, vec);
    return NXS_OK;
}

// scal[slot .. slot+count) <- sum over the ranks
int allreduce(nxs_krylov_handle *h, int slot, int count) {
    // a communicator of ONE rank still reduces (a sum over one rank): it costs an unusual caller a few microseconds per dot and
    // lets a one-GPU box execute the very ncclAllReduce calls of the distributed solve (tests/test_krylov.py)
    if (!distributed(h) && !h->comm) return NXS_OK;
    if (h->allreduce_fn) {
        KCHK(hipMemcpyAsync(h->h_scal, h->d_scal + slot, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        KCHK(hipStreamSynchronize(h->stream));
        const int rc = h->allreduce_fn(h->user, h->h_scal, count);
        if (rc != 0) return fail(NXS_ERR_COMM, "all-reduce callback returned %d", rc);
        KCHK(hipMemcpyAsync(h->d_scal + slot, h->h_scal, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
        KCHK(hipStreamSynchronize(h->stream));  // h_scal is reused by the next call
        return NXS_OK;
    }
    if (!h->comm) return fail(NXS_ERR_STATE, "distributed solve without a communicator");
    const int ncclDouble = 8, ncclSum = 0;
    const int e = h->rccl.AllReduce(h->d_scal + slot, h->d_scal + slot, (size_t)count, ncclDouble, ncclSum, h->comm, h->stream);
    if (e != 0) return fail(NXS_ERR_COMM, "ncclAllReduce: %s", h->rccl.GetErrorString(e));
    h->n_rccl_allreduce++;
    return NXS_OK;
}

template <int ND>
void spmv(nxs_krylov_handle *h, const double *in, double *out, const double *w0, int slot) {
    hipLaunchKernelGGL((k_spmv_sell<ND>), dim3(spmv_grid(h)), dim3(BS), 0, h->stream, h->n, h->nslices, (const int *)h->d_off, (const int *)h->d_col,
                       (const double *)h->d_val, in, out, w0, h->d_partial, h->d_counter, h->d_scal, slot);
}

// b, x on the device (vec[0] = x, vec[1] = b)
int run_solver(nxs_krylov_handle *h, int method, double rtol, int max_iter, int *iterations, double *rel_residual, double *ms_solve) {
    const int n = h->n, g = grid_for(n);
    const dim3 G(g), B(BS);
    double *x = h->vec[0], *b = h->vec[1], *r = h->vec[2], *p = h->vec[3], *q = h->vec[4], *z = h->vec[5], *rhat = h->vec[6], *y = h->vec[7], *sv = h->vec[8];
    double *partial = h->d_partial, *scal = h->d_scal;
    unsigned int *counter = h->d_counter;
    hipEvent_t e0, e1;
    KCHK(hipEventCreate(&e0)); KCHK(hipEventCreate(&e1));
    KCHK(hipEventRecord(e0, h->stream));
    int rc = NXS_OK, it = 0;
    double hs[8] = {0};
    auto read_scal = [&]() -> int {
        KCHK(hipMemcpyAsync(hs, scal, sizeof hs, hipMemcpyDeviceToHost, h->stream));
        KCHK(hipStreamSynchronize(h->stream));
        return NXS_OK;
    };
    const int check_every = 10;
    double bb = 1., rr = 0.;
    if (method == NXS_KRYLOV_CG) {
        hipLaunchKernelGGL(k_cg_init, G, B, 0, h->stream, n, (const double *)b, (const double *)h->d_dinv, x, r, z, p, partial, counter, scal);
        if ((rc = allreduce(h, 0, 2)) || (rc = read_scal())) goto done;
        bb = hs[1] > 0. ? hs[1] : 1.;
        rr = hs[1];
        while (it < max_iter && hs[1] > 0.) {
            const int par = it & 1;
            if ((rc = exchange(h, p))) goto done;
            spmv<1>(h, p, q, p, S_PAP);                                                   // q = A p, (p, Ap)
            if ((rc = allreduce(h, S_PAP, 1))) goto done;
            hipLaunchKernelGGL(k_cg_xr, G, B, 0, h->stream, n, par, (const double *)p, (const double *)q, (const double *)h->d_dinv, x, r, z, partial, counter, scal);
            if ((rc = allreduce(h, 2 * (par ^ 1), 2))) goto done;
            hipLaunchKernelGGL(k_cg_p, G, B, 0, h->stream, n, par, (const double *)scal, (const double *)z, p);
            ++it;
            if (it % check_every == 0 || it == max_iter) {
                if ((rc = read_scal())) goto done;
                rr = hs[2 * (it & 1) + 1];
                if (!(rr == rr) || std::sqrt(rr / bb) <= rtol) break;
            }
        }
    } else {
        hipLaunchKernelGGL(k_bicg_init, G, B, 0, h->stream, n, (const double *)b, x, r, rhat, p, q, partial, counter, scal);
        if ((rc = allreduce(h, 0, 2)) || (rc = read_scal())) goto done;
        bb = hs[1] > 0. ? hs[1] : 1.;
        rr = hs[1];
        double *v = q, *t = b;  // b is consumed once r and rhat hold it: its buffer takes t = A M^-1 s
        while (it < max_iter && hs[1] > 0.) {
            const int par = it & 1;
            hipLaunchKernelGGL(k_bicg_p, G, B, 0, h->stream, n, par, (const double *)scal, (const double *)r, (const double *)v, (const double *)h->d_dinv, p, y);
            if ((rc = exchange(h, y))) goto done;
            spmv<1>(h, y, v, rhat, S_PAP);                                                // v = A M^-1 p, (rhat, v)
            if ((rc = allreduce(h, S_PAP, 1))) goto done;
            hipLaunchKernelGGL(k_bicg_s, G, B, 0, h->stream, n, par, (const double *)scal, (const double *)r, (const double *)v, (const double *)h->d_dinv, sv, z);
            if ((rc = exchange(h, z))) goto done;
            spmv<2>(h, z, t, sv, S_TS);                                                   // t = A M^-1 s, (t, s), (t, t)
            if ((rc = allreduce(h, S_TS, 2))) goto done;
            hipLaunchKernelGGL(k_bicg_x, G, B, 0, h->stream, n, par, (const double *)y, (const double *)z, (const double *)sv, (const double *)t, (const double *)rhat, x, r,
                               partial, counter, scal);
            if ((rc = allreduce(h, 2 * (par ^ 1), 2))) goto done;
            ++it;
            if (it % check_every == 0 || it == max_iter) {
                if ((rc = read_scal())) goto done;
                rr = hs[2 * (it & 1) + 1];
                if (!(rr == rr) || std::sqrt(rr / bb) <= rtol) break;
                if (hs[2 * (it & 1)] == 0.) break;  // breakdown (rho = 0)
            }
        }
    }
done:
    (void)hipEventRecord(e1, h->stream);
    hipError_t err = hipStreamSynchronize(h->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "solver kernels failed: %s", hipGetErrorString(err));
    if (iterations) *iterations = it;
    if (rel_residual) *rel_residual = std::sqrt(rr / bb);
    if (ms_solve) *ms_solve = ms;
    return NXS_OK;
}

int load_rccl(Rccl &r) {
    if (r.lib) return NXS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(NXS_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                       \
    *(void **)(&r.field) = dlsym(r.lib, name);                                 \
    if (!r.field) return fail(NXS_ERR_COMM, "librccl lacks %s", name)
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return NXS_OK;
}

}  // namespace

extern "C" {

const char *nxs_krylov_last_error(void) { return g_err.c_str(); }

int nxs_fem_csr_pattern(const int32_t *indices, int32_t Nn, int32_t Ne, int32_t *rowptr, int32_t *colidx, int64_t *nnz) try {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!rowptr || !nnz) return fail(NXS_ERR_INVALID, "NULL argument");
    std::vector<int> rp, ci;
    build_pattern(indices, Nn, Ne, rp, ci);
    std::copy(rp.begin(), rp.end(), rowptr);
    *nnz = (int64_t)ci.size();
    if (colidx) std::copy(ci.begin(), ci.end(), colidx);
    return NXS_OK;
} catch (...) { return entry_caught("nxs_fem_csr_pattern"); }

int nxs_fem_colour_elements(const int32_t *indices, int32_t Nn, int32_t Ne, int32_t *colour, int32_t *ncolours) try {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!colour || !ncolours) return fail(NXS_ERR_INVALID, "NULL argument");
    std::vector<int> c; int n = 0;
    colour_elements(indices, Nn, Ne, c, n);
    std::copy(c.begin(), c.end(), colour);
    *ncolours = n;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_fem_colour_elements"); }

int nxs_krylov_create(int32_t device, nxs_krylov_handle **out) try {
    if (!out) return fail(NXS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NXS_ERR_NO_DEVICE, "no HIP device visible: the solver has no CPU path");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return fail(NXS_ERR_INVALID, "bad device %d", device);
    nxs_krylov_handle *h = new nxs_krylov_handle;
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&h->d_partial, 2 * MAXG * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&h->d_scal, 16 * sizeof(double)) != hipSuccess || hipMalloc((void **)&h->d_counter, CSTRIDE * (NGRP + 1) * sizeof(unsigned int)) != hipSuccess ||
        hipMalloc((void **)&h->d_bad, sizeof(int)) != hipSuccess || hipHostMalloc((void **)&h->h_scal, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        memset_on(h->stream, h->d_counter, 0, CSTRIDE * (NGRP + 1) * sizeof(unsigned int)) != hipSuccess || memset_on(h->stream, h->d_scal, 0, 16 * sizeof(double)) != hipSuccess) {
        fail(NXS_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        nxs_krylov_destroy(h);
        return NXS_ERR_HIP;
    }
    *out = h;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_create"); }

void nxs_krylov_destroy(nxs_krylov_handle *h) try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm && h->rccl.CommDestroy) (void)h->rccl.CommDestroy(h->comm);
    free_matrix(h);
    free_halo(h);
    dfree(h->d_partial); dfree(h->d_scal); dfree(h->d_counter); dfree(h->d_bad);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
} catch (...) { (void)entry_caught("nxs_krylov_destroy"); }

int nxs_krylov_set_matrix(nxs_krylov_handle *h, int32_t n_rows, int32_t n_cols, const int32_t *rowptr, const int32_t *colidx, const double *val) try {
    if (!h || n_rows < 1 || n_cols < n_rows || !rowptr || !colidx || !val) return fail(NXS_ERR_INVALID, "NULL argument / empty system / n_cols < n_rows");
    if (rowptr[0] != 0) return fail(NXS_ERR_INVALID, "rowptr[0] must be 0");
    for (int i = 0; i < n_rows; ++i) {
        if (rowptr[i + 1] < rowptr[i]) return fail(NXS_ERR_INVALID, "rowptr not monotone at row %d", i);
        bool diag = false;
        for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            if (colidx[q] < 0 || colidx[q] >= n_cols) return fail(NXS_ERR_INVALID, "column index out of range in row %d", i);
            diag = diag || (colidx[q] == i && val[q] != 0.);
        }
        if (!diag) return fail(NXS_ERR_INVALID, "row %d has no diagonal entry (Jacobi preconditioner)", i);
    }
    KCHK(hipSetDevice(h->device));
    std::vector<int> off, col;
    std::vector<double> sval;
    if (!csr_to_sell(n_rows, rowptr, colidx, val, off, col, sval, nullptr)) return fail(NXS_ERR_INVALID, "matrix too large for 32-bit entry offsets");
    int rc = upload_matrix(h, n_rows, n_cols, off, col, sval, (size_t)rowptr[n_rows]);
    if (rc) return rc;
    return finish_matrix(h);
} catch (...) { return entry_caught("nxs_krylov_set_matrix"); }

int nxs_krylov_set_halo(nxs_krylov_handle *h, const nxs_dyn_halo *halo) try {
    if (!h || !halo) return fail(NXS_ERR_INVALID, "NULL argument");
    if (h->n < 1) return fail(NXS_ERR_STATE, "set_halo before set_matrix");
    if (halo->nranks < 1 || halo->rank < 0 || halo->rank >= halo->nranks || halo->num_send_procs < 0 || halo->num_recv_procs < 0)
        return fail(NXS_ERR_INVALID, "bad rank / neighbour counts");
    const int ns = halo->num_send_procs, nr = halo->num_recv_procs;
    if ((ns > 0 && (!halo->send_procs || !halo->send_offsets)) || (nr > 0 && (!halo->recv_procs || !halo->recv_offsets)))
        return fail(NXS_ERR_INVALID, "NULL halo list");
    const int ts = ns > 0 ? halo->send_offsets[ns] : 0, tr = nr > 0 ? halo->recv_offsets[nr] : 0;
    for (int k = 0; k < ns; ++k) if (halo->send_offsets[k + 1] < halo->send_offsets[k] || halo->send_procs[k] < 0 || halo->send_procs[k] >= halo->nranks) return fail(NXS_ERR_INVALID, "bad send list %d", k);
    for (int k = 0; k < nr; ++k) if (halo->recv_offsets[k + 1] < halo->recv_offsets[k] || halo->recv_procs[k] < 0 || halo->recv_procs[k] >= halo->nranks) return fail(NXS_ERR_INVALID, "bad recv list %d", k);
    for (int j = 0; j < ts; ++j) if (halo->send_index[j] < 0 || halo->send_index[j] >= h->n) return fail(NXS_ERR_INVALID, "send_index[%d] is not an owned row", j);
    for (int j = 0; j < tr; ++j) if (halo->recv_index[j] < h->n || halo->recv_index[j] >= h->n_local) return fail(NXS_ERR_INVALID, "recv_index[%d] is not a ghost column", j);
    KCHK(hipSetDevice(h->device));
    free_halo(h);
    h->rank = halo->rank; h->nranks = halo->nranks;
    h->send_procs.assign(halo->send_procs, halo->send_procs + ns);
    h->recv_procs.assign(halo->recv_procs, halo->recv_procs + nr);
    h->send_offsets.assign(1, 0); h->recv_offsets.assign(1, 0);
    if (ns > 0) h->send_offsets.assign(halo->send_offsets, halo->send_offsets + ns + 1);
    if (nr > 0) h->recv_offsets.assign(halo->recv_offsets, halo->recv_offsets + nr + 1);
    KCHK(hipMalloc((void **)&h->d_send_index, std::max(ts, 1) * sizeof(int)));
    KCHK(hipMalloc((void **)&h->d_recv_index, std::max(tr, 1) * sizeof(int)));
    KCHK(hipMalloc((void **)&h->d_send_buf, std::max(ts, 1) * sizeof(double)));
    KCHK(hipMalloc((void **)&h->d_recv_buf, std::max(tr, 1) * sizeof(double)));
    if (ts > 0) KCHK(copy_on(h->stream, h->d_send_index, halo->send_index, (size_t)ts * sizeof(int), hipMemcpyHostToDevice));
    if (tr > 0) KCHK(copy_on(h->stream, h->d_recv_index, halo->recv_index, (size_t)tr * sizeof(int), hipMemcpyHostToDevice));
    KCHK(hipHostMalloc((void **)&h->h_send, std::max(ts, 1) * sizeof(double), hipHostMallocDefault));
    KCHK(hipHostMalloc((void **)&h->h_recv, std::max(tr, 1) * sizeof(double), hipHostMallocDefault));
    h->have_halo = true;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_set_halo"); }

int nxs_krylov_comm_init(nxs_krylov_handle *h, const void *id128, int32_t rank, int32_t nranks) try {
    if (!h || !id128) return fail(NXS_ERR_INVALID, "NULL argument");
    KCHK(hipSetDevice(h->device));
    int rc = load_rccl(h->rccl);
    if (rc) return rc;
    NcclId id;
    std::memcpy(id.internal, id128, 128);
    nccl_comm_init_rank_t init = (nccl_comm_init_rank_t)h->rccl.CommInitRank;
    const int e = init(&h->comm, nranks, id, rank);
    if (e != 0) { h->comm = nullptr; return fail(NXS_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, h->rccl.GetErrorString(e)); }
    h->rank = rank; h->nranks = nranks;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_comm_init"); }

int nxs_krylov_comm_stats(const nxs_krylov_handle *h, int64_t *rccl_allreduces, int64_t *rccl_exchanges) try {
    if (!h) return fail(NXS_ERR_INVALID, "NULL argument");
    if (rccl_allreduces) *rccl_allreduces = h->n_rccl_allreduce;
    if (rccl_exchanges) *rccl_exchanges = h->n_rccl_exchange;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_comm_stats"); }

int nxs_krylov_set_comm_fns(nxs_krylov_handle *h, nxs_krylov_exchange_fn exchange_fn, nxs_krylov_allreduce_fn allreduce_fn, void *user) try {
    if (!h) return fail(NXS_ERR_INVALID, "NULL argument");
    if ((exchange_fn == nullptr) != (allreduce_fn == nullptr)) return fail(NXS_ERR_INVALID, "give both callbacks or neither");
    h->exchange_fn = exchange_fn; h->allreduce_fn = allreduce_fn; h->user = user;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_set_comm_fns"); }

int nxs_krylov_spmv(nxs_krylov_handle *h, const double *in, double *out, int32_t reps, double *ms_per_spmv) try {
    if (!h || !in || !out || reps < 1) return fail(NXS_ERR_INVALID, "NULL argument / reps < 1");
    if (h->n < 1) return fail(NXS_ERR_STATE, "spmv before set_matrix");
    KCHK(hipSetDevice(h->device));
    KCHK(hipMemcpyAsync(h->vec[0], in, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipEvent_t e0, e1;
    KCHK(hipEventCreate(&e0)); KCHK(hipEventCreate(&e1));
    int rc = exchange(h, h->vec[0]);  // (a warm-up SpMV when timing)
    if (!rc && reps > 1) spmv<0>(h, h->vec[0], h->vec[1], nullptr, 0);
    KCHK(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps && !rc; ++i) {
        if (i > 0) rc = exchange(h, h->vec[0]);
        spmv<0>(h, h->vec[0], h->vec[1], nullptr, 0);
    }
    KCHK(hipEventRecord(e1, h->stream));
    hipError_t err = hipStreamSynchronize(h->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (err != hipSuccess) return fail(NXS_ERR_HIP, "spmv failed: %s", hipGetErrorString(err));
    KCHK(copy_on(h->stream, out, h->vec[1], (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost));
    if (ms_per_spmv) *ms_per_spmv = ms / reps;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_spmv"); }

int nxs_krylov_run(nxs_krylov_handle *h, const double *b, double *x, int32_t method, double rtol, int32_t max_iter, int32_t *iterations,
                   double *rel_residual, double *ms_solve) try {
    if (!h || !b || !x) return fail(NXS_ERR_INVALID, "NULL argument");
    if (h->n < 1) return fail(NXS_ERR_STATE, "solve before set_matrix");
    if (method != NXS_KRYLOV_CG && method != NXS_KRYLOV_BICGSTAB) return fail(NXS_ERR_INVALID, "method must be NXS_KRYLOV_CG or NXS_KRYLOV_BICGSTAB");
    KCHK(hipSetDevice(h->device));
    for (auto &v : h->vec) KCHK(hipMemsetAsync(v, 0, (size_t)h->n_local * sizeof(double), h->stream));
    KCHK(hipMemcpyAsync(h->vec[1], b, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = run_solver(h, method, rtol, max_iter, iterations, rel_residual, ms_solve);
    if (rc) return rc;
    KCHK(copy_on(h->stream, x, h->vec[0], (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost));
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_run"); }

int nxs_krylov_info(const nxs_krylov_handle *h, int64_t *nnz, int64_t *stored_entries, int64_t *spmv_bytes) try {
    if (!h || h->n < 1) return fail(NXS_ERR_STATE, "no matrix");
    if (nnz) *nnz = (int64_t)h->nnz;
    if (stored_entries) *stored_entries = (int64_t)h->entries;
    if (spmv_bytes) *spmv_bytes = (int64_t)h->nnz * 12 + (int64_t)h->n * 16;  // value + column per entry, operand read + result write per row
    return NXS_OK;
} catch (...) { return entry_caught("nxs_krylov_info"); }

int nxs_krylov_solve(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *val, const double *b, double *x, int32_t method,
                     double rtol, int32_t max_iter, int32_t device, int32_t *iterations, double *rel_residual, double *ms_solve) try {
    if (n < 1 || !rowptr || !colidx || !val || !b || !x) return fail(NXS_ERR_INVALID, "NULL argument / empty system");
    if (method != NXS_KRYLOV_CG && method != NXS_KRYLOV_BICGSTAB) return fail(NXS_ERR_INVALID, "method must be NXS_KRYLOV_CG or NXS_KRYLOV_BICGSTAB");
    nxs_krylov_handle *h = nullptr;
    int rc = nxs_krylov_create(device, &h);
    if (rc) return rc;
    rc = nxs_krylov_set_matrix(h, n, n, rowptr, colidx, val);
    if (!rc) rc = nxs_krylov_run(h, b, x, method, rtol, max_iter, iterations, rel_residual, ms_solve);
    nxs_krylov_destroy(h);
    return rc;
} catch (...) { return entry_caught("nxs_krylov_solve"); }

int nxs_fem_poisson_solve(const int32_t *indices, const double *x, const double *y, int32_t Nn, int32_t Ne, const uint8_t *dirichlet,
                          const double *f_elem, double *u, double rtol, int32_t max_iter, int32_t device, int32_t *iterations,
                          double *rel_residual, double *ms_assembly, double *ms_solve) try {
    int rc = check_mesh(indices, Nn, Ne);
    if (rc) return rc;
    if (!x || !y || !dirichlet || !f_elem || !u) return fail(NXS_ERR_INVALID, "NULL argument");
    nxs_krylov_handle *h = nullptr;
    if ((rc = nxs_krylov_create(device, &h))) return rc;
    struct Guard { nxs_krylov_handle *h; ~Guard() { nxs_krylov_destroy(h); } } guard{h};

    std::vector<int> rp, ci, colour, off, col, where; int ncol = 0;
    std::vector<double> sval;
    build_pattern(indices, Nn, Ne, rp, ci);
    colour_elements(indices, Nn, Ne, colour, ncol);
    if (!csr_to_sell(Nn, rp.data(), ci.data(), nullptr, off, col, sval, &where)) return fail(NXS_ERR_INVALID, "matrix too large for 32-bit entry offsets");
    // patches of PROWS consecutive rows and, per patch, every element that touches one of its rows
    if (ncol > 255) return fail(NXS_ERR_INVALID, "more than 255 element colours");
    const int npatch = (Nn + PROWS - 1) / PROWS, nslices = (int)off.size() - 1;
    std::vector<int> pel_off(npatch + 1, 0);
    auto patches_of = [&](int e, int (&ps)[3]) {
        int n = 0;
        for (int k = 0; k < 3; ++k) {
            const int q = (indices[3 * e + k] - 1) / PROWS;
            bool dup = false;
            for (int j = 0; j < n; ++j) dup = dup || ps[j] == q;
            if (!dup) ps[n++] = q;
        }
        return n;
    };
    for (int e = 0; e < Ne; ++e) { int ps[3]; const int n = patches_of(e, ps); for (int j = 0; j < n; ++j) ++pel_off[ps[j] + 1]; }
    for (int q = 0; q < npatch; ++q) pel_off[q + 1] += pel_off[q];
    const size_t tot = (size_t)pel_off[npatch];
    std::vector<int> ptri(3 * tot);
    std::vector<double> pf(tot);
    std::vector<unsigned char> pcol(tot), lrow(3 * tot);
    std::vector<unsigned short> lpos(9 * tot);
    int Lmax = 1;
    for (int q = 0; q < npatch; ++q) Lmax = std::max(Lmax, off[std::min(2 * q + 2, nslices)] - off[2 * q]);
    if (Lmax > 0xFFFF) return fail(NXS_ERR_INVALID, "rows too long for the patch assembly");
    {
        std::vector<int> fill(pel_off.begin(), pel_off.end() - 1);
        for (int e = 0; e < Ne; ++e) {  // ascending element order inside every patch
            int ps[3];
            const int n = patches_of(e, ps);
            const int nd[3] = {indices[3 * e] - 1, indices[3 * e + 1] - 1, indices[3 * e + 2] - 1};
            for (int a = 0; a < n; ++a) {
                const int q = ps[a];
                const size_t i = (size_t)fill[q]++;
                pf[i] = f_elem[e];
                pcol[i] = (unsigned char)colour[e];
                for (int j = 0; j < 3; ++j) {
                    ptri[(size_t)j * tot + i] = nd[j];
                    const bool mine = nd[j] / PROWS == q;
                    lrow[(size_t)j * tot + i] = mine ? (unsigned char)(nd[j] - q * PROWS) : (unsigned char)0xFF;
                    for (int k = 0; k < 3; ++k) {
                        unsigned short v = 0xFFFF;
                        if (mine) {
                            const int *b = ci.data() + rp[nd[j]], *en = ci.data() + rp[nd[j] + 1];
                            v = (unsigned short)(where[std::lower_bound(b, en, nd[k]) - ci.data()] - off[2 * q]);
                        }
                        lpos[(size_t)(3 * j + k) * tot + i] = v;
                    }
                }
            }
        }
    }
    if ((rc = upload_matrix(h, Nn, Nn, off, col, sval, ci.size()))) return rc;

    struct Tmp { void *p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } };
    Tmp dpeo, dptri, dpf, dpcol, dlpos, dlrow, ddir, dx, dy;
    auto up = [&](Tmp &t, const void *src, size_t bytes) { return hipMalloc(&t.p, std::max<size_t>(bytes, 1)) == hipSuccess && (bytes == 0 || copy_on(h->stream, t.p, src, bytes, hipMemcpyHostToDevice) == hipSuccess); };
    if (!up(dpeo, pel_off.data(), pel_off.size() * sizeof(int)) || !up(dptri, ptri.data(), ptri.size() * sizeof(int)) || !up(dpf, pf.data(), pf.size() * sizeof(double)) ||
        !up(dpcol, pcol.data(), pcol.size()) || !up(dlpos, lpos.data(), lpos.size() * sizeof(unsigned short)) || !up(dlrow, lrow.data(), lrow.size()) ||
        !up(ddir, dirichlet, Nn) || !up(dx, x, Nn * sizeof(double)) || !up(dy, y, Nn * sizeof(double)))
        return fail(NXS_ERR_HIP, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));

    double *rhs = h->vec[1];
    const size_t lds = (size_t)(Lmax + PROWS) * sizeof(double);
    hipEvent_t e0, e1;
    KCHK(hipEventCreate(&e0)); KCHK(hipEventCreate(&e1));
    // the assembly is run twice and the second pass is the one timed (the first one also pays for first-touch page faults)
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) KCHK(hipEventRecord(e0, h->stream));
        hipLaunchKernelGGL(k_assemble_patches, dim3(npatch), dim3(PT), lds, h->stream, Nn, nslices, ncol, (const int *)h->d_off, (const int *)dpeo.p, (const int *)dptri.p,
                           (const double *)dpf.p, (const unsigned char *)dpcol.p, (const unsigned short *)dlpos.p, (const unsigned char *)dlrow.p, tot, (const double *)dx.p,
                           (const double *)dy.p, Lmax, h->d_val, rhs);
        hipLaunchKernelGGL(k_apply_dirichlet, dim3((Nn + BS - 1) / BS), dim3(BS), 0, h->stream, Nn, (const int *)h->d_off, (const int *)h->d_col, (const unsigned char *)ddir.p, h->d_val, rhs);
        if (pass == 1) KCHK(hipEventRecord(e1, h->stream));
    }
    KCHK(hipStreamSynchronize(h->stream));
    float msa = 0.f;
    (void)hipEventElapsedTime(&msa, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if ((rc = finish_matrix(h))) return rc;
    if ((rc = run_solver(h, NXS_KRYLOV_CG, rtol, max_iter, iterations, rel_residual, ms_solve))) return rc;
    KCHK(copy_on(h->stream, u, h->vec[0], (size_t)Nn * sizeof(double), hipMemcpyDeviceToHost));
    if (ms_assembly) *ms_assembly = msa;
    return NXS_OK;
} catch (...) { return entry_caught("nxs_fem_poisson_solve"); }

}  // extern "C"

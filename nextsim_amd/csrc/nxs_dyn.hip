// nxs_dyn.hip -- libnxsdyn.so: neXtSIM's per-time-step dynamics (explicitSolve + update) on MI355X.
//
// gfx950 only.  fp64 throughout; memory-bound (no MFMA).  One handle = one GPU = one HIP stream.
// Compiled with -ffp-contract=off: the reference is built without FMA contraction
// (model/Makefile:5-8, -O3 and no -march), so every a*b+c below is two roundings, like there.
//
// Kernel inventory ("v1": one kernel per reference loop; FE.cpp = model/finiteelement.cpp):
//   k_prep_elements   FE.cpp:10235-10308  geometry, slab mass, basal C_bu, per-step element constants
//   k_prep_nodes      FE.cpp:10309-10416  node-centric gather of the element->node scatters + prep nodes
//   k_sigma_bbm       FE.cpp:4137-4260    updateSigmaDamage; also emits the 6 corner forces of K4
//   k_sigma_vp        FE.cpp:10649-10726  EVP / mEVP stress
//   k_solve_move      FE.cpp:10445-10553  grad_terms gather + nodal solve + mesh move (owned nodes)
//   k_halo_pack/unpack FE.cpp:13963-13996 updateGhosts (unpack also moves the ghost nodes)
//   k_smooth          FE.cpp:10580-10608  one Jacobi sweep of the open-water smoother
//   k_ow_tail         FE.cpp:10613-10640  D_tau_w, open-water mesh move
//   k_update          FE.cpp:3946-4131    update()
//   k_free_drift      FE.cpp:10140-10176
//   k_regrid_*/k_check_* FE.cpp:8298-8309, 14536-14655 reductions
//
// Determinism: the reference scatters element contributions into nodes in ascending element
// order.  Here every node GATHERS over its element fan sorted ascending, which performs the same
// floating-point additions in the same order: no atomics, bit-reproducible, and identical to the
// serial loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "nxs_dyn.h"

// ------------------------------------------------------------------------------------------------
// constants (model/constants.hpp:56-87, model/finiteelement.hpp:549)
#define NXS_RHOI 917.
#define NXS_RHOW 1025.
#define NXS_RHOS 330.
#define NXS_RHOA 1.22
#define NXS_GRAVITY 9.80616
#define NXS_OMEGA 7.292e-5
#define NXS_PI 3.141592653589793238462643383279502884197169399375105820974944592308
#define NXS_DAYS_IN_SEC 86400.

#define STD_MAX(a, b) (((a) < (b)) ? (b) : (a))  // std::max
#define STD_MIN(a, b) (((b) < (a)) ? (b) : (a))  // std::min

static constexpr int BLOCK = 256;

// node flag bits
#define NF_DIRICHLET 1
#define NF_NEUMANN 2
#define NF_LAT_NEG 4  // signbit(lat): all the solve needs of lat is copysign(sin_theta, lat)
// element flag bits (static, per mesh)
#define EF_ON_NEUMANN 8  // any vertex in M_neumann_flags (FE.cpp:3957-3961)

// everything the kernels need from nxs_dyn_params, plus host-precomputed scalars
struct DevParams {
    double dtime_step, dte;
    int substeps, dynamics_type, basal_stress_type, young_cat, newice_type, equal_ridging, use_young_myi;
    double young, nu0, tan_phi, compr_strength, compaction_param, utrs, ers_m1, compression_factor, ecf;
    double min_h, min_c, min_m, qdw, ldw, qda, lda;
    double cos_ota, sin_ota;
    double k1, k2, Cb, u0;
    double evp_e, evp_Pstar, evp_C, evp_dmin, ralpha1, ralpha2, mevp_beta;
    double sqrt_nu_rhoi;
    double D[9];
    int ers_int;  // exponent_relaxation_sigma - 1 when it is an integer in [0,16], else -1
};

struct DevMesh {
    int Nn, Ne, No, Neo;
    const int *t0, *t1, *t2;     // [Ne] 0-based node ids
    const unsigned char *eflags; // [Ne] bits 0-2 ghostNodes[k], bit 3 on-neumann
    const double *x0, *y0, *lat; // [Nn]
    const unsigned char *nflags; // [Nn]
    int W;  const int *fan;      // [W][Nn] ascending fan: (e<<3)|(ghost<<2)|corner, -1 pad
    int W1; const int *n2e;      // [W1][Nn] bamg row order, 0-based element, -1 pad
    int W2; const int *n2n;      // [W2][Nn] bamg row order, 0-based node
    const int *n2n_cnt;          // [Nn]
};

struct DevState {
    double *VT, *VT2, *UM, *UT;  // VT2: second buffer (Jacobi smoother; ping-pong of the fused sub-step)
    double *conc, *thick, *snow, *damage, *ridge, *s0, *s1, *s2;
    double *damage_b, *s0_b, *s1_b, *s2_b;  // ping-pong partners of damage, sigma (fused sub-step)
    double *cyoung, *hyoung, *hsyoung, *cmyi, *tmyi;
    double *cohesion, *theal, *drag_ui, *drag_ui_young;
    double *wind, *ocean, *ssh, *depth;
};

// v2: node patches.  One workgroup owns up to Pmax nodes ("own" nodes) and processes every element
// that touches one of them; elements shared with a neighbouring patch are recomputed by both (like the
// MPI ghost layer, one level down) and written by exactly one.  Element->node traffic stays in LDS.
struct DevPatches {
    int nP, Pmax, Emax, Mmax, Wp;
    const int *own_cnt, *elem_cnt, *node_cnt;  // [nP]
    const int *pnodes;            // [nP][Mmax] global node id of each patch-local node slot (own nodes first)
    const int *pelem;             // [nP][Emax] global element id, ascending; ~id when another patch writes it
    const unsigned short *ptri;   // [nP][Emax][4] patch-local node slots of the 3 corners (+ pad)
    const unsigned short *pfan;   // [nP][Wp][Pmax] (element slot << 3 | ghost << 2 | corner), 0xFFFF pad
};

// Patches of the several-sub-steps-per-launch kernel (k_substep_multi): D rings of halo around the own nodes.
//   nodes    N_0 = own | N_1 \ N_0 | ... | N_D \ N_(D-1)     N_i = the nodes of the elements E_i
//   elements E_1 | E_2 \ E_1 | ... | E_D \ E_(D-1)           E_i = every element touching a node of N_(i-1), ascending id inside a level
// sub-step k of a launch (k = 0 .. D-1) updates the elements E_(D-k) and solves the nodes N_(D-k-1).
#define NXS_MAX_DEPTH 8
struct DevPatches2 {
    int nP, D, NDmax /*nodes staged*/, NSmax /*nodes ever solved = N_(D-1)*/, EDmax /*elements of sub-step 0*/, ESmax /*elements needed again = E_(D-1)*/, Wp;
    const int *ncnt;              // [nP][D+1] |N_0| .. |N_D|
    const int *ecnt;              // [nP][D]   |E_1| .. |E_D|
    const int *pnodes;            // [nP][NDmax] global node ids
    const int *pelem;             // [nP][EDmax] global element id; ~id when this patch does not write it
    const unsigned short *ptri;   // [nP][EDmax][4] patch-local node slots of the 3 corners (+ pad)
    const unsigned short *pfan;   // [nP][Wp][NSmax] fan of every solved node, ascending element id: (element slot << 3 | ghost << 2 | corner)
};
struct VTOut { double *slot[NXS_MAX_DEPTH]; };  // ring slots of the D velocities a launch produces

struct PingPong {  // buffers a fused sub-step reads (c) and writes (n)
    const double *VTc, *s0c, *s1c, *s2c, *dc;
    double *VTn, *s0n, *s1n, *s2n, *dn;
};

struct DevWork {
    double *delta_x, *surface, *shape /*[6][Ne]*/, *emass, *ecbu;
    double *prec /*[Ne][10]: what k_prep_nodes gathers per fan entry, one record per element (see k_prep_elements)*/;
    double *expC, *pmax, *heal, *dxs, *volume;  // per-step element constants of the sub-step loop
    unsigned char *eskip;                        // conc <= 0.1 (BBM) / thick == 0 (EVP)
    unsigned char *open_blk;                     // [ceil(Nn/BLOCK)] != 0: the block of BLOCK nodes holds a node the open-water smoother changes (zeroed by k_prep_elements, set by k_prep_nodes)
    int *dxi;                                    // BBM, fused kernel: M_delta_x as the integer it is (Q1), ~M_delta_x when the element is skipped
    double *force /*[6][Ne]: fx0,fx1,fx2,fy0,fy1,fy2*/;
    double *rlmass, *node_mass, *C_bu, *grad_ssh /*[2Nn]*/, *fcor, *VTM /*[2Nn]*/;
    double *xs, *ys;  // [Nn] node coordinates on the displaced mesh at step start (frozen over the sub-steps, Q4)
    double *D_tau_a, *D_tau_w, *D_del;
};

// ------------------------------------------------------------------------------------------------
// device helpers

__device__ __forceinline__ void load_vertices(const DevMesh &m, const double *__restrict__ UM, int e,
                                              double vx[3], double vy[3]) {
    // GmshMesh::vertices(indices, um, 1.), gmshmesh.cpp:1929-1939
    const int n[3] = {m.t0[e], m.t1[e], m.t2[e]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vx[i] = m.x0[n[i]] + 1. * UM[n[i]];
        vy[i] = m.y0[n[i]] + 1. * UM[n[i] + m.Nn];
    }
}

__device__ __forceinline__ double jacobian(const double vx[3], const double vy[3]) {  // FE.cpp:1613-1618
    double jac = (vx[1] - vx[0]) * (vy[2] - vy[0]);
    jac -= (vx[2] - vx[0]) * (vy[1] - vy[0]);
    return jac;
}

// ------------------------------------------------------------------------------------------------
// K1a  prep elements, FE.cpp:10235-10308
__global__ void __launch_bounds__(BLOCK) k_prep_elements(DevMesh m, DevState s, DevWork w, DevParams p) {
    // threads past the end redo the last element (identical values to identical places): every thread reaches the barrier
    const int e = min(blockIdx.x * BLOCK + (int)threadIdx.x, m.Ne - 1);
    for (int i = blockIdx.x * BLOCK + (int)threadIdx.x; i < (m.Nn + BLOCK - 1) / BLOCK; i += gridDim.x * BLOCK) w.open_blk[i] = 0;  // k_prep_nodes raises them
    double vx[3], vy[3];
    load_vertices(m, s.UM, e, vx, vy);

    // Q1 (FE.cpp:10239): int accumulator, then unsigned integer division by 3
    const double side0 = hypot(vx[1] - vx[0], vy[1] - vy[0]);
    const double side1 = hypot(vx[2] - vx[1], vy[2] - vy[1]);
    const double side2 = hypot(vx[2] - vx[0], vy[2] - vy[0]);
    int acc = 0;
    acc = (int)(acc + side0);
    acc = (int)(acc + side1);
    acc = (int)(acc + side2);
    const int acc_div3 = (int)((unsigned long)acc / 3ul);
    const double delta_x = (double)((unsigned long)acc / 3ul);
    w.delta_x[e] = delta_x;

    const double jac = jacobian(vx, vy);
    const double surface = (1. / 2) * fabs(jac);  // FE.cpp:1929-1933
    w.surface[e] = surface;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // FE.cpp:1956-1962
        const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
        w.shape[(size_t)k * m.Ne + e] = (vy[kp1] - vy[kp2]) / jac;
        w.shape[(size_t)(k + 3) * m.Ne + e] = (vx[kp2] - vx[kp1]) / jac;
    }

    // slab mass, FE.cpp:10255-10269
    const double conc = s.conc[e], thick = s.thick[e];
    double total_concentration = conc, total_thickness = thick, total_snow = s.snow[e];
    if (p.young_cat) {
        total_concentration += s.cyoung[e];
        total_thickness += s.hyoung[e];
        total_snow += s.hsyoung[e];
    }
    double element_mass = 0.;
    if (total_concentration > 0.)
        element_mass = (NXS_RHOI * total_thickness + NXS_RHOS * total_snow) / total_concentration;
    w.emass[e] = element_mass;

    // basal stress numerator, FE.cpp:10273-10308
    double element_ssh = 0;
    element_ssh += s.ssh[m.t0[e]];
    element_ssh += s.ssh[m.t1[e]];
    element_ssh += s.ssh[m.t2[e]];
    element_ssh /= 3.;
    const double max_keel_depth = 28;
    const double min_water_depth = 2.;
    const double depth_eff = STD_MAX(0., element_ssh + STD_MAX(min_water_depth, s.depth[e]));
    double critical_h = 0., critical_h_mod = 0.;
    if (p.basal_stress_type == NXS_BASAL_LEMIEUX) {
        double mean_keel_depth = p.k1 * thick;
        mean_keel_depth = STD_MIN(mean_keel_depth, conc * max_keel_depth);
        critical_h = conc * depth_eff / p.k1;
        critical_h_mod = mean_keel_depth / p.k1;
    }
    const double ecbu = p.k2 * STD_MAX(0., critical_h_mod - critical_h) * exp(-p.Cb * (1. - conc));
    w.ecbu[e] = ecbu;

    // The record k_prep_nodes gathers for every fan entry -- everything the nodal loops of FE.cpp:10309-10340 and
    // 10578-10602 take from this element, contiguous (80 B) instead of nine arrays: the products are formed with the
    // reference's operand order, so the node side performs the same additions on the same values.
    __shared__ double rec[BLOCK * 10];  // staged so that the 80-byte records leave the block as one contiguous stream
    {
        double *r = rec + threadIdx.x * 10;
        const double meA = element_mass * surface;           // node_mass += element_mass*surface, FE.cpp:10314
        const double m_g_A3rd = meA * (NXS_GRAVITY / 3.);    // FE.cpp:10321
        double dragp = s.drag_ui[e];                          // FE.cpp:10585-10596
        if (p.young_cat) {
            const double cy = s.cyoung[e];
            if (conc + cy > 0.) dragp = (s.drag_ui[e] * conc + s.drag_ui_young[e] * cy) / (conc + cy);
        }
        const double sshn[3] = {s.ssh[m.t0[e]], s.ssh[m.t1[e]], s.ssh[m.t2[e]]};
        r[0] = surface; r[1] = meA; r[2] = ecbu; r[3] = dragp * surface;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                         // FE.cpp:10334-10339
            const int kp1 = (j + 1) % 3, kp2 = (j + 2) % 3;
            r[4 + j] = (vy[kp1] - vy[kp2]) / jac * m_g_A3rd * sshn[j];
            r[7 + j] = (vx[kp2] - vx[kp1]) / jac * m_g_A3rd * sshn[j];
        }
    }
    __syncthreads();
    {
        const size_t base = (size_t)blockIdx.x * BLOCK * 10;
        const int count = min(BLOCK, m.Ne - (int)blockIdx.x * BLOCK) * 10;
        for (int i = threadIdx.x; i < count; i += BLOCK) w.prec[base + i] = rec[i];
    }

    // Per-step constants of the sub-step loop.  M_conc, M_thick, M_delta_x, M_surface do not change
    // while sub-cycling (Q4), so exp/pow of them are evaluated once here instead of S times; the
    // expressions are the reference's, operand for operand.
    if (p.dynamics_type == NXS_DYN_BBM) {
        const double expC = exp(p.compaction_param * (1. - conc));             // FE.cpp:4185
        w.expC[e] = expC;
        w.pmax[e] = pow(thick, p.ecf) * p.compression_factor * expC;          // FE.cpp:4192
        w.heal[e] = p.dte / s.theal[e] * expC;                                 // FE.cpp:4257
        w.dxs[e] = delta_x * p.sqrt_nu_rhoi;                                   // FE.cpp:4232
        w.eskip[e] = (conc <= 0.1) ? 1 : 0;                                    // Q5, FE.cpp:4146-4151
        w.dxi[e] = (conc <= 0.1) ? ~acc_div3 : acc_div3;                       // 4 bytes instead of 9 for the fused kernel
    } else {
        w.expC[e] = p.evp_Pstar * exp(-p.evp_C * (1. - conc));                 // FE.cpp:10684 (P)
        w.eskip[e] = (thick == 0.) ? 1 : 0;                                    // FE.cpp:10656
    }
    w.volume[e] = thick * surface;                                             // FE.cpp:10450
}

// ------------------------------------------------------------------------------------------------
// K1b + K2  the nodal side of prep elements (as a gather) and prep nodes, FE.cpp:10309-10416
__global__ void __launch_bounds__(BLOCK) k_prep_nodes(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.Nn) return;
    const int Nn = m.Nn;
    const bool dirichlet = m.nflags[n] & NF_DIRICHLET;

    double rl = 0., nm = 0., cb = 0., gu = 0., gv = 0.;
    for (int slot = 0; slot < m.W; ++slot) {
        const int ent = m.fan[(size_t)slot * Nn + n];
        if (ent < 0) break;
        const double *r = w.prec + (size_t)(ent >> 3) * 10;
        const bool ghost_corner = ent & 4;
        rl += r[0];                                    // FE.cpp:10313
        nm += r[1];                                    // FE.cpp:10314
        cb = STD_MAX(cb, r[2]);                        // FE.cpp:10317
        // Q7: the skip test sees node_mass as accumulated so far (elements <= e)
        if (dirichlet || nm == 0. || ghost_corner) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                  // FE.cpp:10334-10339
            gu -= r[4 + j];
            gv -= r[7 + j];
        }
    }
    // same expression as load_vertices(): the fused sub-step kernel rebuilds the shape coefficients from these
    w.xs[n] = m.x0[n] + 1. * s.UM[n];
    w.ys[n] = m.y0[n] + 1. * s.UM[n + Nn];
    w.C_bu[n] = cb;
    w.grad_ssh[n] = gu;
    w.grad_ssh[n + Nn] = gv;

    // prep nodes, FE.cpp:10356-10416
    double vu = s.VT[n], vv = s.VT[n + Nn];
    if (nm == 0.) { vu = 0.; vv = 0.; s.VT[n] = 0.; s.VT[n + Nn] = 0.; }

    double drag = 0., surface = 0;
    for (int j = 0; j < m.W1; ++j) {                  // bamg row order (summation order!)
        const int e = m.n2e[(size_t)j * Nn + n];
        if (e < 0) continue;                           // Q2
        const double *r = w.prec + (size_t)e * 10;
        drag += r[3];                                  // dragp * surface
        surface += r[0];
    }
    const double wu = s.wind[n], wv = s.wind[n + Nn];
    drag *= NXS_RHOA * hypot(wu, wv) / surface;        // Q6
    w.D_tau_a[n] = drag * wu;
    w.D_tau_a[n + Nn] = drag * wv;

    w.fcor[n] = 2 * NXS_OMEGA * sin(m.lat[n] * NXS_PI / 180.);

    rl = 1. / rl;                                      // FE.cpp:10400-10402
    nm *= rl;
    rl *= 3.;
    w.rlmass[n] = rl;
    w.node_mass[n] = nm;
    if (n < m.No && !dirichlet && nm == 0.) w.open_blk[n / BLOCK] = 1;  // k_smooth's own test (FE.cpp:10589): its other blocks have nothing to do

    w.VTM[n] = vu;
    w.VTM[n + Nn] = vv;
}

// ------------------------------------------------------------------------------------------------
// Arithmetic shared by the v1 kernels (one per reference loop) and the v2 fused sub-step kernel, so
// that both perform literally the same operations in the same order.

// updateSigmaDamage body for one element, FE.cpp:4161-4257 (the conc <= 0.1 early-out is the caller's)
template <bool POW4>
__device__ __forceinline__ void bbm_stress(const DevParams &p, const double dxN[6], const double u[3], const double v[3],
                                           double sig[3], double &damage, const double expC, const double Pmax,
                                           const double heal, const double dxs, const double cohesion) {
    const double dt = p.dte;
    // M_B0T (FE.cpp:10242-10249) rebuilt in registers, zeros included so that the sums below are the
    // reference's term for term (FE.cpp:4167-4176)
    double B0T[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) B0T[i] = 0.;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        B0T[2 * i] = dxN[i];
        B0T[2 * i + 13] = dxN[i];
        B0T[2 * i + 7] = dxN[i + 3];
        B0T[2 * i + 12] = dxN[i + 3];
    }
    double eps[3] = {0., 0., 0.};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            eps[i] += B0T[i * 6 + 2 * j] * u[j];
            eps[i] += B0T[i * 6 + 2 * j + 1] * v[j];
        }
    double sigma_n = (sig[0] + sig[1]) * 0.5;                                    // FE.cpp:4184
    // FE.cpp:4186: std::pow(x, exponent_relaxation_sigma - 1.).  For the default exponent (5 - 1 = 4) the
    // power is formed by two squarings: <= 1.5 ulp from the correctly rounded value, i.e. inside the error
    // band of any libm pow, and it removes ~400 instructions and ~50 VGPRs from the hot loop.  Any other
    // exponent takes the general pow (POW4 == false).
    const double pw_x = (1. - damage) * expC;
    double pw;
    if (POW4) { const double x2 = pw_x * pw_x; pw = x2 * x2; }
    else pw = pow(pw_x, p.ers_m1);
    const double time_viscous = p.utrs * pw;
    double tildeP;
    if (sigma_n < 0.) {
        tildeP = STD_MIN(1., -Pmax / sigma_n);                                   // FE.cpp:4194
    } else {
        tildeP = 0.;
    }
    const double multiplicator = STD_MIN(1. - 1e-12, time_viscous / (time_viscous + dt * (1. - tildeP)));  // Q3
    const double elasticity = p.young * (1. - damage) * expC;                    // FE.cpp:4202
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                                // FE.cpp:4204-4210
#pragma unroll
        for (int j = 0; j < 3; ++j) sig[i] += dt * elasticity * p.D[3 * i + j] * eps[j];
        sig[i] *= multiplicator;
    }
    const double sigma_s = hypot((sig[0] - sig[1]) / 2., sig[2]);                // FE.cpp:4218
    sigma_n = (sig[0] + sig[1]) * 0.5;
    double dcrit;
    if (sigma_n < -p.compr_strength)
        dcrit = -p.compr_strength / sigma_n;
    else
        dcrit = cohesion / (sigma_s + p.tan_phi * sigma_n);
    if ((0. < dcrit) && (dcrit < 1.)) {                                          // FE.cpp:4229-4243
        const double rtd = sqrt(elasticity) / dxs;
        const double del_damage = (1.0 - damage) * (1.0 - dcrit) * dt * rtd;
        damage += del_damage;
#pragma unroll
        for (int i = 0; i < 3; ++i) sig[i] -= sig[i] * (1. - dcrit) * dt * rtd;
    }
    damage = STD_MAX(0., damage - heal);                                         // FE.cpp:4256
}

// updateSigmaVP body for one element, FE.cpp:10664-10696 (P = Pstar*exp(-C(1-A)) precomputed)
__device__ __forceinline__ void vp_stress(const DevParams &p, const double dxN[6], const double u[3], const double v[3],
                                          double sig[3], const double P) {
    const double re2 = 1. / (p.evp_e * p.evp_e);
    double eps11 = 0., eps22 = 0., eps12 = 0.;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        eps11 += dxN[i] * u[i];
        eps22 += dxN[i + 3] * v[i];
        eps12 += 0.5 * (dxN[i] * v[i] + dxN[i + 3] * u[i]);
    }
    const double eps1 = eps11 + eps22, eps2 = eps11 - eps22;
    const double delta = sqrt(eps1 * eps1 + (eps2 * eps2 + 4 * eps12 * eps12) * re2);
    const double zeta = P / (delta + p.evp_dmin);
    double sigma1 = sig[0] + sig[1], sigma2 = sig[0] - sig[1];
    sigma1 += p.ralpha1 * (zeta * (eps1 - delta) - sigma1);
    sigma2 += p.ralpha2 * (zeta * eps2 * re2 - sigma2);
    sig[2] += p.ralpha2 * (zeta * eps12 * re2 - sig[2]);
    sig[0] = 0.5 * (sigma1 + sigma2);
    sig[1] = 0.5 * (sigma1 - sigma2);
}

// element half of "gradient sigma" (FE.cpp:10449-10465): the term corner i subtracts from its node
__device__ __forceinline__ void corner_forces(const double volume, const double sig[3], const double dxN[6], double F[6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F[i] = volume * (sig[0] * dxN[i] + sig[2] * dxN[i + 3]);
        F[i + 3] = volume * (sig[2] * dxN[i] + sig[1] * dxN[i + 3]);
    }
}

// "sub-solve" for one node, FE.cpp:10481-10528: (uice, vice) in -> new velocity out
__device__ __forceinline__ void nodal_solve(const DevParams &p, const double gx, const double gy, double &uice, double &vice,
                                            const double node_mass, const double rlm, const double C_bu, const double fcor,
                                            const double lat, const double tau_ax, const double tau_ay, const double ou,
                                            const double ov, const double vtm_u, const double vtm_v) {
    double dtep, delu, delv;
    if (p.dynamics_type == NXS_DYN_MEVP) {  // FE.cpp:10483-10493
        const double b_mevp = p.mevp_beta + 1.;
        delu = (vtm_u - uice) / b_mevp;
        delv = (vtm_v - vice) / b_mevp;
        dtep = p.dte / b_mevp;
    } else {
        delu = 0.; delv = 0.; dtep = p.dte;
    }
    const double dte_over_mass = dtep / STD_MAX(p.min_m, node_mass);
    const double c_prime = NXS_RHOW * p.qdw * hypot(ou - uice, ov - vice);
    const double tau_b = C_bu / (hypot(uice, vice) + p.u0);
    const double alpha = 1. + dte_over_mass * (c_prime * p.cos_ota + tau_b);
    const double beta = dtep * fcor + dte_over_mass * c_prime * copysign(p.sin_ota, lat);
    const double rdenom = 1. / (alpha * alpha + beta * beta);
    const double tau_x = tau_ax + c_prime * (ou * p.cos_ota - ov * copysign(p.sin_ota, lat));
    const double tau_y = tau_ay + c_prime * (ov * p.cos_ota + ou * copysign(p.sin_ota, lat));
    const double grad_x = gx * rlm, grad_y = gy * rlm;
    double nu_ = alpha * uice + beta * vice + dte_over_mass * (alpha * (grad_x + tau_x) + beta * (grad_y + tau_y)) + alpha * delu + beta * delv;
    nu_ *= rdenom;
    double nv_ = alpha * vice - beta * uice + dte_over_mass * (alpha * (grad_y + tau_y) - beta * (grad_x + tau_x)) + alpha * delv - beta * delu;
    nv_ *= rdenom;
    uice = nu_;
    vice = nv_;
}

// ------------------------------------------------------------------------------------------------
// K3a  updateSigmaDamage, FE.cpp:4137-4260, + the element half of K4 (corner forces)
template <bool POW4>
__global__ void __launch_bounds__(BLOCK) k_sigma_bbm(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    const int Ne = m.Ne, Nn = m.Nn;
    double dxN[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dxN[k] = w.shape[(size_t)k * Ne + e];
    double sig[3];
    if (w.eskip[e]) {  // FE.cpp:4151-4159
        s.damage[e] = 0.;
        sig[0] = sig[1] = sig[2] = 0.;
    } else {
        const int n[3] = {m.t0[e], m.t1[e], m.t2[e]};
        double u[3], v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { u[j] = s.VT[n[j]]; v[j] = s.VT[n[j] + Nn]; }
        sig[0] = s.s0[e]; sig[1] = s.s1[e]; sig[2] = s.s2[e];
        double damage = s.damage[e];
        bbm_stress<POW4>(p, dxN, u, v, sig, damage, w.expC[e], w.pmax[e], w.heal[e], w.dxs[e], s.cohesion[e]);
        s.damage[e] = damage;
    }
    s.s0[e] = sig[0]; s.s1[e] = sig[1]; s.s2[e] = sig[2];
    double F[6];
    corner_forces(w.volume[e], sig, dxN, F);
#pragma unroll
    for (int i = 0; i < 6; ++i) w.force[(size_t)i * Ne + e] = F[i];
}

// K3b  updateSigmaVP, FE.cpp:10649-10699
__global__ void __launch_bounds__(BLOCK) k_sigma_vp(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    const int Ne = m.Ne, Nn = m.Nn;
    double dxN[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dxN[k] = w.shape[(size_t)k * Ne + e];
    double sig[3];
    if (w.eskip[e]) {
        sig[0] = sig[1] = sig[2] = 0.;
    } else {
        const int n[3] = {m.t0[e], m.t1[e], m.t2[e]};
        double u[3], v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { u[j] = s.VT[n[j]]; v[j] = s.VT[n[j] + Nn]; }
        sig[0] = s.s0[e]; sig[1] = s.s1[e]; sig[2] = s.s2[e];
        vp_stress(p, dxN, u, v, sig, w.expC[e]);
    }
    s.s0[e] = sig[0]; s.s1[e] = sig[1]; s.s2[e] = sig[2];
    double F[6];
    corner_forces(w.volume[e], sig, dxN, F);
#pragma unroll
    for (int i = 0; i < 6; ++i) w.force[(size_t)i * Ne + e] = F[i];
}

// ------------------------------------------------------------------------------------------------
// K4 (node half) + K5 + K7 for owned nodes, FE.cpp:10445-10553
// move_dt == 0 -> no mesh move here (mEVP moves once after the loop, FE.cpp:10559-10573)
__global__ void __launch_bounds__(BLOCK) k_solve_move(DevMesh m, DevState s, DevWork w, DevParams p, double move_dt) {
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.No) return;
    const int Nn = m.Nn, Ne = m.Ne;
    const unsigned char nf = m.nflags[n];
    const double node_mass = w.node_mass[n];
    double uice = s.VT[n], vice = s.VT[n + Nn];

    if (!((nf & NF_DIRICHLET) || node_mass == 0.)) {
        // grad_terms = grad_ssh, then minus the corner forces of the fan in ascending element order
        double gx = w.grad_ssh[n], gy = w.grad_ssh[n + Nn];
        for (int slot = 0; slot < m.W; ++slot) {
            const int ent = m.fan[(size_t)slot * Nn + n];
            if (ent < 0) break;
            if (ent & 4) continue;  // ghostNodes[i] (FE.cpp:10456)
            const int e = ent >> 3, c = ent & 3;
            gx -= w.force[(size_t)c * Ne + e];
            gy -= w.force[(size_t)(c + 3) * Ne + e];
        }
        nodal_solve(p, gx, gy, uice, vice, node_mass, w.rlmass[n], w.C_bu[n], w.fcor[n], m.lat[n], w.D_tau_a[n],
                    w.D_tau_a[n + Nn], s.ocean[n], s.ocean[n + Nn], w.VTM[n], w.VTM[n + Nn]);
        s.VT[n] = uice;
        s.VT[n + Nn] = vice;
    }
    if (move_dt != 0.) {  // FE.cpp:10543-10550; Neumann nodes keep M_UM (restore == skip)
        if (!(nf & NF_NEUMANN)) {
            s.UM[n] += move_dt * uice;
            s.UM[n + Nn] += move_dt * vice;
        }
        s.UT[n] += move_dt * uice;
        s.UT[n + Nn] += move_dt * vice;
    }
}

// mesh move for a node range (mEVP end-of-loop move; ghosts when there is no halo kernel)
__global__ void __launch_bounds__(BLOCK) k_move(DevMesh m, DevState s, int first, int last, double dt) {
    const int n = first + blockIdx.x * BLOCK + threadIdx.x;
    if (n >= last) return;
    const int Nn = m.Nn;
    const double u = s.VT[n], v = s.VT[n + Nn];
    if (!(m.nflags[n] & NF_NEUMANN)) {
        s.UM[n] += dt * u;
        s.UM[n + Nn] += dt * v;
    }
    s.UT[n] += dt * u;
    s.UT[n + Nn] += dt * v;
}

struct IpcDev {
    unsigned long long *seq_push;   // exchanges pushed so far (this rank)
    unsigned long long *seq_pull;   // exchanges pulled so far
    unsigned int *done_push, *done_pull;  // block-completion counters
    int *error;                     // != 0 after a timeout / self-test mismatch
    double *mailbox;                // my mailbox: [2][2*tr] doubles
    unsigned long long *flags;      // my flags: [nr], written by the neighbours
    int tr, ns, nr;
    double *const *peer_seg;        // [ns] neighbour k's mailbox address of MY segment (parity 0)
    const long long *peer_parity_stride;  // [ns] doubles between that neighbour's two buffers (2*tr_k)
    unsigned long long *const *peer_flag; // [ns] address of my flag slot in neighbour k's mailbox
};

__device__ __forceinline__ void sys_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_SYSTEM));
}

// Halo exchange fused into the sub-step kernel (device-direct transport only).  Exchange x = the x-th
// updateGhosts since the mailboxes were connected; *ipc.seq_push counts the exchanges this rank has
// published, a neighbour's flag in my mailbox the exchanges IT has published.
//   kernel of sub-step s (x = *seq_push on entry):
//     boundary patches (own nodes that are sent, or ghost nodes among the staged ones) first wait until every
//     neighbour's flag has reached x, i.e. exchange x-1 -- the velocities this sub-step starts from -- has
//     landed, stage their ghost nodes straight from the mailbox (parity (x-1)&1) and copy them through to the
//     VT buffer (the deferred mesh move and the end of the step read them there);
//     in the node phase every sent node is stored into the neighbours' mailboxes (parity x&1); the last
//     boundary patch to finish raises my flag at the neighbours to x+1; the last patch of the grid advances
//     *seq_push.  Boundary patches come first in the grid, so the data travels while the interior is computed.
//   Why two mailbox halves suffice: a patch that stages ghosts also has sent nodes (if own node n shares an
//   element with a ghost owned by B, then n is a ghost of B), so it is a boundary patch, and my flag x+1 is
//   raised only after all of them have finished reading exchange x-1; a neighbour overwrites that half
//   (exchange x+1) only after it has seen my flag x+1.
struct HaloFused {
    IpcDev ipc;
    int n_boundary;                    // patches [0, n_boundary) are the boundary patches (re-uploaded in that order)
    const int *send_ptr;               // [No+1] CSR over own nodes
    const int *send_k, *send_pos;      // neighbour (index into send_procs) and position inside its segment
    const int *send_off;               // [ns+1] segment offsets (segment length = v offset)
    const int *ghost_off, *ghost_srl;  // [Nn-No] u offset inside a mailbox half, and the v offset from it
    int No;
    int from_mailbox;                  // 0: first sub-step of a step, the ghosts are in the VT buffer
    unsigned int *done_all;            // k_smooth_halo: two-level ticket counters, [0] global, [32 (g+1)] group g
    const int *send_block_rank;        // k_smooth_halo: rank of block b among the blocks that send something, -1: sends nothing
    int n_send_blocks;
};

#ifdef NXS_PHASE_TIMING  // kernel microscope (scripts/phase_timing.py builds a variant of the library with it)
__device__ long long g_phase_t[8 * 8192];
#define NXS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_t[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define NXS_STAMP(k) do { } while (0)
#endif

#ifndef NXS_PF
#define NXS_PF 1
#endif
#ifndef NXS_T256_MAXP
#define NXS_T256_MAXP 128
#endif

// streaming accesses that should not displace the reusable arrays from L2 / Infinity Cache
template <bool NT> __device__ __forceinline__ double ldg(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void stg(double *p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// ------------------------------------------------------------------------------------------------
// v2  ONE launch per sub-step: K3 (stress/damage) + K4 (assembly) + K5 (nodal solve) + K7 (mesh move).
// FE.cpp:10425-10553.  Workgroup = patch.  Phase 0 stages the patch's nodal velocities in LDS,
// phase A updates every patch element from them and leaves its six corner forces in LDS, phase B
// lets each own node subtract the forces of its fan (ascending element order, as the serial scatter)
// and solve.  sigma, damage and VT are ping-pong buffered: a neighbouring patch may still be reading
// the old values of a shared element / node while this one writes the new ones.
template <int T, bool POW4, int NTM, bool HALO>
__global__ void __launch_bounds__(T) k_substep_fused(DevMesh m, DevPatches pp, DevState s, DevWork w, DevParams p,
                                                     PingPong b, double move_dt, HaloFused hf) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lu = lds, *lv = lds + pp.Mmax, *lx = lds + 2 * (size_t)pp.Mmax, *ly = lds + 3 * (size_t)pp.Mmax,
           *lF = lds + 4 * (size_t)pp.Mmax;  // lF[6][Emax]
    // consecutive patches are neighbours in space: keep them on one XCD (blocks are dealt round-robin
    // over the 8 XCDs) so that shared halo elements / nodes hit that XCD's L2.  Speed only.
    auto xcd_remap = [](int pos, int n) {  // position in dispatch order -> index, classes pos%8 -> contiguous index ranges
        const int q = n >> 3, r = n & 7, x = pos & 7;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3);
    };
    int blk = blockIdx.x;
    NXS_STAMP(0);
    unsigned long long xseq = 0ull;
    bool boundary = false;
    if (HALO) {  // boundary patches [0, n_boundary) lead the grid in dispatch order; the remap acts inside each group
        boundary = blk < hf.n_boundary;
        if (boundary || hf.n_boundary == 0) xseq = *hf.ipc.seq_push;  // interior patches never look at it: the last boundary patch may advance it while they run
        blk = boundary ? xcd_remap(blk, hf.n_boundary) : hf.n_boundary + xcd_remap(blk - hf.n_boundary, (int)gridDim.x - hf.n_boundary);
    } else {
        blk = xcd_remap(blk, (int)gridDim.x);
    }
    const int t = threadIdx.x, Nn = m.Nn, Emax = pp.Emax;
    const int nM = pp.node_cnt[blk], nE = pp.elem_cnt[blk], nO = pp.own_cnt[blk];
    const int *pn = pp.pnodes + (size_t)blk * pp.Mmax;
    const int *pe = pp.pelem + (size_t)blk * Emax;
    const ushort4 *pt = reinterpret_cast<const ushort4 *>(pp.ptri) + (size_t)blk * Emax;
    const bool bbm = p.dynamics_type == NXS_DYN_BBM;
    constexpr bool NT_S = NTM & 1, NT_U = NTM & 2, NT_C = NTM & 4;  // sigma/damage, UM/UT, element constants

    // The kernel is latency-bound unless every dependent load hop is overlapped, so all global loads
    // that do not need LDS are issued up front: indices first, then (one hop later) the nodal
    // velocities to stage and this thread's element data, all in flight together before barrier 1.
    // The index rows are padded to Mmax / Emax, so these loads depend on the launch arguments only -- not on the
    // patch's counts (one more dependent hop; the counts arrive meanwhile and mask the uses).
    const int my_node = (t < pp.Mmax) ? pn[t] : 0;  // patch-local slot t (an own node when t < nO)
    const int my_node2 = (t + T < pp.Mmax) ? pn[t + T] : 0;  // a patch stages ~1.25 nodes per own node: second staging slot
    int eraw = 0;
    ushort4 tr = make_ushort4(0, 0, 0, 0);
    if (t < Emax) { eraw = pe[t]; tr = pt[t]; }
#if NXS_PF >= 1
    // element rounds 1 and 2 (a patch holds ~2.2 elements per own node): their indices are fetched now, so
    // that a later round starts with its data loads instead of an index hop
    int eraw1 = 0, eraw2 = 0;
    ushort4 tr1 = tr, tr2 = tr;
    if (t + T < Emax) { eraw1 = pe[t + T]; tr1 = pt[t + T]; }
    if (t + 2 * T < Emax) { eraw2 = pe[t + 2 * T]; tr2 = pt[t + 2 * T]; }
#endif

    const bool mailbox_ghosts = HALO && boundary && hf.from_mailbox;
    if (mailbox_ghosts) {  // exchange xseq-1 must have landed before a ghost node is staged
        if (t == 0) {
            const long long t0 = wall_clock64();  // 100 MHz
            bool ok = true;
            for (int k = 0; k < hf.ipc.nr && ok; ++k)
                while (__hip_atomic_load(hf.ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < xseq) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__hip_atomic_load(hf.ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }  // a wait already timed out: the run is lost, do not wait again
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(hf.ipc.error, 3); break; }  // 10 s
                }
            // no acquire fence: the mailbox is uncached memory and every read of it below is a system-scope load that
            // bypasses the caches; a fence here would invalidate this XCD's caches once per boundary patch and sub-step
        }
        __syncthreads();
    }
    auto stage = [&](int i, int g) {
        if (mailbox_ghosts && g >= hf.No) {
            const double *src = hf.ipc.mailbox + ((xseq - 1ull) & 1ull) * 2ull * (unsigned long long)hf.ipc.tr + hf.ghost_off[g - hf.No];
            const double u = sys_load(src), v = sys_load(src + hf.ghost_srl[g - hf.No]);
            lu[i] = u; lv[i] = v;
            const_cast<double *>(b.VTc)[g] = u;  // every patch that stages g writes the same two values
            const_cast<double *>(b.VTc)[g + Nn] = v;
        } else {
            lu[i] = b.VTc[g];
            lv[i] = b.VTc[g + Nn];
        }
        lx[i] = w.xs[g];
        ly[i] = w.ys[g];
    };
    if (t < nM) stage(t, my_node);
    if (t + T < nM) stage(t + T, my_node2);
    for (int i = t + 2 * T; i < nM; i += T) stage(i, pn[i]);

    for (int base = 0; base < nE; base += T) {
        const int l = base + t;
#if NXS_PF >= 1
        if (base >= 3 * T && l < nE) { eraw = pe[l]; tr = pt[l]; }  // very large patches: rounds beyond the prefetched ones
#else
        if (base > 0 && l < nE) { eraw = pe[l]; tr = pt[l]; }  // patches larger than the block: extra rounds
#endif
        const bool active = l < nE;
        const bool writer = eraw >= 0;
        const int e = writer ? eraw : ~eraw;
        double dxN[6], sig[3] = {0., 0., 0.}, damage = 0., c_expC = 0., c_pmax = 0., c_heal = 0., c_dxs = 1., c_coh = 0., volume = 0.;
        bool skip = true;
        int dxi = 0;
        if (active) {
            if (!bbm) skip = w.eskip[e];
            sig[0] = ldg<NT_S>(b.s0c + e); sig[1] = ldg<NT_S>(b.s1c + e); sig[2] = ldg<NT_S>(b.s2c + e);
            if (bbm) damage = ldg<NT_S>(b.dc + e);
            c_expC = ldg<NT_C>(w.expC + e);
            volume = ldg<NT_C>(w.volume + e);
            if (bbm) {
                c_pmax = ldg<NT_C>(w.pmax + e); c_heal = ldg<NT_C>(w.heal + e); dxi = w.dxi[e]; c_coh = ldg<NT_C>(s.cohesion + e);
            }
        }
        if (base == 0) { __syncthreads(); NXS_STAMP(1); }  // staged velocities / coordinates visible
        if (active && bbm) {  // M_delta_x is an integer number of metres (Q1) and travels as one, with the skip flag in its sign
            skip = dxi < 0;
            c_dxs = (double)(skip ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // == w.dxs[e], FE.cpp:4232
        }
        if (active) {
            {   // shapeCoeff (FE.cpp:1951-1964) from the staged frozen coordinates: the same operations as
                // k_prep_elements, so the same bits as M_shape_coeff -- 48 B/element less to stream
                const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
                const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
                const double jac = jacobian(vx, vy);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                    dxN[k] = (vy[kp1] - vy[kp2]) / jac;
                    dxN[k + 3] = (vx[kp2] - vx[kp1]) / jac;
                }
            }
            if (skip) {
                sig[0] = sig[1] = sig[2] = 0.;
                damage = 0.;
            } else {
                const double u[3] = {lu[tr.x], lu[tr.y], lu[tr.z]};
                const double v[3] = {lv[tr.x], lv[tr.y], lv[tr.z]};
                if (bbm) bbm_stress<POW4>(p, dxN, u, v, sig, damage, c_expC, c_pmax, c_heal, c_dxs, c_coh);
                else vp_stress(p, dxN, u, v, sig, c_expC);
            }
            if (writer) {
                stg<NT_S>(b.s0n + e, sig[0]); stg<NT_S>(b.s1n + e, sig[1]); stg<NT_S>(b.s2n + e, sig[2]);
                if (bbm) stg<NT_S>(b.dn + e, damage);
            }
            double F[6];
            corner_forces(volume, sig, dxN, F);
#pragma unroll
            for (int k = 0; k < 6; ++k) lF[(size_t)k * Emax + l] = F[k];
        }
#if NXS_PF >= 1
        eraw = eraw1; tr = tr1; eraw1 = eraw2; tr1 = tr2;
#endif
    }

    NXS_STAMP(2);
    // node phase: issue this node's loads before barrier 2 so that they overlap the wait
    const unsigned short *pf = pp.pfan + (size_t)blk * pp.Wp * pp.Pmax;
    for (int base = 0; base < nO || base == 0; base += T) {
        const int i = base + t;
        const bool active = i < nO;
        const int n = active ? (base == 0 ? my_node : pn[i]) : 0;
        unsigned char nf = 0;
        double node_mass = 0., gx = 0., gy = 0., rlm = 0., cbu = 0., fcor = 0., lat = 0., tax = 0., tay = 0., ou = 0., ov = 0.,
               vtmu = 0., vtmv = 0., umu = 0., umv = 0., utu = 0., utv = 0.;
        int sq0 = 0, sq1 = 0;
        if (active) {
            nf = m.nflags[n];
            node_mass = w.node_mass[n];
            gx = w.grad_ssh[n]; gy = w.grad_ssh[n + Nn];
            rlm = w.rlmass[n]; cbu = w.C_bu[n]; fcor = w.fcor[n]; lat = (nf & NF_LAT_NEG) ? -1. : 1.;
            tax = w.D_tau_a[n]; tay = w.D_tau_a[n + Nn];
            ou = s.ocean[n]; ov = s.ocean[n + Nn];
            if (p.dynamics_type == NXS_DYN_MEVP) { vtmu = w.VTM[n]; vtmv = w.VTM[n + Nn]; }
            if (move_dt != 0.) { umu = ldg<NT_U>(s.UM + n); umv = ldg<NT_U>(s.UM + n + Nn); utu = ldg<NT_U>(s.UT + n); utv = ldg<NT_U>(s.UT + n + Nn); }
            if (HALO && boundary) { sq0 = hf.send_ptr[n]; sq1 = hf.send_ptr[n + 1]; }
        }
        // the node's fan (element slot, corner) too: 8 entries cover all but the most irregular vertices
        unsigned short fan[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) fan[k] = (active && k < pp.Wp) ? pf[(size_t)k * pp.Pmax + i] : (unsigned short)0xFFFFu;
        if (base == 0) { __syncthreads(); NXS_STAMP(3); }  // corner forces visible
        if (!active) continue;
        double uice = lu[i], vice = lv[i];
        if (!((nf & NF_DIRICHLET) || node_mass == 0.)) {
            bool more = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned ent = fan[k];
                if (!more || ent == 0xFFFFu) { more = false; continue; }
                if (ent & 4u) continue;  // ghostNodes[i] (FE.cpp:10456)
                const int l = ent >> 3, c = ent & 3u;
                gx -= lF[(size_t)c * Emax + l];
                gy -= lF[(size_t)(c + 3) * Emax + l];
            }
            for (int k = 8; more && k < pp.Wp; ++k) {
                const unsigned ent = pf[(size_t)k * pp.Pmax + i];
                if (ent == 0xFFFFu) break;
                if (ent & 4u) continue;  // ghostNodes[i] (FE.cpp:10456)
                const int l = ent >> 3, c = ent & 3u;
                gx -= lF[(size_t)c * Emax + l];
                gy -= lF[(size_t)(c + 3) * Emax + l];
            }
            nodal_solve(p, gx, gy, uice, vice, node_mass, rlm, cbu, fcor, lat, tax, tay, ou, ov, vtmu, vtmv);
        }
        b.VTn[n] = uice;
        b.VTn[n + Nn] = vice;
        if (HALO) {  // updateGhosts, sending side: straight into the neighbours' mailboxes
            for (int q = sq0; q < sq1; ++q) {
                const int k = hf.send_k[q];
                double *dst = hf.ipc.peer_seg[k] + (xseq & 1ull) * hf.ipc.peer_parity_stride[k] + hf.send_pos[q];
                sys_store(dst, uice);
                sys_store(dst + (hf.send_off[k + 1] - hf.send_off[k]), vice);
            }
        }
        if (move_dt != 0.) {  // FE.cpp:10543-10550; Neumann nodes keep M_UM (restore == skip)
            if (!(nf & NF_NEUMANN)) {
                stg<NT_U>(s.UM + n, umu + move_dt * uice);
                stg<NT_U>(s.UM + n + Nn, umv + move_dt * vice);
            }
            stg<NT_U>(s.UT + n, utu + move_dt * uice);
            stg<NT_U>(s.UT + n + Nn, utv + move_dt * vice);
        }
    }
    NXS_STAMP(4);
    if (HALO) {
        if (boundary) {  // publish: the last boundary patch to finish raises my flag at every neighbour
            // The mailbox stores are write-through system-scope stores into uncached memory: nothing of them lives in a cache, so
            // no release fence (= writing this XCD's whole L2 back, per workgroup and sub-step) is needed to make them visible --
            // every wave drains its own stores, the barrier collects the waves, one lane counts the workgroup in, and the last
            // one raises the flags behind ONE release fence per launch.  The separate k_halo_push keeps the fences; bench.py
            // checks both variants against each other on the machine it runs on.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0 && atomicAdd(hf.ipc.done_push, 1u) == (unsigned)hf.n_boundary - 1u) {
                __threadfence_system();  // the one release of the launch
                for (int k = 0; k < hf.ipc.ns; ++k)
                    __hip_atomic_store(hf.ipc.peer_flag[k], xseq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // released by the fence above, once for all flags
                *hf.ipc.done_push = 0u;
                *hf.ipc.seq_push = xseq + 1ull;  // every boundary patch has read it; no counter over the whole grid (same-address atomics are served ~10 ns apart)
            }
        } else if (hf.n_boundary == 0 && blockIdx.x == 0 && t == 0) {
            *hf.ipc.seq_push = xseq + 1ull;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v3  D sub-steps per launch (single rank, deferred mesh move): temporal blocking of the sub-step loop.
// A patch carries D rings of halo (DevPatches2).  Sub-step k of the launch updates the elements E_(D-k) and solves the nodes
// N_(D-k-1): what lies outside the own nodes is recomputed redundantly by the neighbouring patches -- same inputs, same
// operations, same bits -- so the element state and the nodal inputs are read once and written once per D sub-steps and
// the loop needs S/D launches.  The intermediate stresses stay in LDS, the intermediate velocities of the own nodes still go
// to their ring slots: the deferred mesh move needs every sub-step's velocity.
template <int T, bool POW4, int NTM>
__global__ void __launch_bounds__(T) k_substep_multi(DevMesh m, DevPatches2 pp, DevState s, DevWork w, DevParams p, PingPong b, VTOut vout) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int NDm = pp.NDmax, EDm = pp.EDmax, ESm = pp.ESmax, D = pp.D;
    double *lu = lds, *lv = lu + NDm, *lx = lv + NDm, *ly = lx + NDm, *lF = ly + NDm /*[6][EDm]*/, *lS = lF + 6 * (size_t)EDm /*[4][ESm]*/;
    int blk;
    {   // consecutive patches are neighbours in space: keep them on one XCD (see k_substep_fused)
        const int n = (int)gridDim.x, pos = (int)blockIdx.x, q = n >> 3, r = n & 7, x = pos & 7;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3);
    }
    const int t = threadIdx.x, Nn = m.Nn;
    const int *ncnt = pp.ncnt + (size_t)blk * (D + 1), *ecnt = pp.ecnt + (size_t)blk * D;
    const int nO = ncnt[0], nD = ncnt[D];
    const int *pn = pp.pnodes + (size_t)blk * NDm;
    const int *pe = pp.pelem + (size_t)blk * EDm;
    const ushort4 *pt = reinterpret_cast<const ushort4 *>(pp.ptri) + (size_t)blk * EDm;
    const unsigned short *pf = pp.pfan + (size_t)blk * pp.Wp * pp.NSmax;
    const bool bbm = p.dynamics_type == NXS_DYN_BBM;
    constexpr bool NT_S = NTM & 1, NT_C = NTM & 4;

    // index rows are padded: the first loads depend on the launch arguments only
    const int my_node = (t < NDm) ? pn[t] : 0;
    int eraw0 = 0;
    ushort4 tr0 = make_ushort4(0, 0, 0, 0);
    if (t < EDm) { eraw0 = pe[t]; tr0 = pt[t]; }
    for (int i = t; i < nD; i += T) {
        const int g = (i == t) ? my_node : pn[i];
        lu[i] = b.VTc[g]; lv[i] = b.VTc[g + Nn];
        lx[i] = w.xs[g]; ly[i] = w.ys[g];
    }

    // one element of one sub-step (FE.cpp:10425-10467), split into its global loads and the rest so that the barrier
    // between them sits in uniform control flow.  first sub-step of the launch: state from HBM; last: result to HBM (if this
    // patch writes the element); in between the state lives in LDS
    struct ElemIn { int e; bool writer, skip; int dxi; double sig[3], damage, expC, volume, pmax, heal, coh; };
    auto load_element = [&](const int eraw, const bool first, const bool last) {
        ElemIn in;
        in.writer = eraw >= 0;
        in.e = in.writer ? eraw : ~eraw;
        const int e = in.e;
        in.skip = true; in.dxi = 0; in.damage = 0.; in.pmax = 0.; in.heal = 0.; in.coh = 0.; in.sig[0] = in.sig[1] = in.sig[2] = 0.;
        if (!bbm) in.skip = w.eskip[e];
        if (first) {
            in.sig[0] = ldg<NT_S>(b.s0c + e); in.sig[1] = ldg<NT_S>(b.s1c + e); in.sig[2] = ldg<NT_S>(b.s2c + e);
            if (bbm) in.damage = ldg<NT_S>(b.dc + e);
        }
        // the element constants are read by every sub-step of the launch: the later reads hit the L2
        in.expC = last ? ldg<NT_C>(w.expC + e) : w.expC[e];
        in.volume = last ? ldg<NT_C>(w.volume + e) : w.volume[e];
        if (bbm) {
            in.pmax = last ? ldg<NT_C>(w.pmax + e) : w.pmax[e]; in.heal = last ? ldg<NT_C>(w.heal + e) : w.heal[e];
            in.dxi = w.dxi[e]; in.coh = last ? ldg<NT_C>(s.cohesion + e) : s.cohesion[e];
        }
        return in;
    };
    auto compute_element = [&](const int l, const ushort4 tr, ElemIn &in, const bool first, const bool last, const int keep) {
        double dxN[6], sig[3] = {in.sig[0], in.sig[1], in.sig[2]}, damage = in.damage, c_dxs = 1.;
        bool skip = in.skip;
        if (!first) { sig[0] = lS[l]; sig[1] = lS[ESm + l]; sig[2] = lS[2 * (size_t)ESm + l]; damage = lS[3 * (size_t)ESm + l]; }
        if (bbm) {  // M_delta_x is an integer number of metres (Q1) and travels as one, with the skip flag in its sign
            skip = in.dxi < 0;
            c_dxs = (double)(skip ? ~in.dxi : in.dxi) * p.sqrt_nu_rhoi;  // FE.cpp:4232
        }
        {   // shapeCoeff (FE.cpp:1951-1964) from the staged frozen coordinates, as k_substep_fused
            const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
            const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
            const double jac = jacobian(vx, vy);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                dxN[k] = (vy[kp1] - vy[kp2]) / jac;
                dxN[k + 3] = (vx[kp2] - vx[kp1]) / jac;
            }
        }
        if (skip) {
            sig[0] = sig[1] = sig[2] = 0.;
            damage = 0.;
        } else {
            const double u[3] = {lu[tr.x], lu[tr.y], lu[tr.z]};
            const double v[3] = {lv[tr.x], lv[tr.y], lv[tr.z]};
            if (bbm) bbm_stress<POW4>(p, dxN, u, v, sig, damage, in.expC, in.pmax, in.heal, c_dxs, in.coh);
            else vp_stress(p, dxN, u, v, sig, in.expC);
        }
        if (!last) {
            if (l < keep) { lS[l] = sig[0]; lS[ESm + l] = sig[1]; lS[2 * (size_t)ESm + l] = sig[2]; lS[3 * (size_t)ESm + l] = damage; }  // needed by the next sub-step
        } else if (in.writer) {
            stg<NT_S>(b.s0n + in.e, sig[0]); stg<NT_S>(b.s1n + in.e, sig[1]); stg<NT_S>(b.s2n + in.e, sig[2]);
            if (bbm) stg<NT_S>(b.dn + in.e, damage);
        }
        double F[6];
        corner_forces(in.volume, sig, dxN, F);
#pragma unroll
        for (int k = 0; k < 6; ++k) lF[(size_t)k * EDm + l] = F[k];
    };
    // one node of one sub-step (FE.cpp:10472-10529), loads and solve apart for the same reason
    struct NodeIn { unsigned char nf; double node_mass, gx, gy, rlm, cbu, fcor, tax, tay, ou, ov; unsigned short fan[8]; };
    auto load_node = [&](const int i, const int n) {
        NodeIn in;
        in.nf = m.nflags[n];
        in.node_mass = w.node_mass[n];
        in.gx = w.grad_ssh[n]; in.gy = w.grad_ssh[n + Nn];
        in.rlm = w.rlmass[n]; in.cbu = w.C_bu[n]; in.fcor = w.fcor[n];
        in.tax = w.D_tau_a[n]; in.tay = w.D_tau_a[n + Nn];
        in.ou = s.ocean[n]; in.ov = s.ocean[n + Nn];
#pragma unroll
        for (int k = 0; k < 8; ++k) in.fan[k] = (k < pp.Wp) ? pf[(size_t)k * pp.NSmax + i] : (unsigned short)0xFFFFu;
        return in;
    };
    auto solve_node = [&](const int i, NodeIn &in, double &uice, double &vice) {
        uice = lu[i]; vice = lv[i];
        if ((in.nf & NF_DIRICHLET) || in.node_mass == 0.) return;
        double gx = in.gx, gy = in.gy;
        bool more = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned ent = in.fan[k];
            if (!more || ent == 0xFFFFu) { more = false; continue; }
            if (ent & 4u) continue;  // ghostNodes[i] (FE.cpp:10456)
            const int l = ent >> 3, c = ent & 3u;
            gx -= lF[(size_t)c * EDm + l];
            gy -= lF[(size_t)(c + 3) * EDm + l];
        }
        for (int k = 8; more && k < pp.Wp; ++k) {
            const unsigned ent = pf[(size_t)k * pp.NSmax + i];
            if (ent == 0xFFFFu) break;
            if (ent & 4u) continue;
            const int l = ent >> 3, c = ent & 3u;
            gx -= lF[(size_t)c * EDm + l];
            gy -= lF[(size_t)(c + 3) * EDm + l];
        }
        nodal_solve(p, gx, gy, uice, vice, in.node_mass, in.rlm, in.cbu, in.fcor, (in.nf & NF_LAT_NEG) ? -1. : 1., in.tax, in.tay, in.ou, in.ov, 0., 0.);
    };

    for (int k = 0; k < D; ++k) {
        const int ne = ecnt[D - 1 - k], nn = ncnt[D - 1 - k], keep = (k + 1 < D) ? ecnt[D - 2 - k] : 0;
        const bool first = k == 0, last = k == D - 1;
        // ---- elements E_(D-k)
        for (int base = 0; base < ne || base == 0; base += T) {
            const int l = base + t;
            const bool active = l < ne;
            int eraw = eraw0; ushort4 tr = tr0;
            if (base > 0 && active) { eraw = pe[l]; tr = pt[l]; }
            ElemIn in{};
            if (active) in = load_element(eraw, first, last);
            if (base == 0) __syncthreads();  // k == 0: staged velocities / coordinates; k > 0: the velocities of sub-step k on N_(D-k), forces consumed
            if (active) compute_element(l, tr, in, first, last, keep);
        }
        // ---- nodes N_(D-k-1)
        double *vt = vout.slot[k];
        for (int base = 0; base < nn || base == 0; base += T) {
            const int i = base + t;
            const bool active = i < nn;
            const int n = active ? ((base == 0) ? my_node : pn[i]) : 0;
            NodeIn in{};
            if (active) in = load_node(i, n);
            if (base == 0) __syncthreads();  // corner forces of this sub-step visible
            if (active) {
                double u1, v1;
                solve_node(i, in, u1, v1);
                if (i < nO) { vt[n] = u1; vt[n + Nn] = v1; }
                lu[i] = u1; lv[i] = v1;  // a node's solve reads only its own staged velocity: in place
            }
        }
    }
}

// Deferred mesh move of the fused path: the fused kernel leaves every sub-step's velocity in a ring of
// VT buffers; every `count` sub-steps this kernel applies the same sequence of additions
// M_UM += dte*M_VT, M_UT += dte*M_VT (FE.cpp:10543-10550) for all nodes, owned and ghost -- same
// operations in the same order, but UM/UT are streamed once per `count` sub-steps instead of every one.
#define NXS_MAX_RING 129
struct VTRing { double *slot[NXS_MAX_RING]; int R; };

__global__ void __launch_bounds__(BLOCK) k_move_ring(DevMesh m, DevState s, VTRing ring, int first, int count, double dt) {
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.Nn) return;
    const int Nn = m.Nn;
    const bool free_node = !(m.nflags[n] & NF_NEUMANN);  // Neumann nodes keep M_UM (restore == skip)
    double umu = s.UM[n], umv = s.UM[n + Nn], utu = s.UT[n], utv = s.UT[n + Nn];
    int sl = first;
    for (int j = 0; j < count; ++j) {
        const double u = ring.slot[sl][n], v = ring.slot[sl][n + Nn];
        if (free_node) { umu += dt * u; umv += dt * v; }
        utu += dt * u; utv += dt * v;
        sl = (sl + 1 == ring.R) ? 0 : sl + 1;
    }
    if (free_node) { s.UM[n] = umu; s.UM[n + Nn] = umv; }
    s.UT[n] = utu; s.UT[n + Nn] = utv;
}

// odd number of sub-steps: bring the ping-pong result back to the primary buffers
__global__ void __launch_bounds__(BLOCK) k_pingpong_copy_back(DevMesh m, DevState s, int bbm, const double *vt_src, int copy_sigma) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (vt_src && i < 2 * m.Nn) s.VT[i] = vt_src[i];
    if (copy_sigma && i < m.Ne) {
        s.s0[i] = s.s0_b[i]; s.s1[i] = s.s1_b[i]; s.s2[i] = s.s2_b[i];
        if (bbm) s.damage[i] = s.damage_b[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K6 updateGhosts, FE.cpp:13963-13996.  buf holds, per neighbour k, [u-block | v-block] at 2*off[k].
__global__ void __launch_bounds__(BLOCK) k_halo_pack(const double *__restrict__ vec, int Nn, int total,
                                                     const int *__restrict__ index, const int *__restrict__ seg_of,
                                                     const int *__restrict__ offsets, double *__restrict__ buf) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= total) return;
    const int k = seg_of[j];
    const int off = offsets[k], srl = offsets[k + 1] - off;
    const int idx = index[j];
    buf[2 * (size_t)off + (j - off)] = vec[idx];
    buf[2 * (size_t)off + (j - off) + srl] = vec[idx + Nn];
}

// unpack + (optionally) move the ghost nodes: every ghost node is in exactly one recv list
__global__ void __launch_bounds__(BLOCK) k_halo_unpack(double *__restrict__ vec, DevMesh m, DevState s, int total,
                                                       const int *__restrict__ index, const int *__restrict__ seg_of,
                                                       const int *__restrict__ offsets, const double *__restrict__ buf,
                                                       double move_dt) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= total) return;
    const int Nn = m.Nn;
    const int k = seg_of[j];
    const int off = offsets[k], srl = offsets[k + 1] - off;
    const int n = index[j];
    const double u = buf[2 * (size_t)off + (j - off)];
    const double v = buf[2 * (size_t)off + (j - off) + srl];
    vec[n] = u;
    vec[n + Nn] = v;
    if (move_dt != 0.) {
        if (!(m.nflags[n] & NF_NEUMANN)) {
            s.UM[n] += move_dt * u;
            s.UM[n + Nn] += move_dt * v;
        }
        s.UT[n] += move_dt * u;
        s.UT[n + Nn] += move_dt * v;
    }
}

// ------------------------------------------------------------------------------------------------
// K6, device-direct transport: updateGhosts through peer-mapped memory (xGMI P2P stores) instead of
// RCCL launches.  Every rank owns a "mailbox" = two receive buffers (exchange parity) + one flag per
// receive neighbour, exported with hipIpcGetMemHandle and mapped by its neighbours.
//   k_halo_push      pack my boundary values and store them straight into each neighbour's mailbox
//                    (write-through, system scope), then -- after ALL blocks have finished -- set my flag
//                    in each neighbour's mailbox to the exchange's sequence number.
//   k_halo_pull      wait until every neighbour's flag has reached the sequence number, then unpack the
//                    mailbox (reads that bypass the caches) into the ghost slots and move the ghost nodes.
// Sequence numbers live in device memory, so the kernels replay unchanged from a hipGraph.  Two mailbox
// buffers suffice: a neighbour cannot start exchange x+2 before it has received my exchange x+1, which I
// send only after my pull of exchange x.  Every spin is bounded; a timeout raises ipc->error.
// selftest != 0: the payload is a code of (rank, entry, sequence) instead of vec
__global__ void __launch_bounds__(BLOCK) k_halo_push(const double *__restrict__ vec, int Nn, int total, const int *__restrict__ index,
                                                     const int *__restrict__ seg_of, const int *__restrict__ offsets, IpcDev ipc,
                                                     int rank, int selftest) {
    const unsigned long long seq = *ipc.seq_push;
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < total) {
        const int k = seg_of[j];
        const int off = offsets[k], srl = offsets[k + 1] - off;
        double *dst = ipc.peer_seg[k] + (seq & 1ull) * ipc.peer_parity_stride[k];
        double u, v;
        if (selftest) {
            u = (double)rank * 1e6 + (double)(j - off) + (double)seq * 1e-3;
            v = -u;
        } else {
            const int idx = index[j];
            u = vec[idx];
            v = vec[idx + Nn];
        }
        sys_store(dst + (j - off), u);
        sys_store(dst + (j - off) + srl, v);
    }
    // publish: every wave drains its stores, one lane releases for the block, the last block to finish raises the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) { __threadfence_system(); last = (atomicAdd(ipc.done_push, 1u) == gridDim.x - 1); }
    __syncthreads();
    if (last) {
        __threadfence_system();
        for (int k = threadIdx.x; k < ipc.ns; k += BLOCK)
            __hip_atomic_store(ipc.peer_flag[k], seq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // released by the fence above
        if (threadIdx.x == 0) {
            *ipc.done_push = 0u;
            *ipc.seq_push = seq + 1ull;
        }
    }
}

__global__ void __launch_bounds__(BLOCK) k_halo_pull(double *__restrict__ vec, DevMesh m, DevState s, int total,
                                                     const int *__restrict__ index, const int *__restrict__ seg_of,
                                                     const int *__restrict__ offsets, IpcDev ipc, double move_dt, int selftest,
                                                     const int *__restrict__ recv_procs, int latest_pushed) {
    // latest_pushed: pull the exchange this rank published last (the fused kernel pushes by itself and keeps no
    // pull counter); afterwards both counters agree again
    const unsigned long long seq = latest_pushed ? *ipc.seq_push - 1ull : *ipc.seq_pull;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        const long long t0 = wall_clock64();  // 100 MHz
        for (int k = 0; k < ipc.nr; ++k) {
            while (__hip_atomic_load(ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq + 1ull) {
                __builtin_amdgcn_s_sleep(8);
                if (__hip_atomic_load(ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }  // a wait already timed out: do not wait again
                if (wall_clock64() - t0 > 1000000000ll) { ok = 0; atomicExch(ipc.error, 1); break; }  // 10 s
            }
            if (!ok) break;
        }
    }
    __syncthreads();
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (ok && j < total) {
        const int Nn = m.Nn;
        const int k = seg_of[j];
        const int off = offsets[k], srl = offsets[k + 1] - off;
        const double *src = ipc.mailbox + (seq & 1ull) * 2ull * (unsigned long long)ipc.tr + 2 * (size_t)off;
        const double u = sys_load(src + (j - off));
        const double v = sys_load(src + (j - off) + srl);
        if (selftest) {
            const double eu = (double)recv_procs[k] * 1e6 + (double)(j - off) + (double)seq * 1e-3;
            if (u != eu || v != -eu) atomicExch(ipc.error, 2);
        } else {
            const int n = index[j];
            vec[n] = u;
            vec[n + Nn] = v;
            if (move_dt != 0.) {
                if (!(m.nflags[n] & NF_NEUMANN)) {
                    s.UM[n] += move_dt * u;
                    s.UM[n + Nn] += move_dt * v;
                }
                s.UT[n] += move_dt * u;
                s.UT[n + Nn] += move_dt * v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ipc.done_pull, 1u) == gridDim.x - 1) {
        *ipc.done_pull = 0u;
        *ipc.seq_pull = seq + 1ull;
    }
}

// ------------------------------------------------------------------------------------------------
// K8 one Jacobi sweep of the open-water smoother, FE.cpp:10582-10608: src -> dst for every node
// Only ice-free, non-Dirichlet OWNED nodes change (FE.cpp:10589); every other node keeps its value in
// both ping-pong buffers (k_copy_vt makes them equal once before the 50 sweeps; ghosts are refreshed by
// the halo exchange), so a sweep touches 9 B per node plus the open-water nodes' neighbourhoods.
__global__ void __launch_bounds__(BLOCK) k_smooth(DevMesh m, DevWork w, const double *__restrict__ src, double *__restrict__ dst) {
    if (!w.open_blk[blockIdx.x]) return;  // no ice-free node among these BLOCK nodes: 9 B per node not read, 50 times per step
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.No) return;
    if ((m.nflags[n] & NF_DIRICHLET) || w.node_mass[n] != 0.) return;
    const int Nn = m.Nn;
    double u = 0., v = 0.;
    const int num_neighbours = m.n2n_cnt[n];
    for (int j = 0; j < num_neighbours; ++j) {  // Q8: bamg row order
        const int nni = m.n2n[(size_t)j * Nn + n];
        u += src[nni];
        v += src[nni + Nn];
    }
    u /= num_neighbours;
    v /= num_neighbours;
    dst[n] = u;
    dst[n + Nn] = v;
}

// The same sweep with updateGhosts inside (device-direct transport, see HaloFused): ghost neighbours are read from
// the mailbox (exchange x-1), every sent node -- smoothed or not -- is stored into the neighbours' mailboxes
// (exchange x), and the last block raises the flags.  One launch per sweep instead of three.
__global__ void __launch_bounds__(BLOCK) k_smooth_halo(DevMesh m, DevWork w, const double *__restrict__ src, double *__restrict__ dst, HaloFused hf) {
    const unsigned long long xseq = *hf.ipc.seq_push;
    const int Nn = m.Nn, No = m.No;
    if (hf.from_mailbox) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            bool ok = true;
            for (int k = 0; k < hf.ipc.nr && ok; ++k)
                while (__hip_atomic_load(hf.ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < xseq) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__hip_atomic_load(hf.ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(hf.ipc.error, 4); break; }  // 10 s
                }
        }
        __syncthreads();
    }
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n < No) {
        double u = src[n], v = src[n + Nn];
        if (!((m.nflags[n] & NF_DIRICHLET) || w.node_mass[n] != 0.)) {
            const double *mb = hf.ipc.mailbox + ((xseq - 1ull) & 1ull) * 2ull * (unsigned long long)hf.ipc.tr;
            u = 0.; v = 0.;
            const int num_neighbours = m.n2n_cnt[n];
            for (int j = 0; j < num_neighbours; ++j) {  // Q8: bamg row order
                const int nni = m.n2n[(size_t)j * Nn + n];
                if (hf.from_mailbox && nni >= No) {
                    const double *g = mb + hf.ghost_off[nni - No];
                    u += sys_load(g);
                    v += sys_load(g + hf.ghost_srl[nni - No]);
                } else {
                    u += src[nni];
                    v += src[nni + Nn];
                }
            }
            u /= num_neighbours;
            v /= num_neighbours;
            dst[n] = u;
            dst[n + Nn] = v;
        }
        for (int q = hf.send_ptr[n]; q < hf.send_ptr[n + 1]; ++q) {
            const int k = hf.send_k[q];
            double *d = hf.ipc.peer_seg[k] + (xseq & 1ull) * hf.ipc.peer_parity_stride[k] + hf.send_pos[q];
            sys_store(d, u);
            sys_store(d + (hf.send_off[k + 1] - hf.send_off[k]), v);
        }
    }
    // as in the sub-step kernel: drain per wave, count the block in, release once.  Only blocks that sent something take a
    // ticket, and the tickets are two-level (16 group counters, then one): atomics on one address are served ~10 ns apart,
    // a counter over all blocks of a 2-rank 2 km partition (1 400) would cost more than the sweep itself.
    const int sr = hf.send_block_rank[blockIdx.x];
    if (sr < 0 && !(hf.n_send_blocks == 0 && blockIdx.x == 0)) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x != 0) return;
    bool last = hf.n_send_blocks == 0;
    if (!last) {
        const unsigned int total = (unsigned)hf.n_send_blocks, g = (unsigned)sr % 16u, members = (total - g + 15u) / 16u;
        if (__hip_atomic_fetch_add(hf.done_all + 32u * (g + 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            __hip_atomic_store(hf.done_all + 32u * (g + 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int groups = total < 16u ? total : 16u;
            last = __hip_atomic_fetch_add(hf.done_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
        }
    }
    if (last) {
        __threadfence_system();  // the one release of the launch
        for (int k = 0; k < hf.ipc.ns; ++k)
            __hip_atomic_store(hf.ipc.peer_flag[k], xseq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // released by the fence above, once for all flags
        __hip_atomic_store(hf.done_all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *hf.ipc.seq_push = xseq + 1ull;
    }
}

__global__ void __launch_bounds__(BLOCK) k_copy_vt(int n2, const double *__restrict__ src, double *__restrict__ dst) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}

// K9 FE.cpp:10613-10640
__global__ void __launch_bounds__(BLOCK) k_ow_tail(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.Nn) return;
    const int Nn = m.Nn;
    const double vu = s.VT[n], vv = s.VT[n + Nn];
    const double uice = 0.5 * (vu + w.VTM[n]);
    const double vice = 0.5 * (vv + w.VTM[n + Nn]);
    const double ou = s.ocean[n], ov = s.ocean[n + Nn];
    const double c_prime = NXS_RHOW * p.qdw * hypot(ou - uice, ov - vice);
    w.D_tau_w[n] = c_prime * (uice - ou);
    w.D_tau_w[n + Nn] = c_prime * (vice - ov);
    const unsigned char nf = m.nflags[n];
    if ((nf & NF_DIRICHLET) || w.node_mass[n] != 0.) return;
    if (!(nf & NF_NEUMANN)) {
        s.UM[n] += p.dtime_step * vu;
        s.UM[n + Nn] += p.dtime_step * vv;
    }
    s.UT[n] += p.dtime_step * vu;
    s.UT[n + Nn] += p.dtime_step * vv;
}

// ------------------------------------------------------------------------------------------------
// K10 update(), FE.cpp:3946-4131
__global__ void __launch_bounds__(BLOCK) k_update(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    const bool to_be_updated = !(m.eflags[e] & EF_ON_NEUMANN);
    double D_del = 0.;
    const double surface_old = w.surface[e];
    double conc = s.conc[e], thick = s.thick[e], snow = s.snow[e], tmyi = s.tmyi[e], cmyi = s.cmyi[e];
    double ridge = s.ridge[e];
    double cy = 0., hy = 0., hsy = 0.;
    if (p.young_cat) { cy = s.cyoung[e]; hy = s.hyoung[e]; hsy = s.hsyoung[e]; }
    const double old_conc = conc;
    double vx[3], vy[3];
    load_vertices(m, s.UM, e, vx, vy);
    const double surface = (1. / 2) * fabs(jacobian(vx, vy));
    w.surface[e] = surface;
    if ((conc > 0.) && to_be_updated) {
        const double surf_ratio = surface_old / surface;
        conc *= surf_ratio; thick *= surf_ratio; snow *= surf_ratio; tmyi *= surf_ratio;
        s.s0[e] *= surf_ratio; s.s1[e] *= surf_ratio; s.s2[e] *= surf_ratio;
        ridge = 1. - (1. - ridge) * STD_MIN(1., conc) / (old_conc * surf_ratio);
        if (p.young_cat) { hy *= surf_ratio; cy *= surf_ratio; hsy *= surf_ratio; }
        if (p.equal_ridging) {
            const double conc_ratio = STD_MIN(1., conc) / old_conc;
            cmyi *= conc_ratio;
            D_del = 0.;
        } else {
            cmyi *= surf_ratio;
            D_del = -cmyi;
            cmyi = STD_MIN(cmyi, 1.);
            D_del += cmyi;
        }
        D_del *= NXS_DAYS_IN_SEC / p.dtime_step;
    }
    double open_water_concentration = 1. - conc;
    if (p.young_cat) open_water_concentration -= cy;
    open_water_concentration = (open_water_concentration < 0.) ? 0. : open_water_concentration;
    open_water_concentration = (open_water_concentration > 1.) ? 1. : open_water_concentration;
    double new_conc_young = 0., new_h_young = 0., new_hs_young = 0., newice = 0., del_c = 0., newsnow = 0.;
    const double ridge_young_ice_aspect_ratio = 10.;
    if (p.young_cat) {
        if (cy > 0.) {
            new_conc_young = STD_MIN(1., STD_MAX(0., 1. - conc - open_water_concentration));
            if ((conc > p.min_c) && (thick > p.min_h) && (new_conc_young < cy)) {
                new_h_young = new_conc_young * hy / cy;
                new_hs_young = new_conc_young * hsy / cy;
                newice = hy - new_h_young;
                del_c = (cy - new_conc_young) / ridge_young_ice_aspect_ratio;
                newsnow = hsy - new_hs_young;
                hy = new_h_young;
                hsy = new_hs_young;
                ridge = 1. - (1. - ridge) * thick / (thick + newice);
                thick += newice;
                snow += newsnow;
            }
        } else {
            hy = 0.;
            hsy = 0.;
        }
    }
    conc = STD_MIN(1., STD_MAX(0., 1. - new_conc_young - open_water_concentration + del_c));
    if (p.young_cat) {
        new_conc_young = STD_MAX(0., STD_MIN(new_conc_young, 1. - conc));
        cy = new_conc_young;
    }
    const double max_true_thickness = 50.;
    if (conc > 0.) {
        double test_h_thick = thick / conc;
        test_h_thick = (test_h_thick > max_true_thickness) ? max_true_thickness : test_h_thick;
        conc = STD_MIN(1. - new_conc_young, thick / test_h_thick);
    } else {
        ridge = 0.; thick = 0.; snow = 0.;
    }
    conc = ((conc > 0.) ? conc : 0.);
    thick = ((thick > 0.) ? thick : 0.);
    tmyi = ((tmyi > 0.) ? tmyi : 0.);
    snow = ((snow > 0.) ? snow : 0.);
    D_del = -cmyi;
    if (p.newice_type == 4 && p.use_young_myi)
        cmyi = STD_MAX(0., STD_MIN(cmyi, conc + cy));
    else
        cmyi = STD_MAX(0., STD_MIN(cmyi, conc));
    D_del += cmyi;
    s.conc[e] = conc; s.thick[e] = thick; s.snow[e] = snow; s.tmyi[e] = tmyi; s.cmyi[e] = cmyi;
    s.ridge[e] = ridge;
    if (p.young_cat) { s.cyoung[e] = cy; s.hyoung[e] = hy; s.hsyoung[e] = hsy; }
    w.D_del[e] = D_del;
}

// K13 updateFreeDriftVelocity, FE.cpp:10140-10176
__global__ void __launch_bounds__(BLOCK) k_free_drift(DevMesh m, DevState s, DevParams p) {
    const int nd = blockIdx.x * BLOCK + threadIdx.x;
    if (nd >= m.Nn) return;
    if (m.nflags[nd] & NF_DIRICHLET) return;
    const int Nn = m.Nn;
    const double u = s.VT[nd], v = s.VT[nd + Nn];
    const double ou = s.ocean[nd], ov = s.ocean[nd + Nn], wu = s.wind[nd], wv = s.wind[nd + Nn];
    double norm_Voce_ice = hypot(u - ou, v - ov);
    norm_Voce_ice = (norm_Voce_ice > 0.01) ? norm_Voce_ice : 0.01;
    double coef_Voce = p.ldw + p.qdw * norm_Voce_ice;
    coef_Voce *= NXS_RHOW;
    double norm_Vair_ice = hypot(u - wu, v - wv);
    norm_Vair_ice = (norm_Vair_ice > 0.01) ? norm_Vair_ice : 0.01;
    double coef_Vair = p.lda + p.qda * norm_Vair_ice;
    coef_Vair *= NXS_RHOA;
    const double nu_ = (coef_Vair * wu + coef_Voce * ou) / (coef_Vair + coef_Voce);
    const double nv_ = (coef_Vair * wv + coef_Voce * ov) / (coef_Vair + coef_Voce);
    s.VT[nd] = nu_;
    s.VT[nd + Nn] = nv_;
    s.UT[nd] += p.dtime_step * nu_;
    s.UT[nd + Nn] += p.dtime_step * nv_;
}

// ------------------------------------------------------------------------------------------------
// reductions: wave (64 lanes) shuffle -> LDS across the 4 waves of a block -> one partial per block
struct RegridPartial { double min_angle, min_jac, max_jac; };

__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(x, o, 64); x = (y < x) ? y : x; }
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(x, o, 64); x = (x < y) ? y : x; }
    return x;
}

// K11 minAngles (FE.cpp:1758-1768) + flip's jacobians (FE.cpp:1824-1839)
__global__ void __launch_bounds__(BLOCK) k_regrid_partials(DevMesh m, DevState s, RegridPartial *out) {
    __shared__ double sh[3][BLOCK / 64];
    double ang = INFINITY, jmin = INFINITY, jmax = -INFINITY;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < m.Ne; e += gridDim.x * BLOCK) {
        double vx[3], vy[3];
        load_vertices(m, s.UM, e, vx, vy);
        double a = hypot(vx[1] - vx[0], vy[1] - vy[0]);
        double b = hypot(vx[2] - vx[1], vy[2] - vy[1]);
        double c = hypot(vx[2] - vx[0], vy[2] - vy[0]);
        double t;  // std::sort of 3
        if (b < a) { t = a; a = b; b = t; }
        if (c < b) { t = b; b = c; c = t; }
        if (b < a) { t = a; a = b; b = t; }
        double minang = acos((pow(b, 2.) + pow(c, 2.) - pow(a, 2.)) / (2 * b * c));
        minang = minang * 45.0 / atan(1.0);
        ang = (minang < ang) ? minang : ang;
        const double jac = jacobian(vx, vy);
        jmin = (jac < jmin) ? jac : jmin;
        jmax = (jmax < jac) ? jac : jmax;
    }
    ang = wave_min(ang); jmin = wave_min(jmin); jmax = wave_max(jmax);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wv] = ang; sh[1][wv] = jmin; sh[2][wv] = jmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < BLOCK / 64; ++i) {
            ang = (sh[0][i] < ang) ? sh[0][i] : ang;
            jmin = (sh[1][i] < jmin) ? sh[1][i] : jmin;
            jmax = (jmax < sh[2][i]) ? sh[2][i] : jmax;
        }
        out[blockIdx.x] = RegridPartial{ang, jmin, jmax};
    }
}

__global__ void k_regrid_final(const RegridPartial *in, int n, RegridPartial *out) {
    double ang = INFINITY, jmin = INFINITY, jmax = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) {
        ang = (in[i].min_angle < ang) ? in[i].min_angle : ang;
        jmin = (in[i].min_jac < jmin) ? in[i].min_jac : jmin;
        jmax = (jmax < in[i].max_jac) ? in[i].max_jac : jmax;
    }
    ang = wave_min(ang); jmin = wave_min(jmin); jmax = wave_max(jmax);
    if (threadIdx.x == 0) *out = RegridPartial{ang, jmin, jmax};
}

// K12 checkFieldsFast (FE.cpp:14536-14655) restricted to this path's fields
__device__ __forceinline__ bool bad_range(double val, double lo, double hi) {
    return (val > hi) || (val < lo) || isnan(val);
}

__global__ void __launch_bounds__(BLOCK) k_check_fields(DevMesh m, DevState s, DevParams p, int *crash) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    bool bad = false;
    if (i < m.Ne) {
        bad |= bad_range(s.thick[i], 0., 50.);
        bad |= bad_range(s.snow[i], 0., 10.);
        bad |= bad_range(s.conc[i], 0., 1.);
        bad |= bad_range(s.damage[i], 0., 1.);
        bad |= bad_range(s.ridge[i], 0., 1.);
        if (p.young_cat) {
            bad |= bad_range(s.hyoung[i], 0., 2.);
            bad |= bad_range(s.hsyoung[i], 0., 2.);
            bad |= bad_range(s.cyoung[i], 0., 1.);
        }
    }
    if (i < m.Nn) {
        const double u = s.VT[i], v = s.VT[i + m.Nn];
        bad |= hypot(u, v) > 5.;
        bad |= isnan(u + v);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(crash, 1);
}


// ExternalData::get for a dataset that is interpolated linearly in time (model/externaldata.cpp:360-401):
//   value = M_factor*(fcoeff[0]*interpolated_data[0][i] + fcoeff[1]*interpolated_data[1][i]) + M_bias_correction
// for M_wind, M_ocean (2Nn) and M_ssh (Nn), evaluated on the device from two resident snapshots.
struct ForcingBlend { const double *w0, *w1, *o0, *o1, *s0, *s1; double c0, c1, factor[3], bias[3]; };
__global__ void __launch_bounds__(BLOCK) k_blend_forcing(int Nn, ForcingBlend b, double *__restrict__ wind, double *__restrict__ ocean, double *__restrict__ ssh) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < 2 * Nn) {
        wind[i] = b.factor[0] * (b.c0 * b.w0[i] + b.c1 * b.w1[i]) + b.bias[0];
        ocean[i] = b.factor[1] * (b.c0 * b.o0[i] + b.c1 * b.o1[i]) + b.bias[1];
    }
    if (i < Nn) ssh[i] = b.factor[2] * (b.c0 * b.s0[i] + b.c1 * b.s1[i]) + b.bias[2];
}

// ================================================================================================
// host side

namespace {

thread_local std::string g_create_error;

struct Rccl {  // RCCL entry points, resolved at comm_init (no link-time dependency)
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    void *CommInitRank = nullptr;  // ncclCommInitRank(comm*, nranks, ncclUniqueId by value (128 B), rank)
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

}  // namespace

struct NcclId { char internal[128]; };
typedef int (*nccl_comm_init_rank_t)(void **, int, NcclId, int);

struct HostPatches {
    int nP = 0, Pmax = 0, Emax = 0, Mmax = 0, Wp = 0;
    std::vector<int> own_cnt, elem_cnt, node_cnt, pnodes, pelem;
    std::vector<unsigned short> ptri, pfan;
    double avg_elems_per_own_node = 0.;
    bool used_hilbert = false;  // the caller's numbering had no locality: patches cut along a Hilbert curve
};

struct HostPatches2 {
    int nP = 0, D = 0, NDmax = 0, NSmax = 0, EDmax = 0, ESmax = 0, Wp = 0;
    std::vector<int> ncnt, ecnt, pnodes, pelem;
    std::vector<unsigned short> ptri, pfan;
};

struct nxs_dyn_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    nxs_dyn_params params{};
    DevParams dp{};
    bool have_mesh = false, have_state = false, have_forcing = false;
    DevMesh dm{};
    DevState ds{};
    DevWork dw{};
    DevPatches dpch{};
    int fused = 3;          // 3 (default): v3 (two sub-steps per launch) on single-rank meshes that live in the caches, else v2;
                            // 2: v3 wherever it is possible; 1: v2 fused sub-step kernel; 0: v1 two-kernel sub-step
    int pair_nodes = 0;     // v3: own nodes per patch; 0 = auto
    int pair_depth = 0;     // v3: sub-steps per launch, 2..NXS_MAX_DEPTH; 0 = auto
    int pair_depth_built = 0;
    bool pair_failed = false;   // the D-ring patches could not be built for this mesh: v2 instead
    DevPatches2 dpch2{};
    size_t pair_lds = 0;
    int pair_threads = 512;
    bool pair_ready = false;
    std::vector<void *> pair_allocs;
    int patch_nodes = 0;    // own nodes per patch; 0 = auto
    int um_ring = 0;        // fused path: apply the mesh move every um_ring sub-steps from a ring of VT buffers
                            // (1 = every sub-step; 0 = auto: once per step on meshes that stream from HBM, 1 on cache-resident ones)
    VTRing ring{};
    std::vector<void *> ring_allocs;
    int nt_mask = 3;        // non-temporal access classes of the fused kernel (1 sigma/damage, 2 UM/UT, 4 element constants)
    size_t fused_lds = 0;
    std::vector<int> h_t[3];               // kept for rebuilding patches when patch_nodes changes
    std::vector<unsigned char> h_ghost;
    std::vector<double> h_x0, h_y0;
    std::vector<void *> patch_allocs;
    std::vector<void *> mesh_allocs, state_allocs;
    // halo
    bool have_halo = false;
    int rank = 0, nranks = 1;
    std::vector<int> send_procs, send_offsets, recv_procs, recv_offsets;
    int *d_send_index = nullptr, *d_send_seg = nullptr, *d_send_off = nullptr;
    int *d_recv_index = nullptr, *d_recv_seg = nullptr, *d_recv_off = nullptr;
    double *d_send_buf = nullptr, *d_recv_buf = nullptr;
    std::vector<void *> halo_allocs;
    Rccl rccl;
    void *comm = nullptr;
    // device-direct transport (peer-mapped mailboxes)
    bool ipc_ready = false;
    IpcDev ipc{};
    void *ipc_block = nullptr;             // my mailbox allocation (exported)
    size_t ipc_block_bytes = 0;
    std::vector<void *> ipc_peer_base;     // opened peer mailboxes (to close)
    std::vector<void *> ipc_allocs;
    int *d_recv_procs = nullptr;
    // halo exchange fused into the sub-step kernel (device-direct transport + fused path)
    double *f_snap[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // wind0, wind1, ocean0, ocean1, ssh0, ssh1 (forcing pair)
    bool have_pair = false;
    std::vector<void *> forcing_allocs;
    int pin_host = 0;                      // option "pin_host": page-lock the caller's state / forcing vectors on first use
    std::map<const void *, size_t> pinned; // what this handle has registered with hipHostRegister
    int halo_fused = 1;                    // option "halo_fused"
    bool hf_ready = false;
    HaloFused hf{};
    std::vector<void *> hf_allocs;
    std::vector<int> h_send_index, h_recv_index;   // host copies of the halo lists
    std::shared_ptr<HostPatches> hp;  // host copy of the patches (re-uploaded boundary-first for the fused halo)
    nxs_dyn_halo_fn halo_fn = nullptr;  // host-staged exchange through the caller's communicator
    void *halo_ctx = nullptr;
    double *h_send = nullptr, *h_recv = nullptr;  // pinned staging buffers
    // reductions
    RegridPartial *d_partials = nullptr, *d_regrid = nullptr;
    int *d_crash = nullptr;
    int n_partials = 0;
    // graph of the sub-step loop
    int use_graph = 1;
    hipGraphExec_t substep_graph = nullptr, tail_graph = nullptr;
    bool graph_valid = false, tail_graph_valid = false;
    // timing: a ring of event sets so that steps can be enqueued back to back; a set is harvested
    // (its elapsed times added to the sums) when it is about to be reused or when timing is read
    static constexpr int NSETS = 8;
    hipEvent_t ev[NSETS][5] = {};
    bool set_pending[NSETS] = {};
    int set_next = 0;
    double sum_ms[4] = {0, 0, 0, 0};
    int sum_steps = 0;
    hipEvent_t *cur = nullptr;  // event set of the step being enqueued (nullptr: untimed)
    nxs_dyn_timing timing{};
    int timing_enabled = 1;
    std::string err;
};

namespace {

int fail(nxs_dyn_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(h, NXS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(nxs_dyn_handle *h, std::vector<void *> &pool, T **out, size_t count) {
    void *p = nullptr;
    HIPCHK(h, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    pool.push_back(p);
    *out = static_cast<T *>(p);
    return NXS_OK;
}

template <typename T>
int dev_upload(nxs_dyn_handle *h, std::vector<void *> &pool, const T **out, const std::vector<T> &v) {
    T *p = nullptr;
    int rc = dev_alloc(h, pool, &p, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(h, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // v may be a temporary of the caller
    *out = p;
    return NXS_OK;
}

void free_pool(std::vector<void *> &pool) {
    for (void *p : pool) (void)hipFree(p);
    pool.clear();
}

inline int nblocks(int n) { return n > 0 ? (n + BLOCK - 1) / BLOCK : 1; }

void derive_params(nxs_dyn_handle *h) {
    const nxs_dyn_params &p = h->params;
    DevParams &d = h->dp;
    d.dtime_step = p.dtime_step;
    d.substeps = p.substeps;
    d.dte = p.dtime_step / (double)p.substeps;  // FE.cpp:10185
    d.dynamics_type = p.dynamics_type;
    d.basal_stress_type = p.basal_stress_type;
    d.young_cat = p.ice_cat_type == NXS_ICECAT_YOUNG_ICE;
    d.newice_type = p.newice_type;
    d.equal_ridging = p.equal_ridging;
    d.use_young_myi = p.use_young_ice_in_myi_reset;
    d.young = p.young; d.nu0 = p.nu0; d.tan_phi = p.tan_phi; d.compr_strength = p.compr_strength;
    d.compaction_param = p.compaction_param;
    d.utrs = p.undamaged_time_relaxation_sigma;
    d.ers_m1 = p.exponent_relaxation_sigma - 1.;  // FE.cpp:4186
    d.ers_int = (d.ers_m1 >= 0. && d.ers_m1 <= 16. && d.ers_m1 == std::floor(d.ers_m1)) ? (int)d.ers_m1 : -1;
    d.compression_factor = p.compression_factor;
    d.ecf = p.exponent_compression_factor;
    d.min_h = p.min_h; d.min_c = p.min_c;
    d.min_m = NXS_RHOI * p.min_h;  // FE.cpp:10191
    d.qdw = p.quad_drag_coef_water; d.ldw = p.lin_drag_coef_water;
    d.qda = p.quad_drag_coef_air; d.lda = p.lin_drag_coef_air;
    d.cos_ota = std::cos(p.ocean_turning_angle_rad);  // FE.cpp:10187-10188
    d.sin_ota = std::sin(p.ocean_turning_angle_rad);
    d.k1 = p.basal_k1; d.k2 = p.basal_k2; d.Cb = p.basal_Cb; d.u0 = p.basal_u_0;
    d.evp_e = p.evp_e; d.evp_Pstar = p.evp_Pstar; d.evp_C = p.evp_C; d.evp_dmin = p.evp_dmin;
    d.mevp_beta = p.mevp_beta;
    if (p.dynamics_type == NXS_DYN_EVP) {  // FE.cpp:10705-10713
        const double T = p.dtime_step / 3.;
        d.ralpha1 = 0.5 * d.dte / T;
        d.ralpha2 = 0.5 * d.dte / T * p.evp_e * p.evp_e;
    } else {  // FE.cpp:10724
        d.ralpha1 = 1. / p.mevp_alpha;
        d.ralpha2 = 1. / p.mevp_alpha;
    }
    d.sqrt_nu_rhoi = std::sqrt(2. * (1. + p.nu0) * NXS_RHOI);  // FE.cpp:4140
    // initFETensors, FE.cpp:1491-1507
    for (double &x : d.D) x = 0.;
    const double Dunit_factor = 1. / (1. - p.nu0 * p.nu0);
    d.D[0] = Dunit_factor * 1.;
    d.D[1] = Dunit_factor * p.nu0;
    d.D[3] = Dunit_factor * p.nu0;
    d.D[4] = Dunit_factor * 1.;
    d.D[8] = Dunit_factor * (1. - p.nu0) / 2.;
    h->graph_valid = false;       // kernel arguments are baked into the graphs
    h->tail_graph_valid = false;
}

int check_params(nxs_dyn_handle *h, const nxs_dyn_params *p) {
    if (!p) return fail(h, NXS_ERR_INVALID, "params is NULL");
    if (!(p->dtime_step > 0.) || p->substeps < 1) return fail(h, NXS_ERR_INVALID, "dtime_step/substeps invalid");
    if (p->dynamics_type < NXS_DYN_BBM || p->dynamics_type > NXS_DYN_MEVP)
        return fail(h, NXS_ERR_INVALID, "unknown dynamics_type %d (FE.cpp:1352-1358 allows bbm|no_motion|free_drift|evp|mevp)", p->dynamics_type);
    if (p->basal_stress_type != NXS_BASAL_NONE && p->basal_stress_type != NXS_BASAL_LEMIEUX)
        return fail(h, NXS_ERR_INVALID, "unknown basal_stress_type %d", p->basal_stress_type);
    return NXS_OK;
}

int harvest(nxs_dyn_handle *h, int k) {
    if (!h->set_pending[k]) return NXS_OK;
    HIPCHK(h, hipEventSynchronize(h->ev[k][4]));
    for (int i = 0; i < 4; ++i) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev[k][i], h->ev[k][i + 1]));
        h->sum_ms[i] += ms;
    }
    h->sum_steps++;
    h->set_pending[k] = false;
    return NXS_OK;
}


// ------------------------------------------------------------------------------------------------
// Host: node patches for the fused sub-step kernel (see DevPatches).

// order: owned nodes in the order they are cut into patches of P.
bool build_patches_from_order(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int No, int P,
                              const std::vector<int> &order, HostPatches &out) {
    // node -> elements CSR
    std::vector<int> off(Nn + 1, 0);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < Ne; ++e) off[t[k][e] + 1]++;
    for (int n = 0; n < Nn; ++n) off[n + 1] += off[n];
    std::vector<int> adj(off[Nn]), fill(off.begin(), off.end() - 1);
    for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) adj[fill[t[k][e]]++] = e;  // ascending e per node

    const int nNodePatches = (No + P - 1) / P;
    std::vector<int> patch_of(Nn, -1);
    for (int i = 0; i < No; ++i) patch_of[order[i]] = i / P;
    // writer patch of an element = smallest patch id among its owned nodes; none -> orphan
    std::vector<int> writer(Ne, -1);
    std::vector<int> orphans;
    for (int e = 0; e < Ne; ++e) {
        int w = -1;
        for (int k = 0; k < 3; ++k) {
            const int q = patch_of[t[k][e]];
            if (q >= 0 && (w < 0 || q < w)) w = q;
        }
        writer[e] = w;
        if (w < 0) orphans.push_back(e);
    }
    const int EORPH = 2 * P;
    const int nOrphPatches = ((int)orphans.size() + EORPH - 1) / EORPH;
    const int nP = nNodePatches + nOrphPatches;

    std::vector<std::vector<int>> pel(nP), pnd(nP);
    std::vector<int> own_cnt(nP, 0);
    std::vector<int> mark(Ne, -1), slot_of(Nn, -1);
    size_t tot_e = 0;
    for (int q = 0; q < nNodePatches; ++q) {
        const int a = q * P, bnd = std::min(No, a + P);
        own_cnt[q] = bnd - a;
        auto &el = pel[q];
        for (int i = a; i < bnd; ++i) {
            const int n = order[i];
            for (int j = off[n]; j < off[n + 1]; ++j) {
                const int e = adj[j];
                if (mark[e] != q) { mark[e] = q; el.push_back(e); }
            }
        }
        std::sort(el.begin(), el.end());
        tot_e += el.size();
    }
    for (int q = 0; q < nOrphPatches; ++q) {
        auto &el = pel[nNodePatches + q];
        const int a = q * EORPH, bnd = std::min((int)orphans.size(), a + EORPH);
        el.assign(orphans.begin() + a, orphans.begin() + bnd);  // already ascending
    }
    int Emax = 0, Mmax = 0, Wp = 0, Pmax = 0;
    std::vector<std::vector<unsigned short>> tri_l(nP);
    std::vector<std::vector<std::vector<unsigned short>>> fan_l(nP);
    for (int q = 0; q < nP; ++q) {
        auto &nd = pnd[q];
        if (q < nNodePatches) {
            const int a = q * P;
            for (int i = 0; i < own_cnt[q]; ++i) { nd.push_back(order[a + i]); slot_of[order[a + i]] = i; }
        }
        std::vector<int> halo;
        for (int e : pel[q])
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e];
                if (slot_of[n] == -1) { slot_of[n] = -2; halo.push_back(n); }
            }
        std::sort(halo.begin(), halo.end());
        for (int n : halo) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
        if (nd.size() > 65535 || pel[q].size() > 8191) return false;
        auto &tl = tri_l[q];
        tl.resize(4 * pel[q].size());
        auto &fl = fan_l[q];
        fl.assign(own_cnt[q], {});
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e], sl = slot_of[n];
                tl[4 * l + k] = (unsigned short)sl;
                if (sl < own_cnt[q]) fl[sl].push_back((unsigned short)((l << 3) | (ghost3[3 * (size_t)e + k] ? 4 : 0) | k));
            }
            tl[4 * l + 3] = 0;
        }
        for (auto &f : fl) Wp = std::max(Wp, (int)f.size());
        for (int n : nd) slot_of[n] = -1;
        Emax = std::max(Emax, (int)pel[q].size());
        Mmax = std::max(Mmax, (int)nd.size());
        Pmax = std::max(Pmax, own_cnt[q]);
    }
    Emax = (Emax + 1) & ~1;  // keep the ushort4 / double rows 16-byte aligned
    Mmax = (Mmax + 1) & ~1;
    Pmax = std::max(Pmax, 1);
    Wp = std::max(Wp, 1);
    out = HostPatches{};
    out.nP = nP; out.Pmax = Pmax; out.Emax = Emax; out.Mmax = Mmax; out.Wp = Wp;
    out.own_cnt = own_cnt;
    out.elem_cnt.resize(nP); out.node_cnt.resize(nP);
    out.pnodes.assign((size_t)nP * Mmax, 0);
    out.pelem.assign((size_t)nP * Emax, 0);
    out.ptri.assign((size_t)nP * Emax * 4, 0);
    out.pfan.assign((size_t)nP * Wp * Pmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        out.elem_cnt[q] = (int)pel[q].size();
        out.node_cnt[q] = (int)pnd[q].size();
        std::copy(pnd[q].begin(), pnd[q].end(), out.pnodes.begin() + (size_t)q * Mmax);
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            const bool is_writer = (writer[e] == q) || (writer[e] < 0);  // orphans are written by their orphan patch
            out.pelem[(size_t)q * Emax + l] = is_writer ? e : ~e;
        }
        std::copy(tri_l[q].begin(), tri_l[q].end(), out.ptri.begin() + (size_t)q * Emax * 4);
        for (int i = 0; i < own_cnt[q]; ++i)
            for (size_t k = 0; k < fan_l[q][i].size(); ++k)
                out.pfan[(size_t)q * Wp * Pmax + k * Pmax + i] = fan_l[q][i][k];
    }
    out.avg_elems_per_own_node = No > 0 ? (double)tot_e / No : 0.;
    return true;
}

// owned nodes sorted along a Hilbert curve through their coordinates
void hilbert_order(const double *x0, const double *y0, int No, std::vector<int> &order) {
    order.resize(No);
    for (int i = 0; i < No; ++i) order[i] = i;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int n = 0; n < No; ++n) { xmin = std::min(xmin, x0[n]); xmax = std::max(xmax, x0[n]); ymin = std::min(ymin, y0[n]); ymax = std::max(ymax, y0[n]); }
    const double ext = std::max(xmax - xmin, ymax - ymin);
    const double sc = ext > 0. ? 65535. / ext : 0.;
    auto hilbert = [](unsigned x, unsigned y) {
        unsigned long long d = 0;
        for (unsigned s2 = 1u << 15; s2 > 0; s2 >>= 1) {
            const unsigned rx = (x & s2) ? 1u : 0u, ry = (y & s2) ? 1u : 0u;
            d += (unsigned long long)s2 * s2 * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) { x = s2 - 1 - x; y = s2 - 1 - y; }
                const unsigned t2 = x; x = y; y = t2;
            }
        }
        return d;
    };
    std::vector<unsigned long long> key(No);
    for (int n = 0; n < No; ++n) key[n] = hilbert((unsigned)((x0[n] - xmin) * sc), (unsigned)((y0[n] - ymin) * sc));
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return key[a] < key[b2]; });
}

bool build_patches(const std::vector<int> t[3], const unsigned char *ghost3, const double *x0, const double *y0, int Nn, int Ne,
                   int No, int P, HostPatches &out) {
    // 1st try: the caller's node numbering (keeps the patch's nodal accesses contiguous)
    std::vector<int> order(No);
    for (int i = 0; i < No; ++i) order[i] = i;
    bool ok = build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, out);
    if (ok && out.avg_elems_per_own_node <= 3.0) return true;
    // numbering without locality: cut patches along a Hilbert curve through the node coordinates
    // (consecutive runs of a Hilbert curve are compact blobs: small halos)
    hilbert_order(x0, y0, No, order);
    HostPatches alt;
    if (build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, alt) && (!ok || alt.avg_elems_per_own_node < out.avg_elems_per_own_node)) {
        out = std::move(alt);
        out.used_hilbert = true;
        return true;
    }
    return ok;
}

// Host: D-ring patches of k_substep_multi (DevPatches2); single rank (every node owned, no orphan elements).
bool build_patches2(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int P, int D, const std::vector<int> &order, HostPatches2 &out) {
    std::vector<int> off(Nn + 1, 0);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < Ne; ++e) off[t[k][e] + 1]++;
    for (int n = 0; n < Nn; ++n) off[n + 1] += off[n];
    std::vector<int> adj(off[Nn]), fill(off.begin(), off.end() - 1);
    for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) adj[fill[t[k][e]]++] = e;  // ascending e per node
    const int nP = (Nn + P - 1) / P;
    std::vector<int> patch_of(Nn, -1);
    for (int i = 0; i < Nn; ++i) patch_of[order[i]] = i / P;
    std::vector<int> writer(Ne);
    for (int e = 0; e < Ne; ++e) writer[e] = std::min({patch_of[t[0][e]], patch_of[t[1][e]], patch_of[t[2][e]]});

    out = HostPatches2{};
    out.nP = nP; out.D = D;
    out.ncnt.assign((size_t)nP * (D + 1), 0); out.ecnt.assign((size_t)nP * D, 0);
    std::vector<std::vector<int>> pel(nP), pnd(nP);
    std::vector<std::vector<unsigned short>> tri_l(nP);
    std::vector<std::vector<std::vector<unsigned short>>> fan_l(nP);
    std::vector<int> emark(Ne, -1), eslot(Ne, -1), slot_of(Nn, -1);
    for (int q = 0; q < nP; ++q) {
        const int a = q * P, bnd = std::min(Nn, a + P);
        auto &nd = pnd[q];
        auto &el = pel[q];
        int *nc = out.ncnt.data() + (size_t)q * (D + 1), *ec = out.ecnt.data() + (size_t)q * D;
        for (int i = a; i < bnd; ++i) { slot_of[order[i]] = (int)nd.size(); nd.push_back(order[i]); }
        nc[0] = bnd - a;
        int n_prev = 0, e_prev = 0;
        for (int lev = 1; lev <= D; ++lev) {
            // E_lev: the elements touching N_(lev-1) that are not listed yet, ascending
            std::vector<int> add;
            for (int i = n_prev; i < nc[lev - 1]; ++i)
                for (int j = off[nd[i]]; j < off[nd[i] + 1]; ++j) {
                    const int e = adj[j];
                    if (emark[e] != q) { emark[e] = q; add.push_back(e); }
                }
            std::sort(add.begin(), add.end());
            el.insert(el.end(), add.begin(), add.end());
            ec[lev - 1] = (int)el.size();
            // N_lev: their nodes that are not listed yet, ascending
            std::vector<int> addn;
            for (int l = e_prev; l < ec[lev - 1]; ++l)
                for (int k = 0; k < 3; ++k) {
                    const int n = t[k][el[l]];
                    if (slot_of[n] == -1) { slot_of[n] = -2; addn.push_back(n); }
                }
            std::sort(addn.begin(), addn.end());
            for (int n : addn) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
            nc[lev] = (int)nd.size();
            n_prev = nc[lev - 1]; e_prev = ec[lev - 1];
        }
        if (nd.size() > 65535 || el.size() > 8191) return false;
        for (size_t l = 0; l < el.size(); ++l) eslot[el[l]] = (int)l;
        auto &tl = tri_l[q];
        tl.assign(4 * el.size(), 0);
        for (size_t l = 0; l < el.size(); ++l)
            for (int k = 0; k < 3; ++k) tl[4 * l + k] = (unsigned short)slot_of[t[k][el[l]]];
        const int nsolved = nc[D - 1];
        auto &fl = fan_l[q];
        fl.assign(nsolved, {});
        for (int i = 0; i < nsolved; ++i) {
            const int n = nd[i];
            for (int j = off[n]; j < off[n + 1]; ++j) {  // ascending element id = the order of the serial scatter
                const int e = adj[j];
                int k = 0;
                while (t[k][e] != n) ++k;
                fl[i].push_back((unsigned short)((eslot[e] << 3) | (ghost3[3 * (size_t)e + k] ? 4 : 0) | k));
            }
            out.Wp = std::max(out.Wp, (int)fl[i].size());
        }
        for (int n : nd) slot_of[n] = -1;
        out.NDmax = std::max(out.NDmax, nc[D]); out.NSmax = std::max(out.NSmax, nc[D - 1]);
        out.EDmax = std::max(out.EDmax, ec[D - 1]); out.ESmax = std::max(out.ESmax, D >= 2 ? ec[D - 2] : 0);
    }
    out.NDmax = (out.NDmax + 1) & ~1; out.NSmax = (out.NSmax + 1) & ~1; out.EDmax = (out.EDmax + 1) & ~1; out.ESmax = std::max(2, (out.ESmax + 1) & ~1);
    out.Wp = std::max(out.Wp, 1);
    out.pnodes.assign((size_t)nP * out.NDmax, 0);
    out.pelem.assign((size_t)nP * out.EDmax, 0);
    out.ptri.assign((size_t)nP * out.EDmax * 4, 0);
    out.pfan.assign((size_t)nP * out.Wp * out.NSmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        std::copy(pnd[q].begin(), pnd[q].end(), out.pnodes.begin() + (size_t)q * out.NDmax);
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            out.pelem[(size_t)q * out.EDmax + l] = (writer[e] == q) ? e : ~e;
        }
        std::copy(tri_l[q].begin(), tri_l[q].end(), out.ptri.begin() + (size_t)q * out.EDmax * 4);
        for (size_t i = 0; i < fan_l[q].size(); ++i)
            for (size_t k = 0; k < fan_l[q][i].size(); ++k)
                out.pfan[(size_t)q * out.Wp * out.NSmax + k * out.NSmax + i] = fan_l[q][i][k];
    }
    return true;
}

int upload_patches2(nxs_dyn_handle *h, int D, bool single_round_only) {
    free_pool(h->pair_allocs);
    h->dpch2 = DevPatches2{};
    h->pair_ready = false;
    const DevMesh &m = h->dm;
    if (m.No != m.Nn) return fail(h, NXS_ERR_STATE, "multi-sub-step patches need a single-rank mesh");
    std::vector<int> order(m.Nn);
    for (int i = 0; i < m.Nn; ++i) order[i] = i;
    // the caller's numbering if it has locality, else the Hilbert curve the single-ring patches were cut along
    if (h->hp && h->hp->used_hilbert) hilbert_order(h->h_x0.data(), h->h_y0.data(), m.Nn, order);
    HostPatches2 hp;
    auto lds_of = [](const HostPatches2 &x) { return (4 * (size_t)x.NDmax + 6 * (size_t)x.EDmax + 4 * (size_t)x.ESmax) * sizeof(double); };
    int P = 0, threads = 512;
    if (h->pair_nodes > 0) {
        P = h->pair_nodes;
        if (!build_patches2(h->h_t, h->h_ghost.data(), m.Nn, m.Ne, P, D, order, hp)) return fail(h, NXS_ERR_INVALID, "multi-sub-step patch construction failed (pair_nodes=%d)", P);
    } else {
        // as upload_patches: whole rounds of resident workgroups -- j workgroups per CU at a time, j = 1 first (a small mesh
        // is fastest with ONE workgroup on every CU: 10 km, 247 patches of 120 nodes 1.06 ms/step, 265 patches of 112 nodes
        // 1.30)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        cus = std::max(cus, 1);
        bool done = false;
        for (int j = 1; j <= (single_round_only ? 1 : 512) && !done; ++j) {
            P = (int)(((long long)m.Nn + (long long)j * cus - 1) / ((long long)j * cus));
            P = std::max(32, (P + 3) & ~3);
            if (P > 256) continue;
            if (!build_patches2(h->h_t, h->h_ghost.data(), m.Nn, m.Ne, P, D, order, hp)) continue;
            const size_t lds_cap = (j == 1 ? 160 : 80) * 1024;  // one workgroup per CU may take it all; otherwise two must fit
            done = lds_of(hp) <= lds_cap && (hp.nP <= j * cus || P == 32);
        }
        if (!done) return fail(h, NXS_ERR_INVALID, single_round_only ? "the mesh does not fit one multi-sub-step patch per CU" : "no multi-sub-step patch size fits (node numbering without locality?)");
    }
    h->pair_lds = lds_of(hp);
    if (h->pair_lds > 160 * 1024) return fail(h, NXS_ERR_INVALID, "multi-sub-step patches need %zu B of LDS", h->pair_lds);
    // 512 threads at most: a 1 024-thread workgroup runs this kernel at half the speed (10 km, D = 2: 1.98 vs 1.11 ms/step); the
    // outer levels of a deep patch take a second round of the block instead
    threads = hp.EDmax <= 256 ? 256 : 512;
    h->pair_threads = threads;
    if (getenv("NXS_DEBUG_PATCHES")) {
        std::vector<double> se(D, 0.), sn(D + 1, 0.);
        for (int q = 0; q < hp.nP; ++q) { for (int i = 0; i < D; ++i) se[i] += hp.ecnt[(size_t)q * D + i]; for (int i = 0; i <= D; ++i) sn[i] += hp.ncnt[(size_t)q * (D + 1) + i]; }
        fprintf(stderr, "[nxs] multi patches: D=%d P=%d nP=%d EDmax=%d ESmax=%d NDmax=%d NSmax=%d Wp=%d lds=%zu B threads=%d; elements per level x", D, P, hp.nP, hp.EDmax, hp.ESmax,
                hp.NDmax, hp.NSmax, hp.Wp, h->pair_lds, threads);
        for (int i = 0; i < D; ++i) fprintf(stderr, " %.3f", se[i] / std::max(m.Ne, 1));
        fprintf(stderr, "; nodes per level x");
        for (int i = 0; i <= D; ++i) fprintf(stderr, " %.3f", sn[i] / std::max(m.Nn, 1));
        fprintf(stderr, "\n");
    }
    DevPatches2 &d = h->dpch2;
    d.nP = hp.nP; d.D = D; d.NDmax = hp.NDmax; d.NSmax = hp.NSmax; d.EDmax = hp.EDmax; d.ESmax = hp.ESmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ncnt, hp.ncnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ecnt, hp.ecnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pfan, hp.pfan))) return rc;
    h->pair_ready = true;
    h->pair_depth_built = D;
    return NXS_OK;
}

int upload_host_patches(nxs_dyn_handle *h, const HostPatches &hp) {
    free_pool(h->patch_allocs);
    DevPatches &d = h->dpch;
    d = DevPatches{};
    d.nP = hp.nP; d.Pmax = hp.Pmax; d.Emax = hp.Emax; d.Mmax = hp.Mmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.own_cnt, hp.own_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.elem_cnt, hp.elem_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.node_cnt, hp.node_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pfan, hp.pfan))) return rc;
    return NXS_OK;
}

int upload_patches(nxs_dyn_handle *h) {
    free_pool(h->patch_allocs);
    h->dpch = DevPatches{};
    h->fused_lds = 0;
    const DevMesh &m = h->dm;
    const bool automatic = h->patch_nodes <= 0;
    HostPatches hp;
    int P = 0;
    auto build = [&](int PP) -> bool {
        if (!build_patches(h->h_t, h->h_ghost.data(), h->h_x0.data(), h->h_y0.data(), m.Nn, m.Ne, m.No, PP, hp)) return false;
        h->fused_lds = (4 * (size_t)hp.Mmax + 6 * (size_t)hp.Emax) * sizeof(double);
        return true;
    };
    if (!automatic) {
        P = std::max(64, std::min(h->patch_nodes, 1024));
        for (;;) {
            if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
            if (h->fused_lds <= 80 * 1024 || P <= 64) break;
            P = std::max(64, P * 3 / 4);
        }
    } else {
        // Large patches recompute few halo elements; the limits are the LDS of two resident workgroups per CU
        // (160 KiB / 2) and, above all, WHOLE ROUNDS: the grid runs in rounds of `slots` resident workgroups and a
        // last round that is partly empty costs as much as a full one.  So: the smallest number of rounds k whose
        // patch size ceil(No / (k*slots)) fits, e.g. 730 k nodes -> 3 rounds of 512 patches of 476 nodes (not 2.79
        // rounds of 512-node patches); 92 k nodes (one rank of eight) -> one round of 511 patches of 180 nodes.
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        cus = std::max(cus, 1);
        const int slots512 = 2 * cus, slots256 = 4 * cus;  // 16 waves per CU (112 VGPRs): 2 x 512 or 4 x 256 threads
        bool done = false;
        for (int k = 1; k <= 64 && !done; ++k) {
            P = (int)(((long long)m.No + (long long)k * slots512 - 1) / ((long long)k * slots512));
            P = (P + 3) & ~3;
            if (P > 512) continue;
            if (P <= NXS_T256_MAXP) break;  // small mesh: the 256-thread kernel below
            for (int it = 0; it < 4 && !done; ++it) {  // orphan patches (multi-rank) may add a few workgroups
                if (it > 0) P += 4;
                if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
                if (h->fused_lds > 80 * 1024) break;            // does not fit twice: more rounds of smaller patches
                done = hp.nP <= k * slots512;
            }
        }
        if (!done) {
            P = (int)(((long long)m.No + slots256 - 1) / slots256);
            P = std::max(64, std::min((P + 3) & ~3, NXS_T256_MAXP));
            if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
        }
    }
    if (h->fused_lds > 160 * 1024) return fail(h, NXS_ERR_INVALID, "patches need %zu B of LDS", h->fused_lds);
    if (getenv("NXS_DEBUG_PATCHES")) {
        long long se = 0, sm = 0;
        for (int q = 0; q < hp.nP; ++q) { se += hp.elem_cnt[q]; sm += hp.node_cnt[q]; }
        fprintf(stderr, "[nxs] patches: P=%d nP=%d Pmax=%d Emax=%d Mmax=%d Wp=%d avgE=%.1f avgM=%.1f lds=%zu B elems x%.3f\n", P, hp.nP, hp.Pmax,
                hp.Emax, hp.Mmax, hp.Wp, (double)se / hp.nP, (double)sm / hp.nP, h->fused_lds, (double)se / std::max(m.Ne, 1));
    }
    h->hf_ready = false;
    h->hp = std::make_shared<HostPatches>(std::move(hp));
    return upload_host_patches(h, *h->hp);
}

void ipc_release(nxs_dyn_handle *h) {
    for (void *p : h->ipc_peer_base) if (p) (void)hipIpcCloseMemHandle(p);
    h->ipc_peer_base.clear();
    free_pool(h->ipc_allocs);
    if (h->ipc_block) { (void)hipFree(h->ipc_block); h->ipc_block = nullptr; }
    h->ipc_ready = false;
    h->ipc = IpcDev{};
}
// option "pin_host": the caller's vectors (FiniteElement's M_VT, M_conc, ... live as long as the mesh) are page-locked the first
// time they are seen, so that the per-step copies of a host-side thermodynamics run at PCIe speed and overlap; a vector that was
// reallocated simply registers anew, stale registrations are dropped at set_mesh / destroy.  Failure to register is not an error.
void pin_host_buffer(nxs_dyn_handle *h, const void *p, size_t bytes) {
    if (!h->pin_host || !p || bytes == 0) return;
    auto it = h->pinned.find(p);
    if (it != h->pinned.end() && it->second >= bytes) return;
    if (it != h->pinned.end()) { (void)hipHostUnregister(const_cast<void *>(p)); h->pinned.erase(it); }
    if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) h->pinned[p] = bytes;
    else (void)hipGetLastError();
}
void unpin_all(nxs_dyn_handle *h) {
    for (auto &kv : h->pinned) (void)hipHostUnregister(const_cast<void *>(kv.first));
    (void)hipGetLastError();
    h->pinned.clear();
}

void release_graph(nxs_dyn_handle *h) {
    if (h->substep_graph) { (void)hipGraphExecDestroy(h->substep_graph); h->substep_graph = nullptr; }
    if (h->tail_graph) { (void)hipGraphExecDestroy(h->tail_graph); h->tail_graph = nullptr; }
    h->graph_valid = false;
    h->tail_graph_valid = false;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int nxs_dyn_abi_version(void) { return NXS_DYN_ABI_VERSION; }

const char *nxs_dyn_last_error(const nxs_dyn_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nxs_dyn_default_params(nxs_dyn_params *p) {  // model/options.cpp:43,80,109-111,314-376,397,545-547
    if (!p) return NXS_ERR_INVALID;
    std::memset(p, 0, sizeof *p);
    p->dtime_step = 200.; p->substeps = 120;
    p->dynamics_type = NXS_DYN_BBM; p->basal_stress_type = NXS_BASAL_LEMIEUX; p->ice_cat_type = NXS_ICECAT_YOUNG_ICE;
    p->newice_type = 4; p->equal_ridging = 0; p->use_young_ice_in_myi_reset = 1;
    p->young = 5.9605e+08; p->nu0 = 1. / 3.; p->tan_phi = 0.7; p->compr_strength = 1e10; p->compaction_param = -20.;
    p->undamaged_time_relaxation_sigma = 1e7; p->exponent_relaxation_sigma = 5.;
    p->compression_factor = 10e3; p->exponent_compression_factor = 1.5;
    p->min_h = 0.05; p->min_c = 0.01;
    p->quad_drag_coef_water = 0.0055; p->lin_drag_coef_water = 0.; p->quad_drag_coef_air = 0.0049; p->lin_drag_coef_air = 0.;
    p->ocean_turning_angle_rad = (NXS_PI / 180.) * 25.;
    p->basal_k1 = 10.; p->basal_k2 = 15.; p->basal_Cb = 20.; p->basal_u_0 = 5e-5;
    p->evp_e = 2.; p->evp_Pstar = 27.5e3; p->evp_C = 20.; p->evp_dmin = 1e-9;
    p->mevp_alpha = 500.; p->mevp_beta = 500.;
    p->regrid_angle = 10.;
    return NXS_OK;
}

int nxs_dyn_create(const nxs_dyn_params *p, int device, nxs_dyn_handle **out) {
    if (!out) return fail(nullptr, NXS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NXS_ERR_NO_DEVICE, "no HIP device visible (%s): libnxsdyn has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, NXS_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    nxs_dyn_handle *h = new nxs_dyn_handle();
    int rc = check_params(h, p);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    h->device = device;
    h->params = *p;
#define CREATE_CHK(call)                                                                       \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            fail(nullptr, NXS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e));           \
            delete h;                                                                           \
            return NXS_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto &set : h->ev) for (auto &ev : set) CREATE_CHK(hipEventCreate(&ev));
    CREATE_CHK(hipMalloc((void **)&h->d_regrid, sizeof(RegridPartial)));
    CREATE_CHK(hipMalloc((void **)&h->d_crash, sizeof(int)));
#undef CREATE_CHK
    derive_params(h);
    *out = h;
    return NXS_OK;
}

int nxs_dyn_destroy(nxs_dyn_handle *h) {
    if (!h) return NXS_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    release_graph(h);
    if (h->comm && h->rccl.CommDestroy) h->rccl.CommDestroy(h->comm);
    ipc_release(h);
    free_pool(h->mesh_allocs);
    free_pool(h->state_allocs);
    free_pool(h->halo_allocs);
    free_pool(h->patch_allocs);
    free_pool(h->pair_allocs);
    h->pair_ready = false;
    h->pair_failed = false;
    free_pool(h->ring_allocs);
    free_pool(h->hf_allocs);
    h->hf_ready = false;
    free_pool(h->forcing_allocs);
    for (auto &q : h->f_snap) q = nullptr;
    h->have_pair = false;
    unpin_all(h);
    if (h->h_send) (void)hipHostFree(h->h_send);
    if (h->h_recv) (void)hipHostFree(h->h_recv);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_regrid) (void)hipFree(h->d_regrid);
    if (h->d_crash) (void)hipFree(h->d_crash);
    for (auto &set : h->ev) for (auto &ev : set) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return NXS_OK;
}

int nxs_dyn_set_params(nxs_dyn_handle *h, const nxs_dyn_params *p) {
    if (!h) return NXS_ERR_INVALID;
    int rc = check_params(h, p);
    if (rc) return rc;
    h->params = *p;
    derive_params(h);
    return NXS_OK;
}

int nxs_dyn_set_option(nxs_dyn_handle *h, const char *key, int64_t value) {
    if (!h || !key) return NXS_ERR_INVALID;
    if (!std::strcmp(key, "graph")) { h->use_graph = value != 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "timing")) { h->timing_enabled = value != 0; return NXS_OK; }
    if (!std::strcmp(key, "um_ring")) {
        if (value < 0 || value > NXS_MAX_RING - 1) return fail(h, NXS_ERR_INVALID, "um_ring must be in [0,%d]", NXS_MAX_RING - 1);
        h->um_ring = (int)value; release_graph(h);
        return NXS_OK;
    }
    if (!std::strcmp(key, "nt_mask")) { h->nt_mask = (int)value; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "fused")) {
        if (value < 0 || value > 3) return fail(h, NXS_ERR_INVALID, "fused must be 0, 1, 2 or 3");
        h->fused = (int)value; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "substeps_per_launch")) {
        if (value != 0 && (value < 2 || value > NXS_MAX_DEPTH)) return fail(h, NXS_ERR_INVALID, "substeps_per_launch must be 0 (auto) or in [2,%d]", NXS_MAX_DEPTH);
        h->pair_depth = (int)value; h->pair_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "pair_nodes")) {
        if (value != 0 && (value < 16 || value > 512)) return fail(h, NXS_ERR_INVALID, "pair_nodes must be 0 (auto) or in [16,512]");
        h->pair_nodes = (int)value; h->pair_ready = false; h->pair_failed = false; release_graph(h); return NXS_OK;
    }
    if (!std::strcmp(key, "pin_host")) { h->pin_host = value != 0; if (!h->pin_host) unpin_all(h); return NXS_OK; }
    if (!std::strcmp(key, "halo_fused")) { h->halo_fused = value != 0; release_graph(h); return NXS_OK; }
    if (!std::strcmp(key, "patch_nodes")) {
        if (value != 0 && (value < 64 || value > 1024)) return fail(h, NXS_ERR_INVALID, "patch_nodes must be 0 (auto) or in [64,1024]");
        h->patch_nodes = (int)value;
        release_graph(h);
        if (h->have_mesh) { HIPCHK(h, hipSetDevice(h->device)); HIPCHK(h, hipStreamSynchronize(h->stream)); return upload_patches(h); }
        return NXS_OK;
    }
    if (!std::strcmp(key, "timing_reset")) {  // drop what was accumulated so far (e.g. after warm-up)
        for (int k = 0; k < nxs_dyn_handle::NSETS; ++k) { int rc = harvest(h, k); if (rc) return rc; }
        for (double &x : h->sum_ms) x = 0.;
        h->sum_steps = 0;
        return NXS_OK;
    }
    return fail(h, NXS_ERR_INVALID, "unknown option '%s'", key);
}

// ------------------------------------------------------------------------------------------------
int nxs_dyn_set_mesh(nxs_dyn_handle *h, const nxs_dyn_mesh *m) {
    if (!h || !m) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    const int Nn = m->num_nodes, Ne = m->num_elements, No = m->local_ndof, Neo = m->local_nelements;
    if (Nn <= 0 || Ne <= 0 || No < 0 || No > Nn || Neo < 0 || Neo > Ne)
        return fail(h, NXS_ERR_INVALID, "mesh sizes invalid: Nn=%d Ne=%d No=%d Neo=%d", Nn, Ne, No, Neo);
    if (Ne >= (1 << 28)) return fail(h, NXS_ERR_INVALID, "too many elements for the packed fan entries");
    if (!m->indices || !m->ghost_nodes || !m->coord_x || !m->coord_y || !m->lat || !m->mask_dirichlet)
        return fail(h, NXS_ERR_INVALID, "mesh has NULL arrays");
    if (m->num_neumann_flags < 0 || (m->num_neumann_flags > 0 && !m->neumann_flags))
        return fail(h, NXS_ERR_INVALID, "neumann_flags invalid");
    for (int64_t i = 0; i < 3ll * Ne; ++i)
        if (m->indices[i] < 1 || m->indices[i] > Nn)
            return fail(h, NXS_ERR_INVALID, "indices[%lld]=%d outside [1,%d] (1-based local ids expected)", (long long)i, m->indices[i], Nn);
    for (int i = 0; i < m->num_neumann_flags; ++i) {
        if (m->neumann_flags[i] < 0 || m->neumann_flags[i] >= Nn) return fail(h, NXS_ERR_INVALID, "neumann_flags[%d] out of range", i);
        if (i > 0 && m->neumann_flags[i] <= m->neumann_flags[i - 1]) return fail(h, NXS_ERR_INVALID, "neumann_flags must be sorted and unique (FE.cpp:251-252)");
    }

    if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    free_pool(h->mesh_allocs);
    free_pool(h->state_allocs);
    free_pool(h->halo_allocs);
    free_pool(h->patch_allocs);
    free_pool(h->pair_allocs);
    h->pair_ready = false;
    h->pair_failed = false;
    free_pool(h->ring_allocs);
    free_pool(h->hf_allocs);
    h->hf_ready = false;
    free_pool(h->forcing_allocs);
    for (auto &q : h->f_snap) q = nullptr;
    h->have_pair = false;
    unpin_all(h);
    h->ring = VTRing{};
    h->have_mesh = h->have_state = h->have_forcing = h->have_halo = false;
    h->rank = 0; h->nranks = 1;
    h->send_procs.clear(); h->recv_procs.clear(); h->send_offsets.assign(1, 0); h->recv_offsets.assign(1, 0);

    DevMesh &d = h->dm;
    d = DevMesh{};
    d.Nn = Nn; d.Ne = Ne; d.No = No; d.Neo = Neo;
    int rc;
    // triangles, SoA, 0-based
    std::vector<int> *t = h->h_t;
    for (int k = 0; k < 3; ++k) { t[k].resize(Ne); for (int e = 0; e < Ne; ++e) t[k][e] = m->indices[3 * e + k] - 1; }
    h->h_ghost.assign(m->ghost_nodes, m->ghost_nodes + 3 * (size_t)Ne);
    h->h_x0.assign(m->coord_x, m->coord_x + Nn);
    h->h_y0.assign(m->coord_y, m->coord_y + Nn);
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t0, t[0]))) return rc;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t1, t[1]))) return rc;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.t2, t[2]))) return rc;
    // node flags
    std::vector<unsigned char> nf(Nn, 0);
    for (int n = 0; n < Nn; ++n) if (m->mask_dirichlet[n]) nf[n] |= NF_DIRICHLET;
    for (int n = 0; n < Nn; ++n) if (std::signbit(m->lat[n])) nf[n] |= NF_LAT_NEG;
    for (int i = 0; i < m->num_neumann_flags; ++i) nf[m->neumann_flags[i]] |= NF_NEUMANN;
    if ((rc = dev_upload(h, h->mesh_allocs, &d.nflags, nf))) return rc;
    // element flags
    std::vector<unsigned char> ef(Ne, 0);
    for (int e = 0; e < Ne; ++e) {
        for (int k = 0; k < 3; ++k) {
            if (m->ghost_nodes[3 * e + k]) ef[e] |= (1 << k);
            if (nf[t[k][e]] & NF_NEUMANN) ef[e] |= EF_ON_NEUMANN;
        }
    }
    if ((rc = dev_upload(h, h->mesh_allocs, &d.eflags, ef))) return rc;
    // coordinates
    {
        std::vector<double> tmp(m->coord_x, m->coord_x + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.x0, tmp))) return rc;
        tmp.assign(m->coord_y, m->coord_y + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.y0, tmp))) return rc;
        tmp.assign(m->lat, m->lat + Nn);
        if ((rc = dev_upload(h, h->mesh_allocs, &d.lat, tmp))) return rc;
    }
    // ascending element fan of every node (ELL, slot-major)
    {
        std::vector<int> deg(Nn, 0);
        for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) deg[t[k][e]]++;
        int W = 0;
        for (int n = 0; n < Nn; ++n) W = std::max(W, deg[n]);
        std::vector<int> fan((size_t)W * Nn, -1), fill(Nn, 0);
        for (int e = 0; e < Ne; ++e)  // ascending e => ascending rows
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e];
                fan[(size_t)(fill[n]++) * Nn + n] = (e << 3) | (m->ghost_nodes[3 * e + k] ? 4 : 0) | k;
            }
        d.W = W;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.fan, fan))) return rc;
    }
    // bamg tables (given, or built with identical ordering)
    {
        std::vector<double> nec_own, nc_own;
        const double *nec = m->nodal_element_connectivity, *nc = m->nodal_connectivity;
        int w1 = m->nec_width, w2 = m->nc_width;
        if (!nec || !nc) {
            int bw1 = 0, bw2 = 0;
            if (nxs_mesh_connectivity(m->indices, Nn, Ne, &bw1, nullptr, &bw2, nullptr)) return fail(h, NXS_ERR_INVALID, "connectivity build failed");
            nec_own.resize((size_t)bw1 * Nn); nc_own.resize((size_t)bw2 * Nn);
            nxs_mesh_connectivity(m->indices, Nn, Ne, &bw1, nec_own.data(), &bw2, nc_own.data());
            if (!nec) { nec = nec_own.data(); w1 = bw1; }
            if (!nc) { nc = nc_own.data(); w2 = bw2; }
        }
        if (w1 <= 0 || w2 <= 1) return fail(h, NXS_ERR_INVALID, "connectivity widths invalid (%d, %d)", w1, w2);
        std::vector<int> n2e((size_t)w1 * Nn, -1);
        for (int n = 0; n < Nn; ++n)
            for (int j = 0; j < w1; ++j) {
                const double v = nec[(size_t)n * w1 + j];
                if (std::isnan(v)) continue;  // Q2: NaN pad -> skipped
                const int e = (int)(v - 1);
                if (e < 0) continue;
                if (e >= Ne) return fail(h, NXS_ERR_INVALID, "NodalElementConnectivity[%d][%d]=%g beyond %d elements", n, j, v, Ne);
                n2e[(size_t)j * Nn + n] = e;
            }
        std::vector<int> n2n((size_t)(w2 - 1) * Nn, 0), cnt(Nn, 0);
        for (int n = 0; n < Nn; ++n) {
            const int c = (int)nc[(size_t)n * w2 + (w2 - 1)];
            if (c < 0 || c > w2 - 1) return fail(h, NXS_ERR_INVALID, "NodalConnectivity count of node %d invalid (%d)", n, c);
            cnt[n] = c;
            for (int j = 0; j < c; ++j) {
                const int nb = (int)(nc[(size_t)n * w2 + j] - 1);
                if (nb < 0 || nb >= Nn) return fail(h, NXS_ERR_INVALID, "NodalConnectivity[%d][%d] out of range", n, j);
                n2n[(size_t)j * Nn + n] = nb;
            }
        }
        d.W1 = w1; d.W2 = w2 - 1;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2e, n2e))) return rc;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2n, n2n))) return rc;
        if ((rc = dev_upload(h, h->mesh_allocs, &d.n2n_cnt, cnt))) return rc;
    }

    // state + work arrays
    DevState &s = h->ds;
    DevWork &w = h->dw;
    s = DevState{}; w = DevWork{};
    auto &P = h->state_allocs;
    const size_t n2 = 2 * (size_t)Nn, ne = Ne;
#define A(ptr, cnt) if ((rc = dev_alloc(h, P, &(ptr), (cnt)))) return rc
    A(s.VT, n2); A(s.VT2, n2); A(s.UM, n2); A(s.UT, n2);
    A(s.conc, ne); A(s.thick, ne); A(s.snow, ne); A(s.damage, ne); A(s.ridge, ne);
    A(s.s0, ne); A(s.s1, ne); A(s.s2, ne);
    A(s.damage_b, ne); A(s.s0_b, ne); A(s.s1_b, ne); A(s.s2_b, ne);
    A(s.cyoung, ne); A(s.hyoung, ne); A(s.hsyoung, ne); A(s.cmyi, ne); A(s.tmyi, ne);
    A(s.cohesion, ne); A(s.theal, ne); A(s.drag_ui, ne); A(s.drag_ui_young, ne);
    A(s.wind, n2); A(s.ocean, n2); A(s.ssh, (size_t)Nn); A(s.depth, ne);
    A(w.delta_x, ne); A(w.surface, ne); A(w.shape, 6 * ne); A(w.emass, ne); A(w.ecbu, ne); A(w.prec, 10 * ne);
    A(w.expC, ne); A(w.pmax, ne); A(w.heal, ne); A(w.dxs, ne); A(w.volume, ne); A(w.eskip, ne); A(w.dxi, ne); A(w.open_blk, (size_t)nblocks(Nn));
    A(w.force, 6 * ne);
    A(w.rlmass, (size_t)Nn); A(w.node_mass, (size_t)Nn); A(w.C_bu, (size_t)Nn); A(w.grad_ssh, n2);
    A(w.fcor, (size_t)Nn); A(w.VTM, n2); A(w.xs, (size_t)Nn); A(w.ys, (size_t)Nn); A(w.D_tau_a, n2); A(w.D_tau_w, n2); A(w.D_del, ne);
#undef A
    HIPCHK(h, hipMemsetAsync(w.surface, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.delta_x, 0, ne * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_tau_a, 0, n2 * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_tau_w, 0, n2 * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(w.D_del, 0, ne * sizeof(double), h->stream));

    if (h->d_partials) { (void)hipFree(h->d_partials); h->d_partials = nullptr; }
    h->n_partials = std::min(nblocks(Ne), 1024);
    HIPCHK(h, hipMalloc((void **)&h->d_partials, sizeof(RegridPartial) * h->n_partials));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((rc = upload_patches(h))) return rc;
    h->have_mesh = true;
    return NXS_OK;
}

// ------------------------------------------------------------------------------------------------
int nxs_dyn_set_halo(nxs_dyn_handle *h, const nxs_dyn_halo *halo) {
    if (!h || !halo) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_halo before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    free_pool(h->halo_allocs);
    ipc_release(h);
    h->have_halo = false;
    const int Nn = h->dm.Nn, No = h->dm.No;
    if (halo->nranks < 1 || halo->rank < 0 || halo->rank >= halo->nranks) return fail(h, NXS_ERR_INVALID, "rank/nranks invalid");
    const int ns = halo->num_send_procs, nr = halo->num_recv_procs;
    if (ns < 0 || nr < 0) return fail(h, NXS_ERR_INVALID, "negative neighbour count");
    h->rank = halo->rank; h->nranks = halo->nranks;
    h->send_procs.assign(halo->send_procs, halo->send_procs + ns);
    h->recv_procs.assign(halo->recv_procs, halo->recv_procs + nr);
    h->send_offsets.assign(halo->send_offsets, halo->send_offsets + ns + 1);
    h->recv_offsets.assign(halo->recv_offsets, halo->recv_offsets + nr + 1);
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    std::vector<int> sidx(halo->send_index, halo->send_index + ts), ridx(halo->recv_index, halo->recv_index + tr);
    h->h_send_index = sidx; h->h_recv_index = ridx;
    h->hf_ready = false;
    std::vector<int> sseg(ts), rseg(tr);
    for (int k = 0; k < ns; ++k) {
        if (h->send_procs[k] < 0 || h->send_procs[k] >= halo->nranks || h->send_procs[k] == halo->rank) return fail(h, NXS_ERR_INVALID, "send_procs[%d] invalid", k);
        for (int j = h->send_offsets[k]; j < h->send_offsets[k + 1]; ++j) {
            if (sidx[j] < 0 || sidx[j] >= No) return fail(h, NXS_ERR_INVALID, "send_index[%d]=%d is not an owned node", j, sidx[j]);
            sseg[j] = k;
        }
    }
    std::vector<char> seen(Nn, 0);
    for (int k = 0; k < nr; ++k) {
        if (h->recv_procs[k] < 0 || h->recv_procs[k] >= halo->nranks || h->recv_procs[k] == halo->rank) return fail(h, NXS_ERR_INVALID, "recv_procs[%d] invalid", k);
        for (int j = h->recv_offsets[k]; j < h->recv_offsets[k + 1]; ++j) {
            if (ridx[j] < No || ridx[j] >= Nn) return fail(h, NXS_ERR_INVALID, "recv_index[%d]=%d is not a ghost node", j, ridx[j]);
            if (seen[ridx[j]]) return fail(h, NXS_ERR_INVALID, "ghost node %d received twice", ridx[j]);
            seen[ridx[j]] = 1;
            rseg[j] = k;
        }
    }
    if (tr != Nn - No) return fail(h, NXS_ERR_INVALID, "recv lists cover %d of %d ghost nodes", tr, Nn - No);
    int rc;
    const int *cp;
    if ((rc = dev_upload(h, h->halo_allocs, &cp, sidx))) return rc; h->d_send_index = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, sseg))) return rc; h->d_send_seg = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->send_offsets))) return rc; h->d_send_off = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, ridx))) return rc; h->d_recv_index = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, rseg))) return rc; h->d_recv_seg = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->recv_offsets))) return rc; h->d_recv_off = const_cast<int *>(cp);
    if ((rc = dev_upload(h, h->halo_allocs, &cp, h->recv_procs))) return rc; h->d_recv_procs = const_cast<int *>(cp);
    if ((rc = dev_alloc(h, h->halo_allocs, &h->d_send_buf, 2 * (size_t)ts))) return rc;
    if ((rc = dev_alloc(h, h->halo_allocs, &h->d_recv_buf, 2 * (size_t)tr))) return rc;
    if (h->h_send) { (void)hipHostFree(h->h_send); h->h_send = nullptr; }
    if (h->h_recv) { (void)hipHostFree(h->h_recv); h->h_recv = nullptr; }
    HIPCHK(h, hipHostMalloc((void **)&h->h_send, std::max<size_t>(2 * (size_t)ts, 1) * sizeof(double), hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&h->h_recv, std::max<size_t>(2 * (size_t)tr, 1) * sizeof(double), hipHostMallocDefault));
    h->have_halo = true;
    return NXS_OK;
}

static int load_rccl(nxs_dyn_handle *h, Rccl &r) {
    if (r.lib) return NXS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(h, NXS_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                       \
    *(void **)(&r.field) = dlsym(r.lib, name);                                 \
    if (!r.field) return fail(h, NXS_ERR_COMM, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return NXS_OK;
}

int nxs_dyn_comm_unique_id(void *id128) {
    if (!id128) return NXS_ERR_INVALID;
    Rccl r;
    int rc = load_rccl(nullptr, r);
    if (rc) return rc;
    int e = r.GetUniqueId(id128);
    return e == 0 ? NXS_OK : fail(nullptr, NXS_ERR_COMM, "ncclGetUniqueId: %s", r.GetErrorString(e));
}

int nxs_dyn_comm_init(nxs_dyn_handle *h, const void *id128, int rank, int nranks) {
    if (!h || !id128) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = load_rccl(h, h->rccl);
    if (rc) return rc;
    NcclId id;
    std::memcpy(id.internal, id128, 128);
    nccl_comm_init_rank_t init = (nccl_comm_init_rank_t)h->rccl.CommInitRank;
    int e = init(&h->comm, nranks, id, rank);
    if (e != 0) return fail(h, NXS_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, h->rccl.GetErrorString(e));
    return NXS_OK;
}

// Device-direct transport, step 1: allocate my mailbox and export it.  blob receives NXS_IPC_BLOB_BYTES.
int nxs_dyn_ipc_export(nxs_dyn_handle *h, void *blob) {
    if (!h || !blob) return NXS_ERR_INVALID;
    if (!h->have_halo) return fail(h, NXS_ERR_STATE, "ipc_export before set_halo");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    release_graph(h);
    ipc_release(h);
    const int nr = (int)h->recv_procs.size();
    const size_t tr = (size_t)h->recv_offsets[nr];
    const size_t bytes = (4 * tr + (size_t)std::max(nr, 1) + 16) * sizeof(double);  // 2 buffers of 2*tr doubles + flags
    // uncached (MTYPE_UC) device memory: neither my L2 nor a neighbour's can hold a stale copy of a
    // mailbox line or a flag; plain device memory as a fallback (the kernels use system-scope accesses anyway)
    bool uncached = true;
    if (hipExtMallocWithFlags(&h->ipc_block, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        h->ipc_block = nullptr;
        uncached = false;
        HIPCHK(h, hipMalloc(&h->ipc_block, bytes));
    }
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d mailbox: %zu bytes, %s\n", h->rank, bytes, uncached ? "uncached (MTYPE_UC)" : "plain hipMalloc (uncached allocation refused)");
    HIPCHK(h, hipMemset(h->ipc_block, 0, bytes));
    HIPCHK(h, hipDeviceSynchronize());
    h->ipc_block_bytes = bytes;
    hipIpcMemHandle_t mh;
    HIPCHK(h, hipIpcGetMemHandle(&mh, h->ipc_block));
    static_assert(sizeof(hipIpcMemHandle_t) <= NXS_IPC_BLOB_BYTES, "blob too small");
    std::memset(blob, 0, NXS_IPC_BLOB_BYTES);
    std::memcpy(blob, &mh, sizeof mh);
    return NXS_OK;
}

// Step 2: map the neighbours' mailboxes.  For send neighbour k (order of nxs_dyn_halo.send_procs):
// blobs + k*NXS_IPC_BLOB_BYTES is its exported blob, peer_recv_offset[k] the offset (in nodes) of MY
// segment inside its receive lists, peer_recv_total[k] its total number of received nodes and
// peer_flag_slot[k] my position in its recv_procs.
int nxs_dyn_ipc_connect(nxs_dyn_handle *h, const void *blobs, const int32_t *peer_recv_offset, const int32_t *peer_recv_total,
                        const int32_t *peer_flag_slot) {
    if (!h) return NXS_ERR_INVALID;
    if (!h->ipc_block) return fail(h, NXS_ERR_STATE, "ipc_connect before ipc_export");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    if (ns > 0 && (!blobs || !peer_recv_offset || !peer_recv_total || !peer_flag_slot)) return fail(h, NXS_ERR_INVALID, "ipc_connect: NULL tables");
    std::vector<double *> seg(ns);
    std::vector<long long> stride(ns);
    std::vector<unsigned long long *> flag(ns);
    for (int k = 0; k < ns; ++k) {
        hipIpcMemHandle_t mh;
        std::memcpy(&mh, (const char *)blobs + (size_t)k * NXS_IPC_BLOB_BYTES, sizeof mh);
        void *base = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&base, mh, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(h, NXS_ERR_COMM, "hipIpcOpenMemHandle(neighbour %d): %s", h->send_procs[k], hipGetErrorString(e));
        h->ipc_peer_base.push_back(base);
        double *mb = static_cast<double *>(base);
        seg[k] = mb + 2 * (size_t)peer_recv_offset[k];
        stride[k] = 2ll * peer_recv_total[k];
        flag[k] = reinterpret_cast<unsigned long long *>(mb + 4 * (size_t)peer_recv_total[k]) + peer_flag_slot[k];
    }
    IpcDev &d = h->ipc;
    const size_t tr = (size_t)h->recv_offsets[nr];
    d.mailbox = static_cast<double *>(h->ipc_block);
    d.flags = reinterpret_cast<unsigned long long *>(d.mailbox + 4 * tr);
    d.tr = (int)tr; d.ns = ns; d.nr = nr;
    int rc;
    unsigned long long *ctr = nullptr;
    if ((rc = dev_alloc(h, h->ipc_allocs, &ctr, 8))) return rc;
    HIPCHK(h, hipMemset(ctr, 0, 8 * sizeof(unsigned long long)));
    d.seq_push = ctr; d.seq_pull = ctr + 1;
    d.done_push = reinterpret_cast<unsigned int *>(ctr + 2); d.done_pull = reinterpret_cast<unsigned int *>(ctr + 3);
    d.error = reinterpret_cast<int *>(ctr + 4);
    double *const *dseg; const long long *dstr; unsigned long long *const *dfl;
    { const double **tmp; std::vector<const double *> v(seg.begin(), seg.end()); if ((rc = dev_upload(h, h->ipc_allocs, (const double *const **)&tmp, v))) return rc; dseg = (double *const *)tmp; }
    if ((rc = dev_upload(h, h->ipc_allocs, &dstr, stride))) return rc;
    { const unsigned long long **tmp; std::vector<const unsigned long long *> v(flag.begin(), flag.end()); if ((rc = dev_upload(h, h->ipc_allocs, (const unsigned long long *const **)&tmp, v))) return rc; dfl = (unsigned long long *const *)tmp; }
    d.peer_seg = dseg; d.peer_parity_stride = dstr; d.peer_flag = dfl;
    h->ipc_ready = true;
    release_graph(h);
    return NXS_OK;
}

// Step 3 (collective): `rounds` exchanges of a synthetic pattern through the mailboxes; *errors gets
// 0 when every value arrived intact and in time on this rank.
int nxs_dyn_ipc_selftest(nxs_dyn_handle *h, int rounds, int32_t *errors) {
    if (!h || !errors) return NXS_ERR_INVALID;
    if (!h->ipc_ready) return fail(h, NXS_ERR_STATE, "ipc_selftest before ipc_connect");
    HIPCHK(h, hipSetDevice(h->device));
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    for (int it = 0; it < rounds; ++it) {
        hipLaunchKernelGGL(k_halo_push, dim3(nblocks(ts)), dim3(BLOCK), 0, h->stream, (const double *)nullptr, h->dm.Nn, ts,
                           h->d_send_index, h->d_send_seg, h->d_send_off, h->ipc, h->rank, 1);
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, (double *)nullptr, h->dm, h->ds, tr,
                           h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->ipc, 0., 1, h->d_recv_procs, 0);
    }
    int err = 0;
    HIPCHK(h, hipMemcpyAsync(&err, h->ipc.error, sizeof err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *errors = err;
    return NXS_OK;
}

int nxs_dyn_set_halo_exchange_fn(nxs_dyn_handle *h, nxs_dyn_halo_fn fn, void *ctx) {
    if (!h) return NXS_ERR_INVALID;
    h->halo_fn = fn;
    h->halo_ctx = ctx;
    return NXS_OK;
}

// ------------------------------------------------------------------------------------------------
int nxs_dyn_put_state(nxs_dyn_handle *h, const nxs_dyn_state *s) {
    if (!h || !s) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "put_state before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    DevState &d = h->ds;
    struct { double *dst; const double *src; size_t bytes; const char *name; } cp[] = {
        {d.VT, s->VT, n2, "VT"}, {d.UM, s->UM, n2, "UM"}, {d.UT, s->UT, n2, "UT"},
        {d.conc, s->conc, ne, "conc"}, {d.thick, s->thick, ne, "thick"}, {d.snow, s->snow_thick, ne, "snow_thick"},
        {d.damage, s->damage, ne, "damage"}, {d.ridge, s->ridge_ratio, ne, "ridge_ratio"},
        {d.s0, s->sigma[0], ne, "sigma[0]"}, {d.s1, s->sigma[1], ne, "sigma[1]"}, {d.s2, s->sigma[2], ne, "sigma[2]"},
        {d.cyoung, s->conc_young, ne, "conc_young"}, {d.hyoung, s->h_young, ne, "h_young"}, {d.hsyoung, s->hs_young, ne, "hs_young"},
        {d.cmyi, s->conc_myi, ne, "conc_myi"}, {d.tmyi, s->thick_myi, ne, "thick_myi"},
        {d.cohesion, s->cohesion, ne, "cohesion"}, {d.theal, s->time_relaxation_damage, ne, "time_relaxation_damage"},
        {d.drag_ui, s->drag_ui, ne, "drag_ui"}, {d.drag_ui_young, s->drag_ui_young, ne, "drag_ui_young"},
    };
    // the first put after set_mesh must bring every array; later ones may leave members NULL = "the device copy is current"
    // (a host whose thermodynamics only touched concentration and thickness uploads only those)
    if (!h->have_state)
        for (auto &c : cp) if (!c.src) return fail(h, NXS_ERR_INVALID, "put_state: %s is NULL", c.name);
    for (auto &c : cp) if (c.src) pin_host_buffer(h, c.src, c.bytes);
    for (auto &c : cp) if (c.src) HIPCHK(h, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_state = true;
    return NXS_OK;
}

int nxs_dyn_get_state(nxs_dyn_handle *h, nxs_dyn_state *s) {
    if (!h || !s) return NXS_ERR_INVALID;
    if (!h->have_state) return fail(h, NXS_ERR_STATE, "get_state before put_state");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    DevState &d = h->ds;
    struct { double *dst; const double *src; size_t bytes; } cp[] = {
        {s->VT, d.VT, n2}, {s->UM, d.UM, n2}, {s->UT, d.UT, n2},
        {s->conc, d.conc, ne}, {s->thick, d.thick, ne}, {s->snow_thick, d.snow, ne},
        {s->damage, d.damage, ne}, {s->ridge_ratio, d.ridge, ne},
        {s->sigma[0], d.s0, ne}, {s->sigma[1], d.s1, ne}, {s->sigma[2], d.s2, ne},
        {s->conc_young, d.cyoung, ne}, {s->h_young, d.hyoung, ne}, {s->hs_young, d.hsyoung, ne},
        {s->conc_myi, d.cmyi, ne}, {s->thick_myi, d.tmyi, ne},
    };
    for (auto &c : cp) if (c.dst) pin_host_buffer(h, c.dst, c.bytes);
    for (auto &c : cp) if (c.dst) HIPCHK(h, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
}

int nxs_dyn_set_forcing(nxs_dyn_handle *h, const nxs_dyn_forcing *f) {
    if (!h || !f) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_forcing before set_mesh");
    if (!f->wind || !f->ocean || !f->ssh || !f->element_depth) return fail(h, NXS_ERR_INVALID, "forcing has NULL arrays");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    pin_host_buffer(h, f->wind, 2 * Nn * sizeof(double)); pin_host_buffer(h, f->ocean, 2 * Nn * sizeof(double));
    pin_host_buffer(h, f->ssh, Nn * sizeof(double)); pin_host_buffer(h, f->element_depth, Ne * sizeof(double));
    HIPCHK(h, hipMemcpyAsync(h->ds.wind, f->wind, 2 * Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.ocean, f->ocean, 2 * Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.ssh, f->ssh, Nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.depth, f->element_depth, Ne * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_forcing = true;
    return NXS_OK;
}

int nxs_dyn_set_forcing_pair(nxs_dyn_handle *h, const nxs_dyn_forcing *f0, const nxs_dyn_forcing *f1) {
    if (!h || !f0 || !f1) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "set_forcing_pair before set_mesh");
    if (!f0->wind || !f0->ocean || !f0->ssh || !f0->element_depth || !f1->wind || !f1->ocean || !f1->ssh) return fail(h, NXS_ERR_INVALID, "forcing has NULL arrays");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    const size_t len[6] = {2 * Nn, 2 * Nn, 2 * Nn, 2 * Nn, Nn, Nn};
    const double *src[6] = {f0->wind, f1->wind, f0->ocean, f1->ocean, f0->ssh, f1->ssh};
    if (!h->f_snap[0])
        for (int k = 0; k < 6; ++k) { int rc = dev_alloc(h, h->forcing_allocs, &h->f_snap[k], len[k]); if (rc) return rc; }
    for (int k = 0; k < 6; ++k) HIPCHK(h, hipMemcpyAsync(h->f_snap[k], src[k], len[k] * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ds.depth, f0->element_depth, Ne * sizeof(double), hipMemcpyHostToDevice, h->stream));  // a "constant" dataset
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_pair = true;
    return NXS_OK;
}

int nxs_dyn_set_forcing_time(nxs_dyn_handle *h, double fcoeff0, double fcoeff1, const double factor[3], const double bias[3]) {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_pair) return fail(h, NXS_ERR_STATE, "set_forcing_time before set_forcing_pair");
    HIPCHK(h, hipSetDevice(h->device));
    ForcingBlend b{h->f_snap[0], h->f_snap[1], h->f_snap[2], h->f_snap[3], h->f_snap[4], h->f_snap[5], fcoeff0, fcoeff1, {1., 1., 1.}, {0., 0., 0.}};
    for (int k = 0; k < 3; ++k) { if (factor) b.factor[k] = factor[k]; if (bias) b.bias[k] = bias[k]; }
    hipLaunchKernelGGL(k_blend_forcing, dim3(nblocks(2 * h->dm.Nn)), dim3(BLOCK), 0, h->stream, h->dm.Nn, b, h->ds.wind, h->ds.ocean, h->ds.ssh);
    h->have_forcing = true;
    return NXS_OK;
}

int nxs_dyn_get_diag(nxs_dyn_handle *h, nxs_dyn_diag *dg) {
    if (!h || !dg) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "get_diag before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = 2 * (size_t)h->dm.Nn * sizeof(double), ne = (size_t)h->dm.Ne * sizeof(double);
    if (dg->surface) HIPCHK(h, hipMemcpyAsync(dg->surface, h->dw.surface, ne, hipMemcpyDeviceToHost, h->stream));
    if (dg->delta_x) HIPCHK(h, hipMemcpyAsync(dg->delta_x, h->dw.delta_x, ne, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_tau_a) HIPCHK(h, hipMemcpyAsync(dg->D_tau_a, h->dw.D_tau_a, n2, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_tau_w) HIPCHK(h, hipMemcpyAsync(dg->D_tau_w, h->dw.D_tau_w, n2, hipMemcpyDeviceToHost, h->stream));
    if (dg->D_del_ci_ridge_myi) HIPCHK(h, hipMemcpyAsync(dg->D_del_ci_ridge_myi, h->dw.D_del, ne, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return NXS_OK;
}

// debug / test door: copy a named work array to the host (n doubles)
int nxs_dyn_debug_array(nxs_dyn_handle *h, const char *name, double *out, int64_t n) {
    if (!h || !name || !out) return NXS_ERR_INVALID;
    if (!h->have_mesh) return fail(h, NXS_ERR_STATE, "debug_array before set_mesh");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t Nn = h->dm.Nn, Ne = h->dm.Ne;
    struct { const char *nm; const double *p; int64_t len; } tab[] = {
        {"rlmass", h->dw.rlmass, Nn}, {"node_mass", h->dw.node_mass, Nn}, {"C_bu", h->dw.C_bu, Nn},
        {"grad_ssh", h->dw.grad_ssh, 2 * Nn}, {"fcor", h->dw.fcor, Nn}, {"VTM", h->dw.VTM, 2 * Nn},
        {"shape", h->dw.shape, 6 * Ne}, {"emass", h->dw.emass, Ne}, {"ecbu", h->dw.ecbu, Ne},
        {"force", h->dw.force, 6 * Ne}, {"volume", h->dw.volume, Ne}, {"expC", h->dw.expC, Ne},
    };
    for (auto &t : tab)
        if (!std::strcmp(t.nm, name)) {
            if (n != t.len) return fail(h, NXS_ERR_INVALID, "debug_array %s has %lld entries, caller asked %lld", name, (long long)t.len, (long long)n);
            HIPCHK(h, hipMemcpyAsync(out, t.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            return NXS_OK;
        }
#ifdef NXS_PHASE_TIMING
    if (!std::strcmp(name, "phase_times")) {  // [8192][8] timestamps (100 MHz) of the last fused launch, as doubles
        std::vector<long long> t(8 * 8192);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_phase_t), t.size() * sizeof(long long)));
        for (int64_t i = 0; i < n && i < (int64_t)t.size(); ++i) out[i] = (double)t[i];
        return NXS_OK;
    }
#endif
    return fail(h, NXS_ERR_INVALID, "unknown debug array '%s'", name);
}

// ------------------------------------------------------------------------------------------------
// the launches

namespace {

#define LAUNCH(h, kern, n, ...)                                                              \
    do {                                                                                      \
        hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(BLOCK), 0, (h)->stream, __VA_ARGS__); \
    } while (0)

int halo_exchange(nxs_dyn_handle *h, double *vec, double move_dt) {
    // updateGhosts (FE.cpp:13963-13996): pack -> grouped send/recv -> unpack
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    const int ts = h->send_offsets[ns], tr = h->recv_offsets[nr];
    if (h->ipc_ready && !h->halo_fn) {
        // device-direct: pack + peer stores + flags, then wait + unpack (+ ghost-node move); no host work
        hipLaunchKernelGGL(k_halo_push, dim3(nblocks(ts)), dim3(BLOCK), 0, h->stream, (const double *)vec, h->dm.Nn, ts,
                           h->d_send_index, h->d_send_seg, h->d_send_off, h->ipc, h->rank, 0);
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, vec, h->dm, h->ds, tr, h->d_recv_index,
                           h->d_recv_seg, h->d_recv_off, h->ipc, move_dt, 0, h->d_recv_procs, 0);
        return NXS_OK;
    }
    if (!h->comm && !h->halo_fn) return fail(h, NXS_ERR_STATE, "halo exchange needs nxs_dyn_comm_init, nxs_dyn_ipc_connect or nxs_dyn_set_halo_exchange_fn");
    if (ts > 0) LAUNCH(h, k_halo_pack, ts, vec, h->dm.Nn, ts, h->d_send_index, h->d_send_seg, h->d_send_off, h->d_send_buf);
    if (h->halo_fn) {
        // host-staged: exactly the reference's M_comm.send / M_comm.recv of packed std::vector<double>
        if (ts > 0) HIPCHK(h, hipMemcpyAsync(h->h_send, h->d_send_buf, 2 * (size_t)ts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const int rc = h->halo_fn(h->halo_ctx, h->h_send, h->h_recv);
        if (rc != 0) return fail(h, NXS_ERR_COMM, "halo exchange callback returned %d", rc);
        if (tr > 0) {
            HIPCHK(h, hipMemcpyAsync(h->d_recv_buf, h->h_recv, 2 * (size_t)tr * sizeof(double), hipMemcpyHostToDevice, h->stream));
            LAUNCH(h, k_halo_unpack, tr, vec, h->dm, h->ds, tr, h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->d_recv_buf, move_dt);
        }
        return NXS_OK;
    }
    const int ncclDouble = 8;  // ncclFloat64
    int e = h->rccl.GroupStart();
    for (int k = 0; k < ns && e == 0; ++k)
        e = h->rccl.Send(h->d_send_buf + 2 * (size_t)h->send_offsets[k], 2 * (size_t)(h->send_offsets[k + 1] - h->send_offsets[k]),
                         ncclDouble, h->send_procs[k], h->comm, h->stream);
    for (int k = 0; k < nr && e == 0; ++k)
        e = h->rccl.Recv(h->d_recv_buf + 2 * (size_t)h->recv_offsets[k], 2 * (size_t)(h->recv_offsets[k + 1] - h->recv_offsets[k]),
                         ncclDouble, h->recv_procs[k], h->comm, h->stream);
    int e2 = h->rccl.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) return fail(h, NXS_ERR_COMM, "halo send/recv: %s", h->rccl.GetErrorString(e));
    if (tr > 0) LAUNCH(h, k_halo_unpack, tr, vec, h->dm, h->ds, tr, h->d_recv_index, h->d_recv_seg, h->d_recv_off, h->d_recv_buf, move_dt);
    return NXS_OK;
}

bool multi_rank(const nxs_dyn_handle *h) { return h->nranks > 1; }

PingPong pingpong(const nxs_dyn_handle *h, int parity) {
    const DevState &s = h->ds;
    PingPong b;
    if (parity == 0) {
        b.VTc = s.VT; b.s0c = s.s0; b.s1c = s.s1; b.s2c = s.s2; b.dc = s.damage;
        b.VTn = s.VT2; b.s0n = s.s0_b; b.s1n = s.s1_b; b.s2n = s.s2_b; b.dn = s.damage_b;
    } else {
        b.VTc = s.VT2; b.s0c = s.s0_b; b.s1c = s.s1_b; b.s2c = s.s2_b; b.dc = s.damage_b;
        b.VTn = s.VT; b.s0n = s.s0; b.s1n = s.s1; b.s2n = s.s2; b.dn = s.damage;
    }
    return b;
}

// sub-step `sidx` of the fused path: sigma/damage ping-pong by parity; velocities move through the ring
// (ring of 2 == ping-pong between VT and VT2 when the deferred mesh move is off)
// halo != 0: the sub-step also performs updateGhosts (HaloFused); from_mailbox = ghosts come from exchange x-1
void launch_fused(nxs_dyn_handle *h, int sidx, double move_dt, int halo = 0, int from_mailbox = 0) {
    PingPong b = pingpong(h, sidx & 1);
    const int R = h->ring.R;
    b.VTc = h->ring.slot[sidx % R];
    b.VTn = h->ring.slot[(sidx + 1) % R];
    const dim3 grid(h->dpch.nP);
    const bool big = h->dpch.Pmax > NXS_T256_MAXP || h->dpch.Emax > 3 * 256, pow4 = h->dp.ers_int == 4;
    if (halo) {
        HaloFused hf = h->hf;
        hf.ipc = h->ipc;
        hf.from_mailbox = from_mailbox;
#define FUSED_H(TT, PP, NN) hipLaunchKernelGGL((k_substep_fused<TT, PP, NN, true>), grid, dim3(TT), h->fused_lds, h->stream, h->dm, h->dpch, h->ds, h->dw, h->dp, b, move_dt, hf)
        if (big) { if (pow4) { if (h->nt_mask) FUSED_H(512, true, 3); else FUSED_H(512, true, 0); } else { FUSED_H(512, false, 0); } }
        else { if (pow4) { if (h->nt_mask) FUSED_H(256, true, 3); else FUSED_H(256, true, 0); } else { FUSED_H(256, false, 0); } }
#undef FUSED_H
        return;
    }
    const HaloFused none{};
#define FUSED(TT, PP, NN) hipLaunchKernelGGL((k_substep_fused<TT, PP, NN, false>), grid, dim3(TT), h->fused_lds, h->stream, h->dm, h->dpch, h->ds, h->dw, h->dp, b, move_dt, none)
#define FUSED_NT(TT, PP) switch (h->nt_mask) { case 0: FUSED(TT, PP, 0); break; case 1: FUSED(TT, PP, 1); break; case 3: FUSED(TT, PP, 3); break; case 4: FUSED(TT, PP, 4); break; case 5: FUSED(TT, PP, 5); break; default: FUSED(TT, PP, 7); break; }
    if (big) { if (pow4) { FUSED_NT(512, true); } else { FUSED(512, false, 0); } }
    else { if (pow4) { FUSED_NT(256, true); } else { FUSED(256, false, 0); } }
#undef FUSED_NT
#undef FUSED
}

// sub-steps sidx .. sidx+D-1 in one launch (k_substep_multi): sigma/damage ping-pong per LAUNCH, velocities through the ring
void launch_multi(nxs_dyn_handle *h, int sidx, int D) {
    PingPong b = pingpong(h, (sidx / D) & 1);
    const int R = h->ring.R;
    b.VTc = h->ring.slot[sidx % R];
    b.VTn = nullptr;
    VTOut vo{};
    for (int k = 0; k < D; ++k) vo.slot[k] = h->ring.slot[(sidx + 1 + k) % R];
    const dim3 grid(h->dpch2.nP);
    const bool pow4 = h->dp.ers_int == 4;
#define MULTI(TT, PP, NN) hipLaunchKernelGGL((k_substep_multi<TT, PP, NN>), grid, dim3(TT), h->pair_lds, h->stream, h->dm, h->dpch2, h->ds, h->dw, h->dp, b, vo)
#define MULTI_T(TT) do { if (pow4) { if (h->nt_mask) MULTI(TT, true, 5); else MULTI(TT, true, 0); } else MULTI(TT, false, 0); } while (0)
    if (h->pair_threads == 512) MULTI_T(512); else MULTI_T(256);
#undef MULTI_T
#undef MULTI
}

// (re)build the ring of velocity buffers of the fused path: slot 0 is M_VT itself, slot 1 is VT2
int setup_ring(nxs_dyn_handle *h, int K) {
    const int R = K + 1;
    if (h->ring.R == R) return NXS_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    free_pool(h->ring_allocs);
    h->ring = VTRing{};
    h->ring.slot[0] = h->ds.VT;
    h->ring.slot[1] = h->ds.VT2;
    for (int i = 2; i < R; ++i) {
        int rc = dev_alloc(h, h->ring_allocs, &h->ring.slot[i], 2 * (size_t)h->dm.Nn);
        if (rc) return rc;
    }
    h->ring.R = R;
    return NXS_OK;
}

// tables of the halo exchange fused into the sub-step kernel (see HaloFused)
int build_halo_fused(nxs_dyn_handle *h) {
    free_pool(h->hf_allocs);
    h->hf = HaloFused{};
    h->hf_ready = false;
    const int Nn = h->dm.Nn, No = h->dm.No, nP = h->dpch.nP;
    const int ns = (int)h->send_procs.size(), nr = (int)h->recv_procs.size();
    if (!h->hp || h->hp->nP != nP || (int)h->h_recv_index.size() != Nn - No) return fail(h, NXS_ERR_STATE, "fused halo tables: patches / halo lists missing");
    // sending side: CSR over own nodes
    std::vector<int> sptr(No + 1, 0);
    for (int k = 0; k < ns; ++k)
        for (int j = h->send_offsets[k]; j < h->send_offsets[k + 1]; ++j) sptr[h->h_send_index[j] + 1]++;
    for (int n = 0; n < No; ++n) sptr[n + 1] += sptr[n];
    std::vector<int> sk(std::max(sptr[No], 1)), spos(std::max(sptr[No], 1)), fill(sptr.begin(), sptr.end() - 1);
    for (int k = 0; k < ns; ++k)
        for (int j = h->send_offsets[k]; j < h->send_offsets[k + 1]; ++j) {
            const int q = fill[h->h_send_index[j]]++;
            sk[q] = k;
            spos[q] = j - h->send_offsets[k];
        }
    // receiving side: where each ghost node sits inside a mailbox half (layout of k_halo_pull)
    std::vector<int> goff(std::max(Nn - No, 1), 0), gsrl(std::max(Nn - No, 1), 0);
    for (int k = 0; k < nr; ++k) {
        const int off = h->recv_offsets[k], srl = h->recv_offsets[k + 1] - off;
        for (int j = off; j < h->recv_offsets[k + 1]; ++j) {
            goff[h->h_recv_index[j] - No] = 2 * off + (j - off);
            gsrl[h->h_recv_index[j] - No] = srl;
        }
    }
    // boundary patches: send something or stage a ghost node.  The patch arrays are re-uploaded with those patches
    // FIRST, so that the grid starts with them and "boundary" is blk < n_boundary (no lookup on the critical path)
    HostPatches &hp = *h->hp;
    std::vector<int> order_b, order_i;
    for (int q = 0; q < nP; ++q) {
        const int *nd = hp.pnodes.data() + (size_t)q * hp.Mmax;
        bool bnd = false;
        for (int i = 0; i < hp.node_cnt[q] && !bnd; ++i) {
            const int n = nd[i];
            bnd = (n >= No) || (i < hp.own_cnt[q] && sptr[n + 1] > sptr[n]);
        }
        (bnd ? order_b : order_i).push_back(q);
    }
    const int nb = (int)order_b.size();
    bool sorted = true;
    for (int q = 0; q < nb; ++q) sorted = sorted && order_b[q] == q;
    if (!sorted) {
        std::vector<int> order(order_b);
        order.insert(order.end(), order_i.begin(), order_i.end());
        HostPatches r = hp;
        for (int q = 0; q < nP; ++q) {
            const int o = order[q];
            r.own_cnt[q] = hp.own_cnt[o]; r.elem_cnt[q] = hp.elem_cnt[o]; r.node_cnt[q] = hp.node_cnt[o];
            std::copy_n(hp.pnodes.begin() + (size_t)o * hp.Mmax, hp.Mmax, r.pnodes.begin() + (size_t)q * hp.Mmax);
            std::copy_n(hp.pelem.begin() + (size_t)o * hp.Emax, hp.Emax, r.pelem.begin() + (size_t)q * hp.Emax);
            std::copy_n(hp.ptri.begin() + (size_t)o * hp.Emax * 4, (size_t)hp.Emax * 4, r.ptri.begin() + (size_t)q * hp.Emax * 4);
            std::copy_n(hp.pfan.begin() + (size_t)o * hp.Wp * hp.Pmax, (size_t)hp.Wp * hp.Pmax, r.pfan.begin() + (size_t)q * hp.Wp * hp.Pmax);
        }
        hp = std::move(r);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        release_graph(h);
        int rcu = upload_host_patches(h, hp);
        if (rcu) return rcu;
    }
    HaloFused &f = h->hf;
    int rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_ptr, sptr))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_k, sk))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.send_pos, spos))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.ghost_off, goff))) return rc;
    if ((rc = dev_upload(h, h->hf_allocs, &f.ghost_srl, gsrl))) return rc;
    unsigned int *ctr = nullptr;
    if ((rc = dev_alloc(h, h->hf_allocs, &ctr, 32 * 17))) return rc;
    HIPCHK(h, hipMemsetAsync(ctr, 0, 32 * 17 * sizeof(unsigned int), h->stream));
    f.done_all = ctr;
    {   // k_smooth_halo runs BLOCK own nodes per block: which blocks store into a mailbox
        const int nblk = std::max(1, (No + BLOCK - 1) / BLOCK);
        std::vector<int> rank_of(nblk, -1);
        int cnt = 0;
        for (int b = 0; b < nblk; ++b)
            if (sptr[std::min(No, (b + 1) * BLOCK)] > sptr[std::min(No, b * BLOCK)]) rank_of[b] = cnt++;
        if ((rc = dev_upload(h, h->hf_allocs, &f.send_block_rank, rank_of))) return rc;
        f.n_send_blocks = cnt;
    }
    f.send_off = h->d_send_off;
    f.n_boundary = nb;
    f.No = No;
    h->hf_ready = true;
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d fused halo: %d of %d patches on the boundary, %d sent nodes, %d ghosts\n", h->rank, nb, nP, sptr[No], Nn - No);
    return NXS_OK;
}

void launch_substep(nxs_dyn_handle *h, double move_dt) {
    if (h->dp.dynamics_type == NXS_DYN_BBM) {
        if (h->dp.ers_int == 4) LAUNCH(h, k_sigma_bbm<true>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
        else LAUNCH(h, k_sigma_bbm<false>, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    }
    else
        LAUNCH(h, k_sigma_vp, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    LAUNCH(h, k_solve_move, h->dm.No, h->dm, h->ds, h->dw, h->dp, move_dt);
}

int run_substeps(nxs_dyn_handle *h) {
    const int S = h->dp.substeps;
    const double move_dt = (h->dp.dynamics_type == NXS_DYN_MEVP) ? 0. : h->dp.dte;
    const bool fused = h->fused != 0;
    const int bbm = h->dp.dynamics_type == NXS_DYN_BBM;
    const bool mr = multi_rank(h);
    // deferred mesh move (fused path, not mEVP whose single move comes after the loop)
    const bool device_halo = mr && h->ipc_ready && !h->halo_fn;  // no host work inside the loop: graph-capturable
    // auto ring: one flush per step (up to 120 sub-steps) on meshes that stream from HBM -- the flush reads every slot once
    // whatever its period, so a longer ring only saves UM/UT passes (2 km: 7.60 -> 7.49 ms/step from 16 to 120, 1.4 GB of
    // slots); also whenever the halo exchange runs inside the sub-step kernel
    const int want_ring = h->um_ring > 0 ? h->um_ring : ((h->dm.Ne >= 400000 || (device_halo && h->halo_fused)) ? 120 : 1);
    // v3: D sub-steps per launch -- single rank, the deferred mesh move (ring of >= D+1 buffers).  It trades redundant arithmetic
    // on the halo rings for less HBM traffic and fewer launches: a gain where the sub-step is latency-bound (10 km: 1.23 -> 0.97
    // ms/step), a loss as soon as a CU hosts more than one patch (182 k triangles: 1.65 -> 1.90) and where the v2 kernel already
    // runs at 5.5 TB/s with its VALUs half busy (2 km, D = 2: 7.4 -> 8.0).
    // D sub-steps per launch: the requested depth, else (auto) 4 (10 km: D = 2 / 3 / 4 / 5 / 6 / 8: 1.11 / 1.01 / 0.97 / 0.96 / 0.98 / 1.08 ms/step; the
    // rings grow the arithmetic by x2.0 per sub-step at D = 4) -- lowered until it divides the number of sub-steps
    int D = 1;
    // Automatic (fused == 3): only where ONE round of one patch per CU covers the mesh (<= 256 own nodes per patch: 65 k nodes, 130 k
    // triangles on 256 CUs) -- 111 k triangles: 1.47 (v2) -> 1.21 ms/step; 182 k triangles, two patches per CU: 1.65 -> 1.90-2.31.
    if ((h->fused == 2 || (h->fused == 3 && (long long)h->dm.Nn <= 256ll * 1024)) && !mr && move_dt != 0. && S >= 2 && !h->pair_failed) {
        D = std::min(h->pair_depth > 0 ? h->pair_depth : 4, std::min(S, NXS_MAX_DEPTH));
        while (D > 1 && S % D != 0) --D;
        if (D >= 2 && (!h->pair_ready || h->pair_depth_built != D) && upload_patches2(h, D, h->fused == 3) != NXS_OK) {
            h->pair_failed = true;  // no patch size fits (a numbering without any locality, huge fans): one sub-step per launch
            D = 1;
        }
    }
    const bool pair = D >= 2;
    int K = (fused && move_dt != 0.) ? std::max(1, std::min(want_ring, S)) : 1;
    if (pair) K = std::max(D, K - K % D);  // the ring is flushed between launches
    const bool deferred = K > 1;
    if (fused) { int rc = setup_ring(h, K); if (rc) return rc; }
    const int R = h->ring.R;
    // the exchange inside the sub-step kernel: needs the deferred mesh move (ghost nodes are moved from the ring)
    // or no move at all (mEVP)
    const bool halo_in_kernel = device_halo && fused && h->halo_fused && (deferred || move_dt == 0.);
    if (halo_in_kernel && !h->hf_ready) { int rc = build_halo_fused(h); if (rc) return rc; }
    auto pull_latest = [&](double *vec) {
        const int tr = h->recv_offsets[h->recv_procs.size()];
        hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, vec, h->dm, h->ds, tr, h->d_recv_index,
                           h->d_recv_seg, h->d_recv_off, h->ipc, 0., 0, h->d_recv_procs, 1);
    };
    auto loop = [&]() -> int {
        int pending = 0;  // sub-steps whose velocity still has to be applied to UM/UT
        for (int s = 0; s < S; ++s) {
            if (pair) {
                launch_multi(h, s, D);
                s += D - 1;
                pending += D;
                if (pending == K || s == S - 1) {
                    LAUNCH(h, k_move_ring, h->dm.Nn, h->dm, h->ds, h->ring, (s + 1 - (pending - 1)) % R, pending, move_dt);
                    pending = 0;
                }
                continue;
            }
            if (halo_in_kernel) {
                launch_fused(h, s, 0., 1, s > 0);
                const bool flush = deferred && (pending + 1 == K || s == S - 1);
                if (flush || s == S - 1) pull_latest(h->ring.slot[(s + 1) % R]);  // the newest ghosts, for the move / the end of the step
                if (flush) {
                    ++pending;
                    LAUNCH(h, k_move_ring, h->dm.Nn, h->dm, h->ds, h->ring, (s + 1 - (pending - 1)) % R, pending, move_dt);
                    pending = 0;
                } else if (deferred) ++pending;
                continue;
            }
            if (fused) launch_fused(h, s, deferred ? 0. : move_dt); else launch_substep(h, move_dt);
            if (mr) {
                // owned nodes were written to the buffer the next sub-step reads; ghosts must land there too
                double *vec = fused ? h->ring.slot[(s + 1) % R] : h->ds.VT;
                int rc = halo_exchange(h, vec, deferred ? 0. : move_dt);
                if (rc) return rc;
            }
            if (deferred && (++pending == K || s == S - 1)) {
                LAUNCH(h, k_move_ring, h->dm.Nn, h->dm, h->ds, h->ring, (s + 1 - (pending - 1)) % R, pending, move_dt);
                pending = 0;
            }
        }
        if (fused) {  // bring the result back to the primary buffers
            const double *vt_src = (S % R) ? h->ring.slot[S % R] : nullptr;
            const int odd = pair ? ((S / D) & 1) : (S & 1);  // sigma/damage ended in the secondary buffers
            if (vt_src || odd)
                LAUNCH(h, k_pingpong_copy_back, std::max(2 * h->dm.Nn, h->dm.Ne), h->dm, h->ds, bbm, vt_src, odd);
        }
        return NXS_OK;
    };
    h->timing.substep_launches = pair ? S / D : halo_in_kernel ? S + (S + K - 1) / K : S * ((fused ? 1 : 2) + (mr ? 2 : 0));
    if (!h->use_graph || (mr && !device_halo)) return loop();
    if (!h->graph_valid) {
        release_graph(h);
        hipGraph_t g = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        int lrc = loop();
        if (lrc) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(h->stream, &dead); if (dead) (void)hipGraphDestroy(dead); return lrc; }
        HIPCHK(h, hipStreamEndCapture(h->stream, &g));
        hipError_t e = hipGraphInstantiate(&h->substep_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        h->graph_valid = true;
    }
    HIPCHK(h, hipGraphLaunch(h->substep_graph, h->stream));
    return NXS_OK;
}

int ready(nxs_dyn_handle *h) {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state || !h->have_forcing)
        return fail(h, NXS_ERR_STATE, "step needs set_mesh, put_state and set_forcing first");
    if (multi_rank(h) && !h->have_halo) return fail(h, NXS_ERR_STATE, "nranks>1 needs set_halo");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    return NXS_OK;
}

int explicit_solve(nxs_dyn_handle *h) {
    const bool timed = h->cur != nullptr;
    // FE.cpp:10182-10643
    const DevMesh &m = h->dm;
    if (timed) HIPCHK(h, hipEventRecord(h->cur[0], h->stream));
    LAUNCH(h, k_prep_elements, m.Ne, m, h->ds, h->dw, h->dp);
    LAUNCH(h, k_prep_nodes, m.Nn, m, h->ds, h->dw, h->dp);
    if (timed) HIPCHK(h, hipEventRecord(h->cur[1], h->stream));
    int rc = run_substeps(h);
    if (rc) return rc;
    if (h->dp.dynamics_type == NXS_DYN_MEVP)  // FE.cpp:10559-10573
        LAUNCH(h, k_move, m.Nn, m, h->ds, 0, m.Nn, h->dp.dtime_step);
    if (timed) HIPCHK(h, hipEventRecord(h->cur[2], h->stream));
    // Q9: 50 sweeps, hard-coded (FE.cpp:10580); + open-water mesh move.  52+ small launches: replayed from
    // a second hipGraph whenever no host work is needed inside (single rank, or device-direct halo).
    auto smooth_and_tail = [&]() -> int {
        double *a = h->ds.VT, *b = h->ds.VT2;
        LAUNCH(h, k_copy_vt, 2 * m.Nn, 2 * m.Nn, a, b);
        const bool halo_in_kernel = multi_rank(h) && h->ipc_ready && !h->halo_fn && h->halo_fused && h->hf_ready && m.No > 0;
        for (int nit = 0; nit < 50; ++nit) {
            if (halo_in_kernel) {  // updateGhosts inside the sweep; the ghosts land in the array once, after the last sweep
                HaloFused hf = h->hf;
                hf.ipc = h->ipc;
                hf.from_mailbox = nit > 0;
                LAUNCH(h, k_smooth_halo, m.No, m, h->dw, (const double *)a, b, hf);
                if (nit == 49) {
                    const int tr = h->recv_offsets[h->recv_procs.size()];
                    hipLaunchKernelGGL(k_halo_pull, dim3(nblocks(tr)), dim3(BLOCK), 0, h->stream, b, h->dm, h->ds, tr, h->d_recv_index,
                                       h->d_recv_seg, h->d_recv_off, h->ipc, 0., 0, h->d_recv_procs, 1);
                }
            } else {
                LAUNCH(h, k_smooth, m.No, m, h->dw, a, b);
                if (multi_rank(h)) { int r2 = halo_exchange(h, b, 0.); if (r2) return r2; }
            }
            std::swap(a, b);
        }
        // 50 is even: the result is back in ds.VT
        LAUNCH(h, k_ow_tail, m.Nn, m, h->ds, h->dw, h->dp);
        return NXS_OK;
    };
    const bool tail_capturable = h->use_graph && (!multi_rank(h) || (h->ipc_ready && !h->halo_fn));
    if (!tail_capturable) {
        rc = smooth_and_tail();
        if (rc) return rc;
    } else {
        if (!h->tail_graph_valid) {
            if (h->tail_graph) { (void)hipGraphExecDestroy(h->tail_graph); h->tail_graph = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            rc = smooth_and_tail();
            if (rc) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(h->stream, &dead); if (dead) (void)hipGraphDestroy(dead); return rc; }
            HIPCHK(h, hipStreamEndCapture(h->stream, &g));
            hipError_t e = hipGraphInstantiate(&h->tail_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail(h, NXS_ERR_HIP, "hipGraphInstantiate(tail): %s", hipGetErrorString(e));
            h->tail_graph_valid = true;
        }
        HIPCHK(h, hipGraphLaunch(h->tail_graph, h->stream));
    }
    if (timed) HIPCHK(h, hipEventRecord(h->cur[3], h->stream));
    return NXS_OK;
}

}  // namespace

int nxs_dyn_explicit_solve(nxs_dyn_handle *h) {
    int rc = ready(h);
    if (rc) return rc;
    h->cur = nullptr;
    return explicit_solve(h);
}

int nxs_dyn_update(nxs_dyn_handle *h) {
    int rc = ready(h);
    if (rc) return rc;
    LAUNCH(h, k_update, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    return NXS_OK;
}

int nxs_dyn_step(nxs_dyn_handle *h) {  // FE.cpp:8197-8214
    int rc = ready(h);
    if (rc) return rc;
    const int type = h->dp.dynamics_type;
    if (type == NXS_DYN_FREE_DRIFT) {
        LAUNCH(h, k_free_drift, h->dm.Nn, h->dm, h->ds, h->dp);
        return NXS_OK;
    }
    if (type == NXS_DYN_NO_MOTION) return NXS_OK;
    int k = -1;
    h->cur = nullptr;
    if (h->timing_enabled) {
        k = h->set_next;
        if ((rc = harvest(h, k))) return rc;
        h->cur = h->ev[k];
    }
    rc = explicit_solve(h);
    if (rc) { h->cur = nullptr; return rc; }
    LAUNCH(h, k_update, h->dm.Ne, h->dm, h->ds, h->dw, h->dp);
    if (k >= 0) {
        HIPCHK(h, hipEventRecord(h->cur[4], h->stream));
        h->set_pending[k] = true;
        h->set_next = (k + 1) % nxs_dyn_handle::NSETS;
        h->cur = nullptr;
    }
    return NXS_OK;
}

int nxs_dyn_synchronize(nxs_dyn_handle *h) {
    if (!h) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (h->ipc_ready) {
        int err = 0;
        HIPCHK(h, hipMemcpy(&err, h->ipc.error, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(h, NXS_ERR_COMM, "device-direct halo exchange failed (%s)", err == 2 ? "self-test mismatch" : "a neighbour's flag did not arrive within 10 s");
    }
    return NXS_OK;
}

int nxs_dyn_get_timing(nxs_dyn_handle *h, nxs_dyn_timing *t) {
    if (!h || !t) return NXS_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    for (int k = 0; k < nxs_dyn_handle::NSETS; ++k) {
        int rc = harvest(h, k);
        if (rc) return rc;
    }
    const double n = h->sum_steps > 0 ? (double)h->sum_steps : 1.;
    h->timing.prep_ms = h->sum_ms[0] / n; h->timing.substeps_ms = h->sum_ms[1] / n;
    h->timing.smoother_ms = h->sum_ms[2] / n; h->timing.update_ms = h->sum_ms[3] / n;
    h->timing.total_ms = (h->sum_ms[0] + h->sum_ms[1] + h->sum_ms[2] + h->sum_ms[3]) / n;
    h->timing.steps_averaged = h->sum_steps;
    *t = h->timing;
    return NXS_OK;
}

int nxs_dyn_step_host(nxs_dyn_handle *h, nxs_dyn_state *s, const nxs_dyn_forcing *f) {
    int rc;
    if ((rc = nxs_dyn_put_state(h, s))) return rc;
    if ((rc = nxs_dyn_set_forcing(h, f))) return rc;
    if ((rc = nxs_dyn_step(h))) return rc;
    return nxs_dyn_get_state(h, s);
}

int nxs_dyn_check_regridding(nxs_dyn_handle *h, double *min_angle, int32_t *flip, int32_t *regrid_local) {
    if (!h) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state) return fail(h, NXS_ERR_STATE, "check_regridding needs set_mesh and put_state");
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_regrid_partials, dim3(h->n_partials), dim3(BLOCK), 0, h->stream, h->dm, h->ds, h->d_partials);
    hipLaunchKernelGGL(k_regrid_final, dim3(1), dim3(64), 0, h->stream, h->d_partials, h->n_partials, h->d_regrid);
    RegridPartial r;
    HIPCHK(h, hipMemcpyAsync(&r, h->d_regrid, sizeof r, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int fl = (r.min_jac <= 0.) && (r.max_jac >= 0.);  // FE.cpp:1838
    if (min_angle) *min_angle = r.min_angle;
    if (flip) *flip = fl;
    if (regrid_local) *regrid_local = (r.min_angle < h->params.regrid_angle) || fl;  // FE.cpp:8303-8305
    return NXS_OK;
}

int nxs_dyn_check_fields_fast(nxs_dyn_handle *h, int32_t *crash_local) {
    if (!h || !crash_local) return NXS_ERR_INVALID;
    if (!h->have_mesh || !h->have_state) return fail(h, NXS_ERR_STATE, "check_fields_fast needs set_mesh and put_state");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->d_crash, 0, sizeof(int), h->stream));
    LAUNCH(h, k_check_fields, std::max(h->dm.Ne, h->dm.Nn), h->dm, h->ds, h->dp, h->d_crash);
    int c = 0;
    HIPCHK(h, hipMemcpyAsync(&c, h->d_crash, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *crash_local = c;
    return NXS_OK;
}

}  // extern "C"

// nxs_dyn_kernels.inl -- device side of libnxsdyn.so: constants, device structs, every kernel of the dynamics path
// (textually included by nxs_dyn.hip, which holds the inventory and the conventions; gfx950 only).
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
    double *S4a, *S4b;  // [Ne][4] (sigma0, sigma1, sigma2, damage): the element state as the fused sub-step kernels keep it inside the sub-step loop (ping-pong pair)
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
    const int2 *pet;              // [nP][Emax] {pelem, the three corner slots in 10 bits each} -- what k_substep_fused reads: 8 bytes per element instead of 12
    const unsigned short *prow;   // [nP][W1][Pmax] NodalElementConnectivity of the own nodes in patch element slots, bamg row order, 0xFFFF = none (k_prep_fused; nullptr: not built)
    int W1;
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
    const unsigned short *pnbr;   // [nP][W2][NSmax] NodalConnectivity row of every node of N_(D-1) in patch-local slots, bamg order (Q8), 0xFFFF pad;
    int W2;                       //                 NULL when a row leaves its patch (then the smoother runs sweep by sweep)
    const int2 *pet;              // [nP][EDmax] {pelem, the three corner slots in 10 bits each}: what k_substep_pair reads (8 bytes per element instead of 12; NDmax <= 1024), else NULL
    const unsigned int *pfan8;    // [nP][4][NSmax] k_substep_pair: the first eight fan entries of every solved node as what the gather needs -- the LDS index of the corner's force
                                  // (corner * EDmax + element slot; 3 * EDmax, the pair of zeros, for pads and ghost corners), two per word: decoded on the host, once
    int own_is_block;             // the own nodes of patch q are the nodes [256 q, 256 q + 256): k_prep_nodes' open-water flag of that block applies
};
struct VTOut { double *slot[NXS_MAX_DEPTH]; };  // ring slots of the D velocities a launch produces
#define NXS_MAX_RING 129
struct VTRing { double *slot[NXS_MAX_RING]; int R; };  // the ring of velocity buffers of the deferred mesh move (k_move_ring)

struct PingPong {  // buffers a fused sub-step reads (c) and writes (n)
    const double *VTc, *Sc;  // Sc, Sn: [Ne][4] records (sigma0, sigma1, sigma2, damage), see k_pack_state
    double *VTn, *Sn;
};

struct DevWork {
    double *delta_x, *surface, *shape /*[6][Ne]*/, *emass, *ecbu;
    double *prec /*[Ne][8]: what k_prep_nodes gathers per fan entry, one 64-byte record per element (see k_prep_elements)*/;
    double *dragsurf /*[Ne]: air drag coefficient x area, the other thing k_prep_nodes' bamg-order loop sums (FE.cpp:10383-10390)*/;
    double *expC, *pmax, *heal, *dxs, *volume;  // per-step element constants of the sub-step loop
    unsigned char *eskip;                        // conc <= 0.1 (BBM) / thick == 0 (EVP)
    int *shape_range;                            // [1] != 0: a frozen coordinate or a Jacobian of this step lies outside the range in which six divisions by one divisor may share
                                                 // its reciprocal (quotients_by_one_divisor): raised by the prep kernels, lowered by k_update
    unsigned char *open_blk;                     // [ceil(Nn/BLOCK)] != 0: the block of BLOCK nodes holds a node the open-water smoother changes (zeroed by k_prep_elements, set by k_prep_nodes)
    int *dxi;                                    // BBM, fused kernel: M_delta_x as the integer it is (Q1), ~M_delta_x when the element is skipped
    double *erec;                                // [Ne][6]: (expC, volume, pmax, heal, cohesion, {dxi, eskip}) -- the fused kernels' per-step element constants as one record
    double *srec;                                // [Ne][6] M_shape_coeff as one 48-byte record per element for k_substep_multi, or NULL: the coefficients
                                                 // are then rebuilt from the staged coordinates every sub-step, as k_substep_fused always does
    double *nrec;                                // [Nn][10]: (dte / max(min_m, node_mass) -- 0 for a node without mass --, grad_ssh u, v, rlmass, C_bu, fcor, D_tau_a u, v, ocean u, v) -- their nodal inputs
    double *force /*[6][Ne]: fx0,fx1,fx2,fy0,fy1,fy2*/;
    double *rlmass, *node_mass, *C_bu, *grad_ssh /*[2Nn]*/, *fcor, *VTM /*[2Nn]*/;
    double *xy;       // [Nn][2] node coordinates (x, y) on the displaced mesh at step start (frozen over the sub-steps, Q4)
    double *D_tau_a, *D_tau_w, *D_del;
    unsigned long long *trace;  // [Ne][4] branch trace of updateSigmaDamage (option "trace_branches"; layout of ref_work.trace in oracle/dyn_ref.h), else NULL
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

// shapeCoeff (FE.cpp:1951-1964): six quotients with ONE divisor, the Jacobian.  The compiler's division is (gfx950, ROCm 7.2: v_div_scale x 2, v_rcp_f64, four FMAs
// that refine the reciprocal, q = n r, one FMA for the remainder, v_div_fmas, v_div_fixup -- eleven instructions) a reciprocal refined from the DIVISOR alone and
// three operations per numerator.  Where v_div_scale leaves both operands alone (no exponent near the ends of the range) and v_div_fixup has nothing to fix (a
// finite, non-zero divisor) the six divisions can share that reciprocal: the same instructions on the same operands in the same order, 23 instead of 66, the same
// bits as six divisions (nxs_dyn_selftest_quotients compares them bit for bit; tests/test_gpu_parity.py).  2 km: 4.93 -> 4.65 ms of sub-steps.
// The range is checked ONCE PER STEP by the prep kernels, not per element (per-element guards cost what the shared reciprocal saves: gpurun_out/r4_ab16.log): every
// frozen coordinate is zero or has a magnitude in [1e-100, 1e100] -- so every numerator, a difference of two of them, is zero or in [1e-116, 2e100] -- and every
// |Jacobian| is in [1e-100, 1e100]; v_div_scale's thresholds (exponents 768 apart, quotients or reciprocals near the denormals, numerators below 2^-970) are then
// hundreds of binades away.  One violation anywhere raises w.shape_range[0] and every sub-step kernel of the step divides six times (k_update lowers it again).
__device__ __forceinline__ bool shape_value_in_range(const double c) { const double a = fabs(c); return a == 0. || (a >= 1e-100 && a <= 1e100); }
__device__ __forceinline__ bool shape_jacobian_in_range(const double jac) { const double a = fabs(jac); return a >= 1e-100 && a <= 1e100; }
__device__ __forceinline__ void quotients_by_one_divisor(const double num[6], const double jac, double q[6], const bool in_range /*uniform: w.shape_range[0] == 0*/) {
#ifdef NXS_NO_SHARED_RCP
    const bool fast = false;
#else
    const bool fast = in_range;
#endif
    if (fast) {
        double r = __builtin_amdgcn_rcp(jac);
        double e = __builtin_fma(-jac, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-jac, r, 1.0);
        r = __builtin_fma(r, e, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double q0 = num[k] * r;
            const double rem = __builtin_fma(-jac, q0, num[k]);
            q[k] = __builtin_fma(rem, r, q0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = num[k] / jac;
    }
}

template <bool ZEROS> __device__ __forceinline__ void strain_rates(const double dxN[6], const double u[3], const double v[3], double eps[3]);
template <bool ZEROS> __device__ __forceinline__ void elastic_stress_increment(double sig[3], const double dtE, const double Da, const double Db, const double Dc, const double eps[3]);
// nxs_dyn_selftest_quotients: sextuples of numerators over one divisor, both ways, bit for bit.
//   mode 0  triangles as meshes have them: a vertex anywhere within +-4e6 m, edges of 5e2 .. 2e4 m, the numerators and the Jacobian by the kernels' own expressions
//   mode 1  operands spread over the whole range the per-step check admits: numerators zero (one in sixteen) or +-2^[-380, 330], divisors +-2^[-330, 330]
//   mode 2  not the quotients but the two sums of updateSigmaDamage with and without the reference's literal-zero terms (strain_rates, elastic_stress_increment):
//           shape coefficients +-2^[-20, -8], velocities and stresses zero of either sign (one in eight each) or +-2^[-30, 10] / +-2^[-10, 20], dt E > 0
__device__ __forceinline__ unsigned long long selftest_mix(unsigned long long z) {   // splitmix64
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(BLOCK) k_selftest_quotients(long long n, unsigned long long seed, int mode, unsigned long long *mismatches) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    unsigned long long st = seed ^ (0xd1342543de82ef95ull * (unsigned long long)(i + 1));
    auto next = [&]() { st = selftest_mix(st); return st; };
    auto unit = [&]() { return (double)(next() >> 11) * (1. / 9007199254740992.); };   // [0, 1)
    auto pow2 = [&](int lo, int hi) {   // +-2^e (1 + m), e uniform in [lo, hi]
        const unsigned long long r = next();
        const int e = lo + (int)(r % (unsigned long long)(hi - lo + 1));
        const double v = ldexp(1. + unit(), e);
        return (r >> 40) & 1ull ? -v : v;
    };
    if (mode == 2) {
        double dxN[6], u[3], v[3], sa[3], sb[3], ea[3], eb[3];
        auto maybe_zero = [&](int lo, int hi) { const unsigned long long r = next(); return (r & 7ull) == 0ull ? ((r >> 8) & 1ull ? -0. : 0.) : pow2(lo, hi); };
#pragma unroll
        for (int k = 0; k < 6; ++k) dxN[k] = pow2(-20, -8);
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = maybe_zero(-30, 10); v[k] = maybe_zero(-30, 10); sa[k] = sb[k] = maybe_zero(-10, 20); }
        const double dtE = fabs(pow2(-5, 40)), Da = fabs(pow2(0, 1)), Db = fabs(pow2(-2, -1)), Dc = fabs(pow2(-2, -1));
        strain_rates<true>(dxN, u, v, ea);
        strain_rates<false>(dxN, u, v, eb);
        elastic_stress_increment<true>(sa, dtE, Da, Db, Dc, ea);
        elastic_stress_increment<false>(sb, dtE, Da, Db, Dc, eb);
        unsigned bad = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) bad += (__double_as_longlong(ea[k]) != __double_as_longlong(eb[k]) ? 1u : 0u) + (__double_as_longlong(sa[k]) != __double_as_longlong(sb[k]) ? 1u : 0u);
        if (bad) atomicAdd(mismatches, (unsigned long long)bad);
        return;
    }
    double num[6], jac;
    if (mode == 0) {
        double vx[3], vy[3];
        vx[0] = (unit() - .5) * 8e6; vy[0] = (unit() - .5) * 8e6;
        const double h = 5e2 + unit() * 1.95e4, a0 = unit() * 6.283185307179586, a1 = a0 + .3 + unit() * 2.5;
        vx[1] = vx[0] + h * cos(a0); vy[1] = vy[0] + h * sin(a0);
        vx[2] = vx[0] + h * (.5 + unit()) * cos(a1); vy[2] = vy[0] + h * (.5 + unit()) * sin(a1);
        jac = jacobian(vx, vy);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
            num[k] = vy[kp1] - vy[kp2];
            num[k + 3] = vx[kp2] - vx[kp1];
        }
        if (!shape_jacobian_in_range(jac)) return;
    } else {
        jac = pow2(-330, 330);
#pragma unroll
        for (int k = 0; k < 6; ++k) num[k] = (next() & 15ull) == 0ull ? 0. : pow2(-380, 330);
    }
    double fast[6], slow[6];
    quotients_by_one_divisor(num, jac, fast, true);
    quotients_by_one_divisor(num, jac, slow, false);
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) bad += __double_as_longlong(fast[k]) != __double_as_longlong(slow[k]) ? 1u : 0u;
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// first entry of the 80-byte nodal record: the quotient the sub-solve forms from the nodal mass (FE.cpp:10495, nodal_solve), 0 for a node without mass (the solve
// leaves such a node alone: node_mass == 0 <=> this entry == 0, the quotient itself is never 0)
__device__ __forceinline__ double record_dte_over_mass(const DevParams &p, const double node_mass) {
    if (node_mass == 0.) return 0.;
    const double dtep = (p.dynamics_type == NXS_DYN_MEVP) ? p.dte / (p.mevp_beta + 1.) : p.dte;   // FE.cpp:10483-10493
    return dtep / STD_MAX(p.min_m, node_mass);
}

// ------------------------------------------------------------------------------------------------
// K1a  prep elements, FE.cpp:10235-10308
// LEAN: only what the fused sub-step kernels, update() and the diagnostics read is written (records, M_surface, M_delta_x); the
// one-array-per-quantity work vectors of the v1 kernels and of the debug door (M_shape_coeff, element mass, the per-step constants:
// 104 B per element of stores) are left out -- option "work_arrays" brings them back.
template <bool LEAN>
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
    if (!shape_jacobian_in_range(jac)) w.shape_range[0] = 1;   // (see quotients_by_one_divisor)
    const double surface = (1. / 2) * fabs(jac);  // FE.cpp:1929-1933
    w.surface[e] = surface;
    if (!LEAN || w.srec) {
        double sc[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // FE.cpp:1956-1962
            const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
            sc[k] = (vy[kp1] - vy[kp2]) / jac;
            sc[k + 3] = (vx[kp2] - vx[kp1]) / jac;
        }
        if (!LEAN) {
#pragma unroll
            for (int k = 0; k < 6; ++k) w.shape[(size_t)k * m.Ne + e] = sc[k];
        }
        if (w.srec && blockIdx.x * BLOCK + (int)threadIdx.x < m.Ne) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *r = reinterpret_cast<d2 *>(w.srec) + 3 * (size_t)e;
            r[0] = d2{sc[0], sc[1]}; r[1] = d2{sc[2], sc[3]}; r[2] = d2{sc[4], sc[5]};
        }
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
    if (!LEAN) w.emass[e] = element_mass;

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
    if (!LEAN) w.ecbu[e] = ecbu;

    // The record k_prep_nodes gathers for every fan entry -- what the nodal scatter loop of FE.cpp:10309-10340 takes from this
    // element besides its area, contiguous: 64 bytes, one aligned line per fan entry (as an 80-byte record with the area and the
    // drag term inside it straddled two lines, and the bamg-order loop of prep nodes fetched both again for two of its doubles:
    // 1.58 KB per node).  The products are formed with the reference's operand order, so the node side performs the same additions
    // on the same values.  The area (M_surface) and drag coefficient x area travel as compact arrays.
    __shared__ double rec[BLOCK * 8];  // staged so that the records leave the block as one contiguous stream
    {
        double *r = rec + threadIdx.x * 8;
        const double meA = element_mass * surface;           // node_mass += element_mass*surface, FE.cpp:10314
        const double m_g_A3rd = meA * (NXS_GRAVITY / 3.);    // FE.cpp:10321
        double dragp = s.drag_ui[e];                          // FE.cpp:10585-10596
        if (p.young_cat) {
            const double cy = s.cyoung[e];
            if (conc + cy > 0.) dragp = (s.drag_ui[e] * conc + s.drag_ui_young[e] * cy) / (conc + cy);
        }
        w.dragsurf[e] = dragp * surface;
        const double sshn[3] = {s.ssh[m.t0[e]], s.ssh[m.t1[e]], s.ssh[m.t2[e]]};
        r[0] = meA; r[1] = ecbu;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                         // FE.cpp:10334-10339
            const int kp1 = (j + 1) % 3, kp2 = (j + 2) % 3;
            r[2 + j] = (vy[kp1] - vy[kp2]) / jac * m_g_A3rd * sshn[j];
            r[5 + j] = (vx[kp2] - vx[kp1]) / jac * m_g_A3rd * sshn[j];
        }
    }
    __syncthreads();
    {
        const size_t base = (size_t)blockIdx.x * BLOCK * 8;
        const int count = min(BLOCK, m.Ne - (int)blockIdx.x * BLOCK) * 8;
        for (int i = threadIdx.x; i < count; i += BLOCK) w.prec[base + i] = rec[i];
    }

    // Per-step constants of the sub-step loop.  M_conc, M_thick, M_delta_x, M_surface do not change
    // while sub-cycling (Q4), so exp/pow of them are evaluated once here instead of S times; the
    // expressions are the reference's, operand for operand.
    double c_expC, c_pmax = 0., c_heal = 0.;
    int c_dxi = 0, c_skip;
    if (p.dynamics_type == NXS_DYN_BBM) {
        c_expC = exp(p.compaction_param * (1. - conc));                        // FE.cpp:4185
        c_pmax = pow(thick, p.ecf) * p.compression_factor * c_expC;            // FE.cpp:4192
        c_heal = p.dte / s.theal[e] * c_expC;                                  // FE.cpp:4257
        c_skip = (conc <= 0.1) ? 1 : 0;                                        // Q5, FE.cpp:4146-4151
        c_dxi = (conc <= 0.1) ? ~acc_div3 : acc_div3;                          // M_delta_x as the integer it is (Q1), 4 bytes instead of 9
        if (!LEAN) {
            w.expC[e] = c_expC; w.pmax[e] = c_pmax; w.heal[e] = c_heal;
            w.dxs[e] = delta_x * p.sqrt_nu_rhoi;                               // FE.cpp:4232
            w.eskip[e] = (unsigned char)c_skip;
            w.dxi[e] = c_dxi;
        }
    } else {
        c_expC = p.evp_Pstar * exp(-p.evp_C * (1. - conc));                    // FE.cpp:10684 (P)
        c_skip = (thick == 0.) ? 1 : 0;                                        // FE.cpp:10656
        if (!LEAN) {
            w.expC[e] = c_expC;
            w.eskip[e] = (unsigned char)c_skip;
        }
    }
    const double c_vol = thick * surface;                                      // FE.cpp:10450
    if (!LEAN) w.volume[e] = c_vol;
    // the same constants once more as one 48-byte record per element -- what the fused sub-step kernels read (one base pointer and
    // three 16-byte loads instead of six arrays) -- staged through LDS like the records above, so that they leave as one stream
    __syncthreads();
    {
        double *r = rec + 6 * threadIdx.x;
        r[0] = c_expC; r[1] = c_vol; r[2] = c_pmax; r[3] = c_heal; r[4] = s.cohesion[e];
        r[5] = __longlong_as_double(((long long)c_skip << 32) | (long long)(unsigned int)c_dxi);
    }
    __syncthreads();
    {
        const size_t base = (size_t)blockIdx.x * BLOCK * 6;
        const int count = min(BLOCK, m.Ne - (int)blockIdx.x * BLOCK) * 6;
        for (int i = threadIdx.x; i < count; i += BLOCK) w.erec[base + i] = rec[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K1b + K2  the nodal side of prep elements (as a gather) and prep nodes, FE.cpp:10309-10416
template <bool LEAN>
__global__ void __launch_bounds__(BLOCK) k_prep_nodes(DevMesh m, DevState s, DevWork w, DevParams p) {
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of nodes, so that the element records two rows of
    // nodes share are found in that XCD's L2 (each record is gathered by its three corner nodes)
    int blk;
    {
        const int nb = (int)gridDim.x, pos = (int)blockIdx.x, q = nb >> 3, r = nb & 7, x = pos & 7;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3);
    }
    // threads past the end redo the last node (identical values to identical places): every thread reaches the barrier below
    const int n = min(blk * BLOCK + (int)threadIdx.x, m.Nn - 1);
    const int Nn = m.Nn;
    __shared__ double rec[BLOCK * 10];
    const bool dirichlet = m.nflags[n] & NF_DIRICHLET;

    double rl = 0., nm = 0., cb = 0., gu = 0., gv = 0.;
    for (int slot = 0; slot < m.W; ++slot) {
        const int ent = m.fan[(size_t)slot * Nn + n];
        if (ent < 0) break;
        const double *r = w.prec + (size_t)(ent >> 3) * 8;
        const bool ghost_corner = ent & 4;
        rl += w.surface[ent >> 3];                     // FE.cpp:10313
        nm += r[0];                                    // FE.cpp:10314
        cb = STD_MAX(cb, r[1]);                        // FE.cpp:10317
        // Q7: the skip test sees node_mass as accumulated so far (elements <= e)
        if (dirichlet || nm == 0. || ghost_corner) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                  // FE.cpp:10334-10339
            gu -= r[2 + j];
            gv -= r[5 + j];
        }
    }
    // same expression as load_vertices(): the fused sub-step kernel rebuilds the shape coefficients from these
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 c = d2{m.x0[n] + 1. * s.UM[n], m.y0[n] + 1. * s.UM[n + Nn]};
        reinterpret_cast<d2 *>(w.xy)[n] = c;
        if (!shape_value_in_range(c.x) || !shape_value_in_range(c.y)) w.shape_range[0] = 1;   // (see quotients_by_one_divisor)
    }
    if (!LEAN) {
        w.C_bu[n] = cb;
        w.grad_ssh[n] = gu;
        w.grad_ssh[n + Nn] = gv;
    }

    // prep nodes, FE.cpp:10356-10416
    double vu = s.VT[n], vv = s.VT[n + Nn];
    if (nm == 0.) { vu = 0.; vv = 0.; s.VT[n] = 0.; s.VT[n + Nn] = 0.; }

    double drag = 0., surface = 0;
    for (int j = 0; j < m.W1; ++j) {                  // bamg row order (summation order!)
        const int e = m.n2e[(size_t)j * Nn + n];
        if (e < 0) continue;                           // Q2
        drag += w.dragsurf[e];                         // dragp * surface
        surface += w.surface[e];
    }
    const double wu = s.wind[n], wv = s.wind[n + Nn];
    drag *= NXS_RHOA * hypot(wu, wv) / surface;        // Q6
    const double tax = drag * wu, tay = drag * wv;
    w.D_tau_a[n] = tax;
    w.D_tau_a[n + Nn] = tay;

    const double fc = 2 * NXS_OMEGA * sin(m.lat[n] * NXS_PI / 180.);
    if (!LEAN) w.fcor[n] = fc;

    rl = 1. / rl;                                      // FE.cpp:10400-10402
    nm *= rl;
    rl *= 3.;
    if (!LEAN) w.rlmass[n] = rl;
    w.node_mass[n] = nm;
    if (n < m.No && !dirichlet && nm == 0.) w.open_blk[n / BLOCK] = 1;  // k_smooth's own test (FE.cpp:10589): its other blocks have nothing to do

    w.VTM[n] = vu;
    w.VTM[n + Nn] = vv;
    {   // the nodal inputs of the sub-step solve once more, as one 80-byte record per node (what the fused sub-step kernels read: one
        // base pointer, five 16-byte loads), staged through LDS so that the records leave the block as one contiguous stream
        double *r = rec + 10 * threadIdx.x;
        r[0] = record_dte_over_mass(p, nm); r[1] = gu; r[2] = gv; r[3] = rl; r[4] = cb; r[5] = fc; r[6] = tax; r[7] = tay; r[8] = s.ocean[n]; r[9] = s.ocean[n + Nn];
    }
    __syncthreads();
    {
        const size_t base = (size_t)blk * BLOCK * 10;
        const int count = min(BLOCK, Nn - blk * BLOCK) * 10;
        for (int i = threadIdx.x; i < count; i += BLOCK) w.nrec[base + i] = rec[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K1a + K1b + K2 in ONE launch over the node patches of the fused sub-step kernel (single rank, LEAN): k_prep_elements' per-element values are
// formed by the patch that needs them and handed to its own nodes through LDS instead of through 72-byte records in memory that every corner
// node gathers again (k_prep_nodes reads 596 MB on the 2 km mesh, three times the records' size).  A patch computes every element that touches
// an own node (x 1.12 at 476-node patches) and keeps of it what the nodal loops take: mass x area and C_bu (FE.cpp:10314-10317), area and
// drag x area (FE.cpp:10383-10390), and -- instead of the six ssh-gradient products -- the Jacobian and the corner slots: the node side forms
// (vy[kp1] - vy[kp2]) / jac * m_g_A3rd * ssh[j] again from the staged coordinates (the same operations on the same values, so the same bits;
// 48 bytes of LDS per element instead of 96, which is what lets two workgroups share a CU).  The element that a patch WRITES (pelem >= 0; every
// element has one writer) also gets its M_delta_x, M_surface and the 48-byte record of sub-step constants.  Operand for operand the two
// kernels above: tests/test_gpu_parity.py::test_fused_prep_kernel_does_not_change_a_bit.
#ifdef NXS_PHASE_TIMING  // kernel microscope (scripts/phase_timing.py --prep)
__device__ long long g_phase_p[8 * 8192];
#define PSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_p[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_prep_fused(DevMesh m, DevPatches pp, DevState s, DevWork w, DevParams p) {
    constexpr int T = 512;
    typedef double d2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Mmax = pp.Mmax, Emax = pp.Emax, Pmax = pp.Pmax, Mp = (Mmax + 1) & ~1, Nn = m.Nn;
    double *lx = lds, *ly = lx + Mp, *ls = ly + Mp;
    d2 *lA = reinterpret_cast<d2 *>(ls + Mp), *lB = lA + Emax;
    double *lJ = reinterpret_cast<double *>(lB + Emax);
    ushort4 *lT = reinterpret_cast<ushort4 *>(lJ + Emax);
    int blk;
    {
        const int nb = (int)gridDim.x, pos = (int)blockIdx.x, q = nb >> 3, r = nb & 7, x = pos & 7;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3);
    }
    const int t = threadIdx.x;
    const int nM = pp.node_cnt[blk], nE = pp.elem_cnt[blk], nO = pp.own_cnt[blk];
    const int *pn = pp.pnodes + (size_t)blk * Mmax;
    PSTAMP(0);
    for (int i = t; i < nM; i += T) {  // GmshMesh::vertices(indices, um, 1.) per node (load_vertices' expression), and the ssh the elements average
        const int g = pn[i];
        const double x = m.x0[g] + 1. * s.UM[g], y = m.y0[g] + 1. * s.UM[g + Nn];
        lx[i] = x; ly[i] = y; ls[i] = s.ssh[g];
        if (i < nO) {
            reinterpret_cast<d2 *>(w.xy)[g] = d2{x, y};
            if (!shape_value_in_range(x) || !shape_value_in_range(y)) w.shape_range[0] = 1;   // (see quotients_by_one_divisor)
        }
    }
    __syncthreads();
    PSTAMP(1);
    // ---- k_prep_elements (FE.cpp:10235-10308), two elements of this thread at a time: their indices, then all their fields, then the arithmetic --
    // element by element the two dependent load levels would be paid once per element
#ifndef NXS_PREP_NB
#define NXS_PREP_NB 1   // elements of a thread in flight at a time: 2 spills (28 B of scratch per lane) and is no faster once the next round's indices are prefetched
#endif
    constexpr int NB = NXS_PREP_NB;
    // (the indices of the round after this one are asked for before this round's fields: a round then starts with its field loads instead of an index hop)
    int eraw_next[NB];
    ushort4 trs_next[NB];
    auto load_indices = [&](const int l0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int l = l0 + j * T;
            eraw_next[j] = 0; trs_next[j] = make_ushort4(0, 0, 0, 0);
            if (l < nE) { eraw_next[j] = pp.pelem[(size_t)blk * Emax + l]; trs_next[j] = reinterpret_cast<const ushort4 *>(pp.ptri)[(size_t)blk * Emax + l]; }
        }
    };
    load_indices(t);
    for (int l0 = t; l0 < nE; l0 += NB * T) {
        int eraw[NB];
        ushort4 trs[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { eraw[j] = eraw_next[j]; trs[j] = trs_next[j]; }
        if (l0 + NB * T < nE) load_indices(l0 + NB * T);
        double f_conc[NB], f_thick[NB], f_snow[NB], f_cy[NB], f_hy[NB], f_hsy[NB], f_depth[NB], f_drag[NB], f_dragy[NB], f_theal[NB], f_coh[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int l = l0 + j * T;
            const int e = eraw[j] >= 0 ? eraw[j] : ~eraw[j];
            f_conc[j] = f_thick[j] = f_snow[j] = f_cy[j] = f_hy[j] = f_hsy[j] = f_depth[j] = f_drag[j] = f_dragy[j] = 0.; f_theal[j] = 1.; f_coh[j] = 0.;
            if (l < nE) {
                f_conc[j] = s.conc[e]; f_thick[j] = s.thick[e]; f_snow[j] = s.snow[e]; f_depth[j] = s.depth[e]; f_drag[j] = s.drag_ui[e];
                if (p.young_cat) { f_cy[j] = s.cyoung[e]; f_hy[j] = s.hyoung[e]; f_hsy[j] = s.hsyoung[e]; f_dragy[j] = s.drag_ui_young[e]; }
                if (eraw[j] >= 0) { f_coh[j] = s.cohesion[e]; if (p.dynamics_type == NXS_DYN_BBM) f_theal[j] = s.theal[e]; }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
        const int l = l0 + j * T;
        if (l >= nE) continue;
        const bool writer = eraw[j] >= 0;
        const int e = writer ? eraw[j] : ~eraw[j];
        const ushort4 tr = trs[j];
        const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
        const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
        const double jac = jacobian(vx, vy);
        if (!shape_jacobian_in_range(jac)) w.shape_range[0] = 1;   // (see quotients_by_one_divisor)
        const double surface = (1. / 2) * fabs(jac);  // FE.cpp:1929-1933
        const double conc = f_conc[j], thick = f_thick[j];
        double total_concentration = conc, total_thickness = thick, total_snow = f_snow[j];
        if (p.young_cat) {
            total_concentration += f_cy[j];
            total_thickness += f_hy[j];
            total_snow += f_hsy[j];
        }
        double element_mass = 0.;
        if (total_concentration > 0.)
            element_mass = (NXS_RHOI * total_thickness + NXS_RHOS * total_snow) / total_concentration;
        double element_ssh = 0;
        element_ssh += ls[tr.x];
        element_ssh += ls[tr.y];
        element_ssh += ls[tr.z];
        element_ssh /= 3.;
        const double max_keel_depth = 28;
        const double min_water_depth = 2.;
        const double depth_eff = STD_MAX(0., element_ssh + STD_MAX(min_water_depth, f_depth[j]));
        double critical_h = 0., critical_h_mod = 0.;
        if (p.basal_stress_type == NXS_BASAL_LEMIEUX) {
            double mean_keel_depth = p.k1 * thick;
            mean_keel_depth = STD_MIN(mean_keel_depth, conc * max_keel_depth);
            critical_h = conc * depth_eff / p.k1;
            critical_h_mod = mean_keel_depth / p.k1;
        }
        const double ecbu = p.k2 * STD_MAX(0., critical_h_mod - critical_h) * exp(-p.Cb * (1. - conc));
        const double meA = element_mass * surface;           // FE.cpp:10314
        double dragp = f_drag[j];                             // FE.cpp:10585-10596
        if (p.young_cat) {
            const double cy = f_cy[j];
            if (conc + cy > 0.) dragp = (f_drag[j] * conc + f_dragy[j] * cy) / (conc + cy);
        }
        lA[l] = d2{meA, ecbu};
        lB[l] = d2{surface, dragp * surface};
        lJ[l] = jac;
        lT[l] = tr;
        if (!writer) continue;
        // Q1 (FE.cpp:10239): int accumulator, then unsigned integer division by 3
        const double side0 = hypot(vx[1] - vx[0], vy[1] - vy[0]);
        const double side1 = hypot(vx[2] - vx[1], vy[2] - vy[1]);
        const double side2 = hypot(vx[2] - vx[0], vy[2] - vy[0]);
        int acc = 0;
        acc = (int)(acc + side0);
        acc = (int)(acc + side1);
        acc = (int)(acc + side2);
        const int acc_div3 = (int)((unsigned long)acc / 3ul);
        w.delta_x[e] = (double)((unsigned long)acc / 3ul);
        w.surface[e] = surface;
        if (w.srec) {
            double sc[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // FE.cpp:1956-1962
                const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                sc[k] = (vy[kp1] - vy[kp2]) / jac;
                sc[k + 3] = (vx[kp2] - vx[kp1]) / jac;
            }
            d2 *r = reinterpret_cast<d2 *>(w.srec) + 3 * (size_t)e;
            r[0] = d2{sc[0], sc[1]}; r[1] = d2{sc[2], sc[3]}; r[2] = d2{sc[4], sc[5]};
        }
        double c_expC, c_pmax = 0., c_heal = 0.;
        int c_dxi = 0, c_skip;
        if (p.dynamics_type == NXS_DYN_BBM) {
            c_expC = exp(p.compaction_param * (1. - conc));                        // FE.cpp:4185
            c_pmax = pow(thick, p.ecf) * p.compression_factor * c_expC;            // FE.cpp:4192
            c_heal = p.dte / f_theal[j] * c_expC;                                  // FE.cpp:4257
            c_skip = (conc <= 0.1) ? 1 : 0;                                        // Q5, FE.cpp:4146-4151
            c_dxi = (conc <= 0.1) ? ~acc_div3 : acc_div3;
        } else {
            c_expC = p.evp_Pstar * exp(-p.evp_C * (1. - conc));                    // FE.cpp:10684 (P)
            c_skip = (thick == 0.) ? 1 : 0;                                        // FE.cpp:10656
        }
        const double c_vol = thick * surface;                                      // FE.cpp:10450
        d2 *r = reinterpret_cast<d2 *>(w.erec) + 3 * (size_t)e;
        r[0] = d2{c_expC, c_vol}; r[1] = d2{c_pmax, c_heal};
        r[2] = d2{f_coh[j], __longlong_as_double(((long long)c_skip << 32) | (long long)(unsigned int)c_dxi)};
    }
    }
    PSTAMP(2);
    __syncthreads();
    PSTAMP(3);
    const unsigned short *pf = pp.pfan + (size_t)blk * pp.Wp * Pmax;
    const unsigned short *pr = pp.prow + (size_t)blk * pp.W1 * Pmax;
    for (int sl = t; sl < nO; sl += T) {  // ---- k_prep_nodes (FE.cpp:10309-10416)
        const int n = pn[sl];
        // every load of this node is issued before anything is used (the fan loop below ends at the first pad: fetched entry by entry it is a chain
        // of up to Wp dependent global loads per node)
        unsigned ents[8], rows[10];
#pragma unroll
        for (int k = 0; k < 8; ++k) ents[k] = k < pp.Wp ? pf[(size_t)k * Pmax + sl] : 0xFFFFu;
#pragma unroll
        for (int j = 0; j < 10; ++j) rows[j] = j < pp.W1 ? pr[(size_t)j * Pmax + sl] : 0xFFFFu;
        const bool dirichlet = m.nflags[n] & NF_DIRICHLET;
        const double vt_u = s.VT[n], vt_v = s.VT[n + Nn], wu = s.wind[n], wv = s.wind[n + Nn], latn = m.lat[n], ocu = s.ocean[n], ocv = s.ocean[n + Nn];
        double rl = 0., nm = 0., cb = 0., gu = 0., gv = 0.;
        auto fan_entry = [&](const unsigned ent) {
            const unsigned slot = ent >> 3;
            const d2 A = lA[slot];
            rl += lB[slot].x;                              // FE.cpp:10313
            nm += A.x;                                     // FE.cpp:10314
            cb = STD_MAX(cb, A.y);                         // FE.cpp:10317
            // Q7: the skip test sees node_mass as accumulated so far (elements <= e)
            if (dirichlet || nm == 0. || (ent & 4u)) return;
            const ushort4 tr = lT[slot];
            const double jac = lJ[slot];
            const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
            const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
            const double sshn[3] = {ls[tr.x], ls[tr.y], ls[tr.z]};
            const double m_g_A3rd = A.x * (NXS_GRAVITY / 3.);    // FE.cpp:10321
#pragma unroll
            for (int j = 0; j < 3; ++j) {                  // FE.cpp:10334-10339
                const int kp1 = (j + 1) % 3, kp2 = (j + 2) % 3;
                gu -= (vy[kp1] - vy[kp2]) / jac * m_g_A3rd * sshn[j];
                gv -= (vx[kp2] - vx[kp1]) / jac * m_g_A3rd * sshn[j];
            }
        };
        bool more = true;  // (the fan ends at its first pad)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            more = more && ents[k] != 0xFFFFu;
            if (more) fan_entry(ents[k]);
        }
        for (int k = 8; more && k < pp.Wp; ++k) {
            const unsigned ent = pf[(size_t)k * Pmax + sl];
            if (ent == 0xFFFFu) break;
            fan_entry(ent);
        }
        // prep nodes, FE.cpp:10356-10416
        double vu = vt_u, vv = vt_v;
        if (nm == 0.) { vu = 0.; vv = 0.; s.VT[n] = 0.; s.VT[n + Nn] = 0.; }
        double drag = 0., surface = 0;
        auto row_entry = [&](const unsigned se) {          // bamg row order (summation order!)
            if (se == 0xFFFFu) return;                     // Q2
            const d2 B = lB[se];
            drag += B.y;                                   // dragp * surface
            surface += B.x;
        };
#pragma unroll
        for (int j = 0; j < 10; ++j) row_entry(rows[j]);
        for (int j = 10; j < pp.W1; ++j) row_entry(pr[(size_t)j * Pmax + sl]);
        drag *= NXS_RHOA * hypot(wu, wv) / surface;        // Q6
        const double tax = drag * wu, tay = drag * wv;
        w.D_tau_a[n] = tax;
        w.D_tau_a[n + Nn] = tay;
        const double fc = 2 * NXS_OMEGA * sin(latn * NXS_PI / 180.);
        rl = 1. / rl;                                      // FE.cpp:10400-10402
        nm *= rl;
        rl *= 3.;
        w.node_mass[n] = nm;
        if (n < m.No && !dirichlet && nm == 0.) w.open_blk[n / BLOCK] = 1;
        w.VTM[n] = vu;
        w.VTM[n + Nn] = vv;
        d2 *r = reinterpret_cast<d2 *>(w.nrec) + 5 * (size_t)n;
        r[0] = d2{record_dte_over_mass(p, nm), gu}; r[1] = d2{gv, rl}; r[2] = d2{cb, fc}; r[3] = d2{tax, tay}; r[4] = d2{ocu, ocv};
    }
    PSTAMP(4);
}

// Several ranks: k_prep_fused runs over the patches of a rank's OWN nodes (every element touching one is in its patch: the fans are complete); the reference's
// loops also visit the GHOST nodes (FE.cpp:10309, 10356 run over M_num_nodes), whose local fans are partial -- their frozen coordinates are staged by the sub-step
// kernels, their M_VT is zeroed where their local mass is, D_tau_a of all nodes goes to the coupler.  This pass does them: one thread per ghost node, the values of
// the few elements around it formed again with k_prep_elements' expressions (a rank of eight of the 2 km mesh has 793 ghosts).  Operand for operand k_prep_nodes
// on a ghost node: its grad_ssh sums stay zero (every corner a ghost holds is flagged ghostNodes[i], FE.cpp:10328).
__global__ void __launch_bounds__(BLOCK) k_prep_ghost_nodes(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int n = m.No + blockIdx.x * BLOCK + (int)threadIdx.x;
    if (n >= m.Nn) return;
    const int Nn = m.Nn;
    typedef double d2 __attribute__((ext_vector_type(2)));
    // what the nodal loops take from element e: its area, mass x area, C_bu and drag coefficient x area (k_prep_elements, FE.cpp:10235-10308)
    auto element_values = [&](const int e, double &surface, double &meA, double &ecbu, double &dragsurf) {
        double vx[3], vy[3];
        load_vertices(m, s.UM, e, vx, vy);
        const double jac = jacobian(vx, vy);
        surface = (1. / 2) * fabs(jac);  // FE.cpp:1929-1933
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
        ecbu = p.k2 * STD_MAX(0., critical_h_mod - critical_h) * exp(-p.Cb * (1. - conc));
        meA = element_mass * surface;                        // FE.cpp:10314
        double dragp = s.drag_ui[e];                          // FE.cpp:10585-10596
        if (p.young_cat) {
            const double cy = s.cyoung[e];
            if (conc + cy > 0.) dragp = (s.drag_ui[e] * conc + s.drag_ui_young[e] * cy) / (conc + cy);
        }
        dragsurf = dragp * surface;
    };
    double rl = 0., nm = 0., cb = 0.;
    const double gu = 0., gv = 0.;   // (ghost corners are skipped by the ssh-gradient scatter)
    for (int slot = 0; slot < m.W; ++slot) {
        const int ent = m.fan[(size_t)slot * Nn + n];
        if (ent < 0) break;
        double surface, meA, ecbu, dragsurf;
        element_values(ent >> 3, surface, meA, ecbu, dragsurf);
        rl += surface;                                 // FE.cpp:10313
        nm += meA;                                     // FE.cpp:10314
        cb = STD_MAX(cb, ecbu);                        // FE.cpp:10317
    }
    {
        const d2 c = d2{m.x0[n] + 1. * s.UM[n], m.y0[n] + 1. * s.UM[n + Nn]};
        reinterpret_cast<d2 *>(w.xy)[n] = c;
        if (!shape_value_in_range(c.x) || !shape_value_in_range(c.y)) w.shape_range[0] = 1;   // (see quotients_by_one_divisor)
    }
    // prep nodes, FE.cpp:10356-10416
    double vu = s.VT[n], vv = s.VT[n + Nn];
    if (nm == 0.) { vu = 0.; vv = 0.; s.VT[n] = 0.; s.VT[n + Nn] = 0.; }
    double drag = 0., surface_sum = 0;
    for (int j = 0; j < m.W1; ++j) {                  // bamg row order (summation order!)
        const int e = m.n2e[(size_t)j * Nn + n];
        if (e < 0) continue;                           // Q2
        double surface, meA, ecbu, dragsurf;
        element_values(e, surface, meA, ecbu, dragsurf);
        drag += dragsurf;
        surface_sum += surface;
    }
    const double wu = s.wind[n], wv = s.wind[n + Nn];
    drag *= NXS_RHOA * hypot(wu, wv) / surface_sum;   // Q6
    const double tax = drag * wu, tay = drag * wv;
    w.D_tau_a[n] = tax;
    w.D_tau_a[n + Nn] = tay;
    const double fc = 2 * NXS_OMEGA * sin(m.lat[n] * NXS_PI / 180.);
    rl = 1. / rl;                                      // FE.cpp:10400-10402
    nm *= rl;
    rl *= 3.;
    w.node_mass[n] = nm;
    w.VTM[n] = vu;
    w.VTM[n + Nn] = vv;
    d2 *r = reinterpret_cast<d2 *>(w.nrec) + 5 * (size_t)n;
    r[0] = d2{record_dte_over_mass(p, nm), gu}; r[1] = d2{gv, rl}; r[2] = d2{cb, fc}; r[3] = d2{tax, tay}; r[4] = d2{s.ocean[n], s.ocean[n + Nn]};
}

// ------------------------------------------------------------------------------------------------
// Arithmetic shared by the v1 kernels (one per reference loop) and the v2 fused sub-step kernel, so
// that both perform literally the same operations in the same order.

// The two sums of updateSigmaDamage that the reference forms with literal zeros among their terms.
//   strain rates (FE.cpp:4167-4176): eighteen products of M_B0T with the corner velocities, six of them with a literal zero.  A product 0 * x of a finite x is
//   +-0, and a sum that started at +0 and has only ever had products added to it is never -0 (+0 + -0 = +0, a + -a = +0), so adding +-0 returns it unchanged:
//   the twelve remaining operations, in the reference's order, give the same bits for every finite velocity.
//   stress increment (FE.cpp:4204-4210): nine terms dt E D_ij eps_j, four of them with a literal zero of M_Dunit ([[a, b, 0], [b, a, 0], [0, 0, c]] by construction,
//   FE.cpp:1491-1507).  Each of those is +-0; the sum it is added to is -0 only if the stress component and both of its other terms are -0, which takes dt E == 0
//   (damage == 1 exactly) on an exactly zero stress: there, and only there, a zero stress can come out as -0 where the reference has +0.
// ZEROS = true performs every product and addition of the reference's loops (what rounds 1-3 ran); false leaves the literal zeros out: 24 instructions fewer per
// element update and, in k_substep_pair, 16 registers fewer -- the room in which the own nodes' inputs now stay between the two solves.
// nxs_dyn_selftest_quotients mode 2 forms both variants on random operands, zeros of both signs among the velocities and stresses: the same bits.
template <bool ZEROS>
__device__ __forceinline__ void strain_rates(const double dxN[6], const double u[3], const double v[3], double eps[3]) {
    eps[0] = eps[1] = eps[2] = 0.;
    if (ZEROS) {
        double B0T[18];   // M_B0T (FE.cpp:10242-10249) rebuilt in registers
#pragma unroll
        for (int i = 0; i < 18; ++i) B0T[i] = 0.;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            B0T[2 * i] = dxN[i];
            B0T[2 * i + 13] = dxN[i];
            B0T[2 * i + 7] = dxN[i + 3];
            B0T[2 * i + 12] = dxN[i + 3];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                eps[i] += B0T[i * 6 + 2 * j] * u[j];
                eps[i] += B0T[i * 6 + 2 * j + 1] * v[j];
            }
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            eps[0] += dxN[j] * u[j];
            eps[1] += dxN[j + 3] * v[j];
            eps[2] += dxN[j + 3] * u[j];
            eps[2] += dxN[j] * v[j];
        }
    }
}
template <bool ZEROS>
__device__ __forceinline__ void elastic_stress_increment(double sig[3], const double dtE, const double Da, const double Db, const double Dc, const double eps[3]) {
    const double Dm[9] = {Da, Db, 0., Db, Da, 0., 0., 0., Dc};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (ZEROS || (i < 2) == (j < 2)) sig[i] += dtE * Dm[3 * i + j] * eps[j];
}
#ifdef NXS_KEEP_ZERO_TERMS
#define NXS_ZERO_TERMS true
#else
#define NXS_ZERO_TERMS false
#endif

// updateSigmaDamage body for one element, FE.cpp:4161-4257 (the conc <= 0.1 early-out is the caller's)
template <bool POW4>
__device__ __forceinline__ void bbm_stress(const DevParams &p, const double dxN[6], const double u[3], const double v[3],
                                           double sig[3], double &damage, const double expC, const double Pmax,
                                           const double heal, const double dxs, const double cohesion, double *dcrit_out = nullptr) {
    const double dt = p.dte;
    double eps[3];
    strain_rates<NXS_ZERO_TERMS>(dxN, u, v, eps);
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
    // M_Dunit (FE.cpp:1491-1507) is [[a, b, 0], [b, a, 0], [0, 0, c]] by construction (nxs_dyn.hip fills p.D the same way): three
    // uniform values instead of nine keep 12 SGPRs out of the hot loop, whose scalar registers spill into VGPR lanes
    const double Da = p.D[0], Db = p.D[1], Dc = p.D[8];
    elastic_stress_increment<NXS_ZERO_TERMS>(sig, dt * elasticity, Da, Db, Dc, eps);   // FE.cpp:4204-4210
#pragma unroll
    for (int i = 0; i < 3; ++i) sig[i] *= multiplicator;
    const double sigma_s = hypot((sig[0] - sig[1]) / 2., sig[2]);                // FE.cpp:4218
    sigma_n = (sig[0] + sig[1]) * 0.5;
    // FE.cpp:4221-4227: one of two quotients -- the operands are chosen first, then ONE division forms it (lanes of a wave take different sides: as two branches
    // both division sequences ran, 11 instructions each)
    const bool crushed = sigma_n < -p.compr_strength;
    const double dcrit_num = crushed ? -p.compr_strength : cohesion;
    const double dcrit_den = crushed ? sigma_n : sigma_s + p.tan_phi * sigma_n;
    const double dcrit = dcrit_num / dcrit_den;
    if (dcrit_out) *dcrit_out = dcrit;                                           // (branch trace only)
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
// DTM: the caller hands over dtep / max(min_m, node_mass) itself -- the nodal records hold that quotient (it does not change during the sub-steps of a step, so the
// prep kernels divide once per node and step, by the same expression, instead of the solve once per node and sub-step) -- instead of the nodal mass.
template <bool DTM = false>
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
    const double dte_over_mass = DTM ? node_mass : dtep / STD_MAX(p.min_m, node_mass);
    const double c_prime = NXS_RHOW * p.qdw * hypot(ou - uice, ov - vice);
    // (FE.cpp:10498: where no ice is grounded C_bu is +0 and the quotient, over a positive denominator, +0 too: a wave without a grounded node skips the hypot and the division)
    const double tau_b = (C_bu == 0. && p.u0 > 0.) ? 0. : C_bu / (hypot(uice, vice) + p.u0);
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
// one sub-step of one element into the branch trace: the same record as trace_branch() of oracle/dyn_ref.c
__device__ __forceinline__ void trace_branch(unsigned long long *t, int code, double dcrit, double conc) {
    t[0] = t[0] * 0x9E3779B97F4A7C15ull + (unsigned long long)code + 1ull;
    if (code == 1) t[1]++;
    if (code != 2 && fabs(dcrit - 1.) < 1e-9) t[2] |= 1ull;
    if (fabs(conc - 0.1) < 1e-12) t[2] |= 2ull;
    if (code == 2) t[2] |= 4ull;
    t[3]++;
}

template <bool POW4, bool TRACE = false>
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
        if (TRACE) trace_branch(w.trace + 4 * (size_t)e, 2, 0., s.conc[e]);
    } else {
        const int n[3] = {m.t0[e], m.t1[e], m.t2[e]};
        double u[3], v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { u[j] = s.VT[n[j]]; v[j] = s.VT[n[j] + Nn]; }
        sig[0] = s.s0[e]; sig[1] = s.s1[e]; sig[2] = s.s2[e];
        double damage = s.damage[e];
        if (TRACE) {
            double dcrit;
            bbm_stress<POW4>(p, dxN, u, v, sig, damage, w.expC[e], w.pmax[e], w.heal[e], w.dxs[e], s.cohesion[e], &dcrit);
            trace_branch(w.trace + 4 * (size_t)e, ((0. < dcrit) && (dcrit < 1.)) ? 1 : 0, dcrit, s.conc[e]);
        } else {
            bbm_stress<POW4>(p, dxN, u, v, sig, damage, w.expC[e], w.pmax[e], w.heal[e], w.dxs[e], s.cohesion[e]);
        }
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
    // The open-water smoother's own mailbox (k_smooth_halo): one slot per sweep, so a sweep never overwrites what a neighbour may
    // still be reading and needs no acknowledgement -- which is what lets a receiver stop waiting for a direction whose values
    // cannot change during the sweeps ("static this step": no ice-free node among the nodes sent that way).
    double *smb;                    // my slots: [NXS_SMOOTH_SWEEPS][2*tr] doubles, laid out like a mailbox half
    unsigned long long *sflags;     // [nr] written by the neighbours: epoch * 64 + sweeps published
    unsigned long long *sstatic;    // [nr] written by the neighbours with sweep 0: (epoch << 1) | static
    double *const *peer_smb;        // [ns] neighbour k's slot-0 address of MY segment; slot stride = peer_parity_stride[k]
    unsigned long long *const *peer_sflag, *const *peer_sstatic;  // [ns] my entries in neighbour k's sflags / sstatic
    unsigned long long *epoch;      // smoother passes completed by this rank (one per step; the ranks stay aligned)
    int *my_static;                 // [ns] 1 = nothing I send to neighbour k can change during this step's sweeps (reset to 1 by k_smooth_pull)
    int *peer_static;               // [nr] what neighbour k said about its sends to me (cached by sweep 1 for the later ones)
    unsigned delay;                 // test door "ipc_delay" (include/nxs_dyn.h): point << 8 | units of 10 us, set on ONE rank; 0 (always, outside the protocol tests) = none
};
#define NXS_SMOOTH_SWEEPS 50  // FE.cpp:10580 (Q9: hard-coded in the reference)

// Test door "ipc_delay": the named rank sleeps here when `point` is the named point -- a deterministic widening of one window of the exchange protocols, so that an
// ordering the protocol does not enforce shows as wrong bits instead of depending on who wins a race of a few microseconds (tests/test_gpu_protocol_delays.py).
// d is uniform (a kernel argument or a scalar load): one scalar compare per point where the option is off.
__device__ __forceinline__ void nxs_delay_at(const unsigned d, const unsigned point) {
    if (d != 0u && (d >> 8) == point) {
        const long long t0 = wall_clock64(), ticks = (long long)(d & 0xffu) * 1000ll;  // wall_clock64 runs at 100 MHz: units of 10 us
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    }
}

// agent scope (sc1): written through to memory / read past this CU's L1 -- what workgroups of ONE launch hand each other (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// the same for 16-byte records: a buffer resource over the array, byte offsets per lane (buffer_load/store_dwordx4 ... offen sc1)
typedef unsigned int nxs_u4 __attribute__((ext_vector_type(4)));
typedef double nxs_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000 /*gfx9 raw buffer: DATA_FORMAT 32*/);
}
__device__ __forceinline__ nxs_d2 ld_agent16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const nxs_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /*sc1*/);
    nxs_d2 o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}
__device__ __forceinline__ void st_agent16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, nxs_d2 x) {
    nxs_u4 v;
    __builtin_memcpy(&v, &x, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16 /*sc1*/);
}
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
// (the sub-step kernel receives a POINTER to a device copy of this struct: only its boundary patches touch the tables, and held by
// value their fifteen pointers sat in the scalar registers of every patch's hot loop -- 61 spilled SGPRs against 4 without the halo)
struct HaloFused {
    IpcDev ipc;
    int n_boundary;                    // patches [0, n_boundary) are the boundary patches (re-uploaded in that order)
    const int *send_ptr;               // [No+1] CSR over own nodes
    const int *send_k, *send_pos;      // neighbour (index into send_procs) and position inside its segment
    const int *send_off;               // [ns+1] segment offsets (segment length = v offset)
    const int *ghost_off, *ghost_srl;  // [Nn-No] u offset inside a mailbox half, and the v offset from it
    const int *ghost_k;                // [Nn-No] the receive neighbour (index into recv_procs) the ghost comes from
    int No;
    int from_mailbox;                  // 0: first sub-step of a step, the ghosts are in the VT buffer
    unsigned int *done_all;            // k_smooth_halo: two-level ticket counters, [0] global, [32 (g+1)] group g
    const int *send_block_rank;        // k_smooth_halo: rank of block b among the blocks that send something, -1: sends nothing
    int n_send_blocks;
    // Option "resident_release" = 0: the resident loops raise a sub-step's flags WITHOUT the system-scope release in front of them.  What the flags publish are the
    // mailbox stores of OTHER workgroups -- write-through system-scope stores into uncached memory, each drained by its own wave (s_waitcnt vmcnt(0)) before that
    // workgroup took its ticket; the publishing lane's fence (a write-back of ITS XCD's L2 and a wait for ITS wave's stores) reaches none of them, it only costs
    // 0.65 us of the 2.5 us between the last boundary patch's barrier and the flags being seen, in every sub-step (a rank of eight: 1.03 -> 0.95 ms of sub-steps).
    // The default keeps it: no run on more than one device has told the two apart yet; bench.py tries both where every rank has a device of its own and keeps the
    // faster one only if it gives the bits of the separate kernels.
    int no_release;
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
template <int T, bool POW4, int NTM, bool HALO, bool PMEM>
__global__ void __launch_bounds__(T) k_substep_fused(DevMesh m, DevPatches pp, DevState s, DevWork w, DevParams pval, const DevParams *__restrict__ pdev,
                                                     PingPong b, double move_dt, const HaloFused *__restrict__ hfp, int n_boundary, int from_mailbox) {
    // PMEM: the parameters are read from device memory, per phase (the element phase and the node phase need disjoint halves of them);
    // as a by-value argument all of them sit in scalar registers through the whole kernel and push other values into VGPR lanes
    // (0 spilled SGPRs instead of 4, and of 35 with the exchange).  That wins where a workgroup's latency binds (one round of
    // patches: 182 k triangles 1.375 -> 1.344 ms/step) and loses where several rounds stream (2 km: 7.07 -> 7.36), hence both.
    const DevParams &p = PMEM ? *pdev : pval;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lu = lds, *lv = lds + pp.Mmax, *lx = lds + 2 * (size_t)pp.Mmax, *ly = lds + 3 * (size_t)pp.Mmax,
           *lF = lds + 4 * (size_t)pp.Mmax;  // the corner forces as (x, y) pairs, [3][Emax], + one pair of zeros (see fan_gather8)
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 *lF2 = reinterpret_cast<d2 *>(lF);
    const unsigned ZIDX = 3u * (unsigned)pp.Emax;
    if (threadIdx.x == 0) lF2[ZIDX] = d2{0., 0.};
    const bool shape_in_range = w.shape_range[0] == 0;   // (uniform; see quotients_by_one_divisor)
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
        boundary = blk < n_boundary;
        if (boundary || n_boundary == 0) xseq = *hfp->ipc.seq_push;  // interior patches never look at it: the last boundary patch may advance it while they run
        blk = boundary ? xcd_remap(blk, n_boundary) : n_boundary + xcd_remap(blk - n_boundary, (int)gridDim.x - n_boundary);
    } else {
        blk = xcd_remap(blk, (int)gridDim.x);
    }
    const int t = threadIdx.x, Nn = m.Nn, Emax = pp.Emax;
    const int nM = pp.node_cnt[blk], nE = pp.elem_cnt[blk], nO = pp.own_cnt[blk];
    const int *pn = pp.pnodes + (size_t)blk * pp.Mmax;
    const int2 *pet = pp.pet + (size_t)blk * Emax;
    const bool bbm = p.dynamics_type == NXS_DYN_BBM;
    constexpr bool NT_S = NTM & 1, NT_U = NTM & 2, NT_C = NTM & 4;  // sigma/damage, UM/UT, element constants

    // The kernel is latency-bound unless every dependent load hop is overlapped, so all global loads
    // that do not need LDS are issued up front: indices first, then (one hop later) the nodal
    // velocities to stage and this thread's element data, all in flight together before barrier 1.
    // The index rows are padded to Mmax / Emax, so these loads depend on the launch arguments only -- not on the
    // patch's counts (one more dependent hop; the counts arrive meanwhile and mask the uses).
    const int my_node = (t < pp.Mmax) ? pn[t] : 0;  // patch-local slot t (an own node when t < nO)
    const int my_node2 = (t + T < pp.Mmax) ? pn[t + T] : 0;  // a patch stages ~1.25 nodes per own node: second staging slot
    int2 et = make_int2(0, 0);  // {element, corner slots}
    if (t < Emax) et = pet[t];
#if NXS_PF >= 1
    // element rounds 1 and 2 (a patch holds ~2.2 elements per own node): their indices are fetched now, so
    // that a later round starts with its data loads instead of an index hop
    int2 et1 = et, et2 = et;
    if (t + T < Emax) et1 = pet[t + T];
    if (t + 2 * T < Emax) et2 = pet[t + 2 * T];
#endif

    const bool mailbox_ghosts = HALO && boundary && from_mailbox;
    if (mailbox_ghosts) {  // exchange xseq-1 must have landed before a ghost node is staged
        if (t == 0) {
            const long long t0 = wall_clock64();  // 100 MHz
            bool ok = true;
            unsigned polls = 0u;
            for (int k = 0; k < hfp->ipc.nr && ok; ++k)
                while (__hip_atomic_load(hfp->ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < xseq) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++polls & 31u) != 0u) continue;  // the error word is another round trip and the clock a few hundred cycles: every 32nd poll, so that a poll period is ONE round trip
                    if (__hip_atomic_load(hfp->ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }  // a wait already timed out: the run is lost, do not wait again
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(hfp->ipc.error, 3); break; }  // 10 s
                }
            // no acquire fence: the mailbox is uncached memory and every read of it below is a system-scope load that
            // bypasses the caches; a fence here would invalidate this XCD's caches once per boundary patch and sub-step
            nxs_delay_at(hfp->ipc.delay, NXS_DELAY_STAGE_READ);
        }
        __syncthreads();
    }
    auto stage = [&](int i, int g) {
        if (mailbox_ghosts && g >= hfp->No) {
            const double *src = hfp->ipc.mailbox + ((xseq - 1ull) & 1ull) * 2ull * (unsigned long long)hfp->ipc.tr + hfp->ghost_off[g - hfp->No];
            const double u = sys_load(src), v = sys_load(src + hfp->ghost_srl[g - hfp->No]);
            lu[i] = u; lv[i] = v;
            const_cast<double *>(b.VTc)[g] = u;  // every patch that stages g writes the same two values
            const_cast<double *>(b.VTc)[g + Nn] = v;
        } else {
            lu[i] = b.VTc[g];
            lv[i] = b.VTc[g + Nn];
        }
        {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 c = reinterpret_cast<const d2 *>(w.xy)[g];
            lx[i] = c.x; ly[i] = c.y;
        }
    };
    if (t < nM) stage(t, my_node);
    if (t + T < nM) stage(t + T, my_node2);
    for (int i = t + 2 * T; i < nM; i += T) stage(i, pn[i]);

    for (int base = 0; base < nE; base += T) {
        const int l = base + t;
#if NXS_PF >= 1
        if (base >= 3 * T && l < nE) et = pet[l];  // very large patches: rounds beyond the prefetched ones
#else
        if (base > 0 && l < nE) et = pet[l];  // patches larger than the block: extra rounds
#endif
        const bool active = l < nE;
        const int eraw = et.x;
        const ushort4 tr = make_ushort4((unsigned short)(et.y & 1023), (unsigned short)((et.y >> 10) & 1023), (unsigned short)((et.y >> 20) & 1023), 0);
        const bool writer = eraw >= 0;
        const int e = writer ? eraw : ~eraw;
        double dxN[6], sig[3] = {0., 0., 0.}, damage = 0., c_expC = 0., c_pmax = 0., c_heal = 0., c_dxs = 1., c_coh = 0., volume = 0.;
        bool skip = true;
        int dxi = 0;
        if (active) {
            {
                typedef double d2 __attribute__((ext_vector_type(2)));
                const d2 *S = reinterpret_cast<const d2 *>(b.Sc) + 2 * (size_t)e;
                d2 a, c2;
                if (NT_S) { a = __builtin_nontemporal_load(S); c2 = __builtin_nontemporal_load(S + 1); } else { a = S[0]; c2 = S[1]; }
                sig[0] = a.x; sig[1] = a.y; sig[2] = c2.x; damage = c2.y;
            }
            {
                typedef double d2 __attribute__((ext_vector_type(2)));
                const d2 *r = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)e;
                d2 r0, r1, r2;
                if (NT_C) { r0 = __builtin_nontemporal_load(r); r1 = __builtin_nontemporal_load(r + 1); r2 = __builtin_nontemporal_load(r + 2); }
                else { r0 = r[0]; r1 = r[1]; r2 = r[2]; }
                c_expC = r0.x; volume = r0.y; c_pmax = r1.x; c_heal = r1.y; c_coh = r2.x;
                dxi = (int)(__double_as_longlong(r2.y) & 0xffffffffll);
                if (!bbm) skip = (__double_as_longlong(r2.y) >> 32) != 0;  // thick == 0 (FE.cpp:10656), from the record's high word
            }
        }
        if (base == 0) { __syncthreads(); NXS_STAMP(1); }  // staged velocities / coordinates visible
        if (active && bbm) {  // M_delta_x is an integer number of metres (Q1) and travels as one, with the skip flag in its sign
            skip = dxi < 0;
            c_dxs = (double)(skip ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // == w.dxs[e], FE.cpp:4232
        }
        if (active) {
            {   // shapeCoeff (FE.cpp:1951-1964) from the staged frozen coordinates: the same operations as
                // k_prep_elements, so the same bits as M_shape_coeff -- 48 B/element less to stream.  (Reading them from a 48-byte
                // record instead was tried for meshes that live in the caches: three more loads before barrier 1 cost more than the
                // six divisions -- 182 k triangles 1.30 -> 1.36 ms/step, and even the uniform branch around the two variants cost
                // 2-4 % at every size.  The several-sub-steps kernel, bound by VALU issue on its own CU, does read the records.)
                const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
                const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
                const double jac = jacobian(vx, vy);
                double num[6];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                    num[k] = vy[kp1] - vy[kp2];
                    num[k + 3] = vx[kp2] - vx[kp1];
                }
                quotients_by_one_divisor(num, jac, dxN, shape_in_range);
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
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 *S = reinterpret_cast<d2 *>(b.Sn) + 2 * (size_t)e;
                const d2 a = {sig[0], sig[1]}, c2 = {sig[2], damage};
                if (NT_S) { __builtin_nontemporal_store(a, S); __builtin_nontemporal_store(c2, S + 1); } else { S[0] = a; S[1] = c2; }
            }
            double F[6];
            corner_forces(volume, sig, dxN, F);
#pragma unroll
            for (int k = 0; k < 3; ++k) lF2[(size_t)k * Emax + l] = d2{F[k], F[k + 3]};
        }
#if NXS_PF >= 1
        et = et1; et1 = et2;
#endif
    }

    NXS_STAMP(2);
    const DevParams *pn_ = pdev;
    if (PMEM) asm volatile("" : "+s"(pn_));  // the node phase re-reads its parameters (see the top of the kernel)
    const DevParams &q = PMEM ? *pn_ : pval;
    if (HALO) {
        // The element loop above uses nothing of the exchange tables; laundering the pointer makes the compiler re-read the few
        // fields the node phase needs instead of keeping every field it loaded for the staging phase alive -- and spilled into
        // VGPR lanes -- across the hot loop (166 v_readlane/v_writelane in that loop before, 12 in the kernel without the exchange)
        asm volatile("" : "+s"(hfp));
        if (boundary) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_SEND_STORE);
    }
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
            {
                typedef double d2 __attribute__((ext_vector_type(2)));
                const d2 *r = reinterpret_cast<const d2 *>(w.nrec) + 5 * (size_t)n;
                const d2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
                node_mass = r0.x; gx = r0.y; gy = r1.x; rlm = r1.y; cbu = r2.x; fcor = r2.y; tax = r3.x; tay = r3.y; ou = r4.x; ov = r4.y;
            }
            lat = (nf & NF_LAT_NEG) ? -1. : 1.;
            if (q.dynamics_type == NXS_DYN_MEVP) { vtmu = w.VTM[n]; vtmv = w.VTM[n + Nn]; }
            if (move_dt != 0.) { umu = ldg<NT_U>(s.UM + n); umv = ldg<NT_U>(s.UM + n + Nn); utu = ldg<NT_U>(s.UT + n); utv = ldg<NT_U>(s.UT + n + Nn); }
            if (HALO && boundary) { sq0 = hfp->send_ptr[n]; sq1 = hfp->send_ptr[n + 1]; }
        }
        // the node's fan (element slot, corner) too: 8 entries cover all but the most irregular vertices
        unsigned fw[4];  // as LDS indices, two per word: pad entries and ghost corners (ghostNodes[i], FE.cpp:10456) name the pair of zeros
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned e0 = (active && 2 * k < pp.Wp) ? pf[(size_t)(2 * k) * pp.Pmax + i] : 0xFFFFu;
            const unsigned e1 = (active && 2 * k + 1 < pp.Wp) ? pf[(size_t)(2 * k + 1) * pp.Pmax + i] : 0xFFFFu;
            const unsigned i0 = (e0 == 0xFFFFu || (e0 & 4u)) ? ZIDX : (e0 & 3u) * (unsigned)Emax + (e0 >> 3);
            const unsigned i1 = (e1 == 0xFFFFu || (e1 & 4u)) ? ZIDX : (e1 & 3u) * (unsigned)Emax + (e1 >> 3);
            fw[k] = i0 | (i1 << 16);
        }
        if (base == 0) { __syncthreads(); NXS_STAMP(3); }  // corner forces visible
        if (!active) continue;
        double uice = lu[i], vice = lv[i];
        if (!((nf & NF_DIRICHLET) || node_mass == 0.)) {
            {   // eight independent 16-byte reads, then the reference's subtractions in the reference's order (x - (+0) == x bit for bit)
                d2 f[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = lF2[(fw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
#pragma unroll
                for (int k = 0; k < 8; ++k) { gx -= f[k].x; gy -= f[k].y; }
            }
            for (int k = 8; k < pp.Wp; ++k) {  // (fans of more than eight elements)
                const unsigned ent = pf[(size_t)k * pp.Pmax + i];
                if (ent == 0xFFFFu) break;
                if (ent & 4u) continue;  // ghostNodes[i] (FE.cpp:10456)
                const d2 f = lF2[(ent & 3u) * (unsigned)Emax + (ent >> 3)];
                gx -= f.x; gy -= f.y;
            }
            nodal_solve<true>(q, gx, gy, uice, vice, node_mass, rlm, cbu, fcor, lat, tax, tay, ou, ov, vtmu, vtmv);
        }
        b.VTn[n] = uice;
        b.VTn[n + Nn] = vice;
        if (HALO) {  // updateGhosts, sending side: straight into the neighbours' mailboxes
            for (int q = sq0; q < sq1; ++q) {
                const int k = hfp->send_k[q];
                double *dst = hfp->ipc.peer_seg[k] + (xseq & 1ull) * hfp->ipc.peer_parity_stride[k] + hfp->send_pos[q];
                sys_store(dst, uice);
                sys_store(dst + (hfp->send_off[k + 1] - hfp->send_off[k]), vice);
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
            if (t == 0 && atomicAdd(hfp->ipc.done_push, 1u) == (unsigned)n_boundary - 1u) {
                nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PUBLISH_FLAG);
                __threadfence_system();  // the one release of the launch
                for (int k = 0; k < hfp->ipc.ns; ++k)
                    __hip_atomic_store(hfp->ipc.peer_flag[k], xseq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // released by the fence above, once for all flags
                *hfp->ipc.done_push = 0u;
                *hfp->ipc.seq_push = xseq + 1ull;  // every boundary patch has read it; no counter over the whole grid (same-address atomics are served ~10 ns apart)
            }
        } else if (n_boundary == 0 && blockIdx.x == 0 && t == 0) {
            *hfp->ipc.seq_push = xseq + 1ull;
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
// The kernel takes ~170 VGPRs: a 512-thread workgroup has its CU to itself, which is the regime it is used in automatically (one
// patch per CU); capped at 128 (two workgroups per CU, or 1 024 threads) it spills 40 of them and loses more than it gains.
template <int T, bool POW4, int NTM>
__global__ void __launch_bounds__(T) k_substep_multi(DevMesh m, DevPatches2 pp, DevState s, DevWork w, const DevParams *__restrict__ pdev, PingPong b, VTOut vout) {
    const DevParams &p = *pdev;  // read from device memory where they are used (see k_substep_fused)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int NDm = pp.NDmax, EDm = pp.EDmax, ESm = pp.ESmax, D = pp.D;
    double *lu = lds, *lv = lu + NDm, *lx = lv + NDm, *ly = lx + NDm, *lF = ly + NDm /*(x, y) pairs [3][EDm] + a pair of zeros*/, *lS = lF + 6 * (size_t)EDm + 2 /*[4][ESm]*/;
    int blk;
    {   // consecutive patches are neighbours in space: keep them on one XCD (see k_substep_fused)
        const int n = (int)gridDim.x, pos = (int)blockIdx.x, q = n >> 3, r = n & 7, x = pos & 7;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3);
    }
    const int t = threadIdx.x, Nn = m.Nn;
    NXS_STAMP(0);
    const bool shape_in_range = w.shape_range[0] == 0;   // (uniform; see quotients_by_one_divisor)
    const int *ncnt = pp.ncnt + (size_t)blk * (D + 1), *ecnt = pp.ecnt + (size_t)blk * D;
    const int nO = ncnt[0], nD = ncnt[D];
    const int *pn = pp.pnodes + (size_t)blk * NDm;
    const int *pe = pp.pelem + (size_t)blk * EDm;
    const ushort4 *pt = reinterpret_cast<const ushort4 *>(pp.ptri) + (size_t)blk * EDm;
    const unsigned short *pf = pp.pfan + (size_t)blk * pp.Wp * pp.NSmax;
    const unsigned int *pf8 = pp.pfan8 + (size_t)blk * 4 * pp.NSmax;
    const bool bbm = p.dynamics_type == NXS_DYN_BBM;
    constexpr bool NT_S = NTM & 1, NT_C = NTM & 4;

    // (uniform) M_shape_coeff from its per-step records instead of six divisions per element and sub-step: this kernel has its CU to
    // itself and its element phase is bound by VALU issue -- 10 km: 0.739 -> 0.709 ms/step
    const bool shape_mem = w.srec != nullptr;
    // index rows are padded: the first loads depend on the launch arguments only
    const int my_node = (t < NDm) ? pn[t] : 0;
    int eraw0 = 0;
    ushort4 tr0 = make_ushort4(0, 0, 0, 0);
    if (t < EDm) { eraw0 = pe[t]; tr0 = pt[t]; }
    for (int i = t; i < nD; i += T) {
        const int g = (i == t) ? my_node : pn[i];
        lu[i] = b.VTc[g]; lv[i] = b.VTc[g + Nn];
        if (!shape_mem) { typedef double d2 __attribute__((ext_vector_type(2))); const d2 c = reinterpret_cast<const d2 *>(w.xy)[g]; lx[i] = c.x; ly[i] = c.y; }
    }

    // one element of one sub-step (FE.cpp:10425-10467), split into its global loads and the rest so that the barrier
    // between them sits in uniform control flow.  first sub-step of the launch: state from HBM; last: result to HBM (if this
    // patch writes the element); in between the state lives in LDS
    struct ElemIn { int e; bool writer, skip; int dxi; double sig[3], damage, expC, volume, pmax, heal, coh, dxN[6]; };
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 *lF2 = reinterpret_cast<d2 *>(lF);
    const unsigned ZIDX = 3u * (unsigned)EDm;
    if (threadIdx.x == 0) lF2[ZIDX] = d2{0., 0.};  // (read behind the barriers of the first sub-step)
    auto load_element = [&](const int eraw, const bool first, const bool last) {
        ElemIn in;
        in.writer = eraw >= 0;
        in.e = in.writer ? eraw : ~eraw;
        const int e = in.e;
        in.damage = 0.; in.sig[0] = in.sig[1] = in.sig[2] = 0.;
        if (first) {
            const d2 *S = reinterpret_cast<const d2 *>(b.Sc) + 2 * (size_t)e;
            d2 a, c2;
            if (NT_S) { a = __builtin_nontemporal_load(S); c2 = __builtin_nontemporal_load(S + 1); } else { a = S[0]; c2 = S[1]; }
            in.sig[0] = a.x; in.sig[1] = a.y; in.sig[2] = c2.x; in.damage = c2.y;
        }
        // the element constants: one 48-byte record (k_prep_elements), read by every sub-step of the launch -- the later reads hit
        // the L2, the last one is the streaming one
        const d2 *r = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)e;
        d2 r0, r1, r2;
        if (last && NT_C) { r0 = __builtin_nontemporal_load(r); r1 = __builtin_nontemporal_load(r + 1); r2 = __builtin_nontemporal_load(r + 2); }
        else { r0 = r[0]; r1 = r[1]; r2 = r[2]; }
        in.expC = r0.x; in.volume = r0.y; in.pmax = r1.x; in.heal = r1.y; in.coh = r2.x;
        const long long pk = __double_as_longlong(r2.y);
        in.dxi = (int)(pk & 0xffffffffll);
        in.skip = bbm ? true : (int)(pk >> 32) != 0;
        if (shape_mem) {
            const d2 *q = reinterpret_cast<const d2 *>(w.srec) + 3 * (size_t)e;
            const d2 q0 = q[0], q1 = q[1], q2 = q[2];
            in.dxN[0] = q0.x; in.dxN[1] = q0.y; in.dxN[2] = q1.x; in.dxN[3] = q1.y; in.dxN[4] = q2.x; in.dxN[5] = q2.y;
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
        if (shape_mem) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dxN[k] = in.dxN[k];
        } else {   // shapeCoeff (FE.cpp:1951-1964) from the staged frozen coordinates, as k_substep_fused
            const double vx[3] = {lx[tr.x], lx[tr.y], lx[tr.z]};
            const double vy[3] = {ly[tr.x], ly[tr.y], ly[tr.z]};
            const double jac = jacobian(vx, vy);
            double num[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                num[k] = vy[kp1] - vy[kp2];
                num[k + 3] = vx[kp2] - vx[kp1];
            }
            quotients_by_one_divisor(num, jac, dxN, shape_in_range);
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
            d2 *S = reinterpret_cast<d2 *>(b.Sn) + 2 * (size_t)in.e;
            const d2 a = {sig[0], sig[1]}, c2 = {sig[2], damage};
            if (NT_S) { __builtin_nontemporal_store(a, S); __builtin_nontemporal_store(c2, S + 1); } else { S[0] = a; S[1] = c2; }
        }
        double F[6];
        corner_forces(in.volume, sig, dxN, F);
#pragma unroll
        for (int k = 0; k < 3; ++k) lF2[(size_t)k * EDm + l] = d2{F[k], F[k + 3]};
    };
    // one node of one sub-step (FE.cpp:10472-10529), loads and solve apart for the same reason
    struct NodeIn { unsigned char nf; double node_mass, gx, gy, rlm, cbu, fcor, tax, tay, ou, ov; unsigned fw[4]; };  // fw: the first eight fan entries as LDS indices, two per word
    auto load_node = [&](const int i, const int n) {
        NodeIn in;
        in.nf = m.nflags[n];
        const d2 *r = reinterpret_cast<const d2 *>(w.nrec) + 5 * (size_t)n;  // one 80-byte record (k_prep_nodes)
        const d2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
        in.node_mass = r0.x; in.gx = r0.y; in.gy = r1.x; in.rlm = r1.y; in.cbu = r2.x; in.fcor = r2.y;
        in.tax = r3.x; in.tay = r3.y; in.ou = r4.x; in.ov = r4.y;
        // pad entries and ghost corners (ghostNodes[i], FE.cpp:10456) name the pair of zeros: ready-made LDS indices, two per word (DevPatches2::pfan8, as k_substep_pair)
#pragma unroll
        for (int k = 0; k < 4; ++k) in.fw[k] = pf8[(size_t)k * pp.NSmax + i];
        return in;
    };
    auto solve_node = [&](const int i, NodeIn &in, double &uice, double &vice) {
        uice = lu[i]; vice = lv[i];
        if ((in.nf & NF_DIRICHLET) || in.node_mass == 0.) return;
        double gx = in.gx, gy = in.gy;
        {   // eight independent 16-byte reads, then the reference's subtractions in the reference's order (x - (+0) == x bit for bit)
            d2 f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = lF2[(in.fw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
#pragma unroll
            for (int k = 0; k < 8; ++k) { gx -= f[k].x; gy -= f[k].y; }
        }
        for (int k = 8; k < pp.Wp; ++k) {  // (fans of more than eight elements)
            const unsigned ent = pf[(size_t)k * pp.NSmax + i];
            if (ent == 0xFFFFu) break;
            if (ent & 4u) continue;
            const d2 f = lF2[(ent & 3u) * (unsigned)EDm + (ent >> 3)];
            gx -= f.x; gy -= f.y;
        }
        nodal_solve<true>(p, gx, gy, uice, vice, in.node_mass, in.rlm, in.cbu, in.fcor, (in.nf & NF_LAT_NEG) ? -1. : 1., in.tax, in.tay, in.ou, in.ov, 0., 0.);
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
            if (base == 0) { __syncthreads(); if (k == 0) NXS_STAMP(1); }  // k == 0: staged velocities / coordinates; k > 0: the velocities of sub-step k on N_(D-k), forces consumed
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
            if (base == 0) { __syncthreads(); if (k == 0) NXS_STAMP(2); }  // corner forces of this sub-step visible
            if (active) {
                double u1, v1;
                solve_node(i, in, u1, v1);
                if (i < nO) { vt[n] = u1; vt[n + Nn] = v1; }
                lu[i] = u1; lv[i] = v1;  // a node's solve reads only its own staged velocity: in place
            }
        }
        if (k < 4) NXS_STAMP(3 + k);
    }
}

// ------------------------------------------------------------------------------------------------
// v3b  TWO sub-steps per launch on meshes that stream from HBM (single rank, deferred mesh move): k_substep_multi's temporal blocking at
// depth 2 in the shape k_substep_fused has -- 512 threads, 128 registers, two workgroups per CU -- so that patches of ~400 own nodes fit
// (k_substep_multi keeps the stresses between its sub-steps in LDS and takes 170 registers: one workgroup per CU, or patches of 200
// nodes whose two rings cost more arithmetic than the traffic they save: 2 km 8.4-10.6 ms of sub-steps against 6.3).  Here the stress and
// damage of an element stay in the REGISTERS of the thread that updates it in both sub-steps (the element list names E_1 first, so slot
// l is the same element in both), LDS holds the staged velocities / frozen coordinates of N_2 and the corner forces only.  Sub-step 0
// updates E_2 (every element touching N_1) and solves N_1, sub-step 1 updates E_1 and solves the own nodes: sigma / damage, the element
// constants and the nodal inputs cross HBM once per two sub-steps (the second reads hit the L2), at +11 % element and +24 % node
// arithmetic (400-node patches).  Same operations in the same order as k_substep_fused: bit-identical.
// Limits (the host checks them): E_2 <= 3 T, E_1 <= 2 T, N_1 <= 2 T, own nodes <= T.
// HALO (several ranks, device-direct mailboxes): the two updateGhosts of the launch happen inside it.  A rank's patches are cut over its OWN nodes; the first
// sub-step solves the own nodes of N_1 only -- a ghost node's fan is not complete on this rank -- and the ghosts of N_1 arrive from their owners instead:
//   G patches (a ghost among the staged nodes N_2; they lead the grid) first wait until the exchange that ended the launch before has landed
//     (flags >= x, x = *seq_push) and stage their ghosts from mailbox half (x-1)&1, copying them through to the velocity slot as k_substep_fused does;
//   after the first solve the SENDER patches store their sent own nodes into the neighbours' half x&1; the last of the G patches to have staged (and, if it
//     sends, stored) raises the flags to x+1 -- not before: the neighbours answer that flag by overwriting the half the G patches stage from;
//   RECEIVER patches (a ghost in N_1: the band along the partition boundary, and the patches of elements without an own node) wait for the neighbours'
//     flags >= x+1 and take the ghosts of N_1 from half x&1 (also into the first velocity slot: the ghosts' mesh move reads it);
//   after the second solve the senders store into half (x+1)&1; the last band patch to finish raises the flags to x+2 and advances *seq_push by two.
// Every other patch runs the single-rank body.  Same operations on the same values as one k_substep_fused<HALO> per sub-step: the same bits.
// Needs every G patch of the rank on a CU at once (the host checks the count and claims them in the device's registry); every wait is bounded.
struct PairHalo {
    const unsigned char *pflags;   // [nP] bit 0: G (stages a ghost), bit 1: sends, bit 2: a ghost in N_1 (receives between the two sub-steps)
    int nG, nBand;                 // patches [0, nBand) send or receive between the sub-steps, [0, nG) stage a ghost
    int from_mailbox;              // 0: first launch of a step, the ghosts are in the velocity buffer
    unsigned int *tickets;         // [0]: G patches past their first duty, [32]: band patches done, [64..65] (64 bits): the last sequence whose first exchange is published
};
// FLOW (k_substep_flow): the workgroups of ONE launch hand each other velocities and element state -- what a patch stores for the patches around it is written
// through (sc1) and read past the L1 (sc1), everything else as in the launch-per-pair kernel.
// MOVE (single rank, one launch per pair): the mesh move of both sub-steps (FE.cpp:10543-10550) is applied to the own nodes HERE -- M_UM / M_UT read and written
// once per launch, the same two additions per component in the same order as k_move_ring makes them -- instead of once per step from a ring of 120 velocity
// slots: no k_move_ring, no first velocity slot (nobody reads the first sub-step's velocity after the launch).
// KEEPN (the single-rank launch): the inputs of a thread's own node (25 registers) stay in registers from the first solve to the second -- no second read of the
// 97 bytes per own node; the several-rank and the data-flow builds, short of registers, read them again ahead of the second sub-step's last element round.
template <int T, bool POW4, int NTM, bool HALO, bool FLOW = false, bool MOVE = false, bool KEEPN = false>
__device__ __forceinline__ void pair_body(const DevMesh &m, const DevPatches2 &pp, const DevState &s, const DevWork &w, const DevParams &p, const PingPong &b, const VTOut &vout,
                                          const HaloFused *__restrict__ hfp, const PairHalo &ph, const int blk, const unsigned flg, const int t_in = 0) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    static_assert(!(HALO && FLOW), "the flow kernel is a single-rank kernel");
    static_assert(!(MOVE && (HALO || FLOW)), "the mesh move inside the launch: single rank, one launch per pair");
    // (FLOW: the two state buffers as buffer resources, for 16-byte sc1 accesses; unused otherwise)
    const __amdgpu_buffer_rsrc_t rSc = agent_rsrc(b.Sc, FLOW ? 32u * (unsigned)m.Ne : 0u), rSn = agent_rsrc(b.Sn, FLOW ? 32u * (unsigned)m.Ne : 0u);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int NDm = pp.NDmax, EDm = pp.EDmax;
    double *lu = lds, *lv = lu + NDm, *lx = lv + NDm, *ly = lx + NDm;
    d2 *lF2 = reinterpret_cast<d2 *>(ly + NDm);  // [3][EDm] + a pair of zeros
    const unsigned ZIDX = 3u * (unsigned)EDm;
    const int t = FLOW ? t_in : (int)threadIdx.x, Nn = m.Nn;   // (FLOW: the task loop hands the thread number over as a value the compiler cannot hoist address arithmetic out of the loop with)
    NXS_STAMP(0);
    if (t == 0) lF2[ZIDX] = d2{0., 0.};
    unsigned long long xseq = 0ull;
    if (HALO && (flg & 1u)) xseq = *hfp->ipc.seq_push;   // (interior patches never look at it: the last band patch advances it while they run)
    // the neighbours' flags: one lane waits, bounded like every other wait of the transport
    auto wait_flags = [&](const unsigned long long want, const unsigned point) {
        if (t == 0) {
            const long long t0 = wall_clock64();  // 100 MHz
            bool ok = true;
            unsigned polls = 0u;
            for (int k = 0; k < hfp->ipc.nr && ok; ++k)
                while (__hip_atomic_load(hfp->ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++polls & 31u) != 0u) continue;  // the error word is another round trip and the clock a few hundred cycles: every 32nd poll, so that a poll period is ONE round trip
                    if (__hip_atomic_load(hfp->ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }  // a wait already timed out: the run is lost
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(hfp->ipc.error, 3); break; }  // 10 s
                }
            nxs_delay_at(hfp->ipc.delay, point);
        }
        __syncthreads();   // (no acquire: the mailbox is uncached memory, read with system-scope loads -- see k_substep_fused)
    };
    // a G patch is past its first duty (staged; stored, if it sends): the last one tells the neighbours that exchange x is complete AND that half (x-1)&1 is free
    auto ticket_first = [&]() {
        if (t == 0 && atomicAdd(ph.tickets, 1u) == (unsigned)ph.nG - 1u) {
            nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PUBLISH_FLAG);
            __threadfence_system();
            const unsigned long long pub = ph.nBand > 0 ? xseq + 1ull : xseq + 2ull;   // (no band patch: nobody sends, both exchanges are empty)
            for (int k = 0; k < hfp->ipc.ns; ++k) __hip_atomic_store(hfp->ipc.peer_flag[k], pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *ph.tickets = 0u;
            if (ph.nBand == 0) *hfp->ipc.seq_push = xseq + 2ull;
            else __hip_atomic_store(reinterpret_cast<unsigned long long *>(ph.tickets + 64), xseq + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // "the first exchange of sequence x is out"
        }
    };
    auto send_node = [&](const int n, const double u1, const double v1, const unsigned long long x) {   // updateGhosts, sending side (FE.cpp:13967-13985)
        for (int q = hfp->send_ptr[n]; q < hfp->send_ptr[n + 1]; ++q) {
            const int k = hfp->send_k[q];
            double *dst = hfp->ipc.peer_seg[k] + (x & 1ull) * hfp->ipc.peer_parity_stride[k] + hfp->send_pos[q];
            sys_store(dst, u1);
            sys_store(dst + (hfp->send_off[k + 1] - hfp->send_off[k]), v1);
        }
    };
    auto ghost_from_mailbox = [&](const int g, const unsigned long long x, double &u1, double &v1) {
        const double *src = hfp->ipc.mailbox + (x & 1ull) * 2ull * (unsigned long long)hfp->ipc.tr + hfp->ghost_off[g - m.No];
        u1 = sys_load(src); v1 = sys_load(src + hfp->ghost_srl[g - m.No]);
    };
    const bool mailbox_ghosts = HALO && (flg & 1u) && ph.from_mailbox;
    if (mailbox_ghosts) wait_flags(xseq, NXS_DELAY_STAGE_READ);   // the exchange that ended the launch before (x - 1) has landed
    const int *ncnt = pp.ncnt + (size_t)blk * 3, *ecnt = pp.ecnt + (size_t)blk * 2;
    const int nO = ncnt[0], nN1 = ncnt[1], nN2 = ncnt[2], nE1 = ecnt[0], nE2 = ecnt[1];
    const int *pn = pp.pnodes + (size_t)blk * NDm;
    const int2 *pet = pp.pet + (size_t)blk * EDm;   // {element, corner slots in ten bits each}: 8 bytes per element (as k_substep_fused)
    const unsigned short *pf = pp.pfan + (size_t)blk * pp.Wp * pp.NSmax;
    const unsigned int *pf8 = pp.pfan8 + (size_t)blk * 4 * pp.NSmax;
    const bool bbm = p.dynamics_type == NXS_DYN_BBM;
    constexpr bool NT_S = NTM & 1;
    const bool shape_in_range = w.shape_range[0] == 0;   // (uniform; written by the prep kernels, before any launch of the loop)
    // index rows are padded: these loads depend on the launch arguments only
    const int my_node = (t < NDm) ? pn[t] : 0, my_node2 = (t + T < NDm) ? pn[t + T] : 0;
    int eraw[3];
    ushort4 tr[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        eraw[r] = 0; tr[r] = make_ushort4(0, 0, 0, 0);
        if (t + r * T < EDm) {
            const int2 et = pet[t + r * T];
            eraw[r] = et.x;
            tr[r] = make_ushort4((unsigned short)(et.y & 1023), (unsigned short)((et.y >> 10) & 1023), (unsigned short)((et.y >> 20) & 1023), 0);
        }
    }
    auto stage = [&](const int i, const int g) {
        if (mailbox_ghosts && g >= m.No) {
            double u0, v0;
            ghost_from_mailbox(g, xseq - 1ull, u0, v0);
            lu[i] = u0; lv[i] = v0;
            const_cast<double *>(b.VTc)[g] = u0;  // every patch that stages g writes the same two values (the ghosts' mesh move and the end of the step read them here)
            const_cast<double *>(b.VTc)[g + Nn] = v0;
        } else if (FLOW) { lu[i] = ld_agent(b.VTc + g); lv[i] = ld_agent(b.VTc + g + Nn); }
        else { lu[i] = b.VTc[g]; lv[i] = b.VTc[g + Nn]; }
        const d2 c = reinterpret_cast<const d2 *>(w.xy)[g];
        lx[i] = c.x; ly[i] = c.y;
    };
    if (t < nN2) stage(t, my_node);
    if (t + T < nN2) stage(t + T, my_node2);
    for (int i = t + 2 * T; i < nN2; i += T) stage(i, pn[i]);

    double ks[2][4];  // sigma0, sigma1, sigma2, damage of this thread's E_1 elements between the two sub-steps
#pragma unroll
    for (int r = 0; r < 2; ++r) ks[r][0] = ks[r][1] = ks[r][2] = ks[r][3] = 0.;

    // one element update (FE.cpp:4137-4260 / 10649-10726 + the element half of 10445-10467) from the staged velocities
    auto update_element = [&](const int l, const ushort4 trl, double sig[3], double &damage, const d2 r0, const d2 r1, const d2 r2) {
        const double c_expC = r0.x, volume = r0.y, c_pmax = r1.x, c_heal = r1.y, c_coh = r2.x;
        const int dxi = (int)(__double_as_longlong(r2.y) & 0xffffffffll);
        bool skip = bbm ? dxi < 0 : (__double_as_longlong(r2.y) >> 32) != 0;
        double c_dxs = 1.;
        if (bbm) c_dxs = (double)(skip ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // FE.cpp:4232
        double dxN[6];
        {   // shapeCoeff (FE.cpp:1951-1964) from the staged frozen coordinates, as k_substep_fused
            const double vx[3] = {lx[trl.x], lx[trl.y], lx[trl.z]};
            const double vy[3] = {ly[trl.x], ly[trl.y], ly[trl.z]};
            const double jac = jacobian(vx, vy);
            double num[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                num[k] = vy[kp1] - vy[kp2];
                num[k + 3] = vx[kp2] - vx[kp1];
            }
            quotients_by_one_divisor(num, jac, dxN, shape_in_range);
        }
        if (skip) {
            sig[0] = sig[1] = sig[2] = 0.;
            damage = 0.;
        } else {
            const double u[3] = {lu[trl.x], lu[trl.y], lu[trl.z]};
            const double v[3] = {lv[trl.x], lv[trl.y], lv[trl.z]};
            if (bbm) bbm_stress<POW4>(p, dxN, u, v, sig, damage, c_expC, c_pmax, c_heal, c_dxs, c_coh);
            else vp_stress(p, dxN, u, v, sig, c_expC);
        }
        double F[6];
        corner_forces(volume, sig, dxN, F);
#pragma unroll
        for (int k = 0; k < 3; ++k) lF2[(size_t)k * EDm + l] = d2{F[k], F[k + 3]};
    };
    // one node (FE.cpp:10472-10529): its loads, then (behind the barrier) the fan gather in ascending element order and the 2x2 solve
    struct NodeIn { unsigned char nf; d2 r[5]; unsigned fw[4]; };
    auto load_node = [&](const int i, const int n) {
        NodeIn in;
        in.nf = m.nflags[n];
        const d2 *q = reinterpret_cast<const d2 *>(w.nrec) + 5 * (size_t)n;
#pragma unroll
        for (int k = 0; k < 5; ++k) in.r[k] = q[k];
        // the first eight fan entries as LDS indices, two per word (pad entries and ghost corners -- ghostNodes[i], FE.cpp:10456 -- name the pair of zeros): read ready-made
        // (round 4: decoding them here cost ~45 integer instructions per solve)
#pragma unroll
        for (int k = 0; k < 4; ++k) in.fw[k] = pf8[(size_t)k * pp.NSmax + i];
        return in;
    };
    auto solve_node = [&](const int i, const NodeIn &in, double &uice, double &vice) {
        uice = lu[i]; vice = lv[i];
        const double node_mass = in.r[0].x;
        if ((in.nf & NF_DIRICHLET) || node_mass == 0.) return;
        double gx = in.r[0].y, gy = in.r[1].x;
        // (eight 16-byte reads in two halves of four, then the reference's subtractions in the reference's order, x - (+0) == x bit for bit: half the registers of one
        // gather of eight -- they are what lets the elements' constants stay in registers across this phase, see the first sub-step)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            d2 f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = lF2[(in.fw[(4 * h2 + k) >> 1] >> (16 * (k & 1))) & 0xFFFFu];
#pragma unroll
            for (int k = 0; k < 4; ++k) { gx -= f[k].x; gy -= f[k].y; }
        }
        for (int k = 8; k < pp.Wp; ++k) {  // (fans of more than eight elements)
            const unsigned ent = pf[(size_t)k * pp.NSmax + i];
            if (ent == 0xFFFFu) break;
            if (ent & 4u) continue;
            const d2 f = lF2[(ent & 3u) * (unsigned)EDm + (ent >> 3)];
            gx -= f.x; gy -= f.y;
        }
        nodal_solve<true>(p, gx, gy, uice, vice, node_mass, in.r[1].y, in.r[2].x, in.r[2].y, (in.nf & NF_LAT_NEG) ? -1. : 1., in.r[3].x, in.r[3].y, in.r[4].x, in.r[4].y, 0., 0.);
    };

    // ---- sub-step 0: elements E_2 (three rounds of the block), state from HBM
    // Round 4, the second reads of the 48-byte element constants (sub-step 1 needs those of the E_1 elements again) are gone: a line survives ~6-8 us in an XCD's 4 MB
    // L2 at this kernel's rate, the second read used to come ~12 us after the first and missed.  The rounds now run from the LAST to the first, so the constants of
    // round 0 are the last ones loaded and simply STAY in their registers (kc0); those of round 1 are read again right behind this loop -- 3-6 us after their first
    // read: an L2 hit -- and held across the node phase (kc1).  24 registers across the node phase fit since its fan gather runs in two halves of four (122 VGPRs, no
    // scratch); held from their first read they would cross this loop's last round, the kernel's register peak, and spill.  2 km: 5.29 -> 5.05 ms of sub-steps with
    // the early loads of sub-step 1 below, counted traffic 443 -> ~400 MB per launch; same operands, same bits.  What loses: reversing the node rounds as well,
    // asking for a round's state and constants one round ahead (fits, 126 VGPRs, but delays the re-read: 5.27 against 5.07 ms), touching lines to renew them
    // (gpurun_out/r4_ab5.log, r4_ab6.log, r4_ab10.log).
    d2 kc0[3] = {d2{0., 0.}, d2{0., 0.}, d2{0., 0.}}, kc1[3] = {d2{0., 0.}, d2{0., 0.}, d2{0., 0.}};
    NodeIn nin{};   // the inputs of this thread's own node, kept from the first solve for the second (the own nodes lead N_1: thread t solves node t in both)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int r = 2 - rr;
        const int l = t + r * T;
        const bool active = l < nE2;
        const int e = eraw[r] >= 0 ? eraw[r] : ~eraw[r];
        d2 a = d2{0., 0.}, c2 = d2{0., 0.}, r0 = d2{0., 0.}, r1 = d2{0., 0.}, r2 = d2{0., 0.};
        if (active) {
            const d2 *S = reinterpret_cast<const d2 *>(b.Sc) + 2 * (size_t)e;
            if (FLOW) { a = ld_agent16(rSc, 32u * (unsigned)e); c2 = ld_agent16(rSc, 32u * (unsigned)e + 16u); }
            else if (NT_S) { a = __builtin_nontemporal_load(S); c2 = __builtin_nontemporal_load(S + 1); } else { a = S[0]; c2 = S[1]; }
            const d2 *q = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)e;
            r0 = q[0]; r1 = q[1]; r2 = q[2];
        }
        if (rr == 0) {
            __syncthreads(); NXS_STAMP(1);  // staged velocities / coordinates visible
            if (HALO && (flg & 1u) && !(flg & 2u)) ticket_first();   // (a G patch that sends nothing has read its mailbox half: that is all the neighbours wait for from it)
        }
        if (active) {
            double sig[3] = {a.x, a.y, c2.x}, damage = c2.y;
            update_element(l, tr[r], sig, damage, r0, r1, r2);
            if (r < 2) { ks[r][0] = sig[0]; ks[r][1] = sig[1]; ks[r][2] = sig[2]; ks[r][3] = damage; }
        }
        if (r == 0) { kc0[0] = r0; kc0[1] = r1; kc0[2] = r2; }
    }
    if (t + T < nE1) {
        const d2 *q = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)(eraw[1] >= 0 ? eraw[1] : ~eraw[1]);
        kc1[0] = q[0]; kc1[1] = q[1]; kc1[2] = q[2];
    }
    NXS_STAMP(5);
    if (HALO && (flg & 2u)) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_SEND_STORE);
    // ---- sub-step 0: nodes N_1 (two rounds)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = rr;
        const int i = t + r * T;
        const int n = r == 0 ? my_node : my_node2;
        const bool active = i < nN1 && !(HALO && n >= m.No);   // (several ranks: a ghost of N_1 is not solved here, it arrives below)
        NodeIn in{};
        if (active) in = load_node(i, n);
        if (KEEPN && rr == 0) nin = in;   // (25 registers across the second sub-step's elements: they fit since the strain and stress sums lost their literal zeros, bbm_stress)
        if (rr == 0) { __syncthreads(); NXS_STAMP(6); }  // corner forces of sub-step 0 visible
        if (active) {
            double u1, v1;
            solve_node(i, in, u1, v1);
            if (i < nO && !MOVE) {   // (MOVE: the first velocity stays in LDS, where the second solve and the move below find it)
                if (FLOW) { st_agent(vout.slot[0] + n, u1); st_agent(vout.slot[0] + n + Nn, v1); }
                else { vout.slot[0][n] = u1; vout.slot[0][n + Nn] = v1; }
                if (HALO && (flg & 2u)) send_node(n, u1, v1, xseq);
            }
            lu[i] = u1; lv[i] = v1;  // (a node's solve reads only its own staged velocity: in place)
        }
    }
    if (HALO && (flg & 6u)) {   // ---- the exchange between the two sub-steps (band patches only)
        if (flg & 2u) {         // every wave drains its stores into the neighbours' mailboxes, the barrier collects the waves, one lane takes the patch's ticket
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ticket_first();
        }
        // EVERY band patch waits for the neighbours' flags >= x + 1 before it goes on: a receiver because its ghosts arrive with them -- and a patch that only sends
        // (its sent nodes touch no ghost of this rank: they are ghosts of a neighbour through an element none of whose nodes the neighbour owns) because its second
        // store goes into the half (x+1)&1 = (x-1)&1 of the neighbour's mailbox, which the neighbour's G patches stage from (and its k_halo_pull at the end of a
        // step reads) until that flag says they are done with it
        wait_flags(xseq + 1ull, NXS_DELAY_PAIR_MID_READ);
        if (flg & 4u) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = t + r * T;
                const int g = r == 0 ? my_node : my_node2;
                if (i >= nO && i < nN1 && g >= m.No) {
                    double u1, v1;
                    ghost_from_mailbox(g, xseq, u1, v1);
                    lu[i] = u1; lv[i] = v1;
                    vout.slot[0][g] = u1; vout.slot[0][g + Nn] = v1;   // (every patch that holds g in N_1 writes the same two values)
                }
            }
        }
    }
    // ---- sub-step 1: elements E_1 (two rounds), state AND constants from the registers, result to HBM (by the element's writer); the own nodes' inputs are asked
    // for in front of the second round's arithmetic (they were waited for 2.2 us in front of the last barrier: 1.1 now)
    __syncthreads();  // the velocities of sub-step 0 on N_1; the corner forces have been consumed
    NXS_STAMP(2);
    double mv[4] = {0., 0., 0., 0.};   // MOVE: M_UM, M_UT of this thread's own node
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int l = t + r * T;
        if (!KEEPN && r == 1 && t < nO) nin = load_node(t, my_node);
        if (l >= nE1) continue;
        const bool writer = eraw[r] >= 0;
        const int e = writer ? eraw[r] : ~eraw[r];
        double sig[3] = {ks[r][0], ks[r][1], ks[r][2]}, damage = ks[r][3];
        update_element(l, tr[r], sig, damage, r == 0 ? kc0[0] : kc1[0], r == 0 ? kc0[1] : kc1[1], r == 0 ? kc0[2] : kc1[2]);
        if (writer) {
            d2 *S = reinterpret_cast<d2 *>(b.Sn) + 2 * (size_t)e;
            const d2 a = {sig[0], sig[1]}, c2 = {sig[2], damage};
            if (FLOW) { st_agent16(rSn, 32u * (unsigned)e, a); st_agent16(rSn, 32u * (unsigned)e + 16u, c2); }
            else if (NT_S) { __builtin_nontemporal_store(a, S); __builtin_nontemporal_store(c2, S + 1); } else { S[0] = a; S[1] = c2; }
        }
    }
    NXS_STAMP(7);
    // ---- sub-step 1: the own nodes
    {
        const bool active = t < nO;
        // (MOVE: asked for behind the elements' arithmetic -- held across it they spill)
        if (MOVE && active) { mv[0] = s.UM[my_node]; mv[1] = s.UM[my_node + Nn]; mv[2] = s.UT[my_node]; mv[3] = s.UT[my_node + Nn]; }
        __syncthreads();  // corner forces of sub-step 1 visible
        NXS_STAMP(3);
        if (HALO && (flg & 2u)) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PAIR_SECOND_STORE);
        if (active) {
            double u1, v1;
            const double u0 = lu[t], v0 = lv[t];   // (MOVE) the first sub-step's velocity of this node
            solve_node(t, nin, u1, v1);
            if (FLOW) { st_agent(vout.slot[1] + my_node, u1); st_agent(vout.slot[1] + my_node + Nn, v1); }
            else { vout.slot[1][my_node] = u1; vout.slot[1][my_node + Nn] = v1; }
            if (MOVE && vout.slot[0]) { vout.slot[0][my_node] = u1; vout.slot[0][my_node + Nn] = v1; }   // (the last launch of a step: the smoother's second buffer)
            if (MOVE) {   // FE.cpp:10543-10550, twice: Neumann nodes keep M_UM (k_move_ring)
                const double dt = p.dte;
                if (!(nin.nf & NF_NEUMANN)) {
                    mv[0] += dt * u0; mv[1] += dt * v0;
                    mv[0] += dt * u1; mv[1] += dt * v1;
                    s.UM[my_node] = mv[0]; s.UM[my_node + Nn] = mv[1];
                }
                mv[2] += dt * u0; mv[3] += dt * v0;
                mv[2] += dt * u1; mv[3] += dt * v1;
                s.UT[my_node] = mv[2]; s.UT[my_node + Nn] = mv[3];
            }
            if (HALO && (flg & 2u)) send_node(my_node, u1, v1, xseq + 1ull);
        }
    }
    if (HALO) {
        if (flg & 6u) {   // publish the second exchange: the last band patch to finish raises the flags and advances the sequence
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0 && atomicAdd(ph.tickets + 32, 1u) == (unsigned)ph.nBand - 1u) {
                // the flags count upwards: x + 1 (raised by the last G patch to have staged, which may be a patch that neither sends nor receives and started late) goes out
                // before x + 2; and *seq_push may only move once every G patch has read it, which is what that first publication certifies
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(reinterpret_cast<unsigned long long *>(ph.tickets + 64), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < xseq + 1ull) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > 1000000000ll) { atomicExch(hfp->ipc.error, 3); break; }  // 10 s: a G patch of this rank never ran
                }
                nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PAIR_SECOND_FLAG);
                __threadfence_system();
                for (int k = 0; k < hfp->ipc.ns; ++k) __hip_atomic_store(hfp->ipc.peer_flag[k], xseq + 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ph.tickets[32] = 0u;
                *hfp->ipc.seq_push = xseq + 2ull;
            }
        }
    }
    NXS_STAMP(4);
}


// the kernel: which patch, and -- several ranks -- whether it takes part in the exchange: the patches that do not (all but the few along the partition boundary)
// run the single-rank body, so the exchange's tables and tickets cost them no register (the resident kernels branch the same way)
template <int T, bool POW4, int NTM, bool HALO = false, bool MOVE = false>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) k_substep_pair(DevMesh m, DevPatches2 pp, DevState s, DevWork w, DevParams p, PingPong b, VTOut vout,
                                                                                                const HaloFused *__restrict__ hfp, PairHalo ph) {
    int blk;
    {   // consecutive patches are neighbours in space: keep them on one XCD (see k_substep_fused); several ranks: the G patches lead the grid, the remap acts inside each group
        auto xcd_remap = [](const int pos, const int n) { const int q = n >> 3, r = n & 7, x = pos & 7; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (pos >> 3); };
        const int pos = (int)blockIdx.x, n = (int)gridDim.x;
        if (HALO) blk = pos < ph.nG ? xcd_remap(pos, ph.nG) : ph.nG + xcd_remap(pos - ph.nG, n - ph.nG);
        else blk = xcd_remap(pos, n);
    }
    if (HALO) {
        const unsigned flg = ph.pflags[blk];   // this patch's duties in the exchange (uniform over the workgroup)
        if (flg & 1u) { pair_body<T, POW4, NTM, true>(m, pp, s, w, p, b, vout, hfp, ph, blk, flg); return; }
        // a rank whose mesh holds no ghost (no G patch: nobody runs the exchange's body, nobody reads the sequence): the sequence still counts the two exchanges
        if (ph.nG == 0 && blockIdx.x == 0 && threadIdx.x == 0) *hfp->ipc.seq_push += 2ull;
    }
    pair_body<T, POW4, NTM, false, false, MOVE, !HALO>(m, pp, s, w, p, b, vout, hfp, ph, blk, 0u);
}

// ------------------------------------------------------------------------------------------------
// v3c  k_substep_pair as a DATA-FLOW launch (single rank, round 4): ONE launch per step whose workgroups take (pair of sub-steps, patch) tasks from
// queues -- a task starts as soon as the patches it reads from have finished the pair of sub-steps before (per-patch counters), not when a whole launch has.
// Why: a launch of k_substep_pair is 3.33 rounds of the 512 workgroups the device holds at 2 km, so the last third of every launch runs on a part-empty
// device (16 % of its slot-time), and 60 launches a step start and drain 60 times.  Here the device stays full from the first task of the step to the last.
//   tasks      queue x (x = blockIdx & 7: the workgroups the dispatcher deals to one XCD) owns a contiguous range of the patches -- neighbours in space -- and
//              hands out tickets in order: ticket -> (pair-step k, patch) with k = ticket / patches of the queue.
//   waits      task (k, p) waits until done[q] >= k for every q in dep(p): the writers of the elements p reads, the owners of the nodes it stages, the same
//              relation the other way round (p overwrites the state buffer those patches read one pair-step earlier) and p itself (nxs_cut::build_flow_deps).
//   no deadlock, whatever number of workgroups is resident: tickets leave a queue in order, so every task another one waits for is already in the hands of a
//              running workgroup (or finished); a workgroup publishes a finished task BEFORE it waits for its next one.  Every wait is bounded all the same.
//   visibility what patches hand each other (velocities, sigma / damage) is stored write-through and loaded past the L1 (pair_body<FLOW>); a patch's counter
//              is raised by one lane behind every wave's s_waitcnt vmcnt(0) and the workgroup's barrier; the waiting lanes poll it with sc1 loads and the
//              workgroup's barrier stands between the poll and the first load (MI355X_MICROARCH.md, inter-workgroup visibility: the first row of the table).
// Same operations on the same values as k_substep_pair: the same bits (tests/test_gpu_parity.py).
// MEASURED (2 km, gpurun_out/r4_flow1.log, r4_flow_ab1.log, r4_flow2.log): it LOSES -- 7.5-7.8 ms of sub-steps against 5.03-5.09 with one launch per pair; a task takes
// 37 us where a workgroup of the launch takes 22.7.  With plain loads / stores and without the waits (wrong results, timing only) still 5.97: the loop itself costs 7 us a
// task (drain + two barriers + ticket between tasks where the hardware's dispatcher overlaps the end of one workgroup with the start of the next; the parameters read
// from memory; 128 VGPRs, 3 spilled), the write-through stores and L1-bypassing loads another 7.5 (1.5 ms a step).  The 16 % of a launch's slot-time the part-empty
// last round leaves idle do not pay for either.  Kept behind option "pair_flow" = 1 (default: one launch per pair of sub-steps); DESIGN section 4.1.
struct PairFlow {
    unsigned int *queue;      // [8][32] the queues' ticket counters, one per 128-byte line (zeroed before the launch, with done)
    unsigned int *done;       // [nP][16] pair-steps a patch has finished, one per 64-byte line
    const int *dep_ptr;       // [nP + 1]
    const int *dep;           // the patches a patch waits for (itself among them)
    int qstart[9];            // the queues' patch ranges
    int K;                    // pair-steps of the launch (sub-steps / 2)
    int *error;               // != 0: a wait timed out, the step is lost (8)
    double *S[2];             // the element state's two buffers: pair-step k reads S[k & 1] and writes S[(k + 1) & 1]
};
// The rheology parameters (DevParams: ~120 scalar registers' worth) are read from device memory INSIDE the task loop, through a pointer the compiler cannot see
// through: as a kernel argument their scalar loads were hoisted in front of the loop and kept alive across it (93 spilled SGPRs, 36 spilled VGPRs, 84 B of
// scratch per lane).  (Everything that holds POINTERS stays a kernel argument: read from memory they would be generic pointers and every load a flat load.)
template <int T, bool POW4, int NTM>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) k_substep_flow(DevMesh m, DevPatches2 pp, DevState s, DevWork w, const DevParams *__restrict__ pdev, VTRing ring, PairFlow f) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double lds[];
    unsigned int *ctl = reinterpret_cast<unsigned int *>(reinterpret_cast<d2 *>(lds + 4 * (size_t)pp.NDmax) + 3 * (size_t)pp.EDmax + 1);  // behind the pair of zeros: {next ticket, stop}
    const int t = threadIdx.x;
    const int qx = (int)(blockIdx.x & 7u);
    const int q0 = f.qstart[qx], nq = f.qstart[qx + 1] - q0;
    if (nq <= 0) return;
    unsigned int *qc = f.queue + 32 * qx;
    if (t == 0) { ctl[0] = atomicAdd(qc, 1u); ctl[1] = 0u; }
    __syncthreads();
    unsigned int tk = ctl[0];
    const int R = ring.R;
    for (;;) {
        const int k = (int)(tk / (unsigned)nq);
        if (k >= f.K) break;
        const int blk = q0 + (int)(tk - (unsigned)k * (unsigned)nq);
        const DevParams *pp_ = pdev;
        asm volatile("" : "+s"(pp_));
        if (t < 64) {   // one wave waits for the patches this task reads from (a lane each), bounded like every other wait of the library
            const int d1 = f.dep_ptr[blk + 1];
            const long long t0 = wall_clock64();  // 100 MHz
            bool ok = true;
            for (int j = f.dep_ptr[blk] + t; j < d1 && ok; j += 64) {
                const unsigned int *flag = f.done + 16 * (size_t)f.dep[j];
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k) {
                    __builtin_amdgcn_s_sleep(2);
                    if (__hip_atomic_load(f.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }  // a wait already timed out: the run is lost
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(f.error, 8); break; }  // 10 s
                }
            }
            if (!ok) ctl[1] = 1u;
        }
        unsigned int nxt = 0u;
        if (t == 0) nxt = atomicAdd(qc, 1u);   // the next ticket: its answer is not needed before this task ends
        __syncthreads();   // the poll stands before every load of this task; the LDS of the task before is free
        if (ctl[1] != 0u) break;
        PingPong b;
        b.Sc = f.S[k & 1]; b.Sn = f.S[(k + 1) & 1]; b.VTc = ring.slot[(2 * k) % R]; b.VTn = nullptr;
        VTOut vo{};
        vo.slot[0] = ring.slot[(2 * k + 1) % R]; vo.slot[1] = ring.slot[(2 * k + 2) % R];
        int tt = t;
        asm volatile("" : "+v"(tt));   // (address arithmetic on the thread number stays inside the task: hoisted out of the loop it costs the registers the task needs)
        pair_body<T, POW4, NTM, false, true>(m, pp, s, w, *pp_, b, vo, nullptr, PairHalo{}, blk, 0u, tt);
        if (t == 0) ctl[0] = nxt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave drains its write-through stores ...
        __syncthreads();                                    // ... the barrier collects the waves (and the last LDS reads of this patch) ...
        if (t == 0) __hip_atomic_store(f.done + 16 * (size_t)blk, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... one lane says so
        tk = ctl[0];
    }
}

// ------------------------------------------------------------------------------------------------
// v4  The whole sub-step loop in ONE launch: a patch stays on its CU for all the sub-steps and waits for its NEIGHBOURING
// patches only (FE.cpp:10425-10553 as k_substep_fused; same operations in the same order, same bits).
// Where one round of resident workgroups covers the partition the sub-step of k_substep_fused is a latency chain -- launch,
// index hop, staging of velocities / element constants / nodal inputs, two barriers, tail: 10.2 us at 182 k triangles, of which
// ~3 are arithmetic.  Here a workgroup loads all that ONCE per step (stress, damage and element constants in registers, shape
// coefficients, nodal inputs and the running M_UM / M_UT of its own nodes in LDS) and per sub-step only exchanges velocities:
//   node phase -> every own node's new velocity goes to the exchange buffer X[s & 1] with write-through (sc1) stores, every wave
//   drains them, barrier, one lane raises the patch's counter to s + 1;
//   before the next element phase a few lanes wait until the counters of the patches that own this patch's halo nodes have
//   reached s + 1, then the halo velocities are read from X[s & 1] with cache-bypassing (sc1) loads into LDS.
// Two exchange buffers suffice: a patch overwrites X[s & 1] (sub-step s + 2) only after all its neighbours have raised their
// counters to s + 2, i.e. after they have read X[s & 1] (they did that before their sub-step s + 1).  scripts/micro/nbrsync.hip:
// that exchange costs 3.1 us per sub-step with 511 workgroups; a grid-wide barrier costs 6-9 (MI355X_MICROARCH.md, barrier-xcd).
// REQUIRES every workgroup of the grid to be resident at once (the host checks the occupancy and falls back otherwise; every
// wait is bounded and reports through r.error), one element per thread (Emax <= T), not mEVP.
#ifndef NXS_RES_WAVES
#define NXS_RES_WAVES 4   // waves per SIMD the resident kernel is compiled for: 4 = 128 VGPRs = two 512-thread workgroups per CU
#endif
#define NXS_RES_NBR 24
struct DevResident {
    // variant OVL only (option resident_overlap): the patch's elements with the INTERIOR ones first (no corner is a halo node of the patch)
    const int *pelem;             // [nP][Emax] as DevPatches::pelem in that order
    const unsigned short *ptri;   // [nP][Emax][4]
    const unsigned short *pfan;   // [nP][Wp][Pmax] DevPatches::pfan naming the new slots (entry order unchanged: ascending global element)
    const int *ecut;              // [nP] the first ecut elements (a multiple of 64, all interior) are updated for sub-step s + 1 while the exchange of sub-step s is awaited
    const int *pnbr;       // [nP][NXS_RES_NBR] the patches that own this patch's halo nodes
    const int *pnbr_cnt;   // [nP]
    unsigned int *flag;    // [nP * 32] sub-steps published by each patch (one counter per 128-byte line; zeroed before every launch)
    double *X0, *X1;       // [2 Nn] exchange buffers: velocities after an even / an odd sub-step
    int *error;            // != 0: a wait timed out (the workgroups were not all resident): the step is lost, nobody waits again
    // several ranks (the device-direct mailboxes of HaloFused inside the resident loop):
    unsigned int *cnt;     // [NXS_RES_MAXS] boundary patches that have finished each sub-step (zeroed before every launch)
    unsigned int *raised;  // [1] sub-steps whose exchange has been published to the neighbour ranks (keeps the flags monotone)
    double *gring;         // [S - 1][2 NG] the ghost nodes' velocities as they arrived after every sub-step but the last ([u-block | v-block] of the NG = Nn - No
    int NG;                //   ghosts): k_ghost_ring_move applies their mesh moves after the launch -- nothing of that sits in the sub-step loop
    int rank;              // (the microscope of scripts/phase_timing.py stamps rank 0 only: several handles of a process share its table)
};
#define NXS_RES_MAXS 512
#define NXS_RES_MAXNB 16  // neighbour ranks whose mailbox addresses the several-rank variant keeps in LDS


// element phase of the resident kernel for this thread's element (FE.cpp:4137-4260 / 10649-10726 + the element half of 10445-10467);
// the OVL variant calls it from three places, the plain variant keeps its own copy in line
template <bool POW4>
__device__ __forceinline__ void resident_element(const DevParams &p, const double *__restrict__ erec, const bool bbm, const bool skip, const int e,
                                                 const ushort4 tr, const int tt, const int Emax, const double *lu, const double *lv,
                                                 const double *ldx, double *lF, double sig[3], double &damage) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    double c_expC, volume, c_pmax, c_heal, c_coh, c_dxs = 1.;
    {
        const d2 *q = reinterpret_cast<const d2 *>(erec) + 3 * (size_t)e;
        const d2 r0 = q[0], r1 = q[1], r2 = q[2];
        c_expC = r0.x; volume = r0.y; c_pmax = r1.x; c_heal = r1.y; c_coh = r2.x;
        const int dxi = (int)(__double_as_longlong(r2.y) & 0xffffffffll);
        if (bbm) c_dxs = (double)(dxi < 0 ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // FE.cpp:4232
    }
    double dxN[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dxN[k] = ldx[(size_t)k * Emax + tt];
    if (skip) {
        sig[0] = sig[1] = sig[2] = 0.;
        damage = 0.;
    } else {
        const double u[3] = {lu[tr.x], lu[tr.y], lu[tr.z]};
        const double v[3] = {lv[tr.x], lv[tr.y], lv[tr.z]};
        if (bbm) bbm_stress<POW4>(p, dxN, u, v, sig, damage, c_expC, c_pmax, c_heal, c_dxs, c_coh);
        else vp_stress(p, dxN, u, v, sig, c_expC);
    }
    double F[6];
    corner_forces(volume, sig, dxN, F);
    d2 *lF2 = reinterpret_cast<d2 *>(lF);
#pragma unroll
    for (int k = 0; k < 3; ++k) lF2[(size_t)k * Emax + tt] = d2{F[k], F[k + 3]};
}

// HALO: several ranks.  Boundary patches (they lead the grid, see HaloFused) also send their sent nodes into the neighbour ranks'
// mailboxes in the node phase (exchange x0 + s, half (x0 + s) & 1), read their ghost nodes from this rank's mailbox before the next
// element phase -- after the neighbours' flags have reached x0 + s + 1 -- and move the ghost nodes assigned to them; the last
// boundary patch to finish a sub-step raises this rank's flag at the neighbours, in sub-step order.  The exchange of the LAST
// sub-step is taken by k_halo_pull after the launch (with the last mesh move of the ghosts), as in the one-launch-per-sub-step path.
// OVL (option resident_overlap, with HALO): the interior elements of a patch -- no corner is a halo node -- run one exchange ahead: their
// update for sub-step s + 1 is computed while the exchange of sub-step s is awaited, only the rim elements follow the halo loads.  Inside one
// GPU that buys nothing (the other workgroup of the CU fills the wait); between GPUs the wait is a round trip over xGMI.  Same
// operations on the same values in the same order: bit-identical (bench.py keeps whichever variant is faster on the machine it runs on).
// WPE: waves per SIMD the kernel is compiled for -- 4 (128 VGPRs: two 512-thread workgroups per CU) or, where one workgroup per CU covers the
// partition, 2 (the several-rank variant then takes the 148 registers it wants: 15 % faster at that occupancy; the single-rank one gains nothing)
// The body of one patch.  HALO here means "this patch takes part in the exchange between RANKS" (a boundary patch of a several-rank run); the
// interior patches of a several-rank run send nothing and stage no ghost node, so they run the single-rank body (HALO = false) -- the kernel
// below branches once, per workgroup, and each branch keeps the register allocation it has in its own build (the interior patches of the
// several-rank kernel used to carry that build's seven spilled registers and its longer element and node phases: DESIGN 4.1c).
template <int T, bool POW4, bool HALO, bool OVL>
__device__ __forceinline__ void resident_body(const DevMesh &m, const DevPatches &pp, const DevState &s, const DevWork &w, const DevParams *__restrict__ pdev, const DevResident &r,
                                              const double *__restrict__ Sc, double *__restrict__ Sn, const double move_dt,
                                              const HaloFused *__restrict__ hfp, const int n_boundary, const int blk, const bool boundary, const unsigned long long x0) {
    const DevParams &p0 = *pdev;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Mmax = pp.Mmax, Emax = pp.Emax, Pmax = pp.Pmax;
    typedef double d2 __attribute__((ext_vector_type(2)));
    // lF2: the corner forces as (x, y) pairs, [3][Emax] + one pair of zeros behind them: a node's gather is eight independent 16-byte
    // reads (pad entries and ghost corners read the zeros: x - (+0) == x, bit for bit) followed by the reference's subtractions in
    // the reference's order -- not eight dependent rounds of "entry, branch, two reads, two subtractions"
    double *lu = lds, *lv = lu + Mmax, *lF = lv + Mmax /*d2 [3][Emax] + 1*/, *ldx = lF + 6 * (size_t)Emax + 2 /*[6][Emax]*/,
           *lN = ldx + 6 * (size_t)Emax /*[10][Pmax]*/, *lM = lN + 10 * (size_t)Pmax /*[4][Pmax]*/;
    d2 *lF2 = reinterpret_cast<d2 *>(lF);
    uint4 *lFan4 = reinterpret_cast<uint4 *>(lM + 4 * (size_t)Pmax);  // [Pmax] the first eight fan entries of every own node as indices into lF2 (16 bits each)
    // HALO: what a boundary patch needs every sub-step, looked up ONCE (each of these was a chain of two or three dependent loads in the loop):
    int4 *lH = reinterpret_cast<int4 *>(lFan4 + Pmax);   // [min(Mmax, T)] halo slot nO + t: {node, -1, -, -} or, a ghost, {offset of u in a mailbox half, distance to v, ghost number, -}
    double **lPeerSeg = reinterpret_cast<double **>(lH + (Mmax < T ? Mmax : T));  // [NXS_RES_MAXNB] the neighbour ranks' mailbox segments for my values,
    long long *lPeerStride = reinterpret_cast<long long *>(lPeerSeg + NXS_RES_MAXNB);  // the distance of their second half,
    int *lPeerVd = reinterpret_cast<int *>(lPeerStride + NXS_RES_MAXNB);                // and of the v-block inside a segment
    const unsigned ZIDX = 3u * (unsigned)Emax;
    __shared__ int lerr;
    const int t = threadIdx.x, Nn = m.Nn, S = p0.substeps;
    const int nM = pp.node_cnt[blk], nE = pp.elem_cnt[blk], nO = pp.own_cnt[blk], nNb = r.pnbr_cnt[blk];
    const int *pn = pp.pnodes + (size_t)blk * Mmax;
    const bool bbm = p0.dynamics_type == NXS_DYN_BBM;
    if (t == 0) { lerr = 0; lF2[ZIDX] = d2{0., 0.}; }

    // ---- once per step: indices, velocities and frozen coordinates of the staged nodes, this thread's element, this thread's node
    double *sx = lF, *sy = lF + Mmax;  // (scratch: the corner forces are not needed yet)
    for (int i = t; i < nM; i += T) {
        const int g = pn[i];
        lu[i] = s.VT[g]; lv[i] = s.VT[g + Nn];
        const d2 c = reinterpret_cast<const d2 *>(w.xy)[g];
        sx[i] = c.x; sy[i] = c.y;
    }
    const bool has_elem = t < nE;
    int e = 0;
    bool writer = false, skip = true;
    ushort4 tr = make_ushort4(0, 0, 0, 0);
    double sig[3] = {0., 0., 0.}, damage = 0.;
    if (has_elem) {
        const int eraw = (OVL ? r.pelem : pp.pelem)[(size_t)blk * Emax + t];
        tr = reinterpret_cast<const ushort4 *>(OVL ? r.ptri : pp.ptri)[(size_t)blk * Emax + t];
        writer = eraw >= 0;
        e = writer ? eraw : ~eraw;
        const d2 *S4 = reinterpret_cast<const d2 *>(Sc) + 2 * (size_t)e;
        const d2 a = S4[0], c2 = S4[1];
        sig[0] = a.x; sig[1] = a.y; sig[2] = c2.x; damage = c2.y;
        const long long pk = __double_as_longlong(w.erec[6 * (size_t)e + 5]);
        skip = bbm ? (int)(pk & 0xffffffffll) < 0 : (pk >> 32) != 0;
    }
    // OVL: whole wavefronts of interior elements ("early") run one exchange ahead, the others ("late") follow the halo loads
    const bool early = OVL && t < r.ecut[blk], late = has_elem && !early;
    const bool has_node = t < nO;
    const int n = has_node ? pn[t] : 0;
    unsigned char nf = 0;
    const unsigned short *pf = (OVL ? r.pfan : pp.pfan) + (size_t)blk * pp.Wp * Pmax;
    if (has_node) {
        unsigned idx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned ent = (k < pp.Wp) ? pf[(size_t)k * Pmax + t] : 0xFFFFu;
            idx[k] = (ent == 0xFFFFu || (ent & 4u)) ? ZIDX : (ent & 3u) * (unsigned)Emax + (ent >> 3);  // pad, or ghostNodes[i] (FE.cpp:10456)
        }
        lFan4[t] = make_uint4(idx[0] | (idx[1] << 16), idx[2] | (idx[3] << 16), idx[4] | (idx[5] << 16), idx[6] | (idx[7] << 16));
    }
    if (has_node) {
        nf = m.nflags[n];
        const d2 *q = reinterpret_cast<const d2 *>(w.nrec) + 5 * (size_t)n;
#pragma unroll
        for (int k = 0; k < 5; ++k) { const d2 v = q[k]; lN[(size_t)(2 * k) * Pmax + t] = v.x; lN[(size_t)(2 * k + 1) * Pmax + t] = v.y; }
        lM[t] = s.UM[n]; lM[Pmax + t] = s.UM[n + Nn]; lM[2 * (size_t)Pmax + t] = s.UT[n]; lM[3 * (size_t)Pmax + t] = s.UT[n + Nn];
    }
    int nbr = -1;
    if (t < nNb) nbr = r.pnbr[(size_t)blk * NXS_RES_NBR + t];
    // HALO: where this thread's halo slot gets its velocity from, and where this thread's own node is sent to -- bit 31: it is sent somewhere;
    // bit 30: to more than one neighbour rank (a corner of the partition: the tables are walked as in k_substep_fused); else bits 29-26 the
    // neighbour, bits 25-0 the position in its segment
    // A corner of the partition: bit 30 says "and to a second neighbour rank", whose word (same form) sits in the free lane of lH[t]; its bit 30: "and to more"
    // -- only those walk the tables, from the third entry on (the walk is three dependent loads in front of the stores, on the exchange's critical path: the patch
    // with a corner node used to finish its node phase 2 us after the others, every sub-step)
    unsigned sinfo = 0u;
    if (HALO) {
        int second = 0;
        if (boundary && has_node) {
            const int sq0 = hfp->send_ptr[n], sq1 = hfp->send_ptr[n + 1];
            if (sq1 > sq0) sinfo = 0x80000000u | (sq1 - sq0 > 1 ? 0x40000000u : 0u) | ((unsigned)hfp->send_k[sq0] << 26) | (unsigned)hfp->send_pos[sq0];
            if (sq1 - sq0 > 1) second = (int)(0x80000000u | (sq1 - sq0 > 2 ? 0x40000000u : 0u) | ((unsigned)hfp->send_k[sq0 + 1] << 26) | (unsigned)hfp->send_pos[sq0 + 1]);
        }
        if (nO + t < nM) {
            const int g = pn[nO + t];
            lH[t] = (g >= m.No) ? make_int4(hfp->ghost_off[g - m.No], hfp->ghost_srl[g - m.No], g - m.No, second) : make_int4(g, -1, 0, second);
        } else if (has_node) {
            lH[t] = make_int4(0, -1, 0, second);
        }
        if (t < hfp->ipc.ns && t < NXS_RES_MAXNB) {
            lPeerSeg[t] = hfp->ipc.peer_seg[t];
            lPeerStride[t] = hfp->ipc.peer_parity_stride[t];
            lPeerVd[t] = hfp->send_off[t + 1] - hfp->send_off[t];
        }
    }
    __syncthreads();
    if (has_elem) {  // shapeCoeff (FE.cpp:1951-1964): frozen over the sub-steps (Q4), built once, the same quotients as k_prep_elements
        const double vx[3] = {sx[tr.x], sx[tr.y], sx[tr.z]};
        const double vy[3] = {sy[tr.x], sy[tr.y], sy[tr.z]};
        const double jac = jacobian(vx, vy);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
            ldx[(size_t)k * Emax + t] = (vy[kp1] - vy[kp2]) / jac;
            ldx[(size_t)(k + 3) * Emax + t] = (vx[kp2] - vx[kp1]) / jac;
        }
    }
    __syncthreads();  // (sx / sy are read; lF may be written from here on)

    if (OVL && early) {  // sub-step 0 of the interior elements
        int tt = t;
        asm volatile("" : "+v"(tt));
        const DevParams *pl = pdev;
        asm volatile("" : "+s"(pl));
        resident_element<POW4>(*pl, w.erec, bbm, skip, e, tr, tt, Emax, lu, lv, ldx, lF, sig, damage);
    }
    for (int ss = 0; ss < S; ++ss) {
        // (the thread index through an opaque copy: otherwise every LDS address of the loop body -- some forty of them -- is computed
        // once before the loop and kept in a register of its own across all the sub-steps)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // the parameters are re-read where they are used (scalar loads that hit the constant cache): held across the loop their ~40
        // values would push the element's own state out of the registers
        const DevParams *pl = pdev;
        asm volatile("" : "+s"(pl));
        const DevParams &p = *pl;
#ifdef NXS_PHASE_TIMING
#define RSTAMP(k) do { if (ss == 60 && threadIdx.x == 0 && blk < 8192 && r.rank == 0) g_phase_t[8 * blk + (k)] = wall_clock64(); } while (0)
#define RSTAMP_ANY(k) do { if (ss == 60 && blk < 8192 && r.rank == 0) g_phase_t[8 * blk + (k)] = wall_clock64(); } while (0)   /* by whichever thread executes it */
#define RSTAMP_PUB(k) do { if (ss == 60 && r.rank == 0) g_phase_t[8 * 8191 + (k)] = wall_clock64(); } while (0)   /* the publishing patch of the sub-step: row 8191 */
#else
#define RSTAMP(k) do { } while (0)
#define RSTAMP_ANY(k) do { } while (0)
#define RSTAMP_PUB(k) do { } while (0)
#endif
        RSTAMP(0);
        // ---- element phase (FE.cpp:4137-4260 / 10649-10726 + the element half of 10445-10467)
        if (OVL) {
            if (late) resident_element<POW4>(p, w.erec, bbm, skip, e, tr, tt, Emax, lu, lv, ldx, lF, sig, damage);  // (the interior ones are already one sub-step ahead)
        } else if (has_elem) {
            // the element constants: one 48-byte record, re-read every sub-step (it stays in the L2; the loads were issued ahead of
            // the barrier above) -- held in registers across the loop they pushed 36 others out to scratch
            double c_expC, volume, c_pmax, c_heal, c_coh, c_dxs = 1.;
            {
                const d2 *q = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)e;
                const d2 r0 = q[0], r1 = q[1], r2 = q[2];
                c_expC = r0.x; volume = r0.y; c_pmax = r1.x; c_heal = r1.y; c_coh = r2.x;
                const int dxi = (int)(__double_as_longlong(r2.y) & 0xffffffffll);
                if (bbm) c_dxs = (double)(dxi < 0 ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // FE.cpp:4232
            }
            double dxN[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) dxN[k] = ldx[(size_t)k * Emax + tt];
            if (skip) {
                sig[0] = sig[1] = sig[2] = 0.;
                damage = 0.;
            } else {
                const double u[3] = {lu[tr.x], lu[tr.y], lu[tr.z]};
                const double v[3] = {lv[tr.x], lv[tr.y], lv[tr.z]};
                if (bbm) bbm_stress<POW4>(p, dxN, u, v, sig, damage, c_expC, c_pmax, c_heal, c_dxs, c_coh);
                else vp_stress(p, dxN, u, v, sig, c_expC);
            }
            double F[6];
            corner_forces(volume, sig, dxN, F);
#pragma unroll
            for (int k = 0; k < 3; ++k) lF2[(size_t)k * Emax + tt] = d2{F[k], F[k + 3]};
        }
        __syncthreads();
        RSTAMP(1);
        const DevParams *pq = pdev;
        asm volatile("" : "+s"(pq));
        const DevParams &q = *pq;
        if (HALO) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_SEND_STORE);
        // ---- node phase (FE.cpp:10445-10553): fan gather in ascending element order, 2x2 solve, mesh move, publish
        if (has_node) {
            double uice = lu[tt], vice = lv[tt];
            const double node_mass = lN[tt];
            if (!((nf & NF_DIRICHLET) || node_mass == 0.)) {
                double gx = lN[(size_t)Pmax + tt], gy = lN[2 * (size_t)Pmax + tt];
                {
                    const uint4 fw = lFan4[tt];
                    const unsigned w4[4] = {fw.x, fw.y, fw.z, fw.w};
                    d2 f[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = lF2[(w4[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { gx -= f[k].x; gy -= f[k].y; }
                }
                for (int k = 8; k < pp.Wp; ++k) {  // (fans of more than eight elements: the most irregular vertices)
                    const unsigned ent = pf[(size_t)k * Pmax + tt];
                    if (ent == 0xFFFFu) break;
                    if (ent & 4u) continue;
                    const d2 f = lF2[(ent & 3u) * (unsigned)Emax + (ent >> 3)];
                    gx -= f.x; gy -= f.y;
                }
                nodal_solve<true>(q, gx, gy, uice, vice, node_mass, lN[3 * (size_t)Pmax + tt], lN[4 * (size_t)Pmax + tt], lN[5 * (size_t)Pmax + tt],
                            (nf & NF_LAT_NEG) ? -1. : 1., lN[6 * (size_t)Pmax + tt], lN[7 * (size_t)Pmax + tt], lN[8 * (size_t)Pmax + tt],
                            lN[9 * (size_t)Pmax + tt], 0., 0.);
            }
            lu[tt] = uice; lv[tt] = vice;
            if (move_dt != 0.) {  // FE.cpp:10543-10550; Neumann nodes keep M_UM (restore == skip)
                if (!(nf & NF_NEUMANN)) { lM[tt] += move_dt * uice; lM[Pmax + tt] += move_dt * vice; }
                lM[2 * (size_t)Pmax + tt] += move_dt * uice; lM[3 * (size_t)Pmax + tt] += move_dt * vice;
            }
            if (ss == S - 1) { s.VT[n] = uice; s.VT[n + Nn] = vice; }
            else {
                double *X = (ss & 1) ? r.X1 : r.X0;
                st_agent(X + n, uice); st_agent(X + n + Nn, vice);
            }
            unsigned si = sinfo;
            if (HALO) asm volatile("" : "+v"(si));  // (decoded here every sub-step: hoisted out of the loop, neighbour, position and the LDS addresses would each hold a register across it)
            if (HALO && (si & 0x80000000u)) {  // updateGhosts, sending side: straight into the neighbour rank's mailbox
                const unsigned long long half = (x0 + (unsigned long long)ss) & 1ull;
                {
                    const unsigned k = (si >> 26) & 15u;
                    double *dst = lPeerSeg[k] + half * lPeerStride[k] + (si & 0x3FFFFFFu);
                    sys_store(dst, uice);
                    sys_store(dst + lPeerVd[k], vice);
                }
                if (si & 0x40000000u) {  // ... a second one's
                    const unsigned s2 = (unsigned)lH[tt].w;
                    const unsigned k = (s2 >> 26) & 15u;
                    double *dst = lPeerSeg[k] + half * lPeerStride[k] + (s2 & 0x3FFFFFFu);
                    sys_store(dst, uice);
                    sys_store(dst + lPeerVd[k], vice);
                    if (s2 & 0x40000000u) {  // ... and more (as k_substep_fused, from the third entry)
                        const int sq0 = hfp->send_ptr[n], sq1 = hfp->send_ptr[n + 1];
                        for (int qq = sq0 + 2; qq < sq1; ++qq) {
                            const int kk = hfp->send_k[qq];
                            double *d3 = hfp->ipc.peer_seg[kk] + half * hfp->ipc.peer_parity_stride[kk] + hfp->send_pos[qq];
                            sys_store(d3, uice);
                            sys_store(d3 + (hfp->send_off[kk + 1] - hfp->send_off[kk]), vice);
                        }
                    }
                }
            }
        }
        if (!HALO && ss == S - 1) break;
        RSTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's stores have left; the corner forces have been read
        RSTAMP(3);
        // this patch's own counter first: the patches of this rank that wait for it should not wait for the round trip of the atomic below
        if (ss < S - 1 && t == 0) __hip_atomic_store(r.flag + 32 * (size_t)blk, (unsigned)(ss + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (HALO && boundary && t == 0) {
            // the last boundary patch to finish this sub-step publishes it to the neighbour ranks -- in sub-step order: patches far
            // apart may be several sub-steps apart, so the one that completes sub-step ss waits for ss - 1 to have been published
            unsigned raised = __hip_atomic_load(r.raised, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (asked for together with the ticket: one round trip, not two; it only grows)
            if (atomicAdd(r.cnt + ss, 1u) == (unsigned)n_boundary - 1u) {
                RSTAMP_PUB(1);   // (the last boundary patch of the sub-step: its ticket is back)
                const long long t0 = wall_clock64();
                while (raised < (unsigned)ss) {
                    __builtin_amdgcn_s_sleep(1);
                    raised = __hip_atomic_load(r.raised, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wall_clock64() - t0 > 1000000000ll) { lerr = 1; atomicExch(r.error, 6); break; }  // 10 s, as every other inter-rank wait
                }
                nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PUBLISH_FLAG);
                // the one release of the sub-step -- a RELEASE only: __threadfence_system() is an acquire as well, i.e. it also invalidates this
                // XCD's L2, and every patch on the XCD then re-reads its element constants from memory instead of the L2, every sub-step
                if (!hfp->no_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // (option "resident_release" 0 leaves it out: see HaloFused::no_release)
                for (int k = 0; k < hfp->ipc.ns; ++k)
                    __hip_atomic_store(hfp->ipc.peer_flag[k], x0 + (unsigned long long)ss + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ss == S - 1) *hfp->ipc.seq_push = x0 + (unsigned long long)S;
                __hip_atomic_store(r.raised, (unsigned)(ss + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                RSTAMP_PUB(2);   // (... its flags are stored)
            }
        }
        if (ss == S - 1) break;
        if (OVL && early) {  // the interior elements' update for sub-step ss + 1, under the exchange: their corners are own nodes (lu / lv of
                             // the node phase above), their corner forces of sub-step ss have been read (the barrier above)
            const DevParams *pe = pdev;
            asm volatile("" : "+s"(pe));
            resident_element<POW4>(*pe, w.erec, bbm, skip, e, tr, tt, Emax, lu, lv, ldx, lF, sig, damage);
        }
        if (HALO && boundary && tt >= 64 && tt < 64 + hfp->ipc.nr) {  // exchange x0 + ss of every neighbour rank must have landed
            const int k = tt - 64;
            const long long t0 = wall_clock64();
            unsigned polls = 0u;
            while (__hip_atomic_load(hfp->ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < x0 + (unsigned long long)ss + 1ull) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 31u) != 0u) continue;  // (error word and clock every 32nd poll: a poll period is one round trip, not two)
                if (__hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { lerr = 1; break; }
                if (wall_clock64() - t0 > 1000000000ll) { lerr = 1; atomicExch(r.error, 7); break; }  // 10 s: the transport's bound (a neighbour rank may start its step late: output, a regrid, host thermodynamics)
            }
            if (k == 0) RSTAMP_ANY(6);   // (phase microscope: this patch has seen neighbour rank 0's flag)
        }
        int nb = nbr;
        asm volatile("" : "+v"(nb));  // (its counter's address is formed here, not kept across the loop)
        if (nb >= 0) {
            const long long t0 = wall_clock64();  // 100 MHz
            unsigned polls = 0u;
            while (__hip_atomic_load(r.flag + 32 * (size_t)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(ss + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 31u) != 0u) continue;  // (error word and clock every 32nd poll: a poll period is one round trip, not two)
                if (__hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { lerr = 1; break; }
                if (wall_clock64() - t0 > 1200000000ll) { lerr = 1; atomicExch(r.error, 5); break; }  // 12 s: longer than the inter-rank waits, so that a late neighbour RANK is reported as that (7) by the patch that waits for it, not as a missing patch (5) by that patch's neighbours
            }
        }
        __syncthreads();
        RSTAMP(4);
        if (lerr) break;
        if (HALO) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_STAGE_READ);
        {   // the halo nodes' new velocities, past the caches
            const double *X = (ss & 1) ? r.X1 : r.X0;
            if (HALO) {
                // a ghost node comes from this rank's mailbox; what arrived also goes to the ghosts' ring (every patch that stages the node
                // writes the same two values): k_ghost_ring_move makes the ghosts' mesh moves of all sub-steps after the launch
                const double *mb = hfp->ipc.mailbox + ((x0 + (unsigned long long)ss) & 1ull) * 2ull * (unsigned long long)hfp->ipc.tr;
                double *gr = r.gring + (size_t)ss * 2 * (size_t)r.NG;
                if (nO + tt < nM) {
                    const int4 hv = lH[tt];
                    if (hv.y >= 0) {
                        const double gu = sys_load(mb + hv.x), gv = sys_load(mb + hv.x + hv.y);
                        lu[nO + tt] = gu; lv[nO + tt] = gv;
                        if (move_dt != 0.) { gr[hv.z] = gu; gr[r.NG + hv.z] = gv; }
                    } else {
                        lu[nO + tt] = ld_agent(X + hv.x); lv[nO + tt] = ld_agent(X + hv.x + Nn);
                    }
                }
                for (int i = nO + T + t; i < nM; i += T) {  // (more halo nodes than threads: not with the patch sizes this kernel is used for)
                    const int g = pn[i];
                    if (g >= m.No) {
                        const int go = hfp->ghost_off[g - m.No], gs = hfp->ghost_srl[g - m.No];
                        const double gu = sys_load(mb + go), gv = sys_load(mb + go + gs);
                        lu[i] = gu; lv[i] = gv;
                        if (move_dt != 0.) { gr[g - m.No] = gu; gr[r.NG + g - m.No] = gv; }
                    } else {
                        lu[i] = ld_agent(X + g); lv[i] = ld_agent(X + g + Nn);
                    }
                }
            } else {
                for (int i = nO + t; i < nM; i += T) {
                    const int g = pn[i];
                    lu[i] = ld_agent(X + g); lv[i] = ld_agent(X + g + Nn);
                }
            }
        }
        __syncthreads();
        RSTAMP(5);
    }
    // ---- once per step: the element state and the moved mesh go back -- unless a wait of this patch timed out or the error word is already set by another
    // patch: then nothing of THIS patch is written.  Patches that finished all their sub-steps before the time-out elsewhere have written theirs (there is no
    // barrier over the grid), so after an error the state is a mixture of the step's start and its end: nxs_dyn_synchronize and every call that hands state
    // to the host report the lost step, and the caller restores the state (nxs_dyn_put_state) before it goes on
    if (t == 0 && __hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) lerr = 1;
    __syncthreads();
    if (lerr) return;
    if (has_elem && writer) {
        d2 *S4 = reinterpret_cast<d2 *>(Sn) + 2 * (size_t)e;
        S4[0] = d2{sig[0], sig[1]}; S4[1] = d2{sig[2], damage};
    }
    if (has_node && move_dt != 0.) {
        if (!(nf & NF_NEUMANN)) { s.UM[n] = lM[t]; s.UM[n + Nn] = lM[Pmax + t]; }
        s.UT[n] = lM[2 * (size_t)Pmax + t]; s.UT[n + Nn] = lM[3 * (size_t)Pmax + t];
    }
}

template <int T, bool POW4, bool HALO, bool OVL = false, int WPE = NXS_RES_WAVES>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_substep_resident(DevMesh m, DevPatches pp, DevState s, DevWork w, const DevParams *__restrict__ pdev, DevResident r,
                                                        const double *__restrict__ Sc, double *__restrict__ Sn, double move_dt,
                                                        const HaloFused *__restrict__ hfp, int n_boundary) {
    auto xcd_remap = [](int pos, int n) {  // position in dispatch order -> index (see k_substep_fused)
        const int q = n >> 3, rr = n & 7, x = pos & 7;
        return (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (pos >> 3);
    };
    int blk = blockIdx.x;
    if (HALO) {
        const bool boundary = blk < n_boundary;  // boundary patches lead the grid (HaloFused)
        blk = boundary ? xcd_remap(blk, n_boundary) : n_boundary + xcd_remap(blk - n_boundary, (int)gridDim.x - n_boundary);
        if (boundary) {  // exchanges this rank has published so far; changed only after every boundary patch has finished (uniform: kept in scalar registers)
            const unsigned long long xv = *hfp->ipc.seq_push;
            const unsigned long long x0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(xv & 0xffffffffull));
            resident_body<T, POW4, true, OVL>(m, pp, s, w, pdev, r, Sc, Sn, move_dt, hfp, n_boundary, blk, true, x0);
        } else {
            resident_body<T, POW4, false, false>(m, pp, s, w, pdev, r, Sc, Sn, move_dt, nullptr, 0, blk, false, 0ull);
        }
    } else {
        resident_body<T, POW4, false, false>(m, pp, s, w, pdev, r, Sc, Sn, move_dt, nullptr, 0, xcd_remap(blk, (int)gridDim.x), false, 0ull);
    }
}

// ------------------------------------------------------------------------------------------------
// v4b  The resident sub-step loop for partitions of up to ~400 k triangles (a rank of four of the 2 km mesh): ONE 512-thread workgroup per
// CU, compiled for two waves per SIMD (256 VGPRs), EPT = 4 elements and NPT = 2 own nodes per thread.  What k_substep_resident keeps in LDS for a
// patch of 180 nodes does not fit a CU for a patch of 720 (1 600 elements: corner forces 77 KB, shape coefficients 77 KB, nodal inputs 58 KB,
// M_UM / M_UT 23 KB ...), but the register file of a CU is 512 KB: here an element's stress, damage AND its six frozen shape coefficients (Q4) stay
// in registers (20 per element), the running M_UM / M_UT of the own nodes too (8 per node); LDS holds only what crosses threads -- the staged
// velocities, the corner forces, the first eight fan entries of every own node; the element constants (48 B) and the nodal inputs (80 B) are
// re-read from L2 every sub-step, the loads issued ahead of the barrier they wait behind.  The exchange between patches (and, HALO, between
// ranks through the mailboxes) is k_substep_resident's, statement for statement; so are the operations and their order: bit-identical to one
// launch per sub-step (tests/test_gpu_parity.py::test_resident_sub_step_loop_does_not_change_a_bit).
#define NXS_RESB_EPT 4
#define NXS_RESB_NPT 2
template <bool POW4, bool HALO, bool OVL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_substep_resident_big(DevMesh m, DevPatches pp, DevState s, DevWork w, const DevParams *__restrict__ pdev, DevResident r,
                                                        const double *__restrict__ Sc, double *__restrict__ Sn, double move_dt,
                                                        const HaloFused *__restrict__ hfp, int n_boundary) {
    constexpr int T = 512, EPT = NXS_RESB_EPT, NPT = NXS_RESB_NPT;
    const DevParams &p0 = *pdev;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int Mmax = pp.Mmax, Emax = pp.Emax, Pmax = pp.Pmax;
    typedef double d2 __attribute__((ext_vector_type(2)));
    double *lu = lds, *lv = lu + Mmax, *lF = lv + Mmax /*d2 [3][Emax] + 1*/;
    d2 *lF2 = reinterpret_cast<d2 *>(lF);
    uint4 *lFan4 = reinterpret_cast<uint4 *>(lF + 6 * (size_t)Emax + 2);  // [Pmax] the first eight fan entries of every own node as indices into lF2 (16 bits each)
    int4 *lH = reinterpret_cast<int4 *>(lFan4 + Pmax);                    // HALO: [Mmax] halo slot i - nO: {node, -1, -, .} or, a ghost, {offset of u in a mailbox half, distance to v, ghost number, .}; fourth lane: see below
    double **lPeerSeg = reinterpret_cast<double **>(lH + Mmax);           // HALO: [NXS_RES_MAXNB] the neighbour ranks' mailbox segments for my values,
    long long *lPeerStride = reinterpret_cast<long long *>(lPeerSeg + NXS_RES_MAXNB);  // the distance of their second half,
    int *lPeerVd = reinterpret_cast<int *>(lPeerStride + NXS_RES_MAXNB);                // and of the v-block inside a segment
    const unsigned ZIDX = 3u * (unsigned)Emax;
    __shared__ int lerr;
    auto xcd_remap = [](int pos, int n) {  // position in dispatch order -> index (see k_substep_fused)
        const int q = n >> 3, rr = n & 7, x = pos & 7;
        return (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (pos >> 3);
    };
    int blk = blockIdx.x;
    bool boundary = false;
    unsigned long long x0 = 0ull;
    if (HALO) {
        boundary = blk < n_boundary;
        if (boundary) {
            const unsigned long long xv = *hfp->ipc.seq_push;
            x0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(xv & 0xffffffffull));
        }
        blk = boundary ? xcd_remap(blk, n_boundary) : n_boundary + xcd_remap(blk - n_boundary, (int)gridDim.x - n_boundary);
    } else {
        blk = xcd_remap(blk, (int)gridDim.x);
    }
    const int t = threadIdx.x, Nn = m.Nn, S = p0.substeps;
    const int nM = pp.node_cnt[blk], nE = pp.elem_cnt[blk], nO = pp.own_cnt[blk], nNb = r.pnbr_cnt[blk];
    const int *pn = pp.pnodes + (size_t)blk * Mmax;
    // OVL: the element list with the interior elements first (nxs_cut::plan_resident); slices j < JE of it (whole slices of T elements) hold
    // interior elements only, and their next update is computed while the patch waits for its neighbours
    const int *pel = (OVL ? r.pelem : pp.pelem) + (size_t)blk * Emax;
    const ushort4 *ptr4 = reinterpret_cast<const ushort4 *>(OVL ? r.ptri : pp.ptri) + (size_t)blk * Emax;
    const int JE = OVL ? r.ecut[blk] / T : 0, JA = (JE + 1) / 2;
    const bool bbm = p0.dynamics_type == NXS_DYN_BBM;
    if (t == 0) { lerr = 0; lF2[ZIDX] = d2{0., 0.}; }

    // ---- once per step: velocities and frozen coordinates of the staged nodes, this thread's elements, this thread's nodes
    double *sx = lF, *sy = lF + Mmax;  // (scratch: the corner forces are not needed yet; 2 Mmax <= 6 Emax)
    for (int i = t; i < nM; i += T) {
        const int g = pn[i];
        lu[i] = s.VT[g]; lv[i] = s.VT[g + Nn];
        const d2 c = reinterpret_cast<const d2 *>(w.xy)[g];
        sx[i] = c.x; sy[i] = c.y;
    }
    int e[EPT];
    unsigned trp[EPT];            // the three corner slots, ten bits each (Mmax <= 1024)
    unsigned eflags = 0u;         // per element j: bit j has an element, bit 8 + j writer, bit 16 + j skip
    double sig[EPT][3], damage[EPT], dxN[EPT][6];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int l = t + T * j;
        e[j] = 0; trp[j] = 0u;
        sig[j][0] = sig[j][1] = sig[j][2] = 0.; damage[j] = 0.;
#pragma unroll
        for (int k = 0; k < 6; ++k) dxN[j][k] = 0.;
        if (l < nE) {
            const int eraw = pel[l];
            const ushort4 tr = ptr4[l];
            trp[j] = (unsigned)tr.x | ((unsigned)tr.y << 10) | ((unsigned)tr.z << 20);
            const bool writer = eraw >= 0;
            e[j] = writer ? eraw : ~eraw;
            const d2 *S4 = reinterpret_cast<const d2 *>(Sc) + 2 * (size_t)e[j];
            const d2 a = S4[0], c2 = S4[1];
            sig[j][0] = a.x; sig[j][1] = a.y; sig[j][2] = c2.x; damage[j] = c2.y;
            const long long pk = __double_as_longlong(w.erec[6 * (size_t)e[j] + 5]);
            const bool skip = bbm ? (int)(pk & 0xffffffffll) < 0 : (pk >> 32) != 0;
            eflags |= (1u << j) | (writer ? (1u << (8 + j)) : 0u) | (skip ? (1u << (16 + j)) : 0u);
        }
    }
    int n[NPT];
    unsigned nfl = 0u;            // per node i: bits 8 i .. 8 i + 7 the node flags, bit 24 + i has a node
    double um[NPT][4];            // the running M_UM (u, v) and M_UT (u, v) of the own nodes
    unsigned sinfo[NPT];          // HALO: where the own node is sent (see k_substep_resident)
    const unsigned short *pf = (OVL ? r.pfan : pp.pfan) + (size_t)blk * pp.Wp * Pmax;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int sl = t + T * i;
        n[i] = 0; sinfo[i] = 0u;
        um[i][0] = um[i][1] = um[i][2] = um[i][3] = 0.;
        if (sl < nO) {
            n[i] = pn[sl];
            nfl |= ((unsigned)m.nflags[n[i]] << (8 * i)) | (1u << (24 + i));
            unsigned idx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned ent = (k < pp.Wp) ? pf[(size_t)k * Pmax + sl] : 0xFFFFu;
                idx[k] = (ent == 0xFFFFu || (ent & 4u)) ? ZIDX : (ent & 3u) * (unsigned)Emax + (ent >> 3);  // pad, or ghostNodes[i] (FE.cpp:10456)
            }
            lFan4[sl] = make_uint4(idx[0] | (idx[1] << 16), idx[2] | (idx[3] << 16), idx[4] | (idx[5] << 16), idx[6] | (idx[7] << 16));
            um[i][0] = s.UM[n[i]]; um[i][1] = s.UM[n[i] + Nn]; um[i][2] = s.UT[n[i]]; um[i][3] = s.UT[n[i] + Nn];
            if (HALO && boundary) {  // (bit 30: and to a second neighbour rank -- lH[sl].w, see k_substep_resident)
                const int sq0 = hfp->send_ptr[n[i]], sq1 = hfp->send_ptr[n[i] + 1];
                if (sq1 > sq0) sinfo[i] = 0x80000000u | (sq1 - sq0 > 1 ? 0x40000000u : 0u) | ((unsigned)hfp->send_k[sq0] << 26) | (unsigned)hfp->send_pos[sq0];
            }
        }
    }
    int nbr = -1;
    if (t < nNb) nbr = r.pnbr[(size_t)blk * NXS_RES_NBR + t];
    if (HALO) {
        // entry j: halo slot nO + j -- {node, -1, -, .} or, a ghost, {offset of u in a mailbox half, distance to v, ghost number, .} -- and, in its fourth lane,
        // the SECOND neighbour rank own node j is sent to (a corner of the partition; the form of sinfo, bit 30: "and to more": those walk the tables)
        const int nH = nM - nO, top = nH > nO ? nH : nO;
        for (int j = t; j < top; j += T) {
            int second = 0;
            if (boundary && j < nO) {
                const int g = pn[j];
                const int sq0 = hfp->send_ptr[g], sq1 = hfp->send_ptr[g + 1];
                if (sq1 - sq0 > 1) second = (int)(0x80000000u | (sq1 - sq0 > 2 ? 0x40000000u : 0u) | ((unsigned)hfp->send_k[sq0 + 1] << 26) | (unsigned)hfp->send_pos[sq0 + 1]);
            }
            int4 hv = make_int4(0, -1, 0, second);
            if (j < nH) {
                const int g = pn[nO + j];
                hv = (g >= m.No) ? make_int4(hfp->ghost_off[g - m.No], hfp->ghost_srl[g - m.No], g - m.No, second) : make_int4(g, -1, 0, second);
            }
            lH[j] = hv;
        }
        if (t < hfp->ipc.ns && t < NXS_RES_MAXNB) {
            lPeerSeg[t] = hfp->ipc.peer_seg[t];
            lPeerStride[t] = hfp->ipc.peer_parity_stride[t];
            lPeerVd[t] = hfp->send_off[t + 1] - hfp->send_off[t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {  // shapeCoeff (FE.cpp:1951-1964): frozen over the sub-steps (Q4), built once, the same quotients as k_prep_elements
        if (eflags & (1u << j)) {
            const unsigned a = trp[j] & 1023u, b = (trp[j] >> 10) & 1023u, c = (trp[j] >> 20) & 1023u;
            const double vx[3] = {sx[a], sx[b], sx[c]};
            const double vy[3] = {sy[a], sy[b], sy[c]};
            const double jac = jacobian(vx, vy);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int kp1 = (k + 1) % 3, kp2 = (k + 2) % 3;
                dxN[j][k] = (vy[kp1] - vy[kp2]) / jac;
                dxN[j][k + 3] = (vx[kp2] - vx[kp1]) / jac;
            }
        }
    }
    __syncthreads();  // (sx / sy are read; lF may be written from here on)

    // ---- element phase (FE.cpp:4137-4260 / 10649-10726 + the element half of 10445-10467); which: 0 every element of this thread, 1 the interior
    // slices (j < JE), 2 the others, 3 / 4 the interior slices in two parts (j < JA, JA <= j < JE)
    auto element_phase = [&](const DevParams &p, const int which) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            if (!(eflags & (1u << j)) || (which == 1 && j >= JE) || (which == 2 && j < JE) || (which == 3 && j >= JA) || (which == 4 && (j < JA || j >= JE))) continue;
            const int l = t + T * j;
            double c_expC, volume, c_pmax, c_heal, c_coh, c_dxs = 1.;
            {   // the element constants: one 48-byte record, re-read every sub-step (it stays in the L2)
                const d2 *q = reinterpret_cast<const d2 *>(w.erec) + 3 * (size_t)e[j];
                const d2 r0 = q[0], r1 = q[1], r2 = q[2];
                c_expC = r0.x; volume = r0.y; c_pmax = r1.x; c_heal = r1.y; c_coh = r2.x;
                const int dxi = (int)(__double_as_longlong(r2.y) & 0xffffffffll);
                if (bbm) c_dxs = (double)(dxi < 0 ? ~dxi : dxi) * p.sqrt_nu_rhoi;  // FE.cpp:4232
            }
            if (eflags & (1u << (16 + j))) {
                sig[j][0] = sig[j][1] = sig[j][2] = 0.;
                damage[j] = 0.;
            } else {
                const unsigned a = trp[j] & 1023u, b = (trp[j] >> 10) & 1023u, c = (trp[j] >> 20) & 1023u;
                const double u[3] = {lu[a], lu[b], lu[c]};
                const double v[3] = {lv[a], lv[b], lv[c]};
                if (bbm) bbm_stress<POW4>(p, dxN[j], u, v, sig[j], damage[j], c_expC, c_pmax, c_heal, c_dxs, c_coh);
                else vp_stress(p, dxN[j], u, v, sig[j], c_expC);
            }
            double F[6];
            corner_forces(volume, sig[j], dxN[j], F);
#pragma unroll
            for (int k = 0; k < 3; ++k) lF2[(size_t)k * Emax + l] = d2{F[k], F[k + 3]};
        }
    };
    if (OVL) {  // the interior elements' first update
        const DevParams *pl = pdev;
        asm volatile("" : "+s"(pl));
        element_phase(*pl, 1);
    }
    for (int ss = 0; ss < S; ++ss) {
        // the parameters are re-read where they are used (scalar loads that hit the constant cache), not held across the loop
        const DevParams *pl = pdev;
        asm volatile("" : "+s"(pl));
        const DevParams &p = *pl;
        RSTAMP(0);
        element_phase(p, OVL ? 2 : 0);
        // the nodal inputs of this thread's own nodes (80-byte records, from the L2): issued ahead of the barrier
        d2 nr[NPT][5];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
#pragma unroll
            for (int k = 0; k < 5; ++k) nr[i][k] = d2{0., 0.};
            if (nfl & (1u << (24 + i))) {
                const d2 *q = reinterpret_cast<const d2 *>(w.nrec) + 5 * (size_t)n[i];
#pragma unroll
                for (int k = 0; k < 5; ++k) nr[i][k] = q[k];
            }
        }
        __syncthreads();
        RSTAMP(1);
        const DevParams *pq = pdev;
        asm volatile("" : "+s"(pq));
        const DevParams &q = *pq;
        if (HALO && boundary) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_SEND_STORE);
        // ---- node phase (FE.cpp:10445-10553): fan gather in ascending element order, 2x2 solve, mesh move, publish
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            if (!(nfl & (1u << (24 + i)))) continue;
            const int sl = t + T * i;
            const unsigned nf = (nfl >> (8 * i)) & 0xFFu;
            double uice = lu[sl], vice = lv[sl];
            const double node_mass = nr[i][0].x;
            if (!((nf & NF_DIRICHLET) || node_mass == 0.)) {
                double gx = nr[i][0].y, gy = nr[i][1].x;
                {
                    const uint4 fw = lFan4[sl];
                    const unsigned w4[4] = {fw.x, fw.y, fw.z, fw.w};
                    d2 f[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = lF2[(w4[k >> 1] >> (16 * (k & 1))) & 0xFFFFu];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { gx -= f[k].x; gy -= f[k].y; }
                }
                for (int k = 8; k < pp.Wp; ++k) {  // (fans of more than eight elements: the most irregular vertices)
                    const unsigned ent = pf[(size_t)k * Pmax + sl];
                    if (ent == 0xFFFFu) break;
                    if (ent & 4u) continue;
                    const d2 f = lF2[(ent & 3u) * (unsigned)Emax + (ent >> 3)];
                    gx -= f.x; gy -= f.y;
                }
                nodal_solve<true>(q, gx, gy, uice, vice, node_mass, nr[i][1].y, nr[i][2].x, nr[i][2].y, (nf & NF_LAT_NEG) ? -1. : 1., nr[i][3].x, nr[i][3].y, nr[i][4].x, nr[i][4].y, 0., 0.);
            }
            lu[sl] = uice; lv[sl] = vice;
            if (move_dt != 0.) {  // FE.cpp:10543-10550; Neumann nodes keep M_UM (restore == skip)
                if (!(nf & NF_NEUMANN)) { um[i][0] += move_dt * uice; um[i][1] += move_dt * vice; }
                um[i][2] += move_dt * uice; um[i][3] += move_dt * vice;
            }
            if (ss == S - 1) { s.VT[n[i]] = uice; s.VT[n[i] + Nn] = vice; }
            else {
                double *X = (ss & 1) ? r.X1 : r.X0;
                st_agent(X + n[i], uice); st_agent(X + n[i] + Nn, vice);
            }
            if (HALO && (sinfo[i] & 0x80000000u)) {  // updateGhosts, sending side: straight into the neighbour ranks' mailboxes, from what was looked up once
                const unsigned long long half = (x0 + (unsigned long long)ss) & 1ull;
                {
                    const unsigned k = (sinfo[i] >> 26) & 15u;
                    double *dst = lPeerSeg[k] + half * lPeerStride[k] + (sinfo[i] & 0x3FFFFFFu);
                    sys_store(dst, uice);
                    sys_store(dst + lPeerVd[k], vice);
                }
                if (sinfo[i] & 0x40000000u) {
                    const unsigned s2 = (unsigned)lH[sl].w;
                    const unsigned k = (s2 >> 26) & 15u;
                    double *dst = lPeerSeg[k] + half * lPeerStride[k] + (s2 & 0x3FFFFFFu);
                    sys_store(dst, uice);
                    sys_store(dst + lPeerVd[k], vice);
                    if (s2 & 0x40000000u) {  // more than two neighbour ranks share the node: the rest as k_substep_fused
                        const int sq0 = hfp->send_ptr[n[i]], sq1 = hfp->send_ptr[n[i] + 1];
                        for (int qq = sq0 + 2; qq < sq1; ++qq) {
                            const int kk = hfp->send_k[qq];
                            double *d3 = hfp->ipc.peer_seg[kk] + half * hfp->ipc.peer_parity_stride[kk] + hfp->send_pos[qq];
                            sys_store(d3, uice);
                            sys_store(d3 + (hfp->send_off[kk + 1] - hfp->send_off[kk]), vice);
                        }
                    }
                }
            }
        }
        RSTAMP(2);
        if (!HALO && ss == S - 1) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's stores have left; the corner forces have been read
        RSTAMP(3);
        if (ss < S - 1 && t == 0) __hip_atomic_store(r.flag + 32 * (size_t)blk, (unsigned)(ss + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (HALO && boundary && t == 0) {  // the last boundary patch to finish this sub-step publishes it to the neighbour ranks, in sub-step order
            unsigned raised = __hip_atomic_load(r.raised, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (asked for together with the ticket, see k_substep_resident)
            if (atomicAdd(r.cnt + ss, 1u) == (unsigned)n_boundary - 1u) {
                const long long t0 = wall_clock64();
                while (raised < (unsigned)ss) {
                    __builtin_amdgcn_s_sleep(1);
                    raised = __hip_atomic_load(r.raised, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wall_clock64() - t0 > 1000000000ll) { lerr = 1; atomicExch(r.error, 6); break; }
                }
                nxs_delay_at(hfp->ipc.delay, NXS_DELAY_PUBLISH_FLAG);
                if (!hfp->no_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // (option "resident_release" 0 leaves it out: see HaloFused::no_release)
                for (int k = 0; k < hfp->ipc.ns; ++k)
                    __hip_atomic_store(hfp->ipc.peer_flag[k], x0 + (unsigned long long)ss + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ss == S - 1) *hfp->ipc.seq_push = x0 + (unsigned long long)S;
                __hip_atomic_store(r.raised, (unsigned)(ss + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (ss == S - 1) break;
        if (OVL) element_phase(p, 3);  // sub-step ss + 1 of the interior elements: own nodes only, all solved (the barrier above)
        if (HALO && boundary && t >= 64 && t < 64 + hfp->ipc.nr) {  // exchange x0 + ss of every neighbour rank must have landed
            const int k = t - 64;
            const long long t0 = wall_clock64();
            unsigned polls = 0u;
            while (__hip_atomic_load(hfp->ipc.flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < x0 + (unsigned long long)ss + 1ull) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 31u) != 0u) continue;  // (error word and clock every 32nd poll: a poll period is one round trip, not two)
                if (__hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { lerr = 1; break; }
                if (wall_clock64() - t0 > 1000000000ll) { lerr = 1; atomicExch(r.error, 7); break; }
            }
        }
        if (nbr >= 0) {
            const long long t0 = wall_clock64();  // 100 MHz
            unsigned polls = 0u;
            while (__hip_atomic_load(r.flag + 32 * (size_t)nbr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(ss + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 31u) != 0u) continue;  // (error word and clock every 32nd poll: a poll period is one round trip, not two)
                if (__hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { lerr = 1; break; }
                if (wall_clock64() - t0 > 1200000000ll) { lerr = 1; atomicExch(r.error, 5); break; }
            }
        }
        __syncthreads();
        RSTAMP(4);
        if (lerr) break;
        if (HALO && boundary) nxs_delay_at(hfp->ipc.delay, NXS_DELAY_STAGE_READ);
        {   // the halo nodes' new velocities, past the caches; OVL: the first T of them travel while the second part of the interior elements is computed
            const double *X = (ss & 1) ? r.X1 : r.X0;
            const double *mb = nullptr;
            double *gr = nullptr;
            if (HALO) {
                mb = hfp->ipc.mailbox + ((x0 + (unsigned long long)ss) & 1ull) * 2ull * (unsigned long long)hfp->ipc.tr;
                gr = r.gring + (size_t)ss * 2 * (size_t)r.NG;
            }
            auto halo_load = [&](const int i, double &hu, double &hv, int &ghost) {
                ghost = -1;
                if (HALO) {
                    const int4 hv4 = lH[i - nO];
                    if (hv4.y >= 0) { hu = sys_load(mb + hv4.x); hv = sys_load(mb + hv4.x + hv4.y); ghost = hv4.z; }
                    else { hu = ld_agent(X + hv4.x); hv = ld_agent(X + hv4.x + Nn); }
                } else {
                    const int g = pn[i];
                    hu = ld_agent(X + g); hv = ld_agent(X + g + Nn);
                }
            };
            auto halo_keep = [&](const int i, const double hu, const double hv, const int ghost) {
                lu[i] = hu; lv[i] = hv;
                if (HALO && ghost >= 0 && move_dt != 0.) { gr[ghost] = hu; gr[r.NG + ghost] = hv; }
            };
            int i0 = nO + t;
            if (OVL) {
                double hu = 0., hv = 0.;
                int ghost = -1;
                if (i0 < nM) halo_load(i0, hu, hv, ghost);
                element_phase(p, 4);
                if (i0 < nM) halo_keep(i0, hu, hv, ghost);
                i0 += T;
            }
            for (int i = i0; i < nM; i += T) {
                double hu, hv;
                int ghost;
                halo_load(i, hu, hv, ghost);
                halo_keep(i, hu, hv, ghost);
            }
        }
        __syncthreads();
        RSTAMP(5);
    }
    // ---- once per step: the element state and the moved mesh go back -- unless a wait timed out here or anywhere before (see k_substep_resident)
    if (t == 0 && __hip_atomic_load(r.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) lerr = 1;
    __syncthreads();
    if (lerr) return;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if ((eflags & (1u << j)) && (eflags & (1u << (8 + j)))) {
            d2 *S4 = reinterpret_cast<d2 *>(Sn) + 2 * (size_t)e[j];
            S4[0] = d2{sig[j][0], sig[j][1]}; S4[1] = d2{sig[j][2], damage[j]};
        }
    if (move_dt != 0.) {
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (nfl & (1u << (24 + i))) {
                const unsigned nf = (nfl >> (8 * i)) & 0xFFu;
                if (!(nf & NF_NEUMANN)) { s.UM[n[i]] = um[i][0]; s.UM[n[i] + Nn] = um[i][1]; }
                s.UT[n[i]] = um[i][2]; s.UT[n[i] + Nn] = um[i][3];
            }
    }
}

// The ghost nodes' mesh moves of a resident launch (FE.cpp:10543-10550): M_UM += dte * M_VT, M_UT += dte * M_VT with the velocity that arrived after
// each of the first `count` sub-steps, in sub-step order -- the additions the reference makes, from the ring the launch filled (the move of the
// last sub-step follows in k_halo_pull, with the last exchange).
__global__ void __launch_bounds__(BLOCK) k_ghost_ring_move(DevMesh m, DevState s, const double *__restrict__ gring, int NG, int count, double dt, const int *__restrict__ error) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= NG) return;
    if (*error != 0) return;  // the resident launch gave up: the ring is not complete and the own nodes were not moved either
    const int n = m.No + j, Nn = m.Nn;
    const bool free_node = !(m.nflags[n] & NF_NEUMANN);  // Neumann nodes keep M_UM (restore == skip)
    double umu = s.UM[n], umv = s.UM[n + Nn], utu = s.UT[n], utv = s.UT[n + Nn];
    // the additions are sequential (the reference's order), the loads are not: sixteen slots in flight at a time (a few hundred ghosts: latency only)
    int c = 0;
    for (; c + 16 <= count; c += 16) {
        double u[16], v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { u[k] = gring[(size_t)(c + k) * 2 * NG + j]; v[k] = gring[(size_t)(c + k) * 2 * NG + NG + j]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (free_node) { umu += dt * u[k]; umv += dt * v[k]; }
            utu += dt * u[k]; utv += dt * v[k];
        }
    }
    for (; c < count; ++c) {
        const double u = gring[(size_t)c * 2 * NG + j], v = gring[(size_t)c * 2 * NG + NG + j];
        if (free_node) { umu += dt * u; umv += dt * v; }
        utu += dt * u; utv += dt * v;
    }
    if (free_node) { s.UM[n] = umu; s.UM[n + Nn] = umv; }
    s.UT[n] = utu; s.UT[n + Nn] = utv;
}

// K14  updateIceDiagnostics(), FE.cpp:7860-7905: totals over the ice categories, principal stresses, divergence of M_VT on the mesh displaced by
// M_UM (shapeCoeff, FE.cpp:1951-1964).  One [Ne][6] row per element (D_conc, D_thick, D_snow_thick, D_sigma0, D_sigma1, D_divergence), staged through
// LDS so that the rows leave as one stream.  S4 != NULL: sigma lives in the records the sub-step loop left behind (k_pack_state's layout).
__global__ void __launch_bounds__(BLOCK) k_ice_diagnostics(DevMesh m, DevState s, int young_cat, const double *__restrict__ S4, double *__restrict__ out) {
    const int e = min(blockIdx.x * BLOCK + (int)threadIdx.x, m.Ne - 1);  // (threads past the end redo the last element: every thread reaches the barrier)
    __shared__ double rows[BLOCK * 6];
    double dc = s.conc[e], dt = s.thick[e], ds = s.snow[e];
    if (young_cat) { dc += s.cyoung[e]; dt += s.hyoung[e]; ds += s.hsyoung[e]; }
    double s0, s1, s2;
    if (S4) { s0 = S4[4 * (size_t)e]; s1 = S4[4 * (size_t)e + 1]; s2 = S4[4 * (size_t)e + 2]; }
    else { s0 = s.s0[e]; s1 = s.s1[e]; s2 = s.s2[e]; }
    double vx[3], vy[3];
    load_vertices(m, s.UM, e, vx, vy);
    const double jac = jacobian(vx, vy);
    const int n[3] = {m.t0[e], m.t1[e], m.t2[e]};
    double div = 0.;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int kp1 = (j + 1) % 3, kp2 = (j + 2) % 3;
        const double dxN = (vy[kp1] - vy[kp2]) / jac, dyN = (vx[kp2] - vx[kp1]) / jac;
        div += dxN * s.VT[n[j]] + dyN * s.VT[n[j] + m.Nn];
    }
    double *r = rows + 6 * threadIdx.x;
    r[0] = dc; r[1] = dt; r[2] = ds; r[3] = (s0 + s1) / 2.; r[4] = hypot((s0 - s1) / 2., s2); r[5] = div;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * BLOCK * 6;
    const int count = min(BLOCK, m.Ne - (int)blockIdx.x * BLOCK) * 6;
    for (int i = threadIdx.x; i < count; i += BLOCK) out[base + i] = rows[i];
}

// Deferred mesh move of the fused path: the fused kernel leaves every sub-step's velocity in a ring of
// VT buffers; every `count` sub-steps this kernel applies the same sequence of additions
// M_UM += dte*M_VT, M_UT += dte*M_VT (FE.cpp:10543-10550) for all nodes, owned and ghost -- same
// operations in the same order, but UM/UT are streamed once per `count` sub-steps instead of every one.
// the interleaved rows of k_ice_diagnostics as one array per field ([NXS_ICE_DIAG_FIELDS][Ne]): what the host's six vectors are copied from, contiguously
__global__ void __launch_bounds__(BLOCK) k_icediag_soa(int Ne, const double *__restrict__ rows, double *__restrict__ soa) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= Ne) return;
#pragma unroll
    for (int k = 0; k < NXS_ICE_DIAG_FIELDS; ++k) soa[(size_t)k * Ne + e] = rows[(size_t)e * NXS_ICE_DIAG_FIELDS + k];
}


// vt_out != NULL (the last flush of a step): the newest velocity also goes back into M_VT -- the flush has just read it
__global__ void __launch_bounds__(BLOCK) k_move_ring(DevMesh m, DevState s, VTRing ring, int first, int count, double dt, double *__restrict__ vt_out) {
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n >= m.Nn) return;
    const int Nn = m.Nn;
    const bool free_node = !(m.nflags[n] & NF_NEUMANN);  // Neumann nodes keep M_UM (restore == skip)
    double umu = s.UM[n], umv = s.UM[n + Nn], utu = s.UT[n], utv = s.UT[n + Nn];
    int sl = first;
    double lastu = 0., lastv = 0.;
    // the additions are sequential (the reference's order), the loads are not: eight slots in flight at a time
    int j = 0;
    for (; j + 8 <= count; j += 8) {
        double u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double *p = ring.slot[sl];
            u[k] = __builtin_nontemporal_load(p + n); v[k] = __builtin_nontemporal_load(p + n + Nn);
            sl = (sl + 1 == ring.R) ? 0 : sl + 1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (free_node) { umu += dt * u[k]; umv += dt * v[k]; }
            utu += dt * u[k]; utv += dt * v[k];
        }
        lastu = u[7]; lastv = v[7];
    }
    for (; j < count; ++j) {
        const double u = ring.slot[sl][n], v = ring.slot[sl][n + Nn];
        if (free_node) { umu += dt * u; umv += dt * v; }
        utu += dt * u; utv += dt * v;
        sl = (sl + 1 == ring.R) ? 0 : sl + 1;
        lastu = u; lastv = v;
    }
    if (free_node) { s.UM[n] = umu; s.UM[n + Nn] = umv; }
    s.UT[n] = utu; s.UT[n + Nn] = utv;
    if (vt_out) { vt_out[n] = lastu; vt_out[n + Nn] = lastv; }
}

// the velocity of the last sub-step back into M_VT when the ring ended elsewhere
__global__ void __launch_bounds__(BLOCK) k_pingpong_copy_back(DevMesh m, DevState s, const double *vt_src) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < 2 * m.Nn) s.VT[i] = vt_src[i];
}

// The fused sub-step kernels keep (sigma0, sigma1, sigma2, damage) of an element as one 32-byte record: two 16-byte loads and two
// stores per element and sub-step instead of four and four, one base pointer instead of four.  Everything outside the sub-step loop
// (and the C ABI) sees the four arrays M_sigma[0..2], M_damage: packed before the loop, unpacked after it.
__global__ void __launch_bounds__(BLOCK) k_pack_state(DevMesh m, DevState s, int bbm, double *__restrict__ S4) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 *S = reinterpret_cast<d2 *>(S4) + 2 * (size_t)e;
    S[0] = d2{s.s0[e], s.s1[e]};
    S[1] = d2{s.s2[e], bbm ? s.damage[e] : 0.};
}
__global__ void __launch_bounds__(BLOCK) k_unpack_state(DevMesh m, DevState s, int bbm, const double *__restrict__ S4) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *S = reinterpret_cast<const d2 *>(S4) + 2 * (size_t)e;
    const d2 a = S[0], c2 = S[1];
    s.s0[e] = a.x; s.s1[e] = a.y; s.s2[e] = c2.x;
    if (bbm) s.damage[e] = c2.y;
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
// selftest != 0: the payload is a code of (rank, entry, sequence) instead of vec; selftest == 2 publishes with the ONE release per
// launch of the in-kernel exchange (k_substep_fused<HALO>, k_smooth_halo) instead of one per block
__global__ void __launch_bounds__(BLOCK) k_halo_push(const double *__restrict__ vec, int Nn, int total, const int *__restrict__ index,
                                                     const int *__restrict__ seg_of, const int *__restrict__ offsets, IpcDev ipc,
                                                     int rank, int selftest) {
    const unsigned long long seq = *ipc.seq_push;
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    nxs_delay_at(ipc.delay, NXS_DELAY_PUSH_STORE);
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
    if (threadIdx.x == 0) { if (selftest != 2) __threadfence_system(); last = (atomicAdd(ipc.done_push, 1u) == gridDim.x - 1); }
    __syncthreads();
    if (last) {
        nxs_delay_at(ipc.delay, NXS_DELAY_PUSH_FLAG);
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
        nxs_delay_at(ipc.delay, NXS_DELAY_PULL_READ);
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

// The same sweep with updateGhosts inside (device-direct transport).  The smoother has its own mailbox with one slot per sweep
// (IpcDev::smb), so no sweep overwrites what a neighbour may still read and nothing has to be acknowledged.  Sweep j of pass E:
//   j >= 1: wait until every neighbour that still matters has published sweep j-1 (sflags >= 64 E + j); ghost neighbours of the
//           ice-free nodes are read from slot j-1 -- or from slot 0 for a direction that declared itself static;
//   every sent node is stored into the neighbours' slot j (only sweep 0 for a static direction), the last sending block raises the
//   flags; with sweep 0 it first tells each neighbour whether that direction is static this step (k_smooth_static found no node the
//   smoother can change among the nodes sent that way): such a neighbour waits for sweep 0 only, 1 wait per step instead of 50.
// The reference exchanges all 50 times (FE.cpp:10609); the values a static direction would carry are the same 50 times over.
__global__ void __launch_bounds__(BLOCK) k_smooth_static(int total, const int *__restrict__ index, const int *__restrict__ seg_of, DevMesh m, DevWork w, int *my_static) {
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= total) return;
    const int n = index[j];
    if (!((m.nflags[n] & NF_DIRICHLET) || w.node_mass[n] != 0.)) my_static[seg_of[j]] = 0;  // k_smooth's own test (FE.cpp:10589)
}

__global__ void __launch_bounds__(BLOCK) k_smooth_halo(DevMesh m, DevWork w, const double *__restrict__ src, double *__restrict__ dst, HaloFused hf, int sweep) {
    const IpcDev &ipc = hf.ipc;
    const int Nn = m.Nn, No = m.No;
    // What this block of BLOCK own nodes has to do in this sweep: smooth (it holds an ice-free node: the flag k_prep_nodes raised),
    // send (it holds sent nodes, and this is sweep 0 or some direction is not static), or -- block 0 in sweep 1 -- note what the
    // neighbours said about their directions.  Most blocks of most sweeps have nothing to do and leave at once.
    const bool has_open = w.open_blk[blockIdx.x] != 0;
    const int sr = hf.send_block_rank[blockIdx.x];
    bool any_dynamic = sweep == 0;
    for (int k = 0; k < ipc.ns && !any_dynamic; ++k) any_dynamic = !ipc.my_static[k];
    const bool publisher = (sr >= 0 || (hf.n_send_blocks == 0 && blockIdx.x == 0)) && any_dynamic;
    const bool scribe = sweep == 1 && blockIdx.x == 0;
    if (!has_open && !publisher && !scribe) return;
    const unsigned long long E = *ipc.epoch;
    if (sweep >= 1 && (has_open || scribe)) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            bool ok = true;
            for (int k = 0; k < ipc.nr && ok; ++k) {
                if (sweep >= 2 && ipc.peer_static[k]) continue;  // its values are in slot 0 for good
                while (__hip_atomic_load(ipc.sflags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 64ull * E + (unsigned long long)sweep) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__hip_atomic_load(ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(ipc.error, 4); break; }  // 10 s
                }
            }
            if (scribe && ok)  // the neighbours' words came with their sweep 0; sweeps >= 2 read this cached copy
                for (int k = 0; k < ipc.nr; ++k)
                    ipc.peer_static[k] = __hip_atomic_load(ipc.sstatic + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ((E << 1) | 1ull);
            nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_READ);
        }
        __syncthreads();
    }
    if (publisher) nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_STORE);
    const int n = blockIdx.x * BLOCK + threadIdx.x;
    if (n < No && (has_open || publisher)) {
        double u = src[n], v = src[n + Nn];
        if (has_open && !((m.nflags[n] & NF_DIRICHLET) || w.node_mass[n] != 0.)) {
            u = 0.; v = 0.;
            const int num_neighbours = m.n2n_cnt[n];
            for (int j = 0; j < num_neighbours; ++j) {  // Q8: bamg row order
                const int nni = m.n2n[(size_t)j * Nn + n];
                if (sweep >= 1 && nni >= No) {
                    const int slot = (sweep >= 2 && ipc.peer_static[hf.ghost_k[nni - No]]) ? 0 : sweep - 1;
                    const double *g = ipc.smb + (size_t)slot * 2 * (size_t)ipc.tr + hf.ghost_off[nni - No];
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
        if (publisher)
            for (int q = hf.send_ptr[n]; q < hf.send_ptr[n + 1]; ++q) {
                const int k = hf.send_k[q];
                if (sweep >= 1 && ipc.my_static[k]) continue;  // slot 0 already holds this value, and the neighbour knows
                double *d = ipc.peer_smb[k] + (long long)sweep * ipc.peer_parity_stride[k] + hf.send_pos[q];
                sys_store(d, u);
                sys_store(d + (hf.send_off[k + 1] - hf.send_off[k]), v);
            }
    }
    // Publishing as in the sub-step kernel: every wave drains its own stores, the barrier collects the waves, one lane counts the
    // block in, the last one raises the flags behind one release.  Only blocks that can send take a ticket, and the tickets are
    // two-level (16 group counters, then one): atomics on one address are served ~10 ns apart.  After sweep 0 a rank whose
    // directions are all static publishes nothing at all.
    if (!publisher) return;
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
        nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_FLAG);
        if (sweep == 0)
            for (int k = 0; k < ipc.ns; ++k)
                __hip_atomic_store(ipc.peer_sstatic[k], (E << 1) | (unsigned long long)(ipc.my_static[k] != 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();  // the one release of the launch: data and words before the flags
        for (int k = 0; k < ipc.ns; ++k)
            if (sweep == 0 || !ipc.my_static[k])
                __hip_atomic_store(ipc.peer_sflag[k], 64ull * E + (unsigned long long)sweep + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hf.done_all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ALL 50 sweeps with updateGhosts inside in ONE launch (round 5).  Fifty launches of k_smooth_halo cost a rank 0.12-0.16 ms per step -- 10-13 % of a
// rank of eight's step -- although most of them find nothing to do: a launch that returns at once still takes its 2.4 us in the graph.  Here G PERSISTENT workgroups (at
// most 128: all resident whatever else shares the device) walk the rank's blocks of BLOCK own nodes sweep after sweep -- workgroup p takes the blocks p, p + G, ... --
// and meet at a barrier of their own between the sweeps (two-level tickets, a generation word; what a sweep stores for the other workgroups is written through and read
// past the L1: the guide's drained-sc1 hand-off).  The exchange between RANKS is k_smooth_halo's, statement for statement: the workgroup whose ticket completes sweep j
// raises the flags for it (that workgroup cannot reach the ticket of sweep j + 1 before it has done so: the flags count upwards), blocks with an ice-free node wait for
// the neighbours' sweep j - 1 before they read a ghost.  A rank without an ice-free own node whose directions are all static is done after sweep 0.  Same operations on
// the same values in the same (bamg row) order: the same bits as 50 launches.
__global__ void __launch_bounds__(BLOCK) k_smooth_persist(DevMesh m, DevWork w, double *__restrict__ bufA, double *__restrict__ bufB, HaloFused hf, int nblk) {
    const IpcDev &ipc = hf.ipc;
    const int Nn = m.Nn, No = m.No, G = (int)gridDim.x, pid = (int)blockIdx.x, t = (int)threadIdx.x;
    const unsigned long long E = *ipc.epoch;
    unsigned int *gen = hf.done_all + 32u * 17u;
    const unsigned g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read before this workgroup's first ticket: nobody can have raised it yet)
    bool any_dynamic = false;
    for (int k = 0; k < ipc.ns; ++k) any_dynamic |= !ipc.my_static[k];
    int mine_open = 0, rank_open = 0;
    for (int b = t; b < nblk; b += BLOCK) { const int o = w.open_blk[b] != 0; rank_open |= o; if (b % G == pid) mine_open |= o; }
    rank_open = __syncthreads_or(rank_open);
    mine_open = __syncthreads_or(mine_open);
    const int sweeps = (rank_open || any_dynamic) ? NXS_SMOOTH_SWEEPS : 1;
    __shared__ int s_err;
    if (t == 0) s_err = 0;
    __syncthreads();
    // the neighbours' sweep `want - 1` has landed (flags >= 64 E + want) for every direction that still matters; the scribe notes what the neighbours said about theirs
    auto wait_peers = [&](const int sweep, const bool scribe) {
        if (t == 0) {
            const long long t0 = wall_clock64();
            bool ok = true;
            for (int k = 0; k < ipc.nr && ok; ++k) {
                if (sweep >= 2 && __hip_atomic_load(ipc.peer_static + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;  // its values are in slot 0 for good
                while (__hip_atomic_load(ipc.sflags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 64ull * E + (unsigned long long)sweep) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__hip_atomic_load(ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = false; break; }
                    if (wall_clock64() - t0 > 1000000000ll) { ok = false; atomicExch(ipc.error, 4); break; }  // 10 s
                }
            }
            if (scribe && ok)
                for (int k = 0; k < ipc.nr; ++k)
                    __hip_atomic_store(ipc.peer_static + k, (int)(__hip_atomic_load(ipc.sstatic + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ((E << 1) | 1ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!ok) s_err = 1;
            nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_READ);
        }
        __syncthreads();
    };
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        const double *src = (sweep & 1) ? bufB : bufA;
        double *dst = (sweep & 1) ? bufA : bufB;
        const bool scribe = sweep == 1 && pid == 0;
        if (sweep >= 1 && (mine_open || scribe)) wait_peers(sweep, scribe);
        if (s_err) break;
        const bool publishing = sweep == 0 || any_dynamic;
        if (publishing) nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_STORE);
        for (int b = pid; b < nblk; b += G) {
            const bool has_open = w.open_blk[b] != 0, publisher = publishing && hf.send_block_rank[b] >= 0;
            if (!has_open && !publisher) continue;
            const int n = b * BLOCK + t;
            if (n >= No) continue;
            // (sweeps >= 1 read what other workgroups of this launch have written: past the L1)
            double u = sweep ? ld_agent(src + n) : src[n], v = sweep ? ld_agent(src + n + Nn) : src[n + Nn];
            if (has_open && !((m.nflags[n] & NF_DIRICHLET) || w.node_mass[n] != 0.)) {
                u = 0.; v = 0.;
                const int num_neighbours = m.n2n_cnt[n];
                for (int j = 0; j < num_neighbours; ++j) {  // Q8: bamg row order
                    const int nni = m.n2n[(size_t)j * Nn + n];
                    if (sweep >= 1 && nni >= No) {
                        const int slot = (sweep >= 2 && __hip_atomic_load(ipc.peer_static + hf.ghost_k[nni - No], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 0 : sweep - 1;
                        const double *g = ipc.smb + (size_t)slot * 2 * (size_t)ipc.tr + hf.ghost_off[nni - No];
                        u += sys_load(g);
                        v += sys_load(g + hf.ghost_srl[nni - No]);
                    } else if (sweep) {
                        u += ld_agent(src + nni);
                        v += ld_agent(src + nni + Nn);
                    } else {
                        u += src[nni];
                        v += src[nni + Nn];
                    }
                }
                u /= num_neighbours;
                v /= num_neighbours;
                st_agent(dst + n, u);
                st_agent(dst + n + Nn, v);
            }
            if (publisher)
                for (int q = hf.send_ptr[n]; q < hf.send_ptr[n + 1]; ++q) {
                    const int k = hf.send_k[q];
                    if (sweep >= 1 && ipc.my_static[k]) continue;  // slot 0 already holds this value, and the neighbour knows
                    double *d = ipc.peer_smb[k] + (long long)sweep * ipc.peer_parity_stride[k] + hf.send_pos[q];
                    sys_store(d, u);
                    sys_store(d + (hf.send_off[k + 1] - hf.send_off[k]), v);
                }
        }
        // the sweep's barrier: every wave drains its stores, the workgroup's barrier collects the waves, one lane takes the ticket (two-level: 16 group counters, then one)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const unsigned want = g0 + (unsigned)sweep + 1u;
            const unsigned int total = (unsigned)G, g = (unsigned)pid % 16u, members = (total - g + 15u) / 16u;
            bool last = false;
            if (__hip_atomic_fetch_add(hf.done_all + 32u * (g + 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
                __hip_atomic_store(hf.done_all + 32u * (g + 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int groups = total < 16u ? total : 16u;
                last = __hip_atomic_fetch_add(hf.done_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
            }
            if (last) {
                __hip_atomic_store(hf.done_all, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gen, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // this rank's workgroups go on; the flags for the neighbour ranks follow
                if (publishing) {
                    nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_FLAG);
                    if (sweep == 0)
                        for (int k = 0; k < ipc.ns; ++k)
                            __hip_atomic_store(ipc.peer_sstatic[k], (E << 1) | (unsigned long long)(ipc.my_static[k] != 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __threadfence_system();  // the one release of the sweep: data and words before the flags
                    for (int k = 0; k < ipc.ns; ++k)
                        if (sweep == 0 || !ipc.my_static[k])
                            __hip_atomic_store(ipc.peer_sflag[k], 64ull * E + (unsigned long long)sweep + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else {
                const long long t0 = wall_clock64();
                while ((int)(__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 1000000000ll) { s_err = 1; atomicExch(ipc.error, 4); break; }  // 10 s: a workgroup of this launch never arrived
                }
            }
        }
        __syncthreads();
        if (s_err) break;
    }
    // a rank that was done after sweep 0 still notes what the neighbours said about their directions (k_smooth_pull reads it)
    if (sweeps == 1 && pid == 0 && !s_err) wait_peers(1, true);
}

// After the last sweep: the ghosts land in the array (FE.cpp:10609 does it after every sweep) from the last slot -- slot 0 for a
// static direction -- and the pass is closed: epoch + 1, every direction presumed static again until k_smooth_static says otherwise.
__global__ void __launch_bounds__(BLOCK) k_smooth_pull(double *__restrict__ vec, int Nn, int total, const int *__restrict__ index,
                                                       const int *__restrict__ seg_of, const int *__restrict__ offsets, IpcDev ipc) {
    const unsigned long long E = *ipc.epoch;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        const long long t0 = wall_clock64();
        for (int k = 0; k < ipc.nr && ok; ++k) {
            if (ipc.peer_static[k]) continue;
            while (__hip_atomic_load(ipc.sflags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < 64ull * E + (unsigned long long)NXS_SMOOTH_SWEEPS) {
                __builtin_amdgcn_s_sleep(8);
                if (__hip_atomic_load(ipc.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > 1000000000ll) { ok = 0; atomicExch(ipc.error, 4); break; }  // 10 s
            }
        }
        nxs_delay_at(ipc.delay, NXS_DELAY_SMOOTH_PULL_READ);
    }
    __syncthreads();
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (ok && j < total) {
        const int k = seg_of[j];
        const int off = offsets[k], srl = offsets[k + 1] - off;
        const int slot = ipc.peer_static[k] ? 0 : NXS_SMOOTH_SWEEPS - 1;
        const double *src = ipc.smb + (size_t)slot * 2 * (size_t)ipc.tr + 2 * (size_t)off;
        const int n = index[j];
        vec[n] = sys_load(src + (j - off));
        vec[n + Nn] = sys_load(src + (j - off) + srl);
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ipc.done_pull, 1u) == gridDim.x - 1) {
        *ipc.done_pull = 0u;
        *ipc.epoch = E + 1ull;
        for (int k = 0; k < ipc.ns; ++k) ipc.my_static[k] = 1;
    }
}

// KS Jacobi sweeps of the open-water smoother in one launch, on the patches of k_substep_multi: sweep j recomputes the ice-free
// nodes of N_(KS-j-1) from their neighbours in N_(KS-j) -- the rings redo what the neighbouring patches do, same operations in
// the same (bamg row) order, same bits as KS launches of k_smooth -- and only the own nodes are written back.  A patch without an
// ice-free own node has nothing to write and returns at once: with no open water the 50 sweeps cost 13 empty launches instead of 50.
template <int T>
__global__ void __launch_bounds__(T) k_smooth_multi(DevMesh m, DevPatches2 pp, DevWork w, const double *__restrict__ src, double *__restrict__ dst, int KS) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int NDm = pp.NDmax, D = pp.D, Nn = m.Nn, t = threadIdx.x, blk = blockIdx.x;
    double *au = lds, *av = au + NDm, *bu = av + NDm, *bv = bu + NDm;
    unsigned char *open = reinterpret_cast<unsigned char *>(bv + NDm);  // [NSmax]
    const int *ncnt = pp.ncnt + (size_t)blk * (D + 1);
    const int *pn = pp.pnodes + (size_t)blk * NDm;
    const unsigned short *nb = pp.pnbr + (size_t)blk * pp.W2 * pp.NSmax;
    if (pp.own_is_block && !w.open_blk[blk]) return;  // no ice-free own node (flag raised by k_prep_nodes): nothing to write
    const int nO = ncnt[0], nS = ncnt[KS - 1], nK = ncnt[KS];
    int any = 0;
    for (int i = t; i < nO; i += T) {  // the own nodes first: most patches stop here
        const int g = pn[i];
        const unsigned char o = !((m.nflags[g] & NF_DIRICHLET) || w.node_mass[g] != 0.);  // k_smooth's test, FE.cpp:10589
        open[i] = o;
        any |= o;
    }
    if (!__syncthreads_or(any)) return;
    for (int i = nO + t; i < nS; i += T) {
        const int g = pn[i];
        open[i] = !((m.nflags[g] & NF_DIRICHLET) || w.node_mass[g] != 0.);
    }
    __syncthreads();
    for (int i = t; i < nK; i += T) { const int g = pn[i]; au[i] = src[g]; av[i] = src[g + Nn]; }
    __syncthreads();
    for (int j = 0; j < KS; ++j) {
        const int nn = ncnt[KS - 1 - j];
        for (int i = t; i < nn; i += T) {
            double u = au[i], v = av[i];
            if (open[i]) {
                const int num_neighbours = m.n2n_cnt[pn[i]];
                u = 0.; v = 0.;
                for (int k = 0; k < num_neighbours; ++k) {  // Q8: bamg row order
                    const int sl = nb[(size_t)k * pp.NSmax + i];
                    u += au[sl];
                    v += av[sl];
                }
                u /= num_neighbours;
                v /= num_neighbours;
            }
            bu[i] = u; bv[i] = v;
        }
        __syncthreads();
        double *x = au; au = bu; bu = x;
        x = av; av = bv; bv = x;
    }
    for (int i = t; i < nO; i += T)
        if (open[i]) { const int g = pn[i]; dst[g] = au[i]; dst[g + Nn] = av[i]; }
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
// REC: M_sigma lives in the records the sub-step loop left in S4a (see k_pack_state); the arrays are brought up to date on demand
template <bool REC>
__global__ void __launch_bounds__(BLOCK) k_update(DevMesh m, DevState s, DevWork w, DevParams p) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= m.Ne) return;
    if (e == 0) w.shape_range[0] = 0;   // the step's sub-steps are over: the next step's prep kernels judge its own coordinates
    if (e < (m.Nn + BLOCK - 1) / BLOCK) w.open_blk[e] = 0;   // ... and raise the open-water flags of their node blocks anew (k_prep_fused; k_prep_elements lowers them itself)
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
        if (REC) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *S = reinterpret_cast<d2 *>(s.S4a) + 2 * (size_t)e;
            d2 a = S[0], c2 = S[1];
            a.x *= surf_ratio; a.y *= surf_ratio; c2.x *= surf_ratio;
            S[0] = a; S[1] = c2;
        } else {
            s.s0[e] *= surf_ratio; s.s1[e] *= surf_ratio; s.s2[e] *= surf_ratio;
        }
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

__global__ void __launch_bounds__(BLOCK) k_check_fields(DevMesh m, DevState s, DevParams p, int *crash, int damage_in_records) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    bool bad = false;
    if (i < m.Ne) {
        bad |= bad_range(s.thick[i], 0., 50.);
        bad |= bad_range(s.snow[i], 0., 10.);
        bad |= bad_range(s.conc[i], 0., 1.);
        bad |= bad_range(damage_in_records ? s.S4a[4 * (size_t)i + 3] : s.damage[i], 0., 1.);
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


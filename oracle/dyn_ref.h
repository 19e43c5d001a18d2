/*
 * dyn_ref.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A line-faithful, single-threaded, fp64 restatement in plain C of the reference dynamics path of
 * nansencenter/nextsim (FiniteElement::explicitSolve + update and helpers).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (libnxsdyn.so) never links, loads or calls it.
 *
 * Parity status: the reference ships NO golden vectors / known-answer tests for this path and its
 * own translation units cannot be built here (Boost, Gmsh, NetCDF absent) -- see DESIGN.md.  The
 * oracle is pinned by (1) hand-computed single-triangle known answers, (2) the reference's own
 * runtime invariants, (3) the real contrib/bamg compiled into oracle/_ref for the connectivity
 * tables.  For the time-stepping arithmetic itself: PARITY UNPINNED beyond (1)-(2).
 *
 * The struct types are the ABI PODs of include/nxs_dyn.h so that a test fills one set of structs
 * for both sides.  "FE.cpp" = /root/reference/model/finiteelement.cpp.
 */
#ifndef DYN_REF_H
#define DYN_REF_H

#include "../include/nxs_dyn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* scratch + side outputs of explicitSolve()/update() (locals and D_* members of the reference) */
typedef struct ref_work {
    int32_t Nn, Ne;
    double Dunit[9];       /* M_Dunit, FE.cpp:1491-1507 */
    double *delta_x;       /* [Ne] M_delta_x */
    double *surface;       /* [Ne] M_surface */
    double *shape_coeff;   /* [6*Ne] M_shape_coeff[e][0..5] */
    double *B0T;           /* [18*Ne] M_B0T[e][0..17] */
    double *element_mass;  /* [Ne] */
    double *rlmass_matrix; /* [Nn] */
    double *node_mass;     /* [Nn] */
    double *C_bu;          /* [Nn] */
    double *grad_ssh;      /* [2Nn] */
    double *grad_terms;    /* [2Nn] */
    double *fcor;          /* [Nn] */
    double *VTM;           /* [2Nn] */
    double *tmp;           /* [2Nn] copies (UM_P, u) */
    double *D_tau_a;       /* [2Nn] */
    double *D_tau_w;       /* [2Nn] */
    double *D_del_ci_ridge_myi; /* [Ne] */
    /* Branch trace of updateSigmaDamage (NULL = off; ref_work_enable_trace): per element 4 words
     *   [0] hash of the branch taken at every sub-step so far: h = h*0x9E3779B97F4A7C15 + code + 1 with
     *       code 0 = no damage increment, 1 = the 0 < dcrit < 1 branch (FE.cpp:4229), 2 = skipped, conc <= 0.1 (FE.cpp:4151)
     *   [1] number of sub-steps that took the damage branch
     *   [2] bit 0: |dcrit - 1| < 1e-9 at some sub-step, bit 1: |conc - 0.1| < 1e-12 at some sub-step (SURVEY 8d's
     *       threshold-flip set), bit 2: skipped at some sub-step
     *   [3] sub-steps seen
     * The device keeps the same record (option "trace_branches"): two implementations whose hashes agree for an element took the
     * same branches at every sub-step. */
    uint64_t *trace;       /* [4*Ne] */
} ref_work;

typedef void (*ref_ghost_fn)(void *ctx, double *nodal_vec);

ref_work *ref_work_create(int32_t Nn, int32_t Ne);
void ref_work_destroy(ref_work *w);
int ref_work_enable_trace(ref_work *w);   /* allocates and zeroes w->trace; 0 on success */

void ref_default_params(nxs_dyn_params *p);
void ref_physical_constants(double out[8]); /* rhoi, rhow, rhos, rhoa, gravity, omega, PI, days_in_sec as this file uses them */

/* phases of explicitSolve(), split where the reference calls updateGhosts() */
void ref_prep(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
              const nxs_dyn_forcing *f, ref_work *w);                  /* FE.cpp:10213-10418 */
void ref_update_sigma_damage(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                             ref_work *w, double dt);                   /* FE.cpp:4137-4260 */
void ref_update_sigma_vp(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                         ref_work *w, double ralpha1, double ralpha2);  /* FE.cpp:10649-10699 */
void ref_substep_solve(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                       const nxs_dyn_forcing *f, ref_work *w);         /* FE.cpp:10425-10530 */
void ref_move_mesh(const nxs_dyn_mesh *m, nxs_dyn_state *s, ref_work *w, double dt); /* :10539-10553 */
void ref_smoother_sweep(const nxs_dyn_mesh *m, nxs_dyn_state *s, ref_work *w);       /* :10582-10608 */
void ref_ow_tail(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                 const nxs_dyn_forcing *f, ref_work *w);               /* FE.cpp:10613-10640 */

/* whole functions */
void ref_explicit_solve(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                        const nxs_dyn_forcing *f, ref_work *w, ref_ghost_fn ghosts, void *ctx);
void ref_update(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s, ref_work *w);
void ref_free_drift(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                    const nxs_dyn_forcing *f);                         /* FE.cpp:10140-10176 */
/* step(): FE.cpp:8197-8214 */
void ref_step(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
              const nxs_dyn_forcing *f, ref_work *w, ref_ghost_fn ghosts, void *ctx);

int ref_check_regridding(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s,
                         double *min_angle, int32_t *flip);            /* FE.cpp:8298-8309 */
int ref_check_fields_fast(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s); /* :14536 */

/* updateIceDiagnostics(), FE.cpp:7860-7905 (without D_tsurf / FSD); NULL outputs are skipped */
void ref_ice_diagnostics(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s, double *D_conc, double *D_thick,
                         double *D_snow_thick, double *D_sigma0, double *D_sigma1, double *D_divergence);

/* updateGhosts() halves (FE.cpp:13963-13996): pack what I send to neighbour k / unpack what k sent */
void ref_ghosts_pack(const nxs_dyn_halo *h, int32_t Nn, const double *vec, int k, double *buf);
void ref_ghosts_unpack(const nxs_dyn_halo *h, int32_t Nn, double *vec, int k, const double *buf);

/* P in-process ranks in lock-step on `nthreads` threads (thread t runs ranks t, t + nthreads, ...): step() of every rank with every
 * updateGhosts(M_VT) a shared-memory exchange -- the CPU analogue of the reference's MPI run (bench.py's cpu_baseline) */
int ref_multirank_steps(int nranks, const nxs_dyn_mesh *const *m, const nxs_dyn_params *p, nxs_dyn_state *const *s,
                        const nxs_dyn_forcing *const *f, ref_work *const *w, const nxs_dyn_halo *const *h, int nsteps, int nthreads);

/* The same run as a persistent context that uses the host like an MPI run: threads kept across calls and pinned (pin != 0: physical cores first, the sockets
 * in turn), every partition's arrays copied -- first touched -- by the thread that owns it, spin barriers.  ref_mr_run: nsteps lock-step steps (the part
 * bench.py times); ref_mr_destroy(copy_back != 0) writes the state back into the caller's arrays.  Bit for bit ref_multirank_steps. */
typedef struct ref_mr_ctx ref_mr_ctx;
ref_mr_ctx *ref_mr_create(int nranks, const nxs_dyn_mesh *const *m, const nxs_dyn_params *p, nxs_dyn_state *const *s, const nxs_dyn_forcing *const *f,
                          const nxs_dyn_halo *const *h, int nthreads, int pin);
int ref_mr_configure(ref_mr_ctx *c, int barrier_kind /* 0 spin, 1 sleeping (pthread_barrier_t) */, int pin);   /* between runs */
int ref_mr_run(ref_mr_ctx *c, int nsteps);
int ref_mr_info(const ref_mr_ctx *c, int *nthreads, int *sockets_used, int *cpus, int cap);
void ref_mr_destroy(ref_mr_ctx *c, int copy_back);

/* restatement of Mesh::WriteMesh's two connectivity tables (contrib/bamg/src/Mesh.cpp:514-543, 798-865) */
int ref_mesh_connectivity(const int32_t *indices, int32_t num_nodes, int32_t num_elements,
                          int32_t *nec_width, double *nec, int32_t *nc_width, double *nc);

#ifdef __cplusplus
}
#endif
#endif

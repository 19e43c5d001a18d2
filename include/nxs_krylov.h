/*
 * nxs_krylov.h -- EXTENSION.  NOT part of the reference's live path.
 *
 * BASELINE.json's north_star names an "element-stiffness assembly (graph-coloured scatter into CSR) and
 * the Krylov SpMV/axpy/dot that replace the PETSc KSP solve".  The reference at this commit has neither
 * (explicit momentum solver; PETSc only survives in the un-buildable research/laplacian demo, SURVEY.md
 * section 0 F1/F2).  This header provides those kernels as a clearly separate extension so that an implicit
 * solver can be grown on them; PARITY UNPINNED.  The only in-tree known answer is the P1 Laplacian of
 * research/laplacian.cpp:149-266 (element matrix m_jk = (dy_j dy_k + dx_j dx_k)/(4A), load f(x_b)*A/3,
 * homogeneous Dirichlet rows) with exact solution sin(pi x) sin(pi y) (research/laplacian.cpp:348, 429);
 * tests/test_krylov.py checks the discretisation error and its O(h^2) decay against it.
 */
#ifndef NXS_KRYLOV_H
#define NXS_KRYLOV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NXS_KRYLOV_API __attribute__((visibility("default")))
#else
#define NXS_KRYLOV_API
#endif

/* P1 node-to-node sparsity in CSR (one dof per node; columns ascending, diagonal included), the scalar
 * analogue of FiniteElement::createGraph (FE.cpp:13858-13927).  Pass colidx == NULL to get *nnz only. */
NXS_KRYLOV_API int nxs_fem_csr_pattern(const int32_t *indices, int32_t num_nodes, int32_t num_elements, int32_t *rowptr,
                                       int32_t *colidx, int64_t *nnz);

/* Greedy element colouring: no two elements of one colour share a node (host).  colour[e] in [0, *ncolours). */
NXS_KRYLOV_API int nxs_fem_colour_elements(const int32_t *indices, int32_t num_nodes, int32_t num_elements, int32_t *colour,
                                           int32_t *ncolours);

/* -div grad u = f on the mesh, u = 0 on `dirichlet` nodes: graph-coloured scatter of the 3x3 element matrices
 * into the (sliced-ELLPACK) matrix without atomics -- one launch, a workgroup per patch of 128 rows builds its
 * contiguous piece of the matrix in LDS, the element colours taking turns -- then Jacobi-preconditioned CG
 * (SpMV with the dot (p, Ap) fused, one fused x/r/z update carrying (r, z) and (r, r), one p update; dots by
 * wave shuffles + a fixed-order sum of per-block partials).  f_elem[e] = source at the barycentre of element e.
 * Stops at ||r|| <= rtol*||b|| or max_iter.  *ms_assembly: assembly + Dirichlet rows, second of two passes.
 * Returns 0 / negative NXS_ERR_* (no HIP device: -2). */
NXS_KRYLOV_API int nxs_fem_poisson_solve(const int32_t *indices, const double *x, const double *y, int32_t num_nodes,
                                         int32_t num_elements, const uint8_t *dirichlet, const double *f_elem, double *u,
                                         double rtol, int32_t max_iter, int32_t device, int32_t *iterations,
                                         double *rel_residual, double *ms_assembly, double *ms_solve);

/* A x = b for a general CSR matrix (columns in any order, a non-zero diagonal in every row), x0 = 0, Jacobi
 * preconditioner: NXS_KRYLOV_CG for symmetric positive definite A, NXS_KRYLOV_BICGSTAB for non-symmetric A (a momentum
 * matrix with Coriolis terms is).  SpMV / fused vector updates / deterministic dots, scalars kept on the device, the
 * residual read back every 10 iterations.  Stops at ||r|| <= rtol*||b|| or max_iter. */
enum { NXS_KRYLOV_CG = 0, NXS_KRYLOV_BICGSTAB = 1 };
NXS_KRYLOV_API int nxs_krylov_solve(int32_t n, const int32_t *rowptr, const int32_t *colidx, const double *val, const double *b,
                                    double *x, int32_t method, double rtol, int32_t max_iter, int32_t device, int32_t *iterations,
                                    double *rel_residual, double *ms_solve);

/* ---- the same solvers behind a handle: the matrix stays resident, and the rows may be distributed over ranks ----
 * One rank per GPU owns a block of rows.  In a rank's local numbering its own rows come first ([0, n_rows)) and the
 * columns owned by other ranks behind them ([n_rows, n_cols)) -- the owned-nodes-then-ghosts order of the mesh
 * partition (core/src/gmshmesh.cpp:1168-1169), so that the halo lists of nxs_dyn_halo (one dof per entry here) describe
 * the exchange of the SpMV operand.  Per iteration: one neighbour exchange before every SpMV and one all-reduce
 * of 1-2 doubles behind every dot -- over RCCL (nxs_krylov_comm_init: ncclSend/ncclRecv + ncclAllReduce on the
 * solver's stream, no host synchronisation between convergence checks) or through the caller's communicator
 * (nxs_krylov_set_comm_fns, host staged; what the tests use on a one-GPU box).  The matrix is stored as sliced
 * ELLPACK (64-row slices); dots are deterministic on a rank (fixed-order two-stage sums). */
#include "nxs_dyn.h" /* nxs_dyn_halo, error codes */
typedef struct nxs_krylov_handle nxs_krylov_handle;
NXS_KRYLOV_API int nxs_krylov_create(int32_t device, nxs_krylov_handle **out);
NXS_KRYLOV_API void nxs_krylov_destroy(nxs_krylov_handle *h);
/* rows [0, n_rows) in CSR with local column indices < n_cols (n_cols == n_rows on a single rank) */
NXS_KRYLOV_API int nxs_krylov_set_matrix(nxs_krylov_handle *h, int32_t n_rows, int32_t n_cols, const int32_t *rowptr,
                                         const int32_t *colidx, const double *val);
/* send_index: owned rows whose operand value a neighbour needs; recv_index: ghost columns, filled from their owners */
NXS_KRYLOV_API int nxs_krylov_set_halo(nxs_krylov_handle *h, const nxs_dyn_halo *halo);
/* RCCL communicator from the 128-byte ncclUniqueId of nxs_dyn_comm_unique_id (same on every rank) */
NXS_KRYLOV_API int nxs_krylov_comm_init(nxs_krylov_handle *h, const void *id128, int32_t rank, int32_t nranks);
/* how many ncclAllReduce calls and grouped send/recv exchanges the handle has issued (a communicator of one rank still reduces) */
NXS_KRYLOV_API int nxs_krylov_comm_stats(const nxs_krylov_handle *h, int64_t *rccl_allreduces, int64_t *rccl_exchanges);
/* the caller's communicator instead: exchange(send, recv) moves the packed operand values (layout of the halo lists,
 * one double per entry), allreduce(vals, n) sums n doubles over the ranks in place; both return 0 on success */
typedef int (*nxs_krylov_exchange_fn)(void *user, const double *send, double *recv);
typedef int (*nxs_krylov_allreduce_fn)(void *user, double *vals, int32_t n);
NXS_KRYLOV_API int nxs_krylov_set_comm_fns(nxs_krylov_handle *h, nxs_krylov_exchange_fn exchange_fn,
                                           nxs_krylov_allreduce_fn allreduce_fn, void *user);
/* out = A in on the owned rows (in, out: host, n_rows each; the ghost values are exchanged first); repeated `reps`
 * times for timing, *ms_per_spmv = average of the launches after one warm-up */
NXS_KRYLOV_API int nxs_krylov_spmv(nxs_krylov_handle *h, const double *in, double *out, int32_t reps, double *ms_per_spmv);
/* A x = b, x0 = 0, as nxs_krylov_solve; b, x: host, the owned rows; the residual norm is the global one */
NXS_KRYLOV_API int nxs_krylov_run(nxs_krylov_handle *h, const double *b, double *x, int32_t method, double rtol, int32_t max_iter,
                                  int32_t *iterations, double *rel_residual, double *ms_solve);
/* sizes of the resident matrix; *spmv_bytes = algorithmic bytes of one SpMV = 12 B per non-zero (value + column)
 * + 16 B per row (operand read, result write) */
NXS_KRYLOV_API int nxs_krylov_info(const nxs_krylov_handle *h, int64_t *nnz, int64_t *stored_entries, int64_t *spmv_bytes);

NXS_KRYLOV_API const char *nxs_krylov_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

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

/* -div grad u = f on the mesh, u = 0 on `dirichlet` nodes: colour-by-colour scatter of the 3x3 element
 * matrices into CSR (no atomics), then Jacobi-preconditioned CG (SpMV / axpy / dot kernels, dot by wave
 * shuffles + a deterministic two-stage reduction).  f_elem[e] = source at the barycentre of element e.
 * Stops at ||r|| <= rtol*||b|| or max_iter.  Returns 0 / negative NXS_ERR_* (no HIP device: -2). */
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

NXS_KRYLOV_API const char *nxs_krylov_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

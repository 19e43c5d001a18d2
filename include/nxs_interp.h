/*
 * nxs_interp.h -- C ABI of the mesh-to-mesh interpolation used at regrid (SURVEY.md section 8f, row N1;
 * BASELINE config 5).  Replaces the root-serial call
 *
 *   InterpFromMeshToMesh2dx(&interp_out, &mesh_prev.indexTr()[0], &coordX[0], &coordY[0], numNodes, numTriangles,
 *                           &interp_in[0], numNodes, nb_var, &new_coordX[0], &new_coordY[0], new_numNodes, false);
 *
 * of FiniteElement::interpFields (FE.cpp:3131-3139; contrib/bamg/src/InterpFromMeshToMesh2dx.cpp:17-179)
 * with a HIP gather kernel: exact integer point location (bamg's own integer coordinates and
 * determinants, contrib/bamg/src/Mesh.cpp:3441-3468, 3688-3690, include/det.h:8-12) through a uniform
 * bucket grid, then the barycentric (nodal data) or piecewise-constant (element data) gather.
 *
 * Semantics, relative to the reference:
 *   - a target point inside a triangle of the data mesh gets exactly the reference's value: area
 *     coordinates are ratios of the same 64-bit integer determinants, combined in the same order;
 *   - isdefault != 0: points outside the data mesh get defaultvalue (as the reference);
 *   - isdefault == 0 (the regrid call): the reference locates its points in bamg's RECONSTRUCTED mesh -- the data mesh plus
 *     triangles that fill its holes and the concave parts of its boundary up to the convex hull (Mesh.cpp:3135-3440) -- and
 *     projects points beyond the hull on a hull edge (CloseBoundaryEdge, Mesh.cpp:4590-4627).  The same here: the fill
 *     triangles are rebuilt on the host (constrained Delaunay triangulation of every pocket and hole with exact integer
 *     predicates, csrc/nxs_hull.inl: the same triangles as bamg's, checked against the real bamg), numbered behind the
 *     mesh's, and the hull projection is CloseBoundaryEdge's.  A point beyond the hull whose hull edge belongs to a mesh
 *     triangle gets the reference's bits; inside a fill triangle the three products of the P1 sum are the reference's but may
 *     be added in a rotated order (bamg's vertex order inside a fill triangle depends on its insertion history): <= 1 ulp.
 *     Element data outside the mesh (where the reference stops with an error) and meshes the completion does not cover
 *     (several components, pinched boundaries) take the nearest boundary edge instead; nxs_interp_last_info says how many
 *     points went which way.
 */
#ifndef NXS_INTERP_H
#define NXS_INTERP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NXS_INTERP_API __attribute__((visibility("default")))
#else
#define NXS_INTERP_API
#endif

/* Same argument meaning and layout as InterpFromMeshToMesh2dx: index_data 1-based [3*nels_data];
 * data [M_data][N_data] row-major with M_data == nods_data (P1) or nels_data (P0);
 * data_interp [N_interp][N_data] is CALLER-allocated here (the reference allocates it).
 * Returns 0, or a negative NXS_ERR_* code of nxs_dyn.h (no HIP device: -2, never a CPU fallback).
 * kernel_ms (may be NULL) receives the device time of the gather kernel alone. */
NXS_INTERP_API int nxs_interp_mesh_to_mesh_2d(double *data_interp, const int32_t *index_data, const double *x_data,
                                              const double *y_data, int32_t nods_data, int32_t nels_data, const double *data,
                                              int32_t M_data, int32_t N_data, const double *x_interp, const double *y_interp,
                                              int32_t N_interp, int32_t isdefault, double defaultvalue, int32_t device,
                                              int32_t *num_exterior, double *kernel_ms);

/* Mesh -> regular grid sampling of the Moorings output (model/gridoutput.cpp:467-500 calls
 * InterpFromMeshToGridx, contrib/bamg/src/InterpFromMeshToGridx.cpp:11-193).  Same arguments: grid point
 * (i,j), i < nrows along x, j < ncols along y (x_i = xmin + i*xposting, y_j descending from ymax when
 * yposting > 0), griddata[N_data*(i*ncols+j)+k] CALLER-allocated.  A grid point takes the value of the LAST
 * element (highest number) whose area coordinates are all > -1e-11, exactly as the reference's element loop
 * overwrites; points in no element keep default_value; NaN results become default_value.  The area
 * coordinates are the reference's double expressions, so the values are bit-identical. */
NXS_INTERP_API int nxs_interp_mesh_to_grid(double *griddata, const int32_t *index_mesh, const double *x_mesh, const double *y_mesh,
                                           int32_t nods, int32_t nels, const double *data_mesh, int32_t data_length, int32_t N_data,
                                           double xmin, double ymax, double xposting, double yposting, int32_t nrows,
                                           int32_t ncols, double default_value, int32_t device, double *kernel_ms);
/* The same with the mesh data already on `device` (a device pointer to [data_length][N_data] rows, e.g. what nxs_dyn_ice_diagnostics returns):
 * a Moorings record without a round trip of the element state through the host. */
NXS_INTERP_API int nxs_interp_mesh_to_grid_device(double *griddata, const int32_t *index_mesh, const double *x_mesh, const double *y_mesh,
                                           int32_t nods, int32_t nels, const double *data_mesh_device, int32_t data_length, int32_t N_data,
                                           double xmin, double ymax, double xposting, double yposting, int32_t nrows,
                                           int32_t ncols, double default_value, int32_t device, double *kernel_ms);

/* Conservative remapping of the element variables at regrid: replaces the root-serial
 *
 *   ConservativeRemappingMeshToMesh(interp_elt_out, interp_in_elements, nb_var_element, bamgmesh_previous, bamgmesh_root);
 *
 * of FiniteElement::interpFields (FE.cpp:3108; contrib/bamg/src/ConservativeRemapping.cpp:176-328).  One thread per
 * triangle of the new mesh: the old triangle holding its barycentre (the reference finds it with
 * InterpFromMeshToMesh2dx, :243-249), the "same three vertices" shortcut through PreviousNumbering (:263-289),
 * otherwise the walk over the overlapping old triangles (checkTriangle, :330-449) with the polygon-clipping
 * weights, then out = (sum_k in[tri_k]*w_k) * (1/area(new triangle)) in the reference's visiting order
 * (ConservativeRemappingMeshToGrid, :97-131).  Same predicates, tolerances, sort and operand order as the
 * reference, so the result is bit-identical (tests/test_remap.py).
 *
 *   interp_in  [nels_old][nb_var], interp_out [nels_new][nb_var] (caller-allocated; the reference allocates it)
 *   index_     1-based triangles (bamgmesh->Triangles without the 4th column), x_, y_ = bamgmesh->Vertices
 *   nec_old / nec_width / ec_old   bamgmesh_previous->NodalElementConnectivity (+Size[1]) and ->ElementConnectivity
 *              as bamg leaves them (doubles, NaN padding); NULL = built here with nxs_mesh_connectivity /
 *              nxs_mesh_element_connectivity, which reproduce bamg's tables
 *   previous_numbering  bamgmesh_root->PreviousNumbering (1-based old number of every new vertex, 0 = new), may be NULL
 *   n_geom_vertices     bamgmesh_root->VerticesOnGeomVertexSize[0]
 *   num_failed (may be NULL)  new triangles whose barycentre is in no old triangle (the reference asserts) or that overlap
 *              more than 4096 old triangles (up to 96: lists in registers/scratch; beyond: a second pass with lists in
 *              global memory); their rows are NaN
 *   visits (may be NULL) [nels_new]  number of old triangles that contributed (1 = unchanged triangle)
 */
NXS_INTERP_API int nxs_interp_conservative_remap(double *interp_out, const double *interp_in, int32_t nb_var, const int32_t *index_old,
                                                 const double *x_old, const double *y_old, int32_t nods_old, int32_t nels_old,
                                                 const double *nec_old, int32_t nec_width, const double *ec_old,
                                                 const int32_t *index_new, const double *x_new, const double *y_new, int32_t nods_new,
                                                 int32_t nels_new, const double *previous_numbering, int32_t n_geom_vertices,
                                                 int32_t device, int32_t *num_failed, int32_t *visits, double *kernel_ms);

/* Structured grid -> mesh: the forcing ingest.  Replaces
 *
 *   InterpFromGridToMeshx(data_out, &gridX[0], gridX.size(), &gridY[0], gridY.size(), &data_in[0], gridY.size(), gridX.size(),
 *                         nb_var*nb_forcing_step, &RX[0], &RY[0], M_target_size, 100000000., interp_type);
 *
 * of ExternalData::loadDataset (model/externaldata.cpp:1436; contrib/bamg/src/InterpFromGridToMeshx.cpp:14-485), which
 * fills M_wind / M_ocean / M_ssh, the forcing inputs of the dynamics.  Same arguments: x_in / y_in are the pixel centres
 * (x_rows == N, y_rows == M) or their contours (one more entry each: centres are taken); data [M][N][N_data]
 * (row_major == 0, the call above) or [N][M][N_data]; data_mesh [nods][N_data] CALLER-allocated.  A node takes the
 * first grid interval that brackets it in x and in y (either orientation of the axes; the last coordinate belongs to the
 * last interval), then the triangle / bilinear / nearest formula with the reference's operand order -- including its
 * nearest-neighbour rule, which compares the node with the HALF EXTENT of the cell rather than its centre; NaN and
 * nodes outside the grid get default_value.  Bit-identical to the reference (tests/test_grid_to_mesh.py). */
enum { NXS_INTERP_TRIANGLE = 0, NXS_INTERP_BILINEAR = 1, NXS_INTERP_NEAREST = 2 };
NXS_INTERP_API int nxs_interp_grid_to_mesh(double *data_mesh, const double *x_in, int32_t x_rows, const double *y_in, int32_t y_rows,
                                           const double *data, int32_t M, int32_t N, int32_t N_data, const double *x_mesh,
                                           const double *y_mesh, int32_t nods, double default_value, int32_t interp, int32_t row_major,
                                           int32_t device, double *kernel_ms);

/* Of this thread's last nxs_interp_mesh_to_mesh_2d call: size of the completion, and how many target points were in no
 * triangle of the data mesh / of those, inside a fill triangle / projected on the hull / handled by the nearest-boundary-edge
 * stand-in; *completion_refused = why no completion was built (NULL when one was, or when isdefault != 0).  Any pointer may be NULL. */
NXS_INTERP_API int nxs_interp_last_info(int32_t *num_fill_triangles, int32_t *num_hull_edges, int32_t *num_exterior, int32_t *num_in_fill,
                                        int32_t *num_on_hull, int32_t *num_stand_in, const char **completion_refused);

/* Host only: bamg's convex completion of a mesh (index 1-based): the triangles ReconstructExistingMesh adds between the
 * boundary and the convex hull and inside the holes (fill_tri: 1-based, counter-clockwise, [3*cap_fill]) and the hull edges
 * counter-clockwise (hull_edges: 1-based vertex pairs, [2*cap_hull]).  Output arrays may be NULL to query the counts. */
NXS_INTERP_API int nxs_mesh_convex_completion(const int32_t *index, const double *x, const double *y, int32_t nods, int32_t nels,
                                              int32_t *num_fill, int32_t *fill_tri, int32_t cap_fill, int32_t *num_hull,
                                              int32_t *hull_edges, int32_t cap_hull);
/* mode: 0 = as above (the pocket construction; where it does not apply -- several components, a boundary that pinches -- the general one: the
 * constrained Delaunay triangulation of the boundary vertices minus the domain), 1 = the general construction, 2 = the pocket construction only */
NXS_INTERP_API int nxs_mesh_convex_completion_mode(const int32_t *index, const double *x, const double *y, int32_t nods, int32_t nels,
                                              int32_t *num_fill, int32_t *fill_tri, int32_t cap_fill, int32_t *num_hull,
                                              int32_t *hull_edges, int32_t cap_hull, int32_t mode);

NXS_INTERP_API const char *nxs_interp_last_error(void);

/* ---- a regrid's context: the old mesh's search tables built once (on the device) and shared by the two interpolation calls of
 * FiniteElement::interpFields (FE.cpp:3071-3154) and by any later call on the same mesh.  The one-shot functions above make one per call. */
typedef struct nxs_regrid nxs_regrid;
#define NXS_REGRID_IN_DEVICE 1   /* flags: the input data is a device pointer on the context's device */
#define NXS_REGRID_OUT_DEVICE 2  /*        the output array is */
NXS_INTERP_API int nxs_regrid_create(const int32_t *index_old, const double *x_old, const double *y_old, int32_t nods_old, int32_t nels_old, int32_t device,
                                     nxs_regrid **out);
NXS_INTERP_API int nxs_regrid_destroy(nxs_regrid *r);
/* InterpFromMeshToMesh2dx (arguments as nxs_interp_mesh_to_mesh_2d) */
NXS_INTERP_API int nxs_regrid_interp_nodes(nxs_regrid *r, double *data_interp, const double *data, int32_t M_data, int32_t N_data, const double *x_interp,
                                           const double *y_interp, int32_t N_interp, int32_t isdefault, double defaultvalue, int32_t flags,
                                           int32_t *num_exterior, double *kernel_ms);
/* ConservativeRemappingMeshToMesh (arguments as nxs_interp_conservative_remap) */
NXS_INTERP_API int nxs_regrid_remap_elements(nxs_regrid *r, double *interp_out, const double *interp_in, int32_t nb_var, const double *nec_old, int32_t nec_width,
                                             const double *ec_old, const int32_t *index_new, const double *x_new, const double *y_new, int32_t nods_new,
                                             int32_t nels_new, const double *previous_numbering, int32_t n_geom_vertices, int32_t flags,
                                             int32_t *num_failed, int32_t *visits, double *kernel_ms);
/* where the last regrid call of this thread spent its time, ms: [0] connectivity tables, [1] integer plane + bucket grid, [2] convex completion,
 * [3] host -> device copies, [4] kernels, [5] device -> host copies, [6] the whole call */
NXS_INTERP_API int nxs_interp_last_timing(double *ms8);
/* test door: the device-built tables (which = 0 bucket grid offsets, 1 its lists, 2 NodalElementConnectivity, 3 ElementConnectivity; ints, -1 = NaN) */
NXS_INTERP_API int nxs_regrid_debug_tables(nxs_regrid *r, int32_t which, int32_t *out, int64_t cap, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif

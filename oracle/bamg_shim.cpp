/*
 * bamg_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doors into the REAL reference library
 * contrib/bamg (built by oracle/Makefile into oracle/_ref/libbamg_ref.so from the sources where
 * they lie under /root/reference).  Compiled against the reference's headers in place
 * (-I/root/reference/contrib/bamg/include); nothing of the reference is copied here.
 *
 * Used by tests to pin
 *   - the connectivity tables the hot path reads (bamgmesh->NodalElementConnectivity /
 *     NodalConnectivity as built at FE.cpp:77-80 by BamgConvertMeshx), and
 *   - the mesh-to-mesh P1 interpolation used at regrid (FE.cpp:3131, InterpFromMeshToMesh2dx).
 */
#include <cstring>
#include <cmath>

#include "BamgMesh.h"
#include "BamgGeom.h"
#include "BamgConvertMeshx.h"
#include "InterpFromMeshToMesh2dx.h"

extern "C" {

/* Runs BamgConvertMeshx exactly as FiniteElement::distributedMeshProcessing does (FE.cpp:77-80)
 * and copies the two tables out.  Call once with nec/nc == NULL to get the widths. */
int shim_bamg_connectivity(const int *index, const double *x, const double *y, int nods, int nels,
                           int *nec_width, double *nec, int *nc_width, double *nc) {
    BamgMesh *bamgmesh = new BamgMesh();
    BamgGeom *bamggeom = new BamgGeom();
    int rc = BamgConvertMeshx(bamgmesh, bamggeom, const_cast<int *>(index), const_cast<double *>(x),
                              const_cast<double *>(y), nods, nels);
    int w1 = bamgmesh->NodalElementConnectivitySize[1];
    int w2 = bamgmesh->NodalConnectivitySize[1];
    if (nec_width) *nec_width = w1;
    if (nc_width) *nc_width = w2;
    if (nec) std::memcpy(nec, bamgmesh->NodalElementConnectivity, sizeof(double) * (size_t)w1 * nods);
    if (nc) std::memcpy(nc, bamgmesh->NodalConnectivity, sizeof(double) * (size_t)w2 * nods);
    delete bamggeom;
    delete bamgmesh;
    return rc == 1 ? 0 : -1;
}

/* InterpFromMeshToMesh2dx as called at FE.cpp:3131-3139 (M_data = nods_data: nodal data). */
int shim_bamg_interp_mesh_to_mesh(const int *index_data, const double *x_data, const double *y_data,
                                  int nods_data, int nels_data, const double *data, int M_data, int N_data,
                                  const double *x_interp, const double *y_interp, int N_interp,
                                  int isdefault, double defaultvalue, double *out) {
    double *res = NULL;
    int rc = InterpFromMeshToMesh2dx(&res, const_cast<int *>(index_data), const_cast<double *>(x_data),
                                     const_cast<double *>(y_data), nods_data, nels_data,
                                     const_cast<double *>(data), M_data, N_data,
                                     const_cast<double *>(x_interp), const_cast<double *>(y_interp), N_interp,
                                     isdefault != 0, defaultvalue);
    if (res) {
        std::memcpy(out, res, sizeof(double) * (size_t)N_interp * N_data);
        delete[] res; /* xNew<double> == new double[] in contrib/bamg/include/MemOps.h */
    }
    return rc == 1 ? 0 : -1;
}

} /* extern "C" */

#include "InterpFromMeshToGridx.h"

extern "C" {

/* InterpFromMeshToGridx as called by GridOutput (model/gridoutput.cpp:496-505). */
int shim_bamg_interp_mesh_to_grid(const int *index_mesh, const double *x_mesh, const double *y_mesh, int nods, int nels,
                                  const double *data_mesh, int data_length, int N_data, double xmin, double ymax, double xposting,
                                  double yposting, int nrows, int ncols, double default_value, double *out) {
    double *grid = NULL;
    InterpFromMeshToGridx(grid, const_cast<int *>(index_mesh), const_cast<double *>(x_mesh), const_cast<double *>(y_mesh), nods, nels,
                          const_cast<double *>(data_mesh), data_length, N_data, xmin, ymax, xposting, yposting, nrows, ncols,
                          default_value);
    if (!grid) return -1;
    std::memcpy(out, grid, sizeof(double) * (size_t)N_data * nrows * ncols);
    delete[] grid;
    return 0;
}

} /* extern "C" */

#include <vector>
#include "ConservativeRemapping.hpp"

extern "C" {

/* bamgmesh->ElementConnectivity and ->Triangles of BamgConvertMeshx (Mesh.cpp:690-693, 777-796), for the
 * adjacency rows ConservativeRemapping's checkTriangle walks. */
int shim_bamg_element_connectivity(const int *index, const double *x, const double *y, int nods, int nels, double *ec, double *tri) {
    BamgMesh *bamgmesh = new BamgMesh();
    BamgGeom *bamggeom = new BamgGeom();
    int rc = BamgConvertMeshx(bamgmesh, bamggeom, const_cast<int *>(index), const_cast<double *>(x), const_cast<double *>(y), nods, nels);
    if (bamgmesh->ElementConnectivitySize[0] != nels || bamgmesh->TrianglesSize[0] != nels) rc = 0;
    if (rc == 1) {
        std::memcpy(ec, bamgmesh->ElementConnectivity, sizeof(double) * 3 * (size_t)nels);
        for (int t = 0; t < nels; ++t)
            for (int k = 0; k < 3; ++k) tri[3 * t + k] = bamgmesh->Triangles[4 * t + k];
    }
    delete bamggeom;
    delete bamgmesh;
    return rc == 1 ? 0 : -1;
}

/* ConservativeRemappingMeshToMesh as called at FE.cpp:3108.  Both BamgMesh objects come from BamgConvertMeshx;
 * on the new mesh PreviousNumbering (1-based, 0 = new vertex) and VerticesOnGeomVertexSize[0] are set from the
 * arguments, which is what Bamgx leaves there after an adaptation (Mesh.cpp:553-558, 729). */
int shim_bamg_conservative_remap(const int *index_old, const double *x_old, const double *y_old, int nods_old, int nels_old,
                                 const int *index_new, const double *x_new, const double *y_new, int nods_new, int nels_new,
                                 const double *previous_numbering, int n_geom_vertices, const double *in, int nb_var, double *out) {
    BamgMesh *mo = new BamgMesh(), *mn = new BamgMesh();
    BamgGeom *go = new BamgGeom(), *gn = new BamgGeom();
    int rc = BamgConvertMeshx(mo, go, const_cast<int *>(index_old), const_cast<double *>(x_old), const_cast<double *>(y_old), nods_old, nels_old);
    if (rc == 1) rc = BamgConvertMeshx(mn, gn, const_cast<int *>(index_new), const_cast<double *>(x_new), const_cast<double *>(y_new), nods_new, nels_new);
    if (rc == 1 && (mo->TrianglesSize[0] != nels_old || mn->TrianglesSize[0] != nels_new)) rc = 0;
    if (rc == 1) {
        delete[] mn->PreviousNumbering;
        mn->PreviousNumbering = new double[nods_new];
        std::memcpy(mn->PreviousNumbering, previous_numbering, sizeof(double) * (size_t)nods_new);
        mn->VerticesOnGeomVertexSize[0] = n_geom_vertices;
        std::vector<double> vin(in, in + (size_t)nb_var * nels_old);
        double *res = NULL;
        ConservativeRemappingMeshToMesh(res, vin, nb_var, mo, mn);
        std::memcpy(out, res, sizeof(double) * (size_t)nb_var * nels_new);
        delete[] res;
        mn->VerticesOnGeomVertexSize[0] = 0; /* the destructor frees VerticesOnGeomVertex by pointer only */
    }
    delete go; delete gn; delete mo; delete mn;
    return rc == 1 ? 0 : -1;
}

} /* extern "C" */

#include "BamgOpts.h"
#include "Bamgx.h"

extern "C" {

/* One regrid of the reference on the root (FE.cpp:3606-3700 regrid + :3760-3801 adaptMesh), with the REAL remesher:
 *   BamgConvertMeshx of the mesh at rest (as rootMeshProcessing, FE.cpp:300-340) -> bamgopt as initBamg (FE.cpp:992-1038)
 *   with hmin / hmax from the caller (minMaxSide, FE.cpp:343-355) -> Vertices overwritten with the moved coordinates
 *   (FE.cpp:3674-3678) -> Dirichlet edge flags (:3775-3789) -> Bamgx(root, previous).
 * Outputs (caller-allocated with capacities; sizes returned): the adapted mesh, bamgmesh_root->PreviousNumbering and
 * VerticesOnGeomVertexSize[0].  Objects are leaked on purpose (the reference shares pointers between root and previous). */
int shim_bamg_adapt(const int *index, const double *x0, const double *y0, int nods, int nels, const int *dirichlet_flags, int ndir,
                    const double *x_moved, const double *y_moved, double hmin, double hmax, int cap_nods, int cap_nels, int *out_nods,
                    int *out_nels, int *out_index, double *out_x, double *out_y, double *out_prev, int *out_ngeom) {
    BamgOpts *bamgopt = new BamgOpts();
    bamgopt->Crack = 0; bamgopt->anisomax = 1e30; bamgopt->coeff = 1; bamgopt->cutoff = 1e-5; bamgopt->errg = 0.1; bamgopt->field = NULL;
    bamgopt->gradation = 1.5; bamgopt->Hessiantype = 0; bamgopt->hmin = 1e-100; bamgopt->hmax = 1e100; bamgopt->hminVertices = NULL;
    bamgopt->hmaxVertices = NULL; bamgopt->hVertices = NULL; bamgopt->KeepVertices = 1; bamgopt->MaxCornerAngle = 10; bamgopt->maxnbv = 1e7;
    bamgopt->maxsubdiv = 10; bamgopt->metric = NULL; bamgopt->Metrictype = 0; bamgopt->nbjacobi = 1; bamgopt->nbsmooth = 3; bamgopt->omega = 1.8;
    bamgopt->power = 1.; bamgopt->splitcorners = 1; bamgopt->geometricalmetric = 0; bamgopt->random = true; bamgopt->verbose = 0;
    bamgopt->Check();
    bamgopt->hmin = hmin; bamgopt->hmax = hmax;               /* FROM_UNREF meshes, FE.cpp:352-355 */
    bamgopt->KeepVertices = 1; bamgopt->splitcorners = 0;     /* state after the first time step, FE.cpp:391-392 */

    BamgMesh *root = new BamgMesh(); BamgGeom *groot = new BamgGeom();
    if (BamgConvertMeshx(root, groot, const_cast<int *>(index), const_cast<double *>(x0), const_cast<double *>(y0), nods, nels) != 1) return -1;
    for (int id = 0; id < root->VerticesSize[0]; ++id) { root->Vertices[3 * id] = x_moved[id]; root->Vertices[3 * id + 1] = y_moved[id]; }
    BamgMesh *prev = new BamgMesh(); BamgGeom *gprev = new BamgGeom(); BamgOpts *oprev = new BamgOpts();
    *prev = *root; *gprev = *groot; *oprev = *bamgopt;        /* FE.cpp:3770-3772 */
    const int M_flag_fix = 10000;
    for (int edg = 0; edg < prev->EdgesSize[0]; ++edg) {
        const int fnd = (int)prev->Edges[3 * edg];
        bool dir = false;
        for (int k = 0; k < ndir && !dir; ++k) dir = dirichlet_flags[k] == fnd;
        gprev->Edges[3 * edg + 2] = dir ? M_flag_fix : M_flag_fix + 1;
        prev->Edges[3 * edg + 2] = dir ? M_flag_fix : M_flag_fix + 1;
    }
    BamgMesh *out = new BamgMesh(); BamgGeom *gout = new BamgGeom();
    if (Bamgx(out, gout, prev, gprev, oprev) != 1) return -2;
    const int nn = out->VerticesSize[0], ne = out->TrianglesSize[0];
    *out_nods = nn; *out_nels = ne;
    if (nn > cap_nods || ne > cap_nels) return -3;
    for (int i = 0; i < nn; ++i) { out_x[i] = out->Vertices[3 * i]; out_y[i] = out->Vertices[3 * i + 1]; out_prev[i] = out->PreviousNumbering ? out->PreviousNumbering[i] : 0.; }
    for (int t = 0; t < ne; ++t) for (int k = 0; k < 3; ++k) out_index[3 * t + k] = (int)out->Triangles[4 * t + k];
    *out_ngeom = out->VerticesOnGeomVertexSize[0];
    return 0;
}

} /* extern "C" */

#include "InterpFromGridToMeshx.h"

extern "C" {

/* InterpFromGridToMeshx as called by ExternalData::loadDataset (model/externaldata.cpp:1436).  interp: 0 triangle, 1 bilinear, 2 nearest. */
int shim_bamg_interp_grid_to_mesh(const double *x_in, int x_rows, const double *y_in, int y_rows, const double *data, int M, int N, int N_data,
                                  const double *x_mesh, const double *y_mesh, int nods, double default_value, int interp, int row_major, double *out) {
    const int e = interp == 0 ? TriangleInterpEnum : interp == 1 ? BilinearInterpEnum : NearestInterpEnum;
    double *res = NULL;
    InterpFromGridToMeshx(res, const_cast<double *>(x_in), x_rows, const_cast<double *>(y_in), y_rows, const_cast<double *>(data), M, N, N_data,
                          const_cast<double *>(x_mesh), const_cast<double *>(y_mesh), nods, default_value, e, row_major != 0);
    if (!res) return -1;
    std::memcpy(out, res, sizeof(double) * (size_t)nods * N_data);
    delete[] res;
    return 0;
}

} /* extern "C" */

#include "Mesh.h"

extern "C" {

/* The convex completion InterpFromMeshToMesh2dx works on (InterpFromMeshToMesh2dx.cpp:60-65: Mesh(index, x, y, nods, nels) =
 * ReadMesh + SetIntCoor + ReconstructExistingMesh, then TriangleReferenceList): every triangle of bamg's reconstructed mesh in
 * bamg's own order and vertex order -- the given triangles first, then the triangles bamg added to fill holes and concave parts
 * of the boundary, and its boundary ("infinite") triangles with one NULL vertex (-1 here).  tri: [3*max_nbt], reft: [max_nbt]
 * (< 0 = outside the mesh).  Returns the number of triangles (or -1 - needed when max_nbt is too small). */
int shim_bamg_completed_mesh(const int *index, const double *x, const double *y, int nods, int nels, int max_nbt, int *tri, long *reft) {
    bamg::Mesh *Th = new bamg::Mesh(const_cast<int *>(index), const_cast<double *>(x), const_cast<double *>(y), nods, nels);
    const int nbt = (int)Th->nbt;
    if (nbt > max_nbt) { delete Th; return -1 - nbt; }
    long *r = new long[nbt];
    Th->TriangleReferenceList(r);
    for (int i = 0; i < nbt; ++i) {
        bamg::Triangle &t = Th->triangles[i];
        for (int k = 0; k < 3; ++k) tri[3 * i + k] = t(k) ? (int)Th->GetId(t(k)) : -1;
        reft[i] = r[i];
    }
    delete[] r;
    delete Th;
    return nbt;
}

} /* extern "C" */

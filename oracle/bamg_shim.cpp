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

/*
 * remap_host.cpp -- TEST INFRASTRUCTURE.  Host build of nextsim_amd/csrc/nxs_remap_core.inl, the functions the
 * conservative-remapping kernel runs per new triangle, so that they can be compared with the REAL
 * contrib/bamg ConservativeRemappingMeshToMesh (oracle/_ref, through bamg_shim.cpp) in a container without a
 * GPU.  Nothing under nextsim_amd/ loads this library; the product path is the HIP kernel in nxs_interp.hip.
 */
#include <cmath>
#include <vector>

#define NXS_HD
#include "../nextsim_amd/csrc/nxs_remap_core.inl"

extern "C" int remap_host(const int *tri_old, const double *x_old, const double *y_old, int nods_old, int nels_old, const int *nec, int nec_w,
                          const int *ec, const int *tri_new, const double *x_new, const double *y_new, int nels_new,
                          const double *previous_numbering, int n_geom, const int *seed, const double *in, int nb_var, double *out,
                          int *visits) {
    nxs_remap::OldMesh m{nels_old, nods_old, tri_old, x_old, y_old, nec, nec_w, ec};
    int failed = 0;
    std::vector<int> tris(nxs_remap::kMaxVisit);
    std::vector<double> w(nxs_remap::kMaxVisit);
    std::vector<nxs_remap::Frame> stack(nxs_remap::kMaxVisit + 1);
    for (int t = 0; t < nels_new; ++t) {
        double cx[3], cy[3];
        for (int i = 0; i < 3; ++i) { cx[i] = x_new[tri_new[3 * t + i]]; cy[i] = y_new[tri_new[3 * t + i]]; }
        const bool same = nxs_remap::same_triangle(m, seed[t], tri_new + 3 * t, previous_numbering, n_geom);
        const int n = nxs_remap::collect(m, cx, cy, seed[t], same, tris.data(), w.data(), stack.data());
        if (visits) visits[t] = n;
        if (n < 0) {
            ++failed;
            for (int v = 0; v < nb_var; ++v) out[(long long)t * nb_var + v] = std::nan("");
            continue;
        }
        nxs_remap::apply(in, nb_var, cx, cy, tris.data(), w.data(), n, out + (long long)t * nb_var);
    }
    return failed;
}

// nxs_mesh.cpp -- host-side mesh tables for libnxsdyn.so.
//
// nxs_mesh_connectivity() produces the two tables the hot path reads from bamgmesh
// (FE.cpp:10376-10379 NodalElementConnectivity, FE.cpp:10578-10602 NodalConnectivity) with the
// same content AND the same row ordering as BamgConvertMeshx -> Mesh::WriteMesh
// (contrib/bamg/src/Mesh.cpp:514-543, 579-630, 798-865), because the row order is the
// floating-point summation order of the air-drag average and of the open-water smoother.
//
//   element fan of a vertex : elements in DESCENDING number (bamg walks a LIFO chain)
//   neighbours of a vertex  : edge ends in reverse order of edge creation, where edges are created
//                             by first appearance over triangles 0..Ne-1, local edges
//                             (1,2),(2,0),(0,1); an edge is stored oriented as in the triangle
//                             that created it.
//
// Checked against the real contrib/bamg in tests/test_connectivity.py.
#include <cmath>
#include <cstdint>
#include <vector>

#include "nxs_dyn.h"
#include "nxs_guard.hpp"

namespace {

struct EdgeRec {
    int32_t a, b;      // sorted ends (0-based)
    int32_t first, second;  // oriented ends as seen in the creating triangle (0-based)
};

// handler of the function-try-blocks below (nxs_guard.hpp).  These entry points keep no error text of their own.
int entry_caught(const char *entry) noexcept { return nxs_guard::caught(entry, [](int, const char *) {}); }

}  // namespace

extern "C" int nxs_mesh_connectivity(const int32_t *indices, int32_t num_nodes, int32_t num_elements,
                                     int32_t *nec_width, double *nec, int32_t *nc_width, double *nc) try {
    if (!indices || num_nodes <= 0 || num_elements <= 0) return NXS_ERR_INVALID;
    const int64_t Nn = num_nodes, Ne = num_elements;
    for (int64_t i = 0; i < 3 * Ne; ++i)
        if (indices[i] < 1 || indices[i] > num_nodes) return NXS_ERR_INVALID;

    // ---- element fans: count, then fill back to front so rows come out in descending order ----
    std::vector<int32_t> deg(Nn, 0);
    for (int64_t i = 0; i < 3 * Ne; ++i) deg[indices[i] - 1]++;
    int32_t w1 = 0;
    for (int64_t v = 0; v < Nn; ++v) w1 = deg[v] > w1 ? deg[v] : w1;
    if (nec_width) *nec_width = w1;
    if (nec) {
        const double nan = std::nan("");
        for (int64_t i = 0; i < Nn * w1; ++i) nec[i] = nan;
        std::vector<int32_t> left(deg);  // entries still to place per vertex
        for (int64_t e = 0; e < Ne; ++e)
            for (int k = 0; k < 3; ++k) {
                const int64_t v = indices[3 * e + k] - 1;
                // the j-th visit (0-based) of v ends up at column deg-1-j
                nec[v * w1 + (--left[v])] = double(e + 1);
            }
    }

    // ---- unique edges in creation order (hash chain on the smaller end, like SetOfEdges4) ----
    static const int LE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    std::vector<EdgeRec> edges;
    edges.reserve(size_t(3 * Ne / 2 + Nn));
    std::vector<int32_t> head(Nn, -1), nxt;
    nxt.reserve(edges.capacity());
    for (int64_t t = 0; t < Ne; ++t)
        for (int j = 0; j < 3; ++j) {
            const int32_t p = indices[3 * t + LE[j][0]] - 1, q = indices[3 * t + LE[j][1]] - 1;
            const int32_t a = p <= q ? p : q, b = p <= q ? q : p;
            int32_t n = head[a];
            while (n >= 0 && !(edges[n].a == a && edges[n].b == b)) n = nxt[n];
            if (n >= 0) continue;
            // orientation: find the sorted-first end `a` in the triangle; if its successor is `b`
            // keep (a,b) else store (b,a)   (Mesh.cpp:609-627)
            EdgeRec r{a, b, a, b};
            for (int k = 0; k < 3; ++k)
                if (indices[3 * t + k] - 1 == a) {
                    if (indices[3 * t + (k + 1) % 3] - 1 != b) { r.first = b; r.second = a; }
                    break;
                }
            nxt.push_back(head[a]);
            head[a] = int32_t(edges.size());
            edges.push_back(r);
        }

    std::vector<int32_t> ndeg(Nn, 0);
    for (const EdgeRec &r : edges) { ndeg[r.first]++; ndeg[r.second]++; }
    int32_t w2 = 0;
    for (int64_t v = 0; v < Nn; ++v) w2 = ndeg[v] > w2 ? ndeg[v] : w2;
    w2 += 1;  // last column = neighbour count
    if (nc_width) *nc_width = w2;
    if (nc) {
        for (int64_t i = 0; i < Nn * w2; ++i) nc[i] = 0.;
        std::vector<int32_t> left(ndeg);
        // chain insertion order is edge 0 end 0, edge 0 end 1, edge 1 end 0, ... ; rows are read
        // newest first, so the j-th insertion for v lands at column ndeg-1-j
        for (const EdgeRec &r : edges) {
            nc[int64_t(r.first) * w2 + (--left[r.first])] = double(r.second + 1);
            nc[int64_t(r.second) * w2 + (--left[r.second])] = double(r.first + 1);
        }
        for (int64_t v = 0; v < Nn; ++v) nc[v * w2 + (w2 - 1)] = double(ndeg[v]);
    }
    return NXS_OK;
} catch (...) { return entry_caught("nxs_mesh_connectivity"); }

// bamgmesh->ElementConnectivity (contrib/bamg/src/Mesh.cpp:777-796): column j = 1-based number of the
// triangle across local edge j (vertices (j+1)%3, (j+2)%3), NaN on the boundary.  The rows are what
// ConservativeRemapping's checkTriangle walks (ConservativeRemapping.cpp:411-436), which stops at the first
// NaN of a row -- so the column of a neighbour matters, not only the set.
extern "C" int nxs_mesh_element_connectivity(const int32_t *indices, int32_t num_nodes, int32_t num_elements, double *ec) try {
    if (!indices || !ec || num_nodes <= 0 || num_elements <= 0) return NXS_ERR_INVALID;
    const int64_t Ne = num_elements;
    for (int64_t i = 0; i < 3 * Ne; ++i)
        if (indices[i] < 1 || indices[i] > num_nodes) return NXS_ERR_INVALID;
    static const int LE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    const double nan = std::nan("");
    for (int64_t i = 0; i < 3 * Ne; ++i) ec[i] = nan;
    // half-edge chains on the smaller end
    std::vector<int32_t> head(num_nodes, -1), nxt(3 * Ne, -1), other(3 * Ne);
    for (int64_t t = 0; t < Ne; ++t)
        for (int j = 0; j < 3; ++j) {
            const int32_t p = indices[3 * t + LE[j][0]] - 1, q = indices[3 * t + LE[j][1]] - 1;
            const int32_t a = p < q ? p : q, b = p < q ? q : p;
            const int64_t me = 3 * t + j;
            other[me] = b;
            int32_t mate = -1;
            for (int32_t h = head[a]; h >= 0; h = nxt[h])
                if (other[h] == b) { mate = h; break; }
            if (mate >= 0) {
                if (!std::isnan(ec[mate])) return NXS_ERR_INVALID;  // an edge shared by three triangles
                ec[mate] = double(t + 1);
                ec[me] = double(mate / 3 + 1);
            } else {
                nxt[me] = head[a];
                head[a] = int32_t(me);
            }
        }
    return NXS_OK;
} catch (...) { return entry_caught("nxs_mesh_element_connectivity"); }

// calcCohesion (FE.cpp:3909-3914) on top of initIce's random field (FE.cpp:11459-11475): one draw of
// boost::uniform_01<boost::minstd_rand> (default seed 1; x_{k+1} = 48271 x_k mod 2^31-1) per GLOBAL element, indexed by the
// element's global id, so every rank sees the same value for the same triangle.  With an engine as first template argument
// Boost 1.67 (the version pinned in scripts/env_compile_gnu_linux.bash:19) takes uniform_01's backward-compatible class
// (boost/random/uniform_01.hpp, backward_compatible_uniform_01): it stores _factor = 1 / (double(max - min) + 1) =
// 1 / 2147483646.0 once and returns double(x - min) * _factor, drawing again while the result is >= 1.  A MULTIPLICATION by the
// rounded reciprocal, not a division: the two differ in the last bit for about one draw in a hundred, first at draw 142 (tests/test_mesh_partition.py).
extern "C" int nxs_calc_cohesion(double C_fix, double C_alea, const int32_t *global_element_id, int64_t num_elements,
                                 int64_t num_global_elements, double *cohesion) try {
    if (!global_element_id || !cohesion || num_elements < 0 || num_global_elements < 1) return NXS_ERR_INVALID;
    std::vector<double> random_number_root((size_t)num_global_elements);
    unsigned long long x = 1ull;
    const double factor = 1.0 / (double(2147483646ull - 1ull) + 1.0);
    for (int64_t i = 0; i < num_global_elements; ++i) {
        double r;
        do {
            x = (x * 48271ull) % 2147483647ull;
            r = double(x - 1ull) * factor;
        } while (!(r < 1.0));
        random_number_root[(size_t)i] = r;
    }
    for (int64_t i = 0; i < num_elements; ++i) {
        const int32_t id = global_element_id[i];
        if (id < 1 || id > num_global_elements) return NXS_ERR_INVALID;
        cohesion[i] = C_fix + C_alea * (random_number_root[(size_t)id - 1]);
    }
    return NXS_OK;
} catch (...) { return entry_caught("nxs_calc_cohesion"); }

// nxs_regrid_tables.inl -- the search tables of a regrid, built ON THE DEVICE from the old mesh's triangles (textually included by
// nxs_interp.hip): the bucket grid of the exact point locator and the two connectivity tables checkTriangle walks
// (bamgmesh->NodalElementConnectivity, Mesh.cpp:798-828: a vertex's triangles in DESCENDING number; bamgmesh->ElementConnectivity,
// Mesh.cpp:777-796: the triangle across local edge j).  Integer / byte work, a handful of small HBM-bound kernels (count with atomics ->
// exclusive scan -> fill with atomic cursors -> per-row sort, which makes the result independent of the order the atomics were served in):
// the same tables, entry for entry, as the host code builds (nxs_mesh.cpp; tests/test_remap.py compares them through a test door), in
// ~0.2 ms instead of the 60-130 ms the host loops took at 1.5 M triangles -- the regrid calls were 50-600 times their kernels (round 2).

namespace regrid_tables {

// ---- exclusive scan of n ints in place (blocks of 1024 = 256 threads x 4, recursion over the block sums)
__global__ void __launch_bounds__(256) k_scan_blocks(int *data, int *block_sums, int n) {
    __shared__ int sh[256];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? data[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele over the 256 partial sums
        const int add = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - s;  // exclusive prefix of this thread's four
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255 && block_sums) block_sums[blockIdx.x] = sh[255];
}
__global__ void __launch_bounds__(256) k_scan_add(int *data, const int *block_offsets, int n) {
    const int i = blockIdx.x * 1024 + threadIdx.x * 4;
    const int add = block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k < n) data[i + k] += add;
}
// data[0..n) -> exclusive prefix sums; returns hipSuccess; scratch allocated and freed here (a few KB)
inline hipError_t exclusive_scan(int *data, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int nb = (n + 1023) / 1024;
    int *sums = nullptr;
    hipError_t e = hipMalloc((void **)&sums, sizeof(int) * (size_t)std::max(nb, 1));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(256), 0, st, data, sums, n);
    if (nb > 1) {
        e = exclusive_scan(sums, nb, st);
        if (e == hipSuccess) hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, st, data, (const int *)sums, n);
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(sums);
    return e != hipSuccess ? e : e2;
}

// ---- bucket grid over bamg's integer plane: cell (cx, cy) lists every triangle whose bounding box touches it, ascending
__device__ __forceinline__ void cell_range(const int *t0, const int *t1, const int *t2, const int *ix, const int *iy, int e, int shift, int G,
                                           int &cx0, int &cx1, int &cy0, int &cy1) {
    const int a = t0[e], b = t1[e], c = t2[e];
    const int xa = ix[a], xb = ix[b], xc = ix[c], ya = iy[a], yb = iy[b], yc = iy[c];
    cx0 = max(min(min(xa, xb), xc) >> shift, 0); cx1 = min(max(max(xa, xb), xc) >> shift, G - 1);
    cy0 = max(min(min(ya, yb), yc) >> shift, 0); cy1 = min(max(max(ya, yb), yc) >> shift, G - 1);
}
// (a triangle of the mesh touches a handful of cells: one thread each; a FILL triangle of the convex completion may span a bay -- tens of
// thousands of cells: the k_grid_*_wide kernels give such a triangle a whole workgroup.  A triangle is "wide" from NXS_GRID_WIDE cells on.)
#define NXS_GRID_WIDE 64
__global__ void __launch_bounds__(256) k_grid_count(int nels, const int *t0, const int *t1, const int *t2, const int *ix, const int *iy, int shift, int G, int *cnt,
                                                    int *wide_list, int *n_wide) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nels) return;
    int a, b, c, d;
    cell_range(t0, t1, t2, ix, iy, e, shift, G, a, b, c, d);
    if ((long long)(b - a + 1) * (d - c + 1) >= NXS_GRID_WIDE) { wide_list[atomicAdd(n_wide, 1)] = e; return; }
    for (int cy = c; cy <= d; ++cy) for (int cx = a; cx <= b; ++cx) atomicAdd(cnt + (size_t)cy * G + cx, 1);
}
__global__ void __launch_bounds__(256) k_grid_count_wide(const int *wide_list, const int *t0, const int *t1, const int *t2, const int *ix, const int *iy, int shift, int G, int *cnt) {
    const int e = wide_list[blockIdx.x];
    int a, b, c, d;
    cell_range(t0, t1, t2, ix, iy, e, shift, G, a, b, c, d);
    const int w = b - a + 1;
    const long long n = (long long)w * (d - c + 1);
    for (long long k = threadIdx.x; k < n; k += 256) atomicAdd(cnt + (size_t)(c + (int)(k / w)) * G + a + (int)(k % w), 1);
}
__global__ void __launch_bounds__(256) k_grid_fill_wide(const int *wide_list, const int *t0, const int *t1, const int *t2, const int *ix, const int *iy, int shift, int G,
                                                        const int *cell_off, int *cursor, int *cell_tri) {
    const int e = wide_list[blockIdx.x];
    int a, b, c, d;
    cell_range(t0, t1, t2, ix, iy, e, shift, G, a, b, c, d);
    const int w = b - a + 1;
    const long long n = (long long)w * (d - c + 1);
    for (long long k = threadIdx.x; k < n; k += 256) {
        const size_t cell = (size_t)(c + (int)(k / w)) * G + a + (int)(k % w);
        cell_tri[cell_off[cell] + atomicAdd(cursor + cell, 1)] = e;
    }
}
__global__ void __launch_bounds__(256) k_grid_fill(int nels, const int *t0, const int *t1, const int *t2, const int *ix, const int *iy, int shift, int G,
                                                   const int *cell_off, int *cursor, int *cell_tri) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nels) return;
    int a, b, c, d;
    cell_range(t0, t1, t2, ix, iy, e, shift, G, a, b, c, d);
    if ((long long)(b - a + 1) * (d - c + 1) >= NXS_GRID_WIDE) return;  // (k_grid_fill_wide)
    for (int cy = c; cy <= d; ++cy)
        for (int cx = a; cx <= b; ++cx) {
            const size_t cell = (size_t)cy * G + cx;
            cell_tri[cell_off[cell] + atomicAdd(cursor + cell, 1)] = e;
        }
}
__global__ void __launch_bounds__(256) k_rows_sort(int nrows, const int *row_off, int *vals, int descending) {  // insertion sort of every row (a few entries each)
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const int lo = row_off[r], hi = row_off[r + 1];
    for (int i = lo + 1; i < hi; ++i) {
        const int v = vals[i];
        int j = i - 1;
        while (j >= lo && (descending ? vals[j] < v : vals[j] > v)) { vals[j + 1] = vals[j]; --j; }
        vals[j + 1] = v;
    }
}

// ---- NodalElementConnectivity as ints: row v = the triangles holding v, DESCENDING, -1 behind them; width = the longest row
__global__ void __launch_bounds__(256) k_fan_count(int nels, const int *tri /*[3 nels] 0-based*/, int *deg, int *maxdeg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nels) return;
    const int d = atomicAdd(deg + tri[i], 1) + 1;
    atomicMax(maxdeg, d);
}
__global__ void __launch_bounds__(256) k_fan_fill(int nels, const int *tri, int *cursor, int *nec, int w1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nels) return;
    const int v = tri[i];
    nec[(size_t)v * w1 + atomicAdd(cursor + v, 1)] = i / 3;
}
__global__ void __launch_bounds__(256) k_fan_sort_desc(int nods, const int *deg, int *nec, int w1) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nods) return;
    int *row = nec + (size_t)v * w1;
    const int n = deg[v];
    for (int i = 1; i < n; ++i) {
        const int x = row[i];
        int j = i - 1;
        while (j >= 0 && row[j] < x) { row[j + 1] = row[j]; --j; }
        row[j + 1] = x;
    }
    for (int i = n; i < w1; ++i) row[i] = -1;
}
// ---- ElementConnectivity as ints: column j of triangle t = the triangle across its local edge j (vertices (j+1)%3, (j+2)%3), -1 on the boundary;
//      *bad counts edges held by more than two triangles
__global__ void __launch_bounds__(256) k_elem_conn(int nels, const int *tri, const int *deg, const int *nec, int w1, int *ec, int *bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nels) return;
    const int t = i / 3, j = i % 3;
    const int p = tri[3 * t + (j + 1) % 3], q = tri[3 * t + (j + 2) % 3];
    const int *row = nec + (size_t)p * w1;
    int mate = -1, found = 0;
    for (int k = 0; k < deg[p]; ++k) {
        const int e = row[k];
        if (e == t) continue;
        if (tri[3 * e] == q || tri[3 * e + 1] == q || tri[3 * e + 2] == q) { mate = e; ++found; }
    }
    if (found > 1) atomicAdd(bad, 1);
    ec[i] = mate;
}

// The boundary edges of the mesh out of its ElementConnectivity: every local edge i = 3 t + j without a neighbour is appended to `list` (any order:
// the host sorts the few thousand of them).  A neighbour that runs along the shared edge in the SAME direction (an inconsistently oriented mesh)
// raises `bad`: the host's own pass over the triangles (nxs_hull::find_boundary_edges) then decides, as before.
__global__ void __launch_bounds__(256) k_boundary_list(int nels, const int *tri, const int *ec, int *count, int *list, int cap, int *bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nels) return;
    const int t = i / 3, j = i % 3, e = ec[i];
    if (e < 0) {
        const int pos = atomicAdd(count, 1);
        if (pos < cap) list[pos] = i;
        return;
    }
    const int p = tri[3 * t + (j + 1) % 3], q = tri[3 * t + (j + 2) % 3];  // this triangle runs p -> q along the edge: its neighbour must run q -> p
    bool reversed = false;
    for (int k = 0; k < 3; ++k) reversed = reversed || (tri[3 * e + k] == q && tri[3 * e + (k + 1) % 3] == p);
    if (!reversed) atomicAdd(bad, 1);
}

}  // namespace regrid_tables

// Micro-benchmark: what does a grid-wide barrier cost on this GPU, against a kernel boundary in a hipGraph?
//   hipcc --offload-arch=gfx950 -O3 -o gridsync gridsync.hip && ./gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(512) k_coop(double *a, int n, int iters) {
    cg::grid_group g = cg::this_grid();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (i < n) a[i] = a[(i + 4099) % n] * 0.5 + 1.0;
        g.sync();
    }
}

// hand-made barrier: one atomic counter per phase (monotonic), spin on it
__global__ void __launch_bounds__(512) k_manual(double *a, int n, int iters, unsigned int *ctr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (i < n) a[i] = a[(i + 4099) % n] * 0.5 + 1.0;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(ctr, 1u);
            const unsigned target = (unsigned)(it + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) k_plain(double *a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[(i + 4099) % n] * 0.5 + 1.0;
}

int main() {
    int dev = 0; CHK(hipSetDevice(dev));
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, dev));
    printf("%s: %d CUs, cooperativeLaunch=%d\n", pr.name, pr.multiProcessorCount, pr.cooperativeLaunch);
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 1000;
    for (int wgs : {256, 512}) {
        const int n = wgs * 512;
        double *a; CHK(hipMalloc(&a, n * sizeof(double))); CHK(hipMemset(a, 0, n * sizeof(double)));
        unsigned int *ctr; CHK(hipMalloc(&ctr, 4)); CHK(hipMemset(ctr, 0, 4));
        int it = iters; int nn = n;
        void *args[] = {&a, &nn, &it};
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0, s));
            CHK(hipLaunchCooperativeKernel((const void *)k_coop, dim3(wgs), dim3(512), args, 0, s));
            CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s)); CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%d workgroups x 512: cooperative grid.sync  %.2f us per iteration\n", wgs, ms * 1e3 / iters);
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipMemsetAsync(ctr, 0, 4, s));
            CHK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_manual, dim3(wgs), dim3(512), 0, s, a, n, iters, ctr);
            CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s)); CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%d workgroups x 512: atomic-counter barrier  %.2f us per iteration\n", wgs, ms * 1e3 / iters);
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(512), 0, s, a, n);
        CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0, s)); CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
            CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%d workgroups x 512: one kernel per iteration, hipGraph  %.2f us per iteration\n", wgs, ms * 1e3 / iters);
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); CHK(hipFree(a)); CHK(hipFree(ctr));
    }
    return 0;
}

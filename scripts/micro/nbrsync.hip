// Micro-benchmark: what does ONE sub-step of a resident kernel cost in synchronisation, when a workgroup (patch) only waits for
// its neighbouring patches instead of for the whole grid?  Each workgroup, per iteration: "publishes" NB doubles for its
// neighbours (write-through sc1 stores), drains them, raises its own counter; waits until its K neighbours (ids +-1 .. +-K/2)
// have published this iteration; reads NB doubles from each side with sc1 loads.  Two parities of the exchange buffer.
//   hipcc --offload-arch=gfx950 -O3 -o nbrsync nbrsync.hip && ./nbrsync
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int K, int WORK>
__global__ void __launch_bounds__(512) k_nbr(double *xbuf /*[2][nwg][NB]*/, unsigned int *flag /*[nwg*32]*/, int NB, int iters, double *out, int *err) {
    const int b = blockIdx.x, n = gridDim.x, t = threadIdx.x;
    double acc = (double)b;
    for (int it = 0; it < iters; ++it) {
        // dummy "element + node phase"
        double w = acc;
#pragma unroll 1
        for (int q = 0; q < WORK; ++q) w = w * 1.0000001 + 0.5;
        acc = w;
        // publish
        double *mine = xbuf + ((size_t)(it & 1) * n + b) * NB;
        if (t < NB) st_agent(mine + t, acc + t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(flag + 32 * b, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the neighbours
        if (t < K) {
            const int d = (t < K / 2) ? -(t + 1) : (t - K / 2 + 1);
            const int nb = ((b + d) % n + n) % n;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag + 32 * nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ll) { atomicExch(err, 1); break; }  // 2 s
            }
        }
        __syncthreads();
        // read what they published
        if (t < K * 8 && t / 8 < K) {
            const int k = t / 8, d = (k < K / 2) ? -(k + 1) : (k - K / 2 + 1);
            const int nb = ((b + d) % n + n) % n;
            const double *theirs = xbuf + ((size_t)(it & 1) * n + nb) * NB;
            double s = 0.;
            for (int j = t % 8; j < NB; j += 8) s += ld_agent(theirs + j);
            acc += 1e-9 * s;
        }
        __syncthreads();
    }
    if (t == 0) out[b] = acc;
}

__global__ void __launch_bounds__(512) k_plain(double *a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * 0.5 + 1.0;
}

int main() {
    CHK(hipSetDevice(0));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 1200, NB = 48;
    for (int wgs : {255, 511}) {
        double *xbuf, *out; unsigned int *flag; int *err;
        CHK(hipMalloc(&xbuf, 2ull * wgs * NB * sizeof(double))); CHK(hipMalloc(&out, wgs * sizeof(double)));
        CHK(hipMalloc(&flag, wgs * 32 * sizeof(unsigned))); CHK(hipMalloc(&err, 4));
        float ms;
#define RUN(KK, WW) \
        for (int rep = 0; rep < 2; ++rep) { \
            CHK(hipMemset(flag, 0, wgs * 32 * sizeof(unsigned))); CHK(hipMemset(err, 0, 4)); CHK(hipDeviceSynchronize()); \
            CHK(hipEventRecord(e0, s)); \
            hipLaunchKernelGGL((k_nbr<KK, WW>), dim3(wgs), dim3(512), 0, s, xbuf, flag, NB, iters, out, err); \
            CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s)); CHK(hipEventElapsedTime(&ms, e0, e1)); } \
        { int herr = 0; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); \
          printf("%d workgroups x 512, %d neighbours, %d dependent FMAs of work: %.2f us per iteration%s\n", wgs, KK, WW, ms * 1e3 / iters, herr ? "  (TIMEOUT)" : ""); }
        RUN(6, 0) RUN(6, 500) RUN(6, 1500) RUN(8, 0)
        (void)hipFree(xbuf); (void)hipFree(out); (void)hipFree(flag); (void)hipFree(err);
    }
    return 0;
}

// Micro-benchmark: does it matter where a kernel's ~1 KB of arguments live?  (a) passed by value (kernarg segment written by the
// runtime for every launch), (b) one pointer to a struct that stays in device memory.  Every workgroup needs a pointer from the
// arguments before it can issue its first load -- the head of the dependent chain of the fused sub-step kernel.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg kernarg.hip && ./kernarg
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { const double *p[120]; int n; int pad; };

__global__ void __launch_bounds__(512) k_by_value(Big a, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.;
    for (int k = 0; k < 120; k += 17) s += a.p[k][i % a.n];
    out[i] = s;
}
__global__ void __launch_bounds__(512) k_by_pointer(const Big *a, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.;
    for (int k = 0; k < 120; k += 17) s += a->p[k][i % a->n];
    out[i] = s;
}

int main() {
    CHK(hipSetDevice(0));
    hipStream_t s; CHK(hipStreamCreate(&s));
    const int wgs = 512, n = wgs * 512, iters = 2000;
    double *buf, *out; CHK(hipMalloc(&buf, n * sizeof(double))); CHK(hipMalloc(&out, n * sizeof(double))); CHK(hipMemset(buf, 0, n * sizeof(double)));
    Big h; for (int k = 0; k < 120; ++k) h.p[k] = buf; h.n = n; h.pad = 0;
    Big *d; CHK(hipMalloc(&d, sizeof(Big))); CHK(hipMemcpy(d, &h, sizeof(Big), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < iters; ++k) {
            if (mode == 0) hipLaunchKernelGGL(k_by_value, dim3(wgs), dim3(512), 0, s, h, out);
            else hipLaunchKernelGGL(k_by_pointer, dim3(wgs), dim3(512), 0, s, (const Big *)d, out);
        }
        CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, s)); CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
            CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%s: %.3f us per launch (512 workgroups x 512 threads, hipGraph of %d launches)\n", mode == 0 ? "968-byte struct by value " : "pointer to device struct ", ms * 1e3 / iters, iters);
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
    return 0;
}

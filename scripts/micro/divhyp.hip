// Are a shared-reciprocal division and a plain sqrt(fma) hypot bit-identical to the compiler's fp64 '/' and to ocml's
// hypot on the operand ranges of the sub-step kernel?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off divhyp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>

__device__ __forceinline__ double recip_refined(double d) {  // the reciprocal the compiler's fdiv expansion builds (no scaling)
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-d, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
__device__ __forceinline__ double div_shared(double n, double d, double r) {
    const double q0 = n * r;
    const double rem = __builtin_fma(-d, q0, n);
    return __builtin_fma(rem, r, q0);
}

__global__ void k(const double *a, const double *b, int n, unsigned long long *bad_div, unsigned long long *bad_h1, unsigned long long *bad_h2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    const double q_ref = x / y;
    const double q = div_shared(x, y, recip_refined(y));
    if (__double_as_longlong(q) != __double_as_longlong(q_ref)) atomicAdd(bad_div, 1ull);
    const double h_ref = hypot(x, y);
    const double h1 = sqrt(__builtin_fma(x, x, y * y));
    const double ax = fabs(x), ay = fabs(y), mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double h2 = sqrt(__builtin_fma(mx, mx, mn * mn));
    if (__double_as_longlong(h1) != __double_as_longlong(h_ref)) atomicAdd(bad_h1, 1ull);
    if (__double_as_longlong(h2) != __double_as_longlong(h_ref)) atomicAdd(bad_h2, 1ull);
}

int main() {
    const int n = 1 << 24;
    std::vector<double> a(n), b(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1., 1.), E(-30., 30.);
    for (int i = 0; i < n; ++i) {
        const int cls = i & 3;
        if (cls == 0) { a[i] = 3000. * U(rng); b[i] = 1e7 * U(rng); }                 // coordinate difference / jacobian
        else if (cls == 1) { a[i] = U(rng) * std::pow(10., E(rng)); b[i] = U(rng) * std::pow(10., E(rng)); }  // wide exponents
        else if (cls == 2) { a[i] = 0.3 * U(rng); b[i] = 0.3 * U(rng); }              // velocities
        else { a[i] = 1e5 * U(rng); b[i] = 1e5 * U(rng); if ((i & 63) == 3) a[i] = 0.; }  // stresses, some exact zeros
    }
    double *da, *db; unsigned long long *dc, hc[3] = {0, 0, 0};
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, 24);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemset(dc, 0, 24);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, n, dc, dc + 1, dc + 2);
    hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
    // and against the host's correctly rounded results
    printf("n=%d  division mismatches vs '/': %llu   hypot: sqrt(fma(x,x,y*y)) vs ocml %llu, max/min order vs ocml %llu\n", n, hc[0], hc[1], hc[2]);
    return 0;
}

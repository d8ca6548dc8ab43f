// accuracy of v_rsq_f64 and of one / two Newton steps behind it (gfx950): max relative error over random positive doubles
//   hipcc -O3 --offload-arch=gfx950 -o tools/rsq_probe tools/rsq_probe.hip && tools/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rsq(d);
    r0[i] = r;
    double h = 0.5 * d * r;
    r = fma(r, fma(-h, r, 0.5), r);
    r1[i] = r;
    h = 0.5 * d * r;
    r2[i] = fma(r, fma(-h, r, 0.5), r);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-20.0, 20.0);
    for (auto& v : x) v = std::exp2(u(g)) * (1.0 + 0.5 * u(g) / 20.0);
    for (auto& v : x) v = std::fabs(v) + 1e-300;
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
    }
    printf("max rel err: v_rsq_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
    return 0;
}

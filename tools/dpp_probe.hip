// v_fmac_f64_dpp row_newbcast on gfx950: is a 64-lane broadcast mat-vec cheaper through DP-ALU DPP than through LDS broadcast reads?
//   hipcc -O3 --offload-arch=gfx950 -o tools/dpp_probe tools/dpp_probe.hip && tools/dpp_probe
// Each lane l holds row l of a 64 x 64 matrix (64 doubles) and x_l; y_l = sum_c M[l][c] x_c, repeated ITERS times (x <- f(y)).
//   variant 0: x through LDS: 1 ds_write_b64 + 32 broadcast ds_read_b128 + 64 v_fmac_f64        (what admm_wave_body does)
//   variant 1: x through 8 ds_bpermute_b32 (every 16-lane row gets the four 16-value chunks) + 64 v_fmac_f64_dpp row_newbcast
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITERS = 2000;

template <int J>
__device__ __forceinline__ void fmac_bcast(double& acc, double chunk, double k) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(chunk), "v"(k), "n"(J));
}
template <int C, int J>
struct Acc {
    static __device__ __forceinline__ void run(double (&acc)[4], const double (&ch)[4], const double (&m)[64]) {
        fmac_bcast<J>(acc[J & 3], ch[C], m[16 * C + J]);
        if constexpr (J + 1 < 16) Acc<C, J + 1>::run(acc, ch, m);
    }
};
__device__ __forceinline__ double bperm64(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int VAR>
__global__ __launch_bounds__(64, 2) void kern(const double* M, const double* x0, double* y, long long* cyc) {
    __shared__ __attribute__((aligned(16))) double xs[64];
    const int l = threadIdx.x;
    double m[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) m[c] = M[(size_t)blockIdx.x * 4096 + l * 64 + c];
    double x = x0[blockIdx.x * 64 + l];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (VAR == 0) {
            xs[l] = x;
            asm volatile("" ::: "memory");
            const double2* xv = reinterpret_cast<const double2*>(xs);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xv[8 * b + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[(2 * i) & 3] = fma(m[16 * b + 2 * i], v[i].x, acc[(2 * i) & 3]); acc[(2 * i + 1) & 3] = fma(m[16 * b + 2 * i + 1], v[i].y, acc[(2 * i + 1) & 3]); }
            }
            asm volatile("" ::: "memory");
        } else {
            double ch[4];
            const int i = l & 15;
#pragma unroll
            for (int c = 0; c < 4; ++c) ch[c] = bperm64(x, 16 * c + i);
            asm volatile("s_nop 1" ::: "memory");
            Acc<0, 0>::run(acc, ch, m); Acc<1, 0>::run(acc, ch, m); Acc<2, 0>::run(acc, ch, m); Acc<3, 0>::run(acc, ch, m);
        }
        const double yv = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        x = 0.5 * x + 0.5 * yv;             // (M is scaled so that this stays bounded)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    y[blockIdx.x * 64 + l] = x;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int NB = 256 * 8;          // 8 waves per CU (2 per SIMD), as the one-wave kernel runs
    std::vector<double> M((size_t)NB * 4096), x(NB * 64), y0(NB * 64), y1(NB * 64);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (auto& v : M) v = rnd() / 32.0;
    for (auto& v : x) v = rnd();
    double *dM, *dx, *dy; long long* dc;
    CHECK(hipMalloc(&dM, M.size() * 8)); CHECK(hipMalloc(&dx, x.size() * 8)); CHECK(hipMalloc(&dy, x.size() * 8)); CHECK(hipMalloc(&dc, NB * 8));
    CHECK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    std::vector<long long> c(NB);
    for (int grid : {1, NB}) {
        for (int var = 0; var < 2; ++var) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (var == 0) hipLaunchKernelGGL(kern<0>, dim3(grid), dim3(64), 0, 0, dM, dx, dy, dc);
                else hipLaunchKernelGGL(kern<1>, dim3(grid), dim3(64), 0, 0, dM, dx, dy, dc);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(c.data(), dc, grid * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy((var ? y1 : y0).data(), dy, x.size() * 8, hipMemcpyDeviceToHost));
            double mc = 0; for (int i = 0; i < grid; ++i) mc += c[i]; mc /= grid;
            printf("grid %5d variant %d (%s): %.3f ms, %.0f cycles per iteration per wave\n", grid, var, var ? "bpermute + v_fmac_f64_dpp row_newbcast" : "LDS broadcast reads", ms, mc / ITERS);
        }
        double d = 0; for (int i = 0; i < grid * 64; ++i) d = fmax(d, fabs(y0[i] - y1[i]));
        printf("   max |y_lds - y_dpp| = %.3e (|y| ~ %.3e)\n", d, fabs(y0[0]));
    }
    return 0;
}

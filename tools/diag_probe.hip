// 16 x 16 diagonal-tile inversion (S SPD -> W = L^-1, S = L L'): the blocked matrix-core routine the kernels use (diag16_invert_mfma) against the
// column-per-lane DPP elimination (diag16_invert_dpp), one wave alone and eight waves per CU.
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ig1_locomotion_amd/csrc -o tools/diag_probe tools/diag_probe.hip && tools/diag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "srbdqp.h"
#include "srbdqp_common.hpp"
#include "srbdqp_mfma.hpp"
using namespace srbdqp;
template <int VAR>
__global__ __launch_bounds__(64, 2) void k(const double* A, double* W, long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double tile[256];
    const int lane = threadIdx.x, col = lane & 15, g = lane >> 4;
    v4d s;
    for (int r = 0; r < 4; ++r) s[r] = A[(g + 4 * r) * 16 + col];
    bool ok = true;
    v4d w = s;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) {
        v4d in = s;
        in[0] += 1e-30 * w[0];      // dependency between repetitions
        if constexpr (VAR == 0) w = diag16_invert_mfma(in, lane, ok);
        else w = diag16_invert_dpp(in, lane, ok, tile);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0) for (int r = 0; r < 4; ++r) W[(g + 4 * r) * 16 + col] = w[r];
    if (lane == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = ok; }
}
int main() {
    std::vector<double> A(256), W(256);
    unsigned s = 7; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    std::vector<double> G(256); for (auto& v : G) v = rnd();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double a = (i == j) ? 4.0 : 0.0; for (int k2 = 0; k2 < 16; ++k2) a += G[i * 16 + k2] * G[j * 16 + k2]; A[i * 16 + j] = a; }
    double *dA, *dW; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dW, 2048); hipMalloc(&dc, 16);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    const int reps = 1000;
    for (int var = 0; var < 2; ++var) for (int grid : {1, 2048}) {
        for (int it = 0; it < 2; ++it) {
            if (var == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, dA, dW, dc, reps);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, dA, dW, dc, reps);
        }
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost); hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost);
        // check W = L^-1: W A W' = I
        double err = 0, up = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double v = 0;
            for (int p = 0; p <= i; ++p) for (int q = 0; q <= j; ++q) v += W[i * 16 + p] * A[p * 16 + q] * W[j * 16 + q];
            err = fmax(err, fabs(v - (i == j ? 1.0 : 0.0)));
            if (j > i) up = fmax(up, fabs(W[i * 16 + j]));
        }
        printf("%s, %4d workgroups (%s): %.0f cycles per 16x16 inversion, ok=%lld, |W A W' - I| = %.2e, max above the diagonal %.1e\n", var ? "diag16_invert_dpp " : "diag16_invert_mfma", grid,
               grid == 1 ? "one wave alone" : "8 waves per CU", (double)c[0] / reps, c[1], err, up);
    }
    return 0;
}

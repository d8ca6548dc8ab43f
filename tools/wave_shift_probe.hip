// Do the GFX9 whole-wave DPP shifts (wave_shl:1 / wave_shr:1) exist on gfx950, what do they move, and what does a dependent pair cost next to ds_bpermute?
//   hipcc -O3 --offload-arch=gfx950 -o tools/wave_shift_probe tools/wave_shift_probe.hip && tools/wave_shift_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int wshl1(int v) { int r; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int wshr1(int v) { int r; asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v)); return r; }
__global__ void k(int* out, long long* cyc, int reps) {
    const int lane = threadIdx.x;
    out[lane] = wshl1(lane * 10 + 1);
    out[64 + lane] = wshr1(lane * 10 + 1);
    out[128 + lane] = wshl1(wshl1(lane * 10 + 1));
    int v = lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) { v = wshl1(v) + 1; v = wshl1(v) + 1; }
    long long t1 = __builtin_amdgcn_s_memtime();
    int u = lane;
    for (int i = 0; i < reps; ++i) { u = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, u) + 1; u = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, u) + 1; }
    long long t2 = __builtin_amdgcn_s_memtime();
    out[192 + lane] = v + u;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    int* d; long long* c; hipMalloc(&d, 256 * 4); hipMalloc(&c, 16);
    const int reps = 1000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, reps); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, reps);
    hipDeviceSynchronize();
    int h[256]; long long hc[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    printf("wave_shl:1 lane i <- "); for (int i : {0, 1, 14, 15, 16, 31, 32, 47, 48, 62, 63}) printf("[%d]=%d ", i, h[i]); printf("\n");
    printf("wave_shr:1 lane i <- "); for (int i : {0, 1, 14, 15, 16, 31, 32, 47, 48, 62, 63}) printf("[%d]=%d ", i, h[64 + i]); printf("\n");
    printf("shl twice  lane i <- "); for (int i : {0, 1, 14, 15, 16, 31, 32, 47, 48, 62, 63}) printf("[%d]=%d ", i, h[128 + i]); printf("\n");
    printf("dependent pair: wave shift %.1f cycles, ds_bpermute %.1f cycles\n", (double)hc[0] / reps / 2, (double)hc[1] / reps / 2);
    return 0;
}

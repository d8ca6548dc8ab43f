// gfx950's v_permlane16_swap / v_permlane32_swap: what they move, and the four 16-lane rows of a register broadcast to all rows with them
// (the chunk fetch of the one-wave K^-1 product) against four ds_bpermute.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o tools/permlane_probe tools/permlane_probe.hip && tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rows4(unsigned v, unsigned (&ch)[4]) {
    // a = b = v -> permlane32_swap: a = [R0 R1 R0 R1], b = [R2 R3 R2 R3] -> permlane16_swap of each with its own copy: [R0 x4], [R1 x4], [R2 x4], [R3 x4]
    v2u ab = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v2u p = __builtin_amdgcn_permlane16_swap(ab[0], ab[0], false, false);
    v2u q = __builtin_amdgcn_permlane16_swap(ab[1], ab[1], false, false);
    ch[0] = p[0]; ch[1] = p[1]; ch[2] = q[0]; ch[3] = q[1];
}
__global__ void k(unsigned* out, long long* cyc, int reps) {
    const int lane = threadIdx.x;
    unsigned ch[4];
    rows4(lane * 10 + 1, ch);
    for (int c = 0; c < 4; ++c) out[64 * c + lane] = ch[c];
    unsigned v = lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) { rows4(v, ch); v = ch[0] + ch[1] + ch[2] + ch[3] + lane; }
    long long t1 = __builtin_amdgcn_s_memtime();
    unsigned u = lane;
    const int li = lane & 15;
    for (int i = 0; i < reps; ++i) {
        unsigned c0 = __builtin_amdgcn_ds_bpermute(li << 2, u), c1 = __builtin_amdgcn_ds_bpermute((16 + li) << 2, u), c2 = __builtin_amdgcn_ds_bpermute((32 + li) << 2, u), c3 = __builtin_amdgcn_ds_bpermute((48 + li) << 2, u);
        u = c0 + c1 + c2 + c3 + lane;
    }
    long long t2 = __builtin_amdgcn_s_memtime();
    out[256 + lane] = v; out[320 + lane] = u;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    unsigned* d; long long* c; hipMalloc(&d, 384 * 4); hipMalloc(&c, 16);
    const int reps = 1000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, reps); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, reps);
    hipDeviceSynchronize();
    unsigned h[384]; long long hc[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int cc = 0; cc < 4; ++cc) for (int l = 0; l < 64; ++l) ok = ok && (h[64 * cc + l] == (unsigned)((16 * cc + (l & 15)) * 10 + 1));
    for (int cc = 0; cc < 4; ++cc) { printf("ch[%d]: ", cc); for (int l : {0, 1, 15, 16, 17, 32, 47, 48, 63}) printf("[%d]=%u ", l, h[64 * cc + l]); printf("\n"); }
    printf("row broadcast correct: %d; same results both ways: %d\n", (int)ok, (int)(h[256 + 5] == h[320 + 5] && h[256 + 40] == h[320 + 40]));
    printf("dependent 4-row broadcast of one dword: permlane swaps %.1f cycles, 4 x ds_bpermute %.1f cycles\n", (double)hc[0] / reps, (double)hc[1] / reps);
    return 0;
}

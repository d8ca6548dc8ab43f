// Diagnostic micro-benchmarks (one wave, dependent chains) to calibrate the latency model used in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#define PIN(v) asm volatile("" : "+v"(v))
#define T0() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define T1() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ double rl(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ void mb(double* out, long long* cyc, double seed) {
    __shared__ double sm[1024];
    unsigned long long t0, t1;
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-3, y = seed * 0.5;
    sm[threadIdx.x] = x; sm[threadIdx.x + 256] = y;
    __syncthreads();
    // 0: dependent v_fma_f64 chain (64)
    PIN(x); PIN(y);
    T0();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fma(x, 0.999, y);
    PIN(x);
    T1(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // 1: 4 independent chains of 16 (64 fmas)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    PIN(a0); PIN(a1); PIN(a2); PIN(a3);
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0 = fma(a0, 0.999, y); a1 = fma(a1, 0.999, y); a2 = fma(a2, 0.999, y); a3 = fma(a3, 0.999, y); }
    PIN(a0); PIN(a1); PIN(a2); PIN(a3);
    T1(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    x = a0 + a1 + a2 + a3;
    // 2: dependent ds_read_b64 chain (16): address depends on previous value
    int idx = lane & 7;
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) { double v = sm[idx]; idx = ((int)v) & 7; }
    T1(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    x += idx;
    // 3: dependent ds_bpermute chain (16 x f64 = 32 bpermutes)
    PIN(x);
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int lo = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __double2loint(x));
        int hi = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __double2hiint(x));
        x = __hiloint2double(hi, lo) + 1.0;
    }
    PIN(x);
    T1(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // 4: readlane -> fma chain (16)
    PIN(x);
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) { double s = rl(x, i); x = fma(s, 0.5, x); }
    PIN(x);
    T1(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // 5: dpp swap -> add chain (16)
    PIN(x);
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
        x = x + __hiloint2double(hi, lo) * 0.5;
    }
    PIN(x);
    T1(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // 6: 16 barriers (4 waves)
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) __syncthreads();
    T1(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // 7: rcp f64 + 2 newton, dependent chain of 8
    PIN(x);
    T0();
#pragma unroll
    for (int i = 0; i < 8; ++i) { double r = __builtin_amdgcn_rcp(x); double e = fma(-x, r, 1.0); r = fma(r, e, r); e = fma(-x, r, 1.0); x = fma(r, e, r) + 2.0; }
    PIN(x);
    T1(); if (threadIdx.x == 0) cyc[7] = t1 - t0;
    // 8: 16 broadcast ds_read_b128 issued back to back, one wait
    T0();
    double2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = reinterpret_cast<const double2*>(sm)[i + (lane >> 6)];
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    T1(); if (threadIdx.x == 0) cyc[8] = t1 - t0;
    // 9: dependent MFMA f64 chain (16)
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d acc = {x, s, x, s};
    PIN(x); PIN(s);
    T0();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, s, acc, 0, 0, 0);
    { double q0 = acc[0]; PIN(q0); acc[0] = q0; }
    T1(); if (threadIdx.x == 0) cyc[9] = t1 - t0;
    // 10: 16 independent MFMA (4 accumulators x 4)
    v4d b0 = acc, b1 = acc, b2 = acc, b3 = acc;
    T0();
#pragma unroll
    for (int i = 0; i < 4; ++i) { b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, s, b0, 0, 0, 0); b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, s, b1, 0, 0, 0); b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, s, b2, 0, 0, 0); b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, s, b3, 0, 0, 0); }
    { double q0 = b0[0], q1 = b1[0], q2 = b2[0], q3 = b3[0]; PIN(q0); PIN(q1); PIN(q2); PIN(q3); b0[0] = q0; b1[0] = q1; b2[0] = q2; b3[0] = q3; }
    T1(); if (threadIdx.x == 0) cyc[10] = t1 - t0;
    out[threadIdx.x] = x + s + acc[0] + b0[1] + b1[2] + b2[3] + b3[0];
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 16 * 8); hipMemset(cyc, 0, 128);
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(mb, dim3(1), dim3(256), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    const char* nm[] = {"dependent v_fma_f64 x64", "4 indep chains x16 (64 fma)", "dependent ds_read_b64 x16", "dependent bpermute f64 x16", "readlane->fma x16", "dpp swap->fma x16", "s_barrier x16 (4 waves)", "rcp+2newton chain x8", "16 bcast ds_read_b128 + sum", "dependent mfma f64 x16", "16 mfma f64, 4 accumulators"};
    const int cnt[] = {64, 64, 16, 16, 16, 16, 16, 8, 1, 16, 16};
    for (int i = 0; i < 11; ++i) printf("%-32s total %6lld cyc   per op %.1f\n", nm[i], h[i], (double)(h[i] - 40) / cnt[i]);
    return 0;
}

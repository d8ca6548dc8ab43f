// pcie_probe.hip -- what a kernel pays for touching GPU-mapped host memory (the staging slab of the batch-1 path).
//   hipcc -O3 --offload-arch=gfx950 -o tools/pcie_probe tools/pcie_probe.hip && tools/pcie_probe
// One 256-thread workgroup, times from the device's 100 MHz clock (s_memrealtime):
//   rd1      one 8-byte load by one lane (PCIe round trip)
//   rd263    263 doubles, one per lane (the input block of an N = 10 QP), one round
//   rd263v   the same bytes as 16-byte loads (half the lanes)
//   dep3     three dependent rounds of rd1
//   wr263    263 doubles stored, then __threadfence_system() (outputs + visibility fence)
//   wr1f     one store + fence
// each for host-mapped (coherent) memory and for device memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__global__ void probe(const double* src, double* dst, long long* out, double* sink) {
    __shared__ double s[1024];
    const int t = threadIdx.x;
    long long t0, t1;
    double acc = 0.0;
    // rd1
    __syncthreads(); t0 = now();
    if (t == 0) acc += __builtin_nontemporal_load(src);
    s[t] = acc; __syncthreads(); t1 = now(); if (t == 0) out[0] = t1 - t0;
    // rd263
    __syncthreads(); t0 = now();
    for (int i = t; i < 263; i += 256) acc += __builtin_nontemporal_load(src + 512 + i);
    s[t] = acc; __syncthreads(); t1 = now(); if (t == 0) out[1] = t1 - t0;
    // rd263v
    __syncthreads(); t0 = now();
    if (t < 132) { const double2 v = *reinterpret_cast<const double2*>(src + 1024 + 2 * t); acc += v.x + v.y; }
    s[t] = acc; __syncthreads(); t1 = now(); if (t == 0) out[2] = t1 - t0;
    // dep3
    __syncthreads(); t0 = now();
    if (t == 0) {
        int idx = (int)src[2048];
        idx = (int)src[2048 + 64 + idx];
        acc += src[2048 + 128 + idx];
    }
    s[t] = acc; __syncthreads(); t1 = now(); if (t == 0) out[3] = t1 - t0;
    // wr263 + fence
    __syncthreads(); t0 = now();
    for (int i = t; i < 263; i += 256) dst[i] = acc + i;
    __threadfence_system();
    __syncthreads(); t1 = now(); if (t == 0) out[4] = t1 - t0;
    // wr1 + fence
    __syncthreads(); t0 = now();
    if (t == 0) dst[512] = acc;
    __threadfence_system();
    __syncthreads(); t1 = now(); if (t == 0) out[5] = t1 - t0;
    // wr263, no fence (issue cost only)
    __syncthreads(); t0 = now();
    for (int i = t; i < 263; i += 256) dst[1024 + i] = acc + i;
    __syncthreads(); t1 = now(); if (t == 0) out[6] = t1 - t0;
    __threadfence_system();
    sink[t] = acc;
}

__global__ void readback(const double* src, double* dst) {
    for (int i = threadIdx.x; i < 288; i += 256) dst[i] = src[i];
}

int main() {
    const size_t nd = 4096;
    double *hsrc, *hdst, *dsrc, *ddst, *sink; long long* out;
    CK(hipHostMalloc((void**)&hsrc, nd * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&hdst, nd * 8, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipMalloc((void**)&dsrc, nd * 8)); CK(hipMalloc((void**)&ddst, nd * 8)); CK(hipMalloc((void**)&sink, 256 * 8));
    CK(hipHostMalloc((void**)&out, 64 * 8, hipHostMallocMapped));
    memset(hsrc, 0, nd * 8); CK(hipMemset(dsrc, 0, nd * 8));
    const char* names[7] = {"rd1", "rd263", "rd263v", "dep3", "wr263+fence", "wr1+fence", "wr263 issue"};
    for (int where = 0; where < 2; ++where) {
        std::vector<long long> samples[7];
        for (int rep = 0; rep < 60; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, where ? dsrc : hsrc, where ? ddst : hdst, out, sink);
            CK(hipDeviceSynchronize());
            if (rep >= 10) for (int i = 0; i < 7; ++i) samples[i].push_back(out[i]);
        }
        printf("%s memory:", where ? "device" : "host-mapped");
        for (int i = 0; i < 7; ++i) {
            std::sort(samples[i].begin(), samples[i].end());
            printf("  %s %.2f us", names[i], samples[i][samples[i].size() / 2] * 0.01);
        }
        printf("\n");
    }
    // can the CPU store straight into fine-grained device memory (large BAR)?  (run last: may fault)
    double* fg = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fg, nd * 8, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
        e = hipPointerGetAttributes(&at, fg);
        printf("  attributes: %s type %d host %p device %p\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer);
        fflush(stdout);
        {   // the same kernel probe with fine-grained device memory as the source (what a BAR-fed input stage would read)
            CK(hipMemset(fg, 0, nd * 8));
            std::vector<long long> samples[7];
            for (int rep = 0; rep < 60; ++rep) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, fg, ddst, out, sink);
                CK(hipDeviceSynchronize());
                if (rep >= 10) for (int i = 0; i < 7; ++i) samples[i].push_back(out[i]);
            }
            printf("fine-grained device memory (reads):");
            for (int i = 0; i < 4; ++i) { std::sort(samples[i].begin(), samples[i].end()); printf("  %s %.2f us", names[i], samples[i][samples[i].size() / 2] * 0.01); }
            printf("\n");
        }
        if (getenv("PCIE_PROBE_TOUCH")) {
            fg[0] = 42.0; fg[1] = 1.0;                       // CPU store through the BAR
            CK(hipMemcpy(hsrc, fg, 16, hipMemcpyDeviceToHost));
            printf("  CPU store visible to the device: %g %g\n", hsrc[0], hsrc[1]);
            {   // does a later kernel see what the CPU wrote over the same addresses (no stale cache line)?
                int stale = 0;
                for (int rep = 0; rep < 200; ++rep) {
                    for (int i = 0; i < 288; ++i) fg[512 + i] = (double)(rep * 1000 + i);
                    __builtin_ia32_sfence();
                    hipLaunchKernelGGL(readback, dim3(1), dim3(256), 0, 0, fg + 512, hdst);
                    CK(hipDeviceSynchronize());
                    for (int i = 0; i < 288; ++i) stale += (hdst[i] != (double)(rep * 1000 + i));
                }
                printf("  200 x (CPU overwrites 288 doubles through the BAR, kernel reads them): %d stale values\n", stale);
            }
            // what the CPU pays for writing one QP's inputs (2.3 KB) there, and for reading 40 bytes back
            std::vector<double> src(288, 1.0);
            auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 2000; ++rep) { memcpy(fg + (rep & 7) * 288, src.data(), 2304); __builtin_ia32_sfence(); }
            auto t1 = std::chrono::steady_clock::now();
            volatile double sink = 0.0;
            for (int rep = 0; rep < 200; ++rep) for (int i = 0; i < 5; ++i) sink += ((volatile double*)fg)[i + 8 * (rep & 7)];
            auto t2 = std::chrono::steady_clock::now();
            printf("  CPU memcpy of 2304 B into it + sfence: %.3f us; CPU read of 40 B from it: %.3f us\n",
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t2 - t1).count() / 200);
        }
    }
    return 0;
}

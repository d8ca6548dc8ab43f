// Diagnostic: how many workgroups of the hot-path kernels fit per CU (occupancy API + a census kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "srbdqp.h"
#include "../g1_locomotion_amd/csrc/srbdqp_common.hpp"
#include "../g1_locomotion_amd/csrc/srbdqp_gj.hpp"
#include "../g1_locomotion_amd/csrc/srbdqp_mfma.hpp"

__global__ void census(int* counts, int* maxc, int lds_dummy) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        unsigned cu;  // HW_REG_HW_ID: cu_id bits 8..11, sh 12, se 13..15 ; use xcc id too
        unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // XCC_ID bits 0..3
        cu = ((xcc & 15) << 12) | ((hwid >> 8) & 0xFFF);
        int slot = cu & 0xFFFF;
        int c = atomicAdd(&counts[slot], 1) + 1;
        atomicMax(&maxc[slot], c);
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 2000000) { }   // ~20 ms at 100 MHz
        atomicSub(&counts[slot], 1);
    }
    smem[threadIdx.x] = 0;
}

int main() {
    int nb = 0;
    size_t lds = srbdqp::MfmaTraits<10>::lds_bytes;
    hipFuncSetAttribute((const void*)&srbdqp::srbdqp_mfma_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, srbdqp::srbdqp_mfma_kernel<10>, 256, lds);
    printf("mfma<10>: lds=%zu B  occupancy API blocks/CU = %d\n", lds, nb);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)&srbdqp::srbdqp_mfma_kernel<10>);
    printf("  numRegs=%d sharedSizeBytes=%zu maxDyn=%d\n", fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: CUs=%d sharedMemPerBlock=%zu sharedMemPerMultiprocessor=%zu maxSharedMemoryPerMultiProcessor=%zu regsPerBlock=%d clock=%d kHz\n",
           p.multiProcessorCount, p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.clockRate);
    for (int ldsk : {16, 40, 52, 64, 78, 80, 96}) {
        int* counts; int* maxc;
        hipMalloc(&counts, 65536 * 4); hipMalloc(&maxc, 65536 * 4);
        hipMemset(counts, 0, 65536 * 4); hipMemset(maxc, 0, 65536 * 4);
        hipFuncSetAttribute((const void*)&census, hipFuncAttributeMaxDynamicSharedMemorySize, ldsk * 1024);
        hipLaunchKernelGGL(census, dim3(4096), dim3(256), ldsk * 1024, 0, counts, maxc, 0);
        hipDeviceSynchronize();
        std::vector<int> h(65536);
        hipMemcpy(h.data(), maxc, 65536 * 4, hipMemcpyDeviceToHost);
        int used = 0, mx = 0; for (int v : h) { if (v) ++used; if (v > mx) mx = v; }
        printf("census: %d KB LDS/WG, 256 threads -> distinct CU slots %d, max concurrent WGs on a CU %d\n", ldsk, used, mx);
        hipFree(counts); hipFree(maxc);
    }
    return 0;
}

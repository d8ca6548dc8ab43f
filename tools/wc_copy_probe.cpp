// How fast the host can put 2952 bytes of kernel arguments into device memory through the PCIe BAR (write-combining mapping): memcpy, 32-byte and 64-byte
// non-temporal stores, each followed by sfence.   hipcc -O3 -o tools/wc_copy_probe tools/wc_copy_probe.cpp -lhsa-runtime64   (host code only)
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static hsa_agent_t g_gpu, g_cpu; static bool hg = false, hc = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) { hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t); if (t == HSA_DEVICE_TYPE_GPU && !hg) { g_gpu = a; hg = true; } if (t == HSA_DEVICE_TYPE_CPU && !hc) { g_cpu = a; hc = true; } return HSA_STATUS_SUCCESS; }
static hsa_amd_memory_pool_t g_pool; static bool hp = false;
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void*) { hsa_amd_segment_t s; uint32_t f = 0; bool al = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &s); hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &f); hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &al); if (!hp && s == HSA_AMD_SEGMENT_GLOBAL && (f & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && al) { g_pool = p; hp = true; } return HSA_STATUS_SUCCESS; }

__attribute__((target("avx2"))) static void copy32(char* d, const char* s, size_t n) { for (size_t i = 0; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)(d + i), _mm256_loadu_si256((const __m256i*)(s + i))); }
__attribute__((target("avx512f"))) static void copy64(char* d, const char* s, size_t n) { for (size_t i = 0; i + 64 <= n; i += 64) _mm512_stream_si512((__m512i*)(d + i), _mm512_loadu_si512((const void*)(s + i))); }

int main() {
    hsa_init(); hsa_iterate_agents(agent_cb, nullptr); hsa_amd_agent_iterate_memory_pools(g_gpu, pool_cb, nullptr);
    char* k = nullptr; hsa_amd_memory_pool_allocate(g_pool, 16384, 0, (void**)&k);
    if (hsa_amd_agents_allow_access(1, &g_cpu, nullptr, k) != HSA_STATUS_SUCCESS) { printf("no host access\n"); return 1; }
    alignas(64) static char src[3072]; for (int i = 0; i < 3072; ++i) src[i] = (char)i;
    using clk = std::chrono::steady_clock;
    const bool a512 = __builtin_cpu_supports("avx512f");
    for (size_t bytes : {2944, 1536, 576}) {
        for (int mode = 0; mode < (a512 ? 3 : 2); ++mode) {
            std::vector<double> v;
            for (int i = 0; i < 20000; ++i) {
                char* d = k + (i & 3) * 4096;
                const auto t0 = clk::now();
                if (mode == 0) std::memcpy(d, src, bytes); else if (mode == 1) copy32(d, src, bytes); else copy64(d, src, bytes);
                _mm_sfence();
                const auto t1 = clk::now();
                if (i >= 1000) v.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            }
            std::sort(v.begin(), v.end());
            printf("%5zu B  %-28s p50 %.3f us  p90 %.3f us\n", bytes, mode == 0 ? "memcpy" : mode == 1 ? "32-byte non-temporal stores" : "64-byte non-temporal stores", v[v.size() / 2], v[v.size() * 9 / 10]);
        }
    }
    return 0;
}

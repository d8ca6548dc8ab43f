// What a batch-1 call pays for its launch: a kernel with 2952 bytes of by-value arguments (the size of the staged low-latency call: KArgs + StagedIn<10>) that
// publishes a sequence number in GPU-mapped pinned host memory, started (A) by hipLaunchKernelGGL on a stream and (B) by an AQL packet this program writes itself into
// an HSA queue of its own (kernel arguments (i) in the host's kernarg pool, (ii) in device memory written through the PCIe BAR), the host spinning on the word.
//   hipcc -O3 --offload-arch=gfx950 -o tools/aql_probe tools/aql_probe.hip -lhsa-runtime64 && tools/aql_probe [calls]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <elf.h>
#include <immintrin.h>

struct Args {
    int* done;       // pinned host word
    int seq;
    int pad_;
    double* out;     // device
    double v[366];   // 2952 bytes in all
};
static_assert(sizeof(Args) == 2952, "");

extern "C" __global__ void probe_kernel(Args a) {
    if (threadIdx.x == 0) {
        a.out[0] = a.v[365] + a.v[0];
        __hip_atomic_store(a.done, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define HSA_OK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s failed: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool, g_dev_pool; static bool have_kp = false, have_dp = false;
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t fl = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_kp) { g_kernarg_pool = p; have_kp = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t gpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t fl = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    bool alloc = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && alloc && !have_dp) { g_dev_pool = p; have_dp = true; }
    return HSA_STATUS_SUCCESS;
}

// the gfx950 code object inside this executable's .hip_fatbin section (a clang offload bundle)
static std::vector<char> own_code_object() {
    std::ifstream f("/proc/self/exe", std::ios::binary);
    std::vector<char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const Elf64_Ehdr* eh = (const Elf64_Ehdr*)d.data();
    const Elf64_Shdr* sh = (const Elf64_Shdr*)(d.data() + eh->e_shoff);
    const char* names = d.data() + sh[eh->e_shstrndx].sh_offset;
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (std::strcmp(names + sh[i].sh_name, ".hip_fatbin") != 0) continue;
        const char* b = d.data() + sh[i].sh_offset;
        if (std::memcmp(b, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0) { printf("fat binary is not a plain offload bundle\n"); exit(1); }
        uint64_t n; std::memcpy(&n, b + 24, 8);
        const char* p = b + 32;
        for (uint64_t e = 0; e < n; ++e) {
            uint64_t off, sz, tl; std::memcpy(&off, p, 8); std::memcpy(&sz, p + 8, 8); std::memcpy(&tl, p + 16, 8);
            std::string triple(p + 24, tl); p += 24 + tl;
            if (triple.find("gfx950") != std::string::npos && sz > 0) return std::vector<char>(b + off, b + off + sz);
        }
    }
    printf("no gfx950 code object found\n"); exit(1);
}

static double p50(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int calls = argc > 1 ? atoi(argv[1]) : 5000;
    HIP_OK(hipSetDevice(0));
    int* done = nullptr; HIP_OK(hipHostMalloc((void**)&done, 64, hipHostMallocMapped)); *done = 0;
    double* out = nullptr; HIP_OK(hipMalloc((void**)&out, 64));
    hipStream_t st; HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Args a; std::memset(&a, 0, sizeof(a)); a.done = done; a.out = out;
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point t0, clk::time_point t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
    int seq = 0;
    std::vector<double> tot, host;
    // (A) HIP launch
    for (int i = 0; i < calls + 200; ++i) {
        a.seq = ++seq; a.v[365] = (double)seq;
        const auto t0 = clk::now();
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 0, st, a);
        const auto t1 = clk::now();
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) { }
        const auto t2 = clk::now();
        if (i >= 200) { tot.push_back(us(t0, t2)); host.push_back(us(t0, t1)); }
    }
    HIP_OK(hipStreamSynchronize(st));
    printf("A  hipLaunchKernelGGL + spin        : call p50 %.2f us, of which the launch call itself %.2f us\n", p50(tot), p50(host));

    // (B) own AQL queue
    HSA_OK(hsa_init());
    HSA_OK(hsa_iterate_agents(agent_cb, nullptr));
    HSA_OK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
    HSA_OK(hsa_amd_agent_iterate_memory_pools(g_gpu, gpu_pool_cb, nullptr));
    hsa_queue_t* q = nullptr;
    HSA_OK(hsa_queue_create(g_gpu, 64, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, 0, 0, &q));
    std::vector<char> co = own_code_object();
    hsa_code_object_reader_t rd; HSA_OK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &rd));
    hsa_executable_t ex; HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    HSA_OK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    HSA_OK(hsa_executable_freeze(ex, nullptr));
    hsa_executable_symbol_t sym; HSA_OK(hsa_executable_get_symbol_by_name(ex, "probe_kernel.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t kas = 0, gss = 0, pss = 0;
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kas));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gss));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &pss));
    printf("   kernel object %#llx, kernarg %u B, group %u B, private %u B\n", (unsigned long long)kobj, kas, gss, pss);

    const int SLOTS = 16; const size_t SLOT = 4096;
    for (int fence = 2; fence >= 0; --fence)
    for (int mode = 0; mode < 3; ++mode) {
        // 0: host kernarg pool; 1: device memory through the BAR, one fixed slot; 2: device memory, 16 rotating slots
        char* kbuf = nullptr;
        if (mode == 0) { if (!have_kp) continue; HSA_OK(hsa_amd_memory_pool_allocate(g_kernarg_pool, SLOTS * SLOT, 0, (void**)&kbuf)); HSA_OK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kbuf)); }
        else {
            if (!have_dp) continue;
            HSA_OK(hsa_amd_memory_pool_allocate(g_dev_pool, SLOTS * SLOT, 0, (void**)&kbuf));
            hsa_status_t s = hsa_amd_agents_allow_access(1, &g_cpu, nullptr, kbuf);
            if (s != HSA_STATUS_SUCCESS) { printf("B%d device memory is not host-accessible on this box\n", mode); continue; }
        }
        tot.clear(); host.clear();
        int bad = 0;
        for (int i = 0; i < calls + 200; ++i) {
            a.seq = ++seq; a.v[365] = (double)seq; a.v[0] = 1.0;
            char* ka = kbuf + (mode == 2 ? (size_t)(i % SLOTS) * SLOT : 0);
            const auto t0 = clk::now();
            std::memcpy(ka, &a, sizeof(a));
            if (mode != 0) _mm_sfence();
            const uint64_t wi = hsa_queue_add_write_index_relaxed(q, 1);
            hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + (wi & (q->size - 1));
            pk->workgroup_size_x = 256; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1;
            pk->grid_size_x = 256; pk->grid_size_y = 1; pk->grid_size_z = 1;
            pk->private_segment_size = pss; pk->group_segment_size = gss;
            pk->kernel_object = kobj; pk->kernarg_address = ka; pk->reserved2 = 0; pk->completion_signal.handle = 0;
            const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (fence << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                    (fence << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n((uint32_t*)pk, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)wi);
            const auto t1 = clk::now();
            while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) { }
            const auto t2 = clk::now();
            if (i >= 200) { tot.push_back(us(t0, t2)); host.push_back(us(t0, t1)); }
            // the kernel must have seen THIS call's arguments
            HIP_OK(hipMemcpy(&a.v[1], out, 8, hipMemcpyDeviceToHost));
            if (a.v[1] != (double)seq + 1.0) ++bad;
        }
        printf("B%d own AQL queue, fences %s, kernarg %-28s: call p50 %.2f us, of which packet + doorbell %.2f us; stale arguments seen in %d of %d calls\n", mode, fence == 2 ? "system" : fence == 1 ? "agent " : "none  ",
               mode == 0 ? "in the host kernarg pool" : mode == 1 ? "in device memory (one slot)" : "in device memory (16 slots)", p50(tot), p50(host), bad, calls + 200);
    }
    return 0;
}

// The batch-1 call's own way onto the GPU: an HSA user-mode queue the library writes its AQL dispatch packets into itself.
//
// A staged one-QP call (srbdqp_update_f64 / srbdqp_solve_staged_f64, the reference's MPC.update()) is one workgroup whose inputs ride in
// the kernel-argument segment.  Through hipLaunchKernelGGL that launch costs the host 2.8-3.3 us (argument marshalling, the runtime's
// kernel-argument pool with its read-back over PCIe, stream bookkeeping) of a 6.4-7.0 us launch-to-first-store floor; a packet written
// straight into a queue of our own -- 2952 argument bytes stored through the PCIe BAR into device memory, a 64-byte packet, one doorbell
// write -- costs the host 0.4-0.6 us and the floor drops by 1.4 us (tools/aql_probe.hip, profiles/r05_aql_probe.txt).
//
// What is here: the gfx950 code object is taken from the library's own .hip_fatbin section (the file dladdr() names) and loaded a second
// time through the HSA loader, once per process and device; a handle owns a 64-packet queue, four rotating argument slots in device
// memory and one completion signal.  Packets carry the barrier bit and agent-scope acquire / release fences, as the runtime's own
// packets between two kernels of a stream do, so a restart pass queued behind a first pass sees its device-side results.  Only kernels
// without hidden arguments and without a private segment are accepted (the six *_kernel_in instantiations: checked against the code
// object's own descriptors when they are looked up).  Anything that fails at set-up leaves the handle on hipLaunchKernelGGL -- the
// same kernels, the slower door -- and SRBDQP_NO_AQL=1 forces that.
#pragma once
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <dlfcn.h>
#include <elf.h>
#include <immintrin.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace srbdqp {

struct AqlKernel {
    uint64_t object = 0;
    uint32_t kernarg_bytes = 0, group_bytes = 0, private_bytes = 0;
    bool ok = false;
};

// the code object of one device, loaded once per process
struct AqlDevice {
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t dev_pool{};
    hsa_executable_t exe{};
    volatile uint32_t* hdp_flush = nullptr;
    std::vector<char> image;                 // the loader keeps reading it
    std::map<std::string, AqlKernel> kernels;
    std::mutex mu;
    bool ok = false;
    std::string why;
};

namespace aql_detail {

inline bool hsa_ok(hsa_status_t s, const char* what, std::string* why) {
    if (s == HSA_STATUS_SUCCESS || s == HSA_STATUS_INFO_BREAK) return true;
    const char* m = nullptr;
    hsa_status_string(s, &m);
    if (why) *why = std::string(what) + ": " + (m ? m : "HSA error");
    return false;
}

// the gfx950 entry of the clang offload bundle in the .hip_fatbin section of the shared object this function lives in
inline bool own_code_object(std::vector<char>& out, std::string* why) {
    Dl_info di;
    if (!dladdr(reinterpret_cast<void*>(&own_code_object), &di) || !di.dli_fname) { *why = "dladdr() cannot name the library"; return false; }
    std::ifstream f(di.dli_fname, std::ios::binary);
    if (!f) { *why = std::string("cannot read ") + di.dli_fname; return false; }
    std::vector<char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() < sizeof(Elf64_Ehdr) || std::memcmp(d.data(), ELFMAG, SELFMAG) != 0) { *why = "the library is not an ELF file"; return false; }
    const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(d.data());
    if (eh->e_shoff == 0 || eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) > d.size() || eh->e_shstrndx >= eh->e_shnum) { *why = "no section table"; return false; }
    const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(d.data() + eh->e_shoff);
    const char* names = d.data() + sh[eh->e_shstrndx].sh_offset;
    for (int i = 0; i < eh->e_shnum; ++i) {
        if (std::strcmp(names + sh[i].sh_name, ".hip_fatbin") != 0) continue;
        if (sh[i].sh_offset + sh[i].sh_size > d.size() || sh[i].sh_size < 32) break;
        const char* b = d.data() + sh[i].sh_offset;
        if (std::memcmp(b, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0) { *why = "the fat binary is not a plain offload bundle"; return false; }
        uint64_t n;
        std::memcpy(&n, b + 24, 8);
        const char* p = b + 32;
        for (uint64_t e = 0; e < n && p + 24 <= b + sh[i].sh_size; ++e) {
            uint64_t off, sz, tl;
            std::memcpy(&off, p, 8); std::memcpy(&sz, p + 8, 8); std::memcpy(&tl, p + 16, 8);
            if (p + 24 + tl > b + sh[i].sh_size) break;
            const std::string triple(p + 24, (size_t)tl);
            p += 24 + tl;
            if (triple.find("gfx950") != std::string::npos && sz > 0 && off + sz <= sh[i].sh_size) { out.assign(b + off, b + off + sz); return true; }
        }
    }
    *why = "no gfx950 code object in the library's fat binary";
    return false;
}

struct FindAgents { uint32_t bdf, domain; hsa_agent_t gpu, cpu; bool have_gpu = false, have_cpu = false; };
inline hsa_status_t agent_cb(hsa_agent_t a, void* p) {
    auto* f = static_cast<FindAgents*>(p);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) {
        uint32_t bdf = 0, dom = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
        if (((bdf & 0xffffu) >> 3) == ((f->bdf & 0xffffu) >> 3) && dom == f->domain) { f->gpu = a; f->have_gpu = true; }
    }
    return HSA_STATUS_SUCCESS;
}
struct FindPool { hsa_amd_memory_pool_t pool; bool have = false; };
inline hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void* q) {
    auto* f = static_cast<FindPool*>(q);
    hsa_amd_segment_t seg;
    uint32_t fl = 0;
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!f->have && seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && alloc) { f->pool = p; f->have = true; }
    return HSA_STATUS_SUCCESS;
}

}  // namespace aql_detail

// pci_bus / pci_device / pci_domain: of the HIP device the handle runs on (hipDeviceAttributePci*)
inline AqlDevice* aql_device(int pci_domain, int pci_bus, int pci_device) {
    static std::mutex mu;
    static std::map<uint64_t, AqlDevice*> devs;          // never freed: the loader's state must outlive every handle
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t key = ((uint64_t)(uint32_t)pci_domain << 32) | ((uint32_t)pci_bus << 8) | (uint32_t)pci_device;
    auto it = devs.find(key);
    if (it != devs.end()) return it->second;
    AqlDevice* d = new AqlDevice;
    devs[key] = d;
    using namespace aql_detail;
    if (!hsa_ok(hsa_init(), "hsa_init", &d->why)) return d;
    FindAgents fa;
    fa.bdf = ((uint32_t)pci_bus << 8) | ((uint32_t)pci_device << 3);
    fa.domain = (uint32_t)pci_domain;
    if (!hsa_ok(hsa_iterate_agents(agent_cb, &fa), "hsa_iterate_agents", &d->why)) return d;
    if (!fa.have_gpu || !fa.have_cpu) { d->why = "no HSA agent at the HIP device's PCI address"; return d; }
    d->gpu = fa.gpu; d->cpu = fa.cpu;
    FindPool fp;
    if (!hsa_ok(hsa_amd_agent_iterate_memory_pools(d->gpu, pool_cb, &fp), "iterate_memory_pools", &d->why)) return d;
    if (!fp.have) { d->why = "no coarse-grained device pool"; return d; }
    d->dev_pool = fp.pool;
    hsa_amd_hdp_flush_t hdp{};
    if (hsa_agent_get_info(d->gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp) == HSA_STATUS_SUCCESS) d->hdp_flush = hdp.HDP_MEM_FLUSH_CNTL;
    if (!own_code_object(d->image, &d->why)) return d;
    hsa_code_object_reader_t rd;
    if (!hsa_ok(hsa_code_object_reader_create_from_memory(d->image.data(), d->image.size(), &rd), "code_object_reader", &d->why)) return d;
    if (!hsa_ok(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &d->exe), "executable_create", &d->why)) return d;
    if (!hsa_ok(hsa_executable_load_agent_code_object(d->exe, d->gpu, rd, nullptr, nullptr), "load_agent_code_object", &d->why)) return d;
    if (!hsa_ok(hsa_executable_freeze(d->exe, nullptr), "executable_freeze", &d->why)) return d;
    d->ok = true;
    return d;
}

inline const AqlKernel& aql_kernel(AqlDevice* d, const std::string& kd_name, size_t expect_kernarg) {
    std::lock_guard<std::mutex> lk(d->mu);
    auto it = d->kernels.find(kd_name);
    if (it != d->kernels.end()) return it->second;
    AqlKernel& k = d->kernels[kd_name];
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(d->exe, kd_name.c_str(), &d->gpu, &sym) != HSA_STATUS_SUCCESS) return k;
    bool good = hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) == HSA_STATUS_SUCCESS;
    good = good && hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_bytes) == HSA_STATUS_SUCCESS;
    good = good && hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_bytes) == HSA_STATUS_SUCCESS;
    good = good && hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_bytes) == HSA_STATUS_SUCCESS;
    // exactly the explicit arguments (a kernel with hidden arguments has a longer segment) and no scratch memory: nothing here sets those up
    k.ok = good && k.object != 0 && k.kernarg_bytes == expect_kernarg && k.private_bytes == 0;
    return k;
}

class AqlQueue {
public:
    static constexpr int kSlots = 4;
    static constexpr size_t kSlotBytes = 4096;

    static AqlQueue* create(AqlDevice* d, std::string* why) {
        using namespace aql_detail;
        if (!d || !d->ok) { if (why) *why = d ? d->why : "no device"; return nullptr; }
        AqlQueue* q = new AqlQueue;
        q->d_ = d;
        if (!hsa_ok(hsa_queue_create(d->gpu, 64, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, 0, 0, &q->q_), "hsa_queue_create", why)) { delete q; return nullptr; }
        if (!hsa_ok(hsa_amd_memory_pool_allocate(d->dev_pool, kSlots * kSlotBytes, 0, reinterpret_cast<void**>(&q->kbuf_)), "kernarg allocation", why)) { delete q; return nullptr; }
        // the host writes the arguments through the PCIe BAR: refused where device memory is not host-visible
        if (!hsa_ok(hsa_amd_agents_allow_access(1, &d->cpu, nullptr, q->kbuf_), "host access to device memory", why)) { delete q; return nullptr; }
        if (!hsa_ok(hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &q->sig_), "hsa_amd_signal_create", why)) { delete q; return nullptr; }
        q->have_sig_ = true;
        return q;
    }

    ~AqlQueue() {
        wait_end(2000000000ull);
        if (have_sig_) hsa_signal_destroy(sig_);
        if (kbuf_) hsa_amd_memory_pool_free(kbuf_);
        if (q_) hsa_queue_destroy(q_);
    }

    AqlDevice* device() const { return d_; }
    bool in_flight() const { return pending_; }

    // one workgroup of `block` work-items; the argument segment = [a0, n0) followed by [a1, n1) (both multiples of 8 bytes)
    bool launch(const AqlKernel& k, const void* a0, size_t n0, const void* a1, size_t n1, uint32_t block, uint32_t dynamic_lds) {
        if (!k.ok || n0 + n1 != k.kernarg_bytes || n0 + n1 > kSlotBytes) return false;
        char* ka = kbuf_ + (size_t)(slot_++ % kSlots) * kSlotBytes;
        // (through the write-combining mapping of the BAR: whole 32-byte non-temporal stores leave the core in half the time of memcpy()'s -- 0.18 against
        //  0.36 us for 2944 bytes, tools/wc_copy_probe.cpp)
        alignas(64) char img[kSlotBytes];
        std::memcpy(img, a0, n0);
        std::memcpy(img + n0, a1, n1);
        stream_out(ka, img, n0 + n1);
        _mm_sfence();                                                  // the write-combining buffers drain before anything below leaves the core
        if (d_->hdp_flush) *d_->hdp_flush = 1u;                        // (posted, ahead of the doorbell on the same link)
        hsa_signal_add_relaxed(sig_, 1);                                // one per packet in flight: a restart pass may be queued behind a first pass still ending
        // (no wait for a free slot: a staged call is synchronous, so the queue never holds more than the passes of ONE solve -- at most four of its 64 packets)
        const uint64_t wi = hsa_queue_add_write_index_relaxed(q_, 1);
        auto* pk = static_cast<hsa_kernel_dispatch_packet_t*>(q_->base_address) + (wi & (q_->size - 1));
        pk->workgroup_size_x = (uint16_t)block; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1;
        pk->reserved0 = 0;
        pk->grid_size_x = block; pk->grid_size_y = 1; pk->grid_size_z = 1;
        pk->private_segment_size = 0;
        pk->group_segment_size = k.group_bytes + dynamic_lds;
        pk->kernel_object = k.object;
        pk->kernarg_address = ka;
        pk->reserved2 = 0;
        pk->completion_signal = sig_;
        const uint32_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        const uint32_t setup = 1u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n(reinterpret_cast<uint32_t*>(pk), header | (setup << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q_->doorbell_signal, (hsa_signal_value_t)wi);
        pending_ = true;
        return true;
    }

    // every packet's kernel has ENDED (its completion word in host memory comes earlier): before anything is queued elsewhere that must follow it
    bool wait_end(uint64_t timeout_ns) {
        if (!pending_) return true;
        const uint64_t spins = timeout_ns / 50 + 1;
        for (uint64_t i = 0; i < spins; ++i) {
            if (hsa_signal_load_scacquire(sig_) == 0) { pending_ = false; return true; }
            _mm_pause();
        }
        return false;
    }

private:
    AqlQueue() = default;
    __attribute__((target("avx2"))) static void stream_out_avx2(char* d, const char* s, size_t n) {
        size_t i = 0;
        for (; i + 32 <= n; i += 32) _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), _mm256_load_si256(reinterpret_cast<const __m256i*>(s + i)));
        for (; i + 8 <= n; i += 8) _mm_stream_si64(reinterpret_cast<long long*>(d + i), *reinterpret_cast<const long long*>(s + i));
    }
    static void stream_out(char* d, const char* s, size_t n) {          // d, s 32-byte aligned, n a multiple of 8
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) stream_out_avx2(d, s, n); else std::memcpy(d, s, n);
    }
    AqlDevice* d_ = nullptr;
    hsa_queue_t* q_ = nullptr;
    char* kbuf_ = nullptr;
    hsa_signal_t sig_{};
    bool have_sig_ = false, pending_ = false;
    unsigned slot_ = 0;
};

}  // namespace srbdqp

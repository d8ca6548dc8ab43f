// srbdqp.hip -- C-ABI of the batched SRBD convex-MPC QP engine (see include/srbdqp.h) and kernel dispatch.
// Host side: HIP runtime only (stream, workspace, events).  No CPU fallback of any kind.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <chrono>
#include <string>
#include <unordered_set>
#include <vector>

#include "srbdqp.h"
#include "srbdqp_common.hpp"
#include "srbdqp_mfma.hpp"
#include "srbdqp_compact.hpp"
#include "srbdqp_split.hpp"
#include "srbdqp_setup1.hpp"
#include "srbdqp_wrench.hpp"
#include "srbdqp_cascade.hpp"
#include "srbdqp_cascade.h"
#include "srbdqp_aql.hpp"

using srbdqp::KArgs;

struct srbdqp_handle {
    srbdqp_config cfg;
    int maxs_override = 0;         // set by the host-buffer API after scanning the contact flags
    bool io_f32 = false;           // set around a launch by the _f32 entry points: the caller's buffers are float
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;   // ev_mid: between the two kernels of the split pipeline
    bool ev_valid = false, ev_mid_valid = false;
    // device workspace for the host-buffer API
    char* ws = nullptr;
    size_t ws_bytes = 0;
    std::string err;
    const char* kname = "none";
    long long* stamps = nullptr;   // diagnostic stamp buffer (device), see srbdqp_set_stamp_buffer
    const int32_t* sched_hint = nullptr;   // device: previous step's iters[] (srbdqp_set_schedule_hint)
    size_t sched_hint_len = 0;             // its length: batches larger than that are dispatched in natural order
    // per-launch-stream device scratch (a caller may pipeline solves of one handle over several streams: each stream
    // needs its own dispatch order and its own split-pipeline hand-over workspace)
    struct StreamSlot {
        hipStream_t st = nullptr; bool used = false;
        int32_t* perm = nullptr; size_t perm_cap = 0;
        double* ws = nullptr; size_t ws_doubles = 0;
        // rho restart: fp32 maxima of the last check of the QPs that end at the cap of the first pass, and y / status when
        // the caller passes none (the second pass warm-starts from the first pass's outputs)
        char* rs = nullptr; size_t rs_items = 0, rs_rows = 0;
        float* resid = nullptr; double* ybuf = nullptr;
        double* ubuf = nullptr;                   // staged first pass: device copy of u for the pass behind it (KArgs::u_dev)
        int32_t* stbuf = nullptr;
        double* rhobuf[2] = {nullptr, nullptr};   // rho a restart pass ran its QPs with, for the pass behind it (alternating)
        // SRBDQP_FLAG_DEFER_TAIL on the kernels that restart by further LAUNCHES: three sets of the buffers above in rotation (a solve's restart passes run on
        // tail_st, beside the next solves of this stream; a set is reused only behind the event that closes its last user's passes), + per set the lists through
        // which a pass hands its capped QPs to the next one (so that the working workgroups of a pass come first in its grid)
        struct RSet { float* resid = nullptr; double* ybuf = nullptr; int32_t* stbuf = nullptr; double* rhobuf[2] = {nullptr, nullptr};
                      int32_t* list[4] = {nullptr, nullptr, nullptr, nullptr}; int32_t* cnt = nullptr; hipEvent_t ev_tail = nullptr; bool ev_used = false; };
        RSet rsets[3];
        int rs_nsets = 0;
        unsigned long long rs_k = 0;
        hipStream_t tail_st = nullptr; hipEvent_t ev_main = nullptr;
        hipEvent_t last_tail = nullptr;    // closes the restart passes of the last deferred solve on this stream (srbdqp_flush waits for it), or null
        // deferred tails (SRBDQP_FLAG_DEFER_TAIL): three rotating lists of continuation records and their counts
        char* tail = nullptr; int32_t* tail_cnt = nullptr; size_t tail_cap = 0;
        unsigned long long tail_k = 0;     // launches so far: list k % 3 is appended to, (k + 2) % 3 read, (k + 1) % 3 zeroed
        static constexpr int kTailHist = 8;
        long long tail_hist[kTailHist] = {0, 0, 0, 0, 0, 0, 0, 0};   // batch sizes of the last launches on this stream, newest first (bounds the records a launch can find)
        bool tail_live = false;            // records may be pending (a solve since the last flush)
    };
    static constexpr int kMaxSlots = 8;
    StreamSlot slots[kMaxSlots];
    // low-latency staging: one pinned, GPU-mapped slab carved into the arrays of srbdqp_stage
    char* stage_host = nullptr;
    char* stage_dev = nullptr;
    srbdqp_stage stage_h{};        // host addresses
    srbdqp_stage stage_d{};        // device addresses of the same memory
    // completion word of the staged path (last 64 bytes of the slab) + device counter of finished workgroups
    volatile int32_t* done_host = nullptr;
    int32_t* done_dev = nullptr;
    int32_t* done_count = nullptr;
    int32_t done_seq = 0;
    bool signal_next = false;      // set by srbdqp_solve_staged_f64 around its launch
    bool done_cs = false;          // the last launch publishes its completion word with the checksum of its outputs (KArgs::done_cs): wait_done() verifies it
    bool done_cs_x = false;        // ... which cover x_out
    bool staged_call = false;      // inside srbdqp_solve_staged_f64 (with or without the completion word)
    int staged_neff = 0;           // ... with the largest number of presolved variables (3 x stance contacts) among its QPs
    bool lazy_restart = false;     // staged path: run only the first pass; the host starts the second one if a status asks for it
    bool lazy_pending = false;     // ... and the last solve really was such a first pass: last_args are its arguments,
    int lazy_rcount = 1;           //     this many restart passes may follow it,
    StreamSlot* lazy_slot = nullptr;   //  with the buffers of this launch-stream slot
    int32_t prepared_B = 0; int prepared_maxs = 4; bool prepared_pcom = false;   // two-phase call: a set-up is pending
    KArgs last_args;               // arguments of that first pass (for the lazily started second pass)
    // kernels whose dynamic-LDS limit has been raised on this handle's device (function attributes are per device, and a
    // process may hold handles on several)
    std::unordered_set<const void*> lds_attr_done;
    // the staged one-QP call's own AQL queue (srbdqp_aql.hpp); null: hipLaunchKernelGGL (set-up failed -- aql_why says how -- or SRBDQP_NO_AQL=1)
    srbdqp::AqlQueue* aql = nullptr;
    bool aql_tried = false;
    std::string aql_why;
};

// slot of a launch stream (at most kMaxSlots distinct streams per handle; null when exhausted)
srbdqp_handle::StreamSlot* stream_slot(srbdqp_handle* h, hipStream_t st) {
    for (auto& s : h->slots) if (s.used && s.st == st) return &s;
    for (auto& s : h->slots) if (!s.used) { s.used = true; s.st = st; return &s; }
    h->err = "more than 8 distinct launch streams on one handle";
    return nullptr;
}

namespace {

// Longest-first dispatch order from the previous step's iteration counts (one workgroup; counting sort by iters/4,
// descending).  QPs that needed many ADMM iterations last time are started first, so the straggler tail of a launch
// overlaps the bulk instead of trailing it.  The hint only orders work; every QP is solved in full either way.
__global__ __launch_bounds__(1024) void srbdqp_schedule_kernel(const int32_t* iters_prev, int32_t* perm, int B) {
    __shared__ int cnt[128];
    __shared__ int base[128];
    const int t = threadIdx.x;
    if (t < 128) cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) {
        int k = iters_prev[i] >> 2;
        k = k < 0 ? 0 : (k > 127 ? 127 : k);
        atomicAdd(&cnt[127 - k], 1);
    }
    __syncthreads();
    if (t == 0) { int acc = 0; for (int k = 0; k < 128; ++k) { base[k] = acc; acc += cnt[k]; } }
    __syncthreads();
    for (int i = t; i < B; i += 1024) {
        int k = iters_prev[i] >> 2;
        k = k < 0 ? 0 : (k > 127 ? 127 : k);
        perm[atomicAdd(&base[127 - k], 1)] = i;
    }
}

std::string g_create_err;

#define HIP_TRY(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return SRBDQP_E_HIP;                                                                 \
        }                                                                                        \
    } while (0)

// horizons with an instantiation (the 4-wave compact kernel: N in {4, 8, 10} with up to 4 stance contacts per step, N in
// {12, 16, 20} with at most 2; the one-wave kernel: <= 64 presolved variables; the general kernel: every horizon and pattern)
bool horizon_supported(int N) { return N == 4 || N == 8 || N == 10 || N == 12 || N == 16 || N == 20 || N == 24; }

int resolve_kernel(const srbdqp_config& c) {
    if (c.kernel == SRBDQP_KERNEL_WRENCH) return SRBDQP_KERNEL_WRENCH;
    return SRBDQP_KERNEL_COMPACT;   // AUTO, COMPACT, SPLIT and WAVE: the presolved family of srbdqp_compact.hpp
}

void fill_args(const srbdqp_config& c, KArgs& a) {
    a.max_iter = c.max_iter;
    a.check_every = c.check_every;
    a.dt = c.dt;
    a.inv_mass = 1.0 / c.mass;
    for (int i = 0; i < 3; ++i) a.iinv[i] = 1.0 / c.inertia[i];
    a.mu = c.mu;
    a.s = c.force_scale;
    a.fzmin_s = c.fz_min / c.force_scale;
    a.fzmax_s = c.fz_max / c.force_scale;
    for (int i = 0; i < 12; ++i) a.sqrtq[i] = std::sqrt(c.q_diag[i]);
    a.rs2 = c.r_diag * c.force_scale * c.force_scale;
    a.rho = c.rho;
    a.rho_eq = c.rho * c.rho_eq_scale;
    a.rho_fz = c.rho_fz_scale;
    a.sigma = c.sigma;
    a.alpha = c.alpha;
    a.eps_abs = c.eps_abs;
    a.eps_rel = c.eps_rel;
}

template <typename K>
int set_lds_once(srbdqp_handle* h, K kernel, size_t lds) {
    const void* fn = reinterpret_cast<const void*>(kernel);
    if (!h->lds_attr_done.count(fn)) {
        HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->lds_attr_done.insert(fn);
    }
    return SRBDQP_OK;
}

// Batches of at least this many QPs of the small instantiations (<= 64 presolved variables) run with one wave per QP
// (launch_wave); the staged (completion-word) path and the big instantiations use the 4-wave kernel.  Until round 4 the cross-over of the per-call time was 512
// QPs; with the rho restart on at every batch size (in place on the one-wave kernel, one more launch per pass on the 4-wave one) the one-wave kernel is the
// faster one at EVERY size (tools/threshold_probe.py, us per synchronised call, 4-wave / one-wave: B = 1 60 / 55, 32 67 / 61, 128 72 / 64, 256 163 / 149,
// 512 238 / 192, 4096 483 / 289).
constexpr int kSplitMinBatch = 1;

// Batches of at least this many QPs with more than 2 stance contacts in a step go to the general kernel at N <= 10 too
// (measured, tools/schedule_bench.py, 4096 QPs: N = 10 double support 13.5 M QP/s against 4.2 M on the 4-wave compact kernel,
// mixed gait 13.9 M against 6.5 M); smaller ones stay on the 4-wave kernel (lowest latency).
constexpr int kWrenchMinBatch = 512;      // re-measured in round 4 (uniform rho restart; tools/schedule_bench.py, M QP/s 4-wave / general): N = 10 mixed gait 256 QPs 1.72 / 1.63,
                                          // 512 3.10 / 3.28, 768 3.98 / 4.67, 1024 4.63 / 6.06; double support 256 1.70 / 2.45, 512 2.61 / 4.28 (round 2: 768)
constexpr int kRestartMinBatch = 4096;         // the one-wave kernel's automatic rho restart in place: batches that fill the chip twice over (restart_iter_of)
constexpr int kTail1MaxBatch = 8;              // staged calls of up to this many QPs on <= 2 stance contacts per step: the 4-wave set-up + one-wave iteration kernel
constexpr int kStagedWrenchMinVars = 60;   // staged call: presolved variables (3 per stance contact) above which the wrench-space kernel's low-latency
                                          // instantiation wins (B = 1, N = 10: mixed gait, 72 variables, 74 us compact / 69 us; double support, 120, 107 / 69)
constexpr int kWrenchMinBatchN20 = 256;   // N = 20: one workgroup per CU on the compact kernel, two on the general one

// fp32 calls of at least this many QPs are split by tile precision (two launches + the classification kernel); smaller
// ones run on fp64 tiles, where the third workgroup per CU would stay empty anyway.
constexpr int kTileClassMinBatch = 512;

// does a solve of B QPs on this handle go to the general kernel (srbdqp_wrench.hpp)?  launch() / launch_long() ask this.
inline bool uses_wrench(const srbdqp_handle* h, int maxs, int B) {
    const int N = h->cfg.horizon;
    if (h->cfg.kernel == SRBDQP_KERNEL_WRENCH || h->io_f32 || N == 24) return true;
    // N = 20 single support too, for batches: the general kernel holds 2 workgroups per CU there, the compact one 1
    // (tools/schedule_bench.py, 16,384 QPs: 2.85 M QP/s against 1.95 M; at N = 12 / 16 the compact kernel wins, 7.1 / 4.9 M
    // against 5.7 / 4.2 M)
    if (N > 10) return maxs > 2 || (N == 20 && h->cfg.kernel == SRBDQP_KERNEL_AUTO && B >= kWrenchMinBatchN20 && !h->stamps && !h->staged_call);
    // N <= 10 with more than 2 stance contacts in a step: batches (the 4-wave kernel wins up to two QPs per CU) AND the staged
    // low-latency path -- the reference's own call feeds full double support on every step (run_simulation.py:100-101), where the
    // wrench-space problem is 60 x 60 against the 120 x 120 dense K of the compact kernel (round 3, tools/latency_patterns.py:
    // B = 1 double support p50 82 us against 112 us)
    if (h->cfg.kernel != SRBDQP_KERNEL_AUTO || maxs <= 2 || h->stamps) return false;
    // (the low-latency instantiation is one workgroup per CU, tuned and measured at B = 1: the same bound as the compact kernel's TAIL1 path)
    return B >= kWrenchMinBatch || (h->staged_call && B <= kTail1MaxBatch && N >= 8 && h->staged_neff > kStagedWrenchMinVars);
}

// does a solve of B QPs on this handle run on the one-wave kernel (launch_wave)?  launch_compact() and the restart plan (solve_device_impl) ask this.
inline bool uses_wave(const srbdqp_handle* h, int maxs, int B, bool stamps, bool signalled) {
    const int N = h->cfg.horizon, k = h->cfg.kernel;
    if (N > 10 || !(N == 4 || maxs <= 2) || uses_wrench(h, maxs, B)) return false;     // (<= 64 presolved variables: Setup1Smem::supported)
    const bool want = k == SRBDQP_KERNEL_WAVE || (k == SRBDQP_KERNEL_AUTO && B >= kSplitMinBatch);
    return want && (!stamps || k == SRBDQP_KERNEL_WAVE) && !signalled;
}

// KERNEL_SPLIT (A/B): the one-wave set-up and the one-wave ADMM as two kernels with the hand-over through HBM.

template <int N, int MAXS>
int launch_split(srbdqp_handle* h, KArgs a, hipStream_t st) {
    using W = srbdqp::SplitWs<N, MAXS>;
    const size_t need = (size_t)(a.qp_span > a.B ? a.qp_span : a.B) * W::doubles;   // indexed by QP, not by workgroup
    auto* slot = stream_slot(h, st);
    if (!slot) return SRBDQP_E_INVALID;
    if (need > slot->ws_doubles) {
        HIP_TRY(h, hipStreamSynchronize(st));               // a previous launch on this stream may still use the old buffer
        if (slot->ws) { HIP_TRY(h, hipFree(slot->ws)); slot->ws = nullptr; slot->ws_doubles = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&slot->ws), need * sizeof(double));
        if (e != hipSuccess) { h->err = std::string("hipMalloc split workspace: ") + hipGetErrorString(e); return SRBDQP_E_NOMEM; }
        slot->ws_doubles = need;
    }
    a.ws = slot->ws;
    h->prepared_B = 0;                                      // the split pipeline overwrites the workspace a pending two-phase set-up lives in
    constexpr size_t ldsA = srbdqp::CompactTraits<N, MAXS>::lds_bytes, ldsB = srbdqp::SplitSmem<N, MAXS>::bytes;
    int rc = set_lds_once(h, &srbdqp::srbdqp_compact_kernel<N, MAXS, true>, ldsA);
    if (rc != SRBDQP_OK) return rc;
    static const std::string nm = "split_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
    h->kname = nm.c_str();
    if constexpr (srbdqp::Setup1Smem<N, MAXS>::supported) {
        if (!(h->cfg.flags & SRBDQP_FLAG_SETUP4)) {          // set-up with one wave per QP (default)
            constexpr size_t lds1 = srbdqp::Setup1Smem<N, MAXS>::bytes;
            hipLaunchKernelGGL((srbdqp::srbdqp_setup1_kernel<N, MAXS, false>), dim3((unsigned)a.B), dim3(64), lds1, st, a);
        } else {
            hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel<N, MAXS, true>), dim3((unsigned)a.B), dim3(srbdqp::kThreads), ldsA, st, a);
        }
    } else {
        hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel<N, MAXS, true>), dim3((unsigned)a.B), dim3(srbdqp::kThreads), ldsA, st, a);
    }
    if ((h->cfg.flags & SRBDQP_FLAG_TIMING) && !a.count_ptr) { HIP_TRY(h, hipEventRecord(h->ev_mid, st)); h->ev_mid_valid = true; }
    hipLaunchKernelGGL((srbdqp::srbdqp_admm_kernel<N, MAXS>), dim3((unsigned)a.B), dim3(64), ldsB, st, a);
    HIP_TRY(h, hipGetLastError());
    return SRBDQP_OK;
}

// Two-phase call (srbdqp_prepare_staged_f64 / srbdqp_solve_prepared_f64): the split pipeline's two kernels launched apart.
// phase 0 = set-up (+ dq/dx0) from a predicted x0, phase 1 = gradient patch for the measured x0 + ADMM + roll-out.
template <int N, int MAXS>
int launch_two_phase(srbdqp_handle* h, KArgs a, hipStream_t st, int phase) {
    if constexpr (srbdqp::Setup1Smem<N, MAXS>::supported && srbdqp::SplitWs<N, MAXS>::supported) {
        using W = srbdqp::SplitWs<N, MAXS>;
        const size_t need = (size_t)a.B * W::doubles;
        auto* slot = stream_slot(h, st);
        if (!slot) return SRBDQP_E_INVALID;
        if (need > slot->ws_doubles) {
            if (phase == 1) { h->err = "srbdqp_solve_prepared_f64 without a matching srbdqp_prepare_staged_f64"; return SRBDQP_E_INVALID; }
            HIP_TRY(h, hipStreamSynchronize(st));
            if (slot->ws) { HIP_TRY(h, hipFree(slot->ws)); slot->ws = nullptr; slot->ws_doubles = 0; }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&slot->ws), need * sizeof(double));
            if (e != hipSuccess) { h->err = std::string("hipMalloc split workspace: ") + hipGetErrorString(e); return SRBDQP_E_NOMEM; }
            slot->ws_doubles = need;
        }
        a.ws = slot->ws;
        if (phase == 0) {
            constexpr size_t lds1 = srbdqp::Setup1Smem<N, MAXS>::bytes;
            hipLaunchKernelGGL((srbdqp::srbdqp_setup1_kernel<N, MAXS, false, false, true>), dim3((unsigned)a.B), dim3(64), lds1, st, a);
            static const std::string nm = "prepare_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
            h->kname = nm.c_str();
        } else {
            constexpr size_t ldsB = srbdqp::SplitSmem<N, MAXS>::bytes;
            a.defer_x0 = 1;
            hipLaunchKernelGGL((srbdqp::srbdqp_admm_kernel<N, MAXS>), dim3((unsigned)a.B), dim3(64), ldsB, st, a);
            static const std::string nm = "prepared_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
            h->kname = nm.c_str();
        }
        HIP_TRY(h, hipGetLastError());
        return SRBDQP_OK;
    } else {
        h->err = "the two-phase call needs at most 64 presolved variables (N <= 10 with <= 2 stance contacts per step, N = 4 with 4)";
        return SRBDQP_E_INVALID;
    }
}

int launch_two_phase_any(srbdqp_handle* h, const KArgs& a, hipStream_t st, int maxs, int phase) {
    switch (h->cfg.horizon) {
        case 4: return maxs <= 2 ? launch_two_phase<4, 2>(h, a, st, phase) : launch_two_phase<4, 4>(h, a, st, phase);
        case 8: return maxs <= 2 ? launch_two_phase<8, 2>(h, a, st, phase) : launch_two_phase<8, 4>(h, a, st, phase);
        case 10: return maxs <= 2 ? launch_two_phase<10, 2>(h, a, st, phase) : launch_two_phase<10, 4>(h, a, st, phase);
        default: h->err = "the two-phase call is built for N in {4, 8, 10}"; return SRBDQP_E_INVALID;
    }
}

// One wave per QP for the whole solve (srbdqp_setup1.hpp, FUSED): the default for large batches of the small
// instantiations; nothing but inputs and outputs touches HBM.
template <int N, int MAXS>
int launch_wave(srbdqp_handle* h, const KArgs& a, hipStream_t st) {
    constexpr size_t lds1 = srbdqp::Setup1Smem<N, MAXS>::bytes;
    static const std::string nm = "wave_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
    h->kname = nm.c_str();
    if (a.mode == 1) hipLaunchKernelGGL((srbdqp::srbdqp_setup1_kernel<N, MAXS, true, true>), dim3((unsigned)a.B), dim3(64), lds1, st, a);
    else if (a.restart_every > 0) {   // the rho restart in place: (x, y) of a pass wait in [3][64] doubles behind the kernel's own LDS
        constexpr size_t ldsr = lds1 + 3 * 64 * sizeof(double);
        static_assert(8 * ((ldsr + 1279) / 1280) * 1280 <= 163840, "eight QPs per CU");
        hipLaunchKernelGGL((srbdqp::srbdqp_setup1_kernel<N, MAXS, true, false, false, true>), dim3((unsigned)a.B), dim3(64), ldsr, st, a);
    }
    else hipLaunchKernelGGL((srbdqp::srbdqp_setup1_kernel<N, MAXS, true>), dim3((unsigned)a.B), dim3(64), lds1, st, a);
    HIP_TRY(h, hipGetLastError());
    return SRBDQP_OK;
}

// The one-wave kernel with deferred tails (srbdqp_setup1.hpp, srbdqp_wave_defer_kernel): grid = tail workgroups (continuations of earlier launches on this
// stream, first) + one workgroup per QP of this launch.  a.B = 0: a flush launch (continuations only, one workgroup per record the lists can hold).
template <int N, int MAXS>
int launch_wave_defer(srbdqp_handle* h, KArgs a, hipStream_t st, srbdqp_handle::StreamSlot* slot) {
    constexpr size_t lds1 = srbdqp::Setup1Smem<N, MAXS>::bytes + 3 * 64 * sizeof(double);
    static_assert(8 * ((lds1 + 1279) / 1280) * 1280 <= 163840, "eight QPs per CU");
    static const std::string nm = "wave_defer_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
    h->kname = nm.c_str();
    const unsigned long long k = slot->tail_k++;
    a.tail_lists = slot->tail; a.tail_cnt = slot->tail_cnt; a.tail_cap = (int32_t)slot->tail_cap;
    a.tail_iout = (int32_t)(k % 3); a.tail_iin = (int32_t)((k + 2) % 3); a.tail_izero = (int32_t)((k + 1) % 3);
    // tail workgroups: one per record the list this launch reads CAN hold -- a bound the host knows without asking the device: a record was written by the launch
    // before this one on the stream, as a first pass that reached its mark (at most that launch's QPs) or as a continuation that reached another (at most the QPs
    // of the launches before it that may still re-balance): the sum of the last rho_restart_count batch sizes.  Workgroups without a record leave after one
    // scalar load; 8192 of them in front of a 4096-QP launch cost 0.7 % (16,384: 1.5 %, tools/defer_bench.py with SRBDQP_TAIL_WGS_MIN).  A first version sized
    // this from the record count the device reported through a host-mapped word and moved surplus records on to the next list: the report is stale by however
    // far the host runs ahead of the device, and with 95 % of every batch continuing and the host 300 launches ahead the lists overflowed (tools/defer_fuzz.py).
    long long T = 0;
    const int rmax = a.restart_max > 0 ? a.restart_max : 1;
    for (int j = 0; j < rmax && j < srbdqp_handle::StreamSlot::kTailHist; ++j) T += slot->tail_hist[j];
    if (T < 64) T = 64;
    static const long long tail_wgs_min = [] { const char* e = getenv("SRBDQP_TAIL_WGS_MIN"); return e ? atoll(e) : 0LL; }();   // (experiments: the cost of empty tail workgroups; read once)
    if (a.B > 0 && T < tail_wgs_min) T = tail_wgs_min;
    if (T > (long long)slot->tail_cap) T = (long long)slot->tail_cap;
    for (int j = srbdqp_handle::StreamSlot::kTailHist - 1; j > 0; --j) slot->tail_hist[j] = slot->tail_hist[j - 1];
    slot->tail_hist[0] = a.B;
    if (a.B == 0) for (auto& v : slot->tail_hist) v = 0;      // (a flush launch finishes every record in place: the lists are empty behind it)
    a.tail_wgs = (int32_t)T;
    const bool timing = (h->cfg.flags & SRBDQP_FLAG_TIMING) != 0;          // (srbdqp_last_kernel_ms: this launch, continuations of earlier solves included)
    if (timing) { HIP_TRY(h, hipEventRecord(h->ev0, st)); h->ev_mid_valid = false; }
    if (a.B > 0) hipLaunchKernelGGL((srbdqp::srbdqp_wave_defer_kernel<N, MAXS, false>), dim3((unsigned)(T + a.B)), dim3(64), lds1, st, a);
    else hipLaunchKernelGGL((srbdqp::srbdqp_wave_defer_kernel<N, MAXS, true>), dim3((unsigned)T), dim3(64), lds1, st, a);
    HIP_TRY(h, hipGetLastError());
    if (timing) { HIP_TRY(h, hipEventRecord(h->ev1, st)); h->ev_valid = true; }
    slot->tail_live = true;
    return SRBDQP_OK;
}

int launch_wave_defer_any(srbdqp_handle* h, const KArgs& a, hipStream_t st, srbdqp_handle::StreamSlot* slot, int maxs) {
    const bool s2 = maxs <= 2;
    switch (h->cfg.horizon) {
        // N = 4: ONE instantiation for every launch and for the flush, whatever bound on the stance contacts the call came with -- a record written by a <4, 4>
        // launch and continued by a <4, 2> one (the flush used to pick its MAXS from cfg.max_contacts_per_step, 0 -> 2, while the device API assumes 4 and the
        // host API scans the flags per call) rebuilt the QP with the wrong bound and returned SRBDQP_CONTACT_BOUND with zero forces
        case 4: return launch_wave_defer<4, 4>(h, a, st, slot);
        case 8: if (s2) return launch_wave_defer<8, 2>(h, a, st, slot); break;
        case 10: if (s2) return launch_wave_defer<10, 2>(h, a, st, slot); break;
        default: break;
    }
    h->err = "deferred tails exist for the one-wave kernel only (N <= 10, at most 2 stance contacts per step)";
    return SRBDQP_E_INVALID;
}

// lists for launches of up to B QPs that may re-balance up to rmax times (sized for the share that really continues; a full list is not an error)
int ensure_tail_lists(srbdqp_handle* h, srbdqp_handle::StreamSlot* slot, hipStream_t st, size_t B, int rmax);
int flush_slot(srbdqp_handle* h, srbdqp_handle::StreamSlot* slot, hipStream_t st);

// the handle's AQL queue, made at the first staged one-QP call (a 5 MB code object goes through the HSA loader once per process and device)
srbdqp::AqlQueue* aql_queue(srbdqp_handle* h) {
    if (h->aql_tried) return h->aql;
    h->aql_tried = true;
    const char* off = std::getenv("SRBDQP_NO_AQL");
    if (off && off[0] && off[0] != '0') { h->aql_why = "SRBDQP_NO_AQL is set"; return nullptr; }
    int bus = 0, dev = 0, dom = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->cfg.device) != hipSuccess || hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, h->cfg.device) != hipSuccess ||
        hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->cfg.device) != hipSuccess) { (void)hipGetLastError(); h->aql_why = "no PCI address for the HIP device"; return nullptr; }
    h->aql = srbdqp::AqlQueue::create(srbdqp::aql_device(dom, bus, dev), &h->aql_why);
    return h->aql;
}

// every kernel this handle started through its own queue has ended: before anything that must follow it is handed to a HIP stream
int aql_quiesce(srbdqp_handle* h) {
    if (!h->aql || !h->aql->in_flight()) return SRBDQP_OK;
    if (h->aql->wait_end(5000000000ull)) return SRBDQP_OK;
    h->err = "a kernel on the handle's AQL queue did not end within 5 s";
    return SRBDQP_E_HIP;
}

// one workgroup of a *_kernel_in instantiation (arguments: KArgs, then StagedIn<N>) through the handle's own queue; false: the caller launches through HIP
template <class In>
bool aql_launch_in(srbdqp_handle* h, hipStream_t st, const char* kd_format, int n, int x, const KArgs& a, const In& in, unsigned block, size_t lds) {
    // (events and tail passes live on HIP streams; a call that does not spin on the completion word waits on its stream)
    if (!h->staged_call || st != h->stream || !a.done_flag || (h->cfg.flags & (SRBDQP_FLAG_DEFER_TAIL | SRBDQP_FLAG_TIMING | SRBDQP_FLAG_NO_SPIN))) return false;
    srbdqp::AqlQueue* q = aql_queue(h);
    if (!q) return false;
    char name[160];
    std::snprintf(name, sizeof(name), kd_format, n, x);
    const srbdqp::AqlKernel& k = srbdqp::aql_kernel(q->device(), name, sizeof(KArgs) + sizeof(In));
    return q->launch(k, &a, sizeof(KArgs), &in, sizeof(In), block, (uint32_t)lds);
}

// the one staged QP of a *_in launch whose host-visible outputs are u, x, status and iters in the staging arrays: completion word with their checksum, no fence
// (srbdqp_common.hpp signal_done_checksum; SRBDQP_DONE_FENCE=1 in the environment keeps the fence: A/B)
void staged_done_checksum(srbdqp_handle* h, KArgs& ai) {
    static const bool off = [] { const char* e = std::getenv("SRBDQP_DONE_FENCE"); return e && e[0] && e[0] != '0'; }();
    h->done_cs = false;
    if (off || !ai.done_flag || ai.B != 1 || ai.u_out != h->stage_d.u || ai.status != h->stage_d.status || ai.iters != h->stage_d.iters ||
        (ai.x_out && ai.x_out != h->stage_d.x) || (ai.y_out && !ai.y_capped_only)) return;
    ai.done_cs = 1;
    h->done_cs = true;
    h->done_cs_x = ai.x_out != nullptr;
}

// one staged QP whose inputs still sit in the library's own staging arrays: they ride in the kernel-argument segment (srbdqp_common.hpp StagedIn)
template <int N>
bool staged_inline_inputs(const srbdqp_handle* h, const KArgs& a, srbdqp::StagedIn<N>& in) {
    if (!(h->staged_call && a.B == 1 && a.x0 == h->stage_d.x0 && a.xref == h->stage_d.x_ref && a.foot == h->stage_d.foot && a.contact == h->stage_d.contact) ||
        a.perm || a.row_off || (a.pcom && a.pcom != h->stage_d.pcom)) return false;     // (a restart pass of the staged call too: the host starts it only for a QP at the cap)
    std::memcpy(in.x0, h->stage_h.x0, sizeof(in.x0));
    std::memcpy(in.xref, h->stage_h.x_ref, sizeof(in.xref));
    std::memcpy(in.foot, h->stage_h.foot, sizeof(in.foot));
    std::memcpy(in.contact, h->stage_h.contact, sizeof(in.contact));
    if (a.pcom) std::memcpy(in.pcom, h->stage_h.pcom, sizeof(in.pcom));
    return true;
}

template <int N, int MAXS>
int launch_compact(srbdqp_handle* h, const KArgs& a, hipStream_t st) {
    if constexpr (srbdqp::Setup1Smem<N, MAXS>::supported) {
        if (uses_wave(h, MAXS, a.B, a.stamps != nullptr, a.done_flag != nullptr)) return launch_wave<N, MAXS>(h, a, st);
    }
    if constexpr (srbdqp::SplitWs<N, MAXS>::supported) {
        if (h->cfg.kernel == SRBDQP_KERNEL_SPLIT && a.mode == 0 && !a.stamps && !a.done_flag) return launch_split<N, MAXS>(h, a, st);
    }
    constexpr size_t lds = srbdqp::CompactTraits<N, MAXS>::lds_bytes;
    static const std::string nm = "compact_f64_n" + std::to_string(N) + "_s" + std::to_string(MAXS);
    h->kname = nm.c_str();
    if (a.mode == 1) {   // assembly dump: the same kernel, stopped before its factorisation
        int rc = set_lds_once(h, &srbdqp::srbdqp_compact_kernel<N, MAXS, false, true>, lds);
        if (rc != SRBDQP_OK) return rc;
        hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel<N, MAXS, false, true>), dim3((unsigned)a.B), dim3(srbdqp::kThreads), lds, st, a);
        return SRBDQP_OK;
    }
    if constexpr (MAXS == 2 && srbdqp::SplitWs<N, MAXS>::supported) {
        // staged batch-1 path (completion word): four waves for the set-up, then the one-wave iteration on wave 0 (srbdqp_compact.hpp, TAIL1) --
        // compiled for one workgroup's worth of registers.  tools/latency_patterns.py, tools/batch1_kernel_probe.py
        if (a.done_flag && a.B <= kTail1MaxBatch && !(h->cfg.flags & SRBDQP_FLAG_NO_LAT)) {
            constexpr size_t lds1 = srbdqp::CompactTraits<N, MAXS>::lds_bytes_tail1;
            static_assert(lds1 <= 163840 && srbdqp::SplitWs<N, MAXS>::KS <= 64, "TAIL1: K^-1 rows behind the kernel's own LDS");
            int rc1 = set_lds_once(h, &srbdqp::srbdqp_compact_kernel<N, MAXS, false, false, true>, lds1);
            if (rc1 != SRBDQP_OK) return rc1;
            static const std::string nml = nm + "_lat";
            h->kname = nml.c_str();
            srbdqp::StagedIn<N> in;
            if (!a.count_ptr && staged_inline_inputs<N>(h, a, in)) {
                rc1 = set_lds_once(h, &srbdqp::srbdqp_compact_kernel_in<N, MAXS>, lds1);
                if (rc1 != SRBDQP_OK) return rc1;
                KArgs ai = a;
                ai.inline_in = 1;
                staged_done_checksum(h, ai);
                if (aql_launch_in(h, st, "_ZN6srbdqp24srbdqp_compact_kernel_inILi%dELi%dEEEvNS_5KArgsENS_8StagedInIXT_EEE.kd", N, MAXS, ai, in, srbdqp::kThreads, lds1)) return SRBDQP_OK;
                rc1 = aql_quiesce(h);
                if (rc1 != SRBDQP_OK) return rc1;
                hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel_in<N, MAXS>), dim3(1), dim3(srbdqp::kThreads), lds1, st, ai, in);
                return SRBDQP_OK;
            }
            hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel<N, MAXS, false, false, true>), dim3((unsigned)a.B), dim3(srbdqp::kThreads), lds1, st, a);
            return SRBDQP_OK;
        }
    }
    int rc = set_lds_once(h, &srbdqp::srbdqp_compact_kernel<N, MAXS>, lds);
    if (rc != SRBDQP_OK) return rc;
    hipLaunchKernelGGL((srbdqp::srbdqp_compact_kernel<N, MAXS>), dim3((unsigned)a.B), dim3(srbdqp::kThreads), lds, st, a);
    return SRBDQP_OK;
}

template <int N>
int launch_n(srbdqp_handle* h, const KArgs& a, hipStream_t st, int maxs) {
    int rc = (maxs <= 2) ? launch_compact<N, 2>(h, a, st) : launch_compact<N, 4>(h, a, st);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipGetLastError());
    return SRBDQP_OK;
}

// The general kernel (srbdqp_wrench.hpp): any contact pattern, fp64 or fp32 iterations / buffers.
template <int N, typename R, int TB = 8>
struct WrenchTraits {
    using S = srbdqp::WrenchSmem<N, TB>;
    static constexpr int by_lds = (S::lds_wgs * S::NW + 3) / 4 > 0 ? (S::lds_wgs * S::NW + 3) / 4 : 1;   // waves per SIMD LDS admits (rounded up: 3 workgroups of 3 waves put 3 waves on one SIMD)
#ifdef SRBDQP_F32TILE_WPS   // experiments: waves per SIMD the fp32-tile instantiation is compiled for
    static constexpr int want = (TB == 4) ? SRBDQP_F32TILE_WPS : ((sizeof(R) == 4) ? 3 : (S::CHMAX <= 36 ? 2 : 1));
#else
#ifndef SRBDQP_F64_SMALL_WPS
#define SRBDQP_F64_SMALL_WPS 3
#endif
#ifndef SRBDQP_F32_ON_F64_WPS
#define SRBDQP_F32_ON_F64_WPS 3
#endif
    static constexpr int want = (sizeof(R) == 4) ? ((TB == 8 && S::CHMAX > 30) ? SRBDQP_F32_ON_F64_WPS : 3) : (S::CHMAX <= 36 ? SRBDQP_F64_SMALL_WPS : (S::CHMAX <= 60 ? 2 : 1));   // register budget
#endif
    static constexpr int wps = by_lds < want ? by_lds : want;
};

template <int N, typename R, typename TIO>
int launch_wrench_t(srbdqp_handle* h, const KArgs& a, hipStream_t st) {
    using S = srbdqp::WrenchSmem<N>;
    constexpr int WPS = WrenchTraits<N, R>::wps;
    constexpr size_t lds = S::bytes;
    static_assert(lds <= 163840, "one QP must fit the LDS of a CU");
    // N = 24 on fp64 tiles holds ONE workgroup of 5 waves per CU (102 KB of LDS): three extra waves that take part in the set-up only
    // (45 tiles over 8 waves instead of 5; they end before the iterations) cost nothing the CU was using (round 3)
    constexpr int BXW = (N == 24) ? 3 : 0;
    using SB = srbdqp::WrenchSmem<N, 8, 5, BXW>;
    constexpr size_t ldsb = SB::bytes;
    static_assert(ldsb <= 163840, "one QP must fit the LDS of a CU");
    static const std::string nm = std::string("wrench_") + (sizeof(R) == 4 ? "f32" : "f64") + "_n" + std::to_string(N);
    if (a.mode == 1) {
        if constexpr (sizeof(R) == 8) {
            int rc = set_lds_once(h, &srbdqp::srbdqp_wrench_kernel<N, double, double, 1, WPS>, lds);
            if (rc != SRBDQP_OK) return rc;
            h->kname = nm.c_str();
            hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel<N, double, double, 1, WPS>), dim3((unsigned)a.B), dim3(S::BT), lds, st, a);
        } else { h->err = "the assembly dump is fp64 only"; return SRBDQP_E_INVALID; }
    } else {
        if constexpr (sizeof(R) == 8 && N <= 10) {
            // staged batch-1 path (completion word): the low-latency instantiation -- two extra waves for the set-up (tables, T
            // assembly, tile phases) that end before the iterations, one workgroup's worth of registers (no scratch, V in
            // registers, every broadcast read of the T^-1 product in flight).  tools/latency_patterns.py
            if (a.done_flag && a.B <= kTail1MaxBatch && !(h->cfg.flags & SRBDQP_FLAG_NO_LAT)) {
                constexpr int XW = (N >= 8) ? 2 : 1;
                using SL = srbdqp::WrenchSmem<N, 8, 5, XW>;
                constexpr size_t ldsl = SL::bytes;
                int rcl = set_lds_once(h, &srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, 1, double, 5, XW>, ldsl);
                if (rcl != SRBDQP_OK) return rcl;
                static const std::string nml = nm + "_lat";
                h->kname = nml.c_str();
                if constexpr (std::is_same<TIO, double>::value) {
                    srbdqp::StagedIn<N> in;
                    if (!a.count_ptr && !a.tile_sel && staged_inline_inputs<N>(h, a, in)) {
                        rcl = set_lds_once(h, &srbdqp::srbdqp_wrench_kernel_in<N, XW>, ldsl);
                        if (rcl != SRBDQP_OK) return rcl;
                        KArgs ai = a;
                        ai.inline_in = 1;
                        staged_done_checksum(h, ai);
                        if (aql_launch_in(h, st, "_ZN6srbdqp23srbdqp_wrench_kernel_inILi%dELi%dEEEvNS_5KArgsENS_8StagedInIXT_EEE.kd", N, XW, ai, in, SL::BT, ldsl)) return SRBDQP_OK;
                        rcl = aql_quiesce(h);
                        if (rcl != SRBDQP_OK) return rcl;
                        hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel_in<N, XW>), dim3(1), dim3(SL::BT), ldsl, st, ai, in);
                        HIP_TRY(h, hipGetLastError());
                        return SRBDQP_OK;
                    }
                }
                hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, 1, double, 5, XW>), dim3((unsigned)a.B), dim3(SL::BT), ldsl, st, a);
                HIP_TRY(h, hipGetLastError());
                return SRBDQP_OK;
            }
        }
        int rc = set_lds_once(h, &srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, WPS, double, 5, BXW>, ldsb);
        if (rc != SRBDQP_OK) return rc;
        h->kname = nm.c_str();
        if constexpr (sizeof(R) == 4) {
            // fp32 iterations: QPs whose steps all have 0 or >= 3 stance contacts (every g coordinate a wrench coordinate,
            // cond(T) ~ 5e4) factor T in fp32 tiles -- half the LDS, one more workgroup per CU; a step kept in force
            // variables carries the conditioning of K (1e8) into T and needs fp64 tiles.  Two launches over the same grid,
            // each workgroup looks at its QP's contact flags and leaves at once if the QP belongs to the other launch.
            // Not on the staged path (its workgroups are counted), not in the restart pass (few QPs), not for ragged batches.
            if (!a.done_flag && !a.count_ptr && !a.resid_in && !a.row_off && !(h->cfg.flags & SRBDQP_FLAG_F64_TILES) && (a.B >= kTileClassMinBatch || (h->cfg.flags & SRBDQP_FLAG_F32_TILES))) {
                using S4 = srbdqp::WrenchSmem<N, 4>;
                constexpr int WPS4 = WrenchTraits<N, R, 4>::wps;
                constexpr size_t lds4 = S4::bytes;
                rc = set_lds_once(h, &srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, WPS4, float>, lds4);
                if (rc != SRBDQP_OK) return rc;
                KArgs a4 = a, a8 = a;
                a4.tile_sel = 1; a8.tile_sel = 2;
                hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, WPS4, float>), dim3((unsigned)a.B), dim3(S::BT), lds4, st, a4);
                hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, WPS, double, 5, BXW>), dim3((unsigned)a.B), dim3(SB::BT), ldsb, st, a8);
                HIP_TRY(h, hipGetLastError());
                return SRBDQP_OK;
            }
        }
        hipLaunchKernelGGL((srbdqp::srbdqp_wrench_kernel<N, R, TIO, 0, WPS, double, 5, BXW>), dim3((unsigned)a.B), dim3(SB::BT), ldsb, st, a);
    }
    HIP_TRY(h, hipGetLastError());
    return SRBDQP_OK;
}

template <int N>
int launch_wrench(srbdqp_handle* h, const KArgs& a, hipStream_t st) {
    if (h->io_f32) return launch_wrench_t<N, float, float>(h, a, st);
    return launch_wrench_t<N, double, double>(h, a, st);
}

template <int N>
int launch_long(srbdqp_handle* h, const KArgs& a, hipStream_t st, int maxs) {
    // the compact kernel exists only with <= 2 stance contacts per step at these horizons; anything else (and every
    // fp32 call) goes to the general kernel
    if (uses_wrench(h, maxs, a.qp_span > a.B ? a.qp_span : a.B)) return launch_wrench<N>(h, a, st);
    int rc = launch_compact<N, 2>(h, a, st);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipGetLastError());
    return SRBDQP_OK;
}

// pass: 0 = the only launch of a solve, 1 = first of two (restart follows), 2 = second of two
int launch(srbdqp_handle* h, const KArgs& a, hipStream_t st, int maxs = 4, int pass = 0) {
    if (a.B <= 0) return SRBDQP_OK;
    const bool force_wrench = uses_wrench(h, maxs, a.qp_span > a.B ? a.qp_span : a.B);
    const bool timing = (h->cfg.flags & SRBDQP_FLAG_TIMING) != 0;
    if (timing && pass != 2) { HIP_TRY(h, hipEventRecord(h->ev0, st)); h->ev_mid_valid = false; }
    int rc;
    switch (h->cfg.horizon) {
        case 4: rc = force_wrench ? launch_wrench<4>(h, a, st) : launch_n<4>(h, a, st, maxs); break;
        case 8: rc = force_wrench ? launch_wrench<8>(h, a, st) : launch_n<8>(h, a, st, maxs); break;
        case 10: rc = force_wrench ? launch_wrench<10>(h, a, st) : launch_n<10>(h, a, st, maxs); break;
        case 12: rc = launch_long<12>(h, a, st, maxs); break;
        case 16: rc = launch_long<16>(h, a, st, maxs); break;
        case 20: rc = launch_long<20>(h, a, st, maxs); break;
        case 24: rc = launch_wrench<24>(h, a, st); break;
        default: h->err = "unsupported horizon"; return SRBDQP_E_INVALID;
    }
    if (rc != SRBDQP_OK) return rc;
    if (timing && pass != 1) {
        HIP_TRY(h, hipEventRecord(h->ev1, st));
        h->ev_valid = true;
    }
    return SRBDQP_OK;
}

int ensure_ws(srbdqp_handle* h, size_t bytes) {
    if (bytes <= h->ws_bytes) return SRBDQP_OK;
    if (h->ws) { HIP_TRY(h, hipFree(h->ws)); h->ws = nullptr; h->ws_bytes = 0; }
    size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->ws), want);
    if (e != hipSuccess) { h->err = std::string("hipMalloc workspace: ") + hipGetErrorString(e); return SRBDQP_E_NOMEM; }
    h->ws_bytes = want;
    return SRBDQP_OK;
}

struct Carver {
    char* base; size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <typename T> T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};

// Iteration at which a solve of this handle re-balances rho (0 = never), and how many times it may (*count).  srbdqp_config.rho_restart_iter: > 0 that
// iteration, < 0 off, 0 = automatic:
//  * N > 10: the long horizons have a 1 - 3 % tail of slow QPs and a set-up that is two thirds of a solve: N = 12: 70 x 2, N = 16: 80 x 3, N = 20: 125 x 1,
//    N = 24: 100 x 2 (below) -- N = 20 single support: 98.4 % solved without, 99.3 % with; N = 20 double support with the (0.7, 4) penalties: 17 % of the QPs run past
//    80 iterations, 8 % past 100, 3 % past 125 -- an earlier restart sends too many through a second set-up: 80 instead of 125 cost configs[2] 10 % for 99.90 %
//    instead of 99.89 % solved;
//  * N <= 10: every 55 iterations, up to twice, each time from the rho of the pass before it -- 99.3 % -> 99.93 % of the configs[1] QPs solved inside the same
//    250-iteration cap, at fewer iterations in total (34.8 -> 33.8).  Measured on the one-wave kernel (configs[1], 4 x 4096 QPs; M QP/s, solved, duration of one
//    isolated launch): off 30.6, 0.9929, 0.19 ms; 80 x 1 30.2, 0.9981, 0.21; 65 x 2 30.0, 0.9991, 0.24; 55 x 2 29.6, 0.9993, 0.24; 70 x 3 29.2, 0.9996, 0.27;
//    50 x 3 29.5, 0.9999.
// Since round 4 the rule is the SAME for every kernel and batch size (round 3: at N <= 10 only where the one-wave kernel ran a call of >= 4096 QPs, so one QP
// could end SOLVED in a large batch and MAX_ITER alone): what differs is how the passes run --
//    one-wave kernel:      in place, inside the kernel (srbdqp_setup1.hpp RST), or handed to the next launch on the stream (SRBDQP_FLAG_DEFER_TAIL);
//    every other kernel:   one more launch over the same grid per pass, in which only the workgroups of the QPs the pass before left at its cap do anything
//                          (the rho of a pass reaches the next one through StreamSlot::rhobuf);
//    staged (batch-1) call: the host starts a further pass only when a status[] asks for it (4 % of the calls take a second launch, 1 % a third).
// The price where a call is small: it cannot end before its slowest QP, and a restarted one is a chain of up to three set-ups and 250 iterations (0.22 ms
// against 0.16 ms for 250 iterations at one rho): 512 QPs per call on the one-wave kernel 7.2 -> 5.2 M QP/s.  rho_restart_iter = -1 buys that back.
inline int restart_iter_of(const srbdqp_handle* h, int maxs, int B, bool wave = false, int* count = nullptr) {
    (void)maxs; (void)B; (void)wave;
    const srbdqp_config& c = h->cfg;
    const int rk = resolve_kernel(c);
    if (count) *count = 1;
    if (rk != SRBDQP_KERNEL_COMPACT && rk != SRBDQP_KERNEL_WRENCH) return 0;   // v0 / v1 have no restart
    int r = c.rho_restart_iter;
    const bool automatic = r == 0;
    // automatic, by horizon (C oracle sweeps, 768 - 1024 QPs per case, round 4; solved share and restart events -- i.e. repeated set-ups -- per QP):
    //   N <= 10  55 x 2;   N = 12  70 x 2 (mixed 99.6 -> 100 %, single 99.3 -> 99.9 %, 0.025 -> 0.05 events);   N = 16  80 x 3 (99.4 -> 100 %, 0.04 -> 0.08);
    //   N = 20  125 x 1 (more re-balancings buy nothing inside the 250-iteration cap);   N = 24  100 x 2 (mixed 98.05 -> 98.96 %, double 99.6 -> 99.7 %,
    //   0.13 -> 0.22 events; 80 x 3 would give 99.6 / 99.9 % for 0.32 events: a set-up is two thirds of a solve there)
    const int N = c.horizon;
    if (automatic) r = (N <= 10) ? 55 : (N == 12 ? 70 : (N == 16 ? 80 : (N == 24 ? 100 : 125)));
    if (count) *count = c.rho_restart_count > 0 ? c.rho_restart_count : (!automatic ? 1 : ((N <= 12) ? 2 : (N == 16 ? 3 : (N == 24 ? 2 : 1))));
    // at most three re-balancings, on every kernel (round 5): the one-wave kernel runs its continued passes as three straight copies of the body -- a loop around it
    // costs the whole kernel 30 registers and puts 52 - 72 bytes per lane in scratch memory -- and the rule is the same for every kernel and batch size
    if (count && *count > 3) *count = 3;
    return (r > 0 && r < c.max_iter) ? r : 0;
}

// per-stream restart buffers for batches of up to B QPs with m rows
int ensure_restart_buffers(srbdqp_handle* h, srbdqp_handle::StreamSlot* slot, hipStream_t st, size_t B, size_t m, int nsets = 1) {
    if (slot->rs && slot->rs_items >= B && slot->rs_rows >= m && slot->rs_nsets >= nsets) return SRBDQP_OK;
    HIP_TRY(h, hipStreamSynchronize(st));
    if (slot->tail_st) HIP_TRY(h, hipStreamSynchronize(slot->tail_st));
    if (slot->rs) { HIP_TRY(h, hipFree(slot->rs)); slot->rs = nullptr; }
    auto carve = [&](Carver& c) {
        for (int i = 0; i < nsets; ++i) {
            auto& r = slot->rsets[i];
            r.resid = c.take<float>(B * 4); r.ybuf = c.take<double>(B * m); r.stbuf = c.take<int32_t>(B);
            r.rhobuf[0] = c.take<double>(B); r.rhobuf[1] = c.take<double>(B);
            if (nsets > 1) { for (int j = 0; j < 4; ++j) r.list[j] = c.take<int32_t>(B); r.cnt = c.take<int32_t>(16); }
            r.ev_used = false;
        }
        const auto& r0 = slot->rsets[0];
        slot->resid = r0.resid; slot->ybuf = r0.ybuf; slot->stbuf = r0.stbuf; slot->rhobuf[0] = r0.rhobuf[0]; slot->rhobuf[1] = r0.rhobuf[1];
        slot->ubuf = (B <= 64) ? c.take<double>(B * (m * 12 / 20)) : nullptr;       // (only the staged path uses it: small batches)
    };
    slot->rs_nsets = nsets; slot->last_tail = nullptr;
    Carver sz(nullptr);
    carve(sz);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&slot->rs), sz.off);
    if (e != hipSuccess) { h->err = std::string("hipMalloc restart buffers: ") + hipGetErrorString(e); return SRBDQP_E_NOMEM; }
    Carver cv(slot->rs);
    carve(cv);
    slot->rs_items = B; slot->rs_rows = m;
    return SRBDQP_OK;
}

int ensure_tail_lists(srbdqp_handle* h, srbdqp_handle::StreamSlot* slot, hipStream_t st, size_t B, int rmax) {
    // Round 5: sized for the share of a launch that really continues, not for every QP of it (rmax + 1) times over (1.2 GB per stream at 65,536 QPs).  About 4 % of a
    // configs[1] batch reach the first mark and 1 % the second; a list holds a QUARTER of the largest launch (at least 8192 records, never more than the
    // (rmax + 1) B that can exist): 16,448 records x 3 lists x 1664 B = 82 MB at 65,536 QPs, 41 MB at 4096.  A QP that finds its list full runs its remaining
    // passes in place (srbdqp_setup1.hpp, srbdqp_wave_defer_kernel): same results, it only holds its own launch up as the restart in place would.
    size_t want = B / 4 > 8192 ? B / 4 : 8192;
    if (want > (size_t)(rmax + 1) * B) want = (size_t)(rmax + 1) * B;
    want += 64;
    if (slot->tail && slot->tail_cap >= want) return SRBDQP_OK;
    if (slot->tail) {                                               // growing: finish what the old lists hold, then let go of them
        int rc = flush_slot(h, slot, st);
        if (rc != SRBDQP_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(st));
        HIP_TRY(h, hipFree(slot->tail)); slot->tail = nullptr; slot->tail_cap = 0;
    }
    if (!slot->tail_cnt) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&slot->tail_cnt), 64));
    }
    const size_t cap = want;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&slot->tail), 3 * cap * srbdqp::kTailRecDoubles * sizeof(double));
    if (e != hipSuccess) { h->err = std::string("hipMalloc tail lists: ") + hipGetErrorString(e); return SRBDQP_E_NOMEM; }
    slot->tail_cap = cap;
    slot->tail_k = 0;
    for (auto& v : slot->tail_hist) v = 0;
    HIP_TRY(h, hipMemsetAsync(slot->tail_cnt, 0, 64, st));
    return SRBDQP_OK;
}

// run what the lists of this launch stream still hold: one launch, one workgroup per record the lists can hold, each running every pass its QP has left;
// enqueued on st, no host synchronisation
int flush_slot(srbdqp_handle* h, srbdqp_handle::StreamSlot* slot, hipStream_t st) {
    if (!slot->tail || !slot->tail_live) return SRBDQP_OK;
    int rcount = 1;
    const int maxs = h->cfg.max_contacts_per_step > 0 ? h->cfg.max_contacts_per_step : 2;
    const int restart = restart_iter_of(h, maxs, kRestartMinBatch, true, &rcount);
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_args(h->cfg, a);
    a.B = 0; a.restart_every = restart; a.restart_max = rcount;
    {   // ONE launch: a flush workgroup runs every pass its QP has left (srbdqp_setup1.hpp, FLUSH)
        const int rc = launch_wave_defer_any(h, a, st, slot, maxs <= 2 ? 2 : 4);
        if (rc != SRBDQP_OK) return rc;
    }
    slot->tail_live = false;
    return SRBDQP_OK;
}

// Restart pass p (1 = the first re-balancing) of a solve whose first pass ran with the arguments a1 (status / y_out / resid_out set, max_iter = the restart
// period): the same grid again; the workgroup of a QP that the pass before left at its cap re-balances its rho from the maxima of its last check -- from the rho of
// THAT pass -- and continues from its own (x, y), every other workgroup leaves at once.  No selection kernel and no list between the passes: a one-workgroup kernel
// queued behind a chip-filling launch of another stream waits ~150 us for its turn at the dispatcher (rocprofv3 timeline, round 2), which cost more than the
// pass itself.  *last_out: no further pass may follow (p = rcount, or the cap on the total comes first: oracle solve_with_restart).
int srbdqp_restart_pass(srbdqp_handle* h, const KArgs& a1, hipStream_t st, int maxs, bool signal, srbdqp_handle::StreamSlot* slot, int p, int rcount, bool* last_out) {
    const int every = a1.max_iter;
    const int done = p * every;                             // iterations of a QP that every pass so far left at its cap
    const int left = h->cfg.max_iter - done;                // the cap is on the total
    const bool last = p >= rcount || every >= left;
    if (last_out) *last_out = last;
    KArgs a2 = a1;
    a2.resid_in = a1.resid_out;
    a2.warm_u = a1.u_dev ? a1.u_dev : a1.u_out;             // newtons, as a caller's warm start would be (the device copy of a staged first pass)
    a2.u_dev = (a1.u_dev && !last) ? a1.u_dev : nullptr;    // ... refreshed by every pass another one may follow
    a2.warm_y = a1.y_out;
    a2.max_iter = last ? left : every;
    a2.iters_base = done;
    a2.resid_out = last ? nullptr : a1.resid_out;           // (a workgroup reads its entry when it starts and writes it when it ends)
    a2.rho_qp = (p == 1) ? a1.rho_qp : slot->rhobuf[p % 2];
    a2.rho_out = last ? nullptr : slot->rhobuf[(p + 1) % 2];
    if (signal) { a2.done_flag = h->done_dev; a2.done_count = h->done_count; a2.done_value = h->done_seq; }
    else { a2.done_flag = nullptr; a2.done_count = nullptr; }
    return launch(h, a2, st, maxs, 2);
}

}  // namespace

extern "C" {

const char* srbdqp_version(void) { return "srbdqp 0.1 (gfx950, fp64)"; }

int srbdqp_default_config(srbdqp_config* c) {
    if (!c) return SRBDQP_E_INVALID;
    // the caller says how large ITS struct is: a binding built against an older header (a shorter struct) is refused here instead of
    // being written past its end
    if (c->struct_size != (int32_t)sizeof(srbdqp_config)) { g_create_err = "srbdqp_default_config: set cfg->struct_size = sizeof(srbdqp_config) before the call (ABI check)"; return SRBDQP_E_INVALID; }
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (int32_t)sizeof(srbdqp_config);
    c->horizon = 10;
    c->device = 0;
    c->flags = 0;
    c->kernel = SRBDQP_KERNEL_AUTO;
    c->max_iter = 250;
    c->check_every = 5;
    c->dt = 0.04;
    c->mass = 34.13385728;
    c->inertia[0] = 8.20564e-2; c->inertia[1] = 8.05015e-2; c->inertia[2] = 0.32353e-2;
    c->mu = 0.8;
    c->fz_min = 10.0; c->fz_max = 1000.0;
    const double q[13] = {300.0, 300.0, 150.0, 400.0, 400.0, 600.0, 1.0, 1.0, 1.0, 20.0, 20.0, 20.0, 0.0};
    for (int i = 0; i < 13; ++i) c->q_diag[i] = q[i];
    c->r_diag = 1.0e-4;
    c->force_scale = 100.0;
    c->rho = 0.0; c->rho_eq_scale = 1.0e3; c->sigma = 1.0e-6; c->alpha = 1.6;
    c->eps_abs = 1.0e-6; c->eps_rel = 1.0e-6;
    c->rho_restart_iter = 0; c->rho_restart_count = 0;
    c->rho_fz_scale = 0.0;
    return SRBDQP_OK;
}

int srbdqp_create(const srbdqp_config* cfg, srbdqp_handle** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return SRBDQP_E_INVALID; }
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(srbdqp_config)) { g_create_err = "srbdqp_config.struct_size mismatch"; return SRBDQP_E_INVALID; }
    if (!horizon_supported(cfg->horizon)) { g_create_err = "unsupported horizon (N in {4, 8, 10, 12, 16, 20, 24})"; return SRBDQP_E_INVALID; }
    if (cfg->kernel != SRBDQP_KERNEL_AUTO && cfg->kernel != SRBDQP_KERNEL_COMPACT && cfg->kernel != SRBDQP_KERNEL_SPLIT &&
        cfg->kernel != SRBDQP_KERNEL_WAVE && cfg->kernel != SRBDQP_KERNEL_WRENCH) { g_create_err = "unknown srbdqp_config.kernel (the round-1 baselines v0 / v1 are retired)"; return SRBDQP_E_INVALID; }
    if (!(cfg->dt > 0) || !(cfg->mass > 0) || !(cfg->force_scale > 0) || !(cfg->rho >= 0) || !(cfg->sigma > 0) ||
        cfg->max_iter < 1 || cfg->check_every < 1 || !(cfg->mu >= 0) || cfg->max_contacts_per_step < 0 || cfg->max_contacts_per_step > 4) { g_create_err = "invalid constants"; return SRBDQP_E_INVALID; }
    for (int i = 0; i < 13; ++i) if (!(cfg->q_diag[i] >= 0)) { g_create_err = "negative q_diag"; return SRBDQP_E_INVALID; }
    for (int i = 0; i < 3; ++i) if (!(cfg->inertia[i] > 0)) { g_create_err = "inertia must be positive"; return SRBDQP_E_INVALID; }
    if (!(cfg->alpha > 0 && cfg->alpha < 2) || !(cfg->eps_abs >= 0) || !(cfg->eps_rel >= 0) || !(cfg->eps_abs + cfg->eps_rel > 0) ||
        !(cfg->fz_min >= 0) || !(cfg->fz_min <= cfg->fz_max) || !(cfg->r_diag >= 0) || !(cfg->rho_eq_scale > 0) || !(cfg->rho_fz_scale >= 0)) {
        g_create_err = "invalid constants (alpha in (0, 2), eps >= 0, 0 <= fz_min <= fz_max, r_diag >= 0)"; return SRBDQP_E_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_create_err = "no usable HIP device (this engine has no CPU fallback)";
        return SRBDQP_E_NO_DEVICE;
    }
    srbdqp_handle* h = new (std::nothrow) srbdqp_handle();
    if (!h) { g_create_err = "out of host memory"; return SRBDQP_E_NOMEM; }
    h->cfg = *cfg;
    if (h->cfg.rho == 0.0) h->cfg.rho = 0.7;                    // auto (oracle auto_rho()): friction rows
    if (h->cfg.rho_fz_scale == 0.0) h->cfg.rho_fz_scale = 4.0;   // auto (oracle auto_rho_fz_scale()): normal-force rows at 4 rho
    auto fail = [&](const char* what, hipError_t er) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(er);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        if (h->stage_host) (void)hipHostFree(h->stage_host);
        if (h->done_count) (void)hipFree(h->done_count);
        delete h;
        return SRBDQP_E_HIP;
    };
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&h->ev_mid)) != hipSuccess) return fail("hipEventCreate", e);
    {   // staging slab for the low-latency path
        const size_t N = (size_t)cfg->horizon, n = 12 * N, m = 20 * N, cap = 16;
        auto carve = [&](char* base, srbdqp_stage& st) {
            Carver c(base);
            st.capacity = (int32_t)cap;
            st.x0 = c.take<double>(cap * 13); st.x_ref = c.take<double>(cap * N * 13); st.foot = c.take<double>(cap * N * 12);
            st.contact = c.take<uint8_t>(cap * N * 4); st.pcom = c.take<double>(cap * N * 3);
            st.warm_u = c.take<double>(cap * n); st.warm_y = c.take<double>(cap * m);
            st.u = c.take<double>(cap * n); st.x = c.take<double>(cap * (N + 1) * 13); st.y = c.take<double>(cap * m);
            st.status = c.take<int32_t>(cap); st.iters = c.take<int32_t>(cap);
            return c.off;
        };
        srbdqp_stage tmp{};
        const size_t body = (carve(reinterpret_cast<char*>(4096), tmp) + 255) & ~size_t(255);   // dry run for the size
        const size_t bytes = body + 256;
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->stage_host), bytes, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return fail("hipHostMalloc(staging)", e);
        if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->stage_dev), h->stage_host, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
        std::memset(h->stage_host, 0, bytes);
        carve(h->stage_host, h->stage_h);
        carve(h->stage_dev, h->stage_d);
        h->done_host = reinterpret_cast<volatile int32_t*>(h->stage_host + body);
        h->done_dev = reinterpret_cast<int32_t*>(h->stage_dev + body);
        if ((e = hipMalloc(reinterpret_cast<void**>(&h->done_count), 64)) != hipSuccess) return fail("hipMalloc(done counter)", e);
        if ((e = hipMemset(h->done_count, 0, 64)) != hipSuccess) return fail("hipMemset(done counter)", e);
    }
    *out = h;
    return SRBDQP_OK;
}

int srbdqp_destroy(srbdqp_handle* h) {
    if (!h) return SRBDQP_OK;
    (void)hipSetDevice(h->cfg.device);
    delete h->aql;                                 // (waits for its last kernel)
    h->aql = nullptr;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ws) (void)hipFree(h->ws);
    for (auto& sl : h->slots) {
        if (sl.perm) (void)hipFree(sl.perm); if (sl.ws) (void)hipFree(sl.ws); if (sl.rs) (void)hipFree(sl.rs);
        if (sl.tail) (void)hipFree(sl.tail); if (sl.tail_cnt) (void)hipFree(sl.tail_cnt);
        if (sl.tail_st) { (void)hipStreamSynchronize(sl.tail_st); (void)hipStreamDestroy(sl.tail_st); }
        if (sl.ev_main) (void)hipEventDestroy(sl.ev_main);
        for (auto& r : sl.rsets) if (r.ev_tail) (void)hipEventDestroy(r.ev_tail);
    }
    if (h->done_count) (void)hipFree(h->done_count);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SRBDQP_OK;
}

const char* srbdqp_last_error(const srbdqp_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

const char* srbdqp_kernel_name(const srbdqp_handle* h) { return h ? h->kname : "none"; }

const char* srbdqp_batch1_launch_path(const srbdqp_handle* h) {
    if (!h || !h->aql_tried) return "undecided";
    if (h->aql) return "aql";
    static thread_local std::string s;
    s = "hip: " + h->aql_why;
    return s.c_str();
}

int srbdqp_stage_ptrs(srbdqp_handle* h, srbdqp_stage* out) {
    if (!h || !out) return SRBDQP_E_INVALID;
    *out = h->stage_h;
    return SRBDQP_OK;
}

int srbdqp_solve_staged_f64(srbdqp_handle* h, int32_t B, int32_t use_pcom, int32_t use_warm, int32_t want_x, int32_t want_y) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || B > h->stage_h.capacity) { h->err = "staged batch exceeds the staging capacity"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    {   // (the kernel of the call before this one published its completion word before it ended)
        const int rq = aql_quiesce(h);
        if (rq != SRBDQP_OK) return rq;
    }
    const srbdqp_stage& d = h->stage_d;
    h->staged_neff = 0;
    h->staged_call = true;
    h->done_cs = false;            // (set again by a launch that publishes a checksum)
    struct Reset { srbdqp_handle* h; ~Reset() { h->signal_next = false; h->staged_call = false; h->maxs_override = 0; h->staged_neff = 0; } } reset_on_return{h};
    {   // same per-batch kernel choice as the host-buffer API, from the staged contact flags
        int worst = 0;
        const uint8_t* c = h->stage_h.contact;
        const size_t N = (size_t)h->cfg.horizon;
        for (size_t b = 0; b < (size_t)B; ++b) {
            int na = 0;
            for (size_t q = b * N; q < (b + 1) * N; ++q) {
                const int cnt = (c[4 * q] != 0) + (c[4 * q + 1] != 0) + (c[4 * q + 2] != 0) + (c[4 * q + 3] != 0);
                if (cnt > worst) worst = cnt;
                na += cnt;
            }
            if (3 * na > h->staged_neff) h->staged_neff = 3 * na;
        }
        if (h->cfg.max_contacts_per_step <= 0) h->maxs_override = (worst <= 2) ? 2 : 4;
    }
    // completion: the compact kernel publishes a sequence number in host memory after its outputs (signal_done());
    // spinning on it skips the stream's completion interrupt (~15 us).  Other kernel variants: stream synchronise.
    const int rk = resolve_kernel(h->cfg);
    const bool spin = (rk == SRBDQP_KERNEL_COMPACT || rk == SRBDQP_KERNEL_WRENCH) && !(h->cfg.flags & SRBDQP_FLAG_NO_SPIN);
    // (a word that came with a checksum -- KArgs::done_cs -- counts once the outputs read back agree with it: they travel without a fence in front of the word)
    auto outputs_there = [&]() -> bool {
        if (!h->done_cs) return true;
        const size_t Nn = (size_t)h->cfg.horizon;
        const volatile uint64_t* u = reinterpret_cast<const volatile uint64_t*>(h->stage_h.u);
        uint64_t x = 0;
        for (size_t i = 0; i < 12 * Nn; ++i) x ^= u[i];
        if (h->done_cs_x) {
            const volatile uint64_t* xs = reinterpret_cast<const volatile uint64_t*>(h->stage_h.x);
            for (size_t i = 0; i < 13 * (Nn + 1); ++i) x ^= xs[i];
        }
        x ^= (uint64_t)(uint32_t)*reinterpret_cast<const volatile int32_t*>(h->stage_h.status) | ((uint64_t)(uint32_t)*reinterpret_cast<const volatile int32_t*>(h->stage_h.iters) << 32);
        // the records: {sequence number, how many records, XOR of one row of 16 lanes} every 16 bytes (srbdqp_common.hpp signal_done_checksum)
        const volatile int32_t* rec = h->done_host;
        const int32_t nrec = rec[1];
        if (nrec < 1 || nrec > 16) return false;
        for (int32_t r = 0; r < nrec; ++r) {
            if (rec[4 * r] != h->done_seq || rec[4 * r + 1] != nrec) return false;
            x ^= (uint64_t)(uint32_t)rec[4 * r + 2] | ((uint64_t)(uint32_t)rec[4 * r + 3] << 32);
        }
        return x == 0;
    };
    auto wait_done = [&]() -> int {
        if (spin) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned polls = 0;
            while (*h->done_host != h->done_seq || !outputs_there()) {
                if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    // slow or failed launch: hand over to the runtime (reports a fault, or returns once the kernel is done)
                    const int rq = aql_quiesce(h);
                    if (rq != SRBDQP_OK) return rq;
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            return SRBDQP_OK;
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return SRBDQP_OK;
    };
    if (spin) { h->done_seq = (h->done_seq == INT32_MAX) ? 1 : h->done_seq + 1; h->signal_next = true; }
    const int maxs = h->maxs_override ? h->maxs_override : (h->cfg.max_contacts_per_step > 0 ? h->cfg.max_contacts_per_step : 4);
    h->lazy_restart = true;                                 // the rho restart costs two more launches: only when needed
    const int32_t* hint_keep = h->sched_hint;               // the dispatch hint belongs to the device-buffer API
    h->sched_hint = nullptr;
    int rc = srbdqp_solve_batch_device_f64(h, B, d.x0, d.x_ref, d.foot, d.contact, use_pcom ? d.pcom : nullptr,
                                           use_warm ? d.warm_u : nullptr, use_warm ? d.warm_y : nullptr, d.u,
                                           want_x ? d.x : nullptr, want_y ? d.y : nullptr, d.status, d.iters, h->stream);
    h->sched_hint = hint_keep;
    h->lazy_restart = false;
    // (staged_call / maxs_override / staged_neff stay set until this call returns: a restart pass below must choose the same kernel)
    if (rc != SRBDQP_OK) return rc;
    rc = wait_done();
    if (rc != SRBDQP_OK) return rc;
    if (h->lazy_pending) {          // the solve above was the first pass of a multi-pass solve (not: restarted in place, or no restart at all)
        h->lazy_pending = false;
        for (int p = 1; p <= h->lazy_rcount; ++p) {         // a further pass only when a status asks for it
            bool capped = false;
            for (int32_t q = 0; q < B; ++q) capped |= (h->stage_h.status[q] == SRBDQP_MAX_ITER);
            if (!capped) break;
            if (spin) h->done_seq = (h->done_seq == INT32_MAX) ? 1 : h->done_seq + 1;
            bool last = true;
            rc = srbdqp_restart_pass(h, h->last_args, h->stream, maxs, spin, h->lazy_slot, p, h->lazy_rcount, &last);
            if (rc != SRBDQP_OK) return rc;
            rc = wait_done();
            if (rc != SRBDQP_OK || last) break;
        }
    }
    return rc;
}

int srbdqp_update_f64(srbdqp_handle* h, const double* x0, const double* x_ref, const double* foot, const uint8_t* contact,
                      const double* pcom, double* u0_out, double* u_out, double* x_out, int32_t* status, int32_t* iters) {
    if (!h) return SRBDQP_E_INVALID;
    if (!x0 || !x_ref || !foot || !contact || !u0_out) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    const size_t N = (size_t)h->cfg.horizon;
    const srbdqp_stage& s = h->stage_h;
    // into the pinned, GPU-mapped staging arrays (2.4 KB at N = 10; an argument that IS the staging array is left where it is)
    if (x0 != s.x0) std::memcpy(s.x0, x0, 13 * sizeof(double));
    if (x_ref != s.x_ref) std::memcpy(s.x_ref, x_ref, N * 13 * sizeof(double));
    if (foot != s.foot) std::memcpy(s.foot, foot, N * 12 * sizeof(double));
    if (contact != s.contact) std::memcpy(s.contact, contact, N * 4);
    if (pcom && pcom != s.pcom) std::memcpy(s.pcom, pcom, N * 3 * sizeof(double));
    const int rc = srbdqp_solve_staged_f64(h, 1, pcom != nullptr, 0, x_out != nullptr, 0);
    if (rc != SRBDQP_OK) return rc;
    if (u0_out != s.u) std::memcpy(u0_out, s.u, 12 * sizeof(double));
    if (u_out && u_out != s.u) std::memcpy(u_out, s.u, N * 12 * sizeof(double));
    if (x_out && x_out != s.x) std::memcpy(x_out, s.x, (N + 1) * 13 * sizeof(double));
    if (status) *status = s.status[0];
    if (iters) *iters = s.iters[0];
    return SRBDQP_OK;
}

namespace {
int staged_maxs(srbdqp_handle* h, int32_t B) {
    if (h->cfg.max_contacts_per_step > 0) return h->cfg.max_contacts_per_step;
    int worst = 0;
    const uint8_t* c = h->stage_h.contact;
    for (size_t q = 0; q < (size_t)B * h->cfg.horizon; ++q) {
        const int cnt = (c[4 * q] != 0) + (c[4 * q + 1] != 0) + (c[4 * q + 2] != 0) + (c[4 * q + 3] != 0);
        if (cnt > worst) worst = cnt;
    }
    return (worst <= 2) ? 2 : 4;
}
KArgs staged_args(srbdqp_handle* h, int32_t B, bool use_pcom, bool want_x, bool want_y) {
    const srbdqp_stage& d = h->stage_d;
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_args(h->cfg, a);
    a.x0 = d.x0; a.xref = d.x_ref; a.foot = d.foot; a.contact = d.contact; a.pcom = use_pcom ? d.pcom : nullptr;
    a.u_out = d.u; a.x_out = want_x ? d.x : nullptr; a.y_out = want_y ? d.y : nullptr;
    a.status = d.status; a.iters = d.iters;
    a.B = B; a.mode = 0;
    return a;
}
}  // namespace

int srbdqp_prepare_staged_f64(srbdqp_handle* h, int32_t B, int32_t use_pcom) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || B > h->stage_h.capacity) { h->err = "staged batch exceeds the staging capacity"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    {
        const int rq = aql_quiesce(h);
        if (rq != SRBDQP_OK) return rq;
    }
    const int maxs = staged_maxs(h, B);
    KArgs a = staged_args(h, B, use_pcom != 0, true, false);
    const int rc = launch_two_phase_any(h, a, h->stream, maxs, 0);
    if (rc != SRBDQP_OK) return rc;
    h->prepared_B = B; h->prepared_maxs = maxs; h->prepared_pcom = use_pcom != 0;
    return SRBDQP_OK;   // asynchronous: the set-up runs behind the handle's stream; srbdqp_solve_prepared_f64 queues behind it
}

int srbdqp_solve_prepared_f64(srbdqp_handle* h, int32_t B, int32_t want_x, int32_t want_y) {
    if (!h) return SRBDQP_E_INVALID;
    if (B <= 0 || B != h->prepared_B) { h->err = "srbdqp_solve_prepared_f64: no set-up of this batch size is pending (srbdqp_prepare_staged_f64)"; return SRBDQP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const bool spin = !(h->cfg.flags & SRBDQP_FLAG_NO_SPIN);
    KArgs a = staged_args(h, B, h->prepared_pcom, want_x != 0, want_y != 0);
    if (spin) {
        h->done_seq = (h->done_seq == INT32_MAX) ? 1 : h->done_seq + 1;
        a.done_flag = h->done_dev; a.done_count = h->done_count; a.done_value = h->done_seq;
    }
    h->prepared_B = 0;
    const int rc = launch_two_phase_any(h, a, h->stream, h->prepared_maxs, 1);
    if (rc != SRBDQP_OK) return rc;
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned polls = 0;
        while (*h->done_host != h->done_seq) {
            if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return SRBDQP_OK;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SRBDQP_OK;
}

int srbdqp_set_schedule_hint(srbdqp_handle* h, const int32_t* device_iters_prev, int32_t length) {
    if (!h) return SRBDQP_E_INVALID;
    if (device_iters_prev && length < 0) { h->err = "negative hint length"; return SRBDQP_E_INVALID; }
    h->sched_hint = device_iters_prev;
    h->sched_hint_len = device_iters_prev ? (size_t)length : 0;
    return SRBDQP_OK;
}

int srbdqp_set_stamp_buffer(srbdqp_handle* h, void* device_ptr) {
    if (!h) return SRBDQP_E_INVALID;
    h->stamps = reinterpret_cast<long long*>(device_ptr);
    return SRBDQP_OK;
}

int srbdqp_shard_range(int64_t total, int32_t world, int32_t rank, int64_t* first, int64_t* count) {
    if (world < 1 || rank < 0 || rank >= world || total < 0 || !first || !count) return SRBDQP_E_INVALID;
    const int64_t base = total / world, rem = total % world;
    *first = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1 : 0);
    return SRBDQP_OK;
}

int srbdqp_gather_u0_f64(srbdqp_handle* h, const double* u_local, int64_t B_local, double* u0_all, void* rccl_comm, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    if (!u_local || !u0_all || !rccl_comm || B_local < 0) { h->err = "srbdqp_gather_u0_f64: null pointer or negative count"; return SRBDQP_E_INVALID; }
    // RCCL by name, once per process: the library does not link it (a single-GPU consumer never needs it)
    struct Rccl {
        int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
        int (*user_rank)(void*, int*) = nullptr;
        int (*count)(void*, int*) = nullptr;
        const char* (*error_string)(int) = nullptr;
        bool ok = false;
        Rccl() {
            void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) return;
            all_gather = reinterpret_cast<decltype(all_gather)>(dlsym(lib, "ncclAllGather"));
            user_rank = reinterpret_cast<decltype(user_rank)>(dlsym(lib, "ncclCommUserRank"));
            count = reinterpret_cast<decltype(count)>(dlsym(lib, "ncclCommCount"));
            error_string = reinterpret_cast<decltype(error_string)>(dlsym(lib, "ncclGetErrorString"));
            ok = all_gather && user_rank && count;
        }
    };
    static const Rccl rccl;
    if (!rccl.ok) { h->err = "srbdqp_gather_u0_f64: librccl.so.1 (ncclAllGather, ncclCommUserRank, ncclCommCount) is not loadable"; return SRBDQP_E_HIP; }
    auto fail = [&](const char* what, int rc) { h->err = std::string("srbdqp_gather_u0_f64: ") + what + ": " + (rccl.error_string ? rccl.error_string(rc) : "RCCL error"); return SRBDQP_E_HIP; };
    int rank = 0, world = 0, rc;
    if ((rc = rccl.user_rank(rccl_comm, &rank)) != 0) return fail("ncclCommUserRank", rc);
    if ((rc = rccl.count(rccl_comm, &world)) != 0) return fail("ncclCommCount", rc);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
    if (B_local == 0) return SRBDQP_OK;
    double* slot = u0_all + (size_t)rank * (size_t)B_local * 12;
    const long long items = (long long)B_local * 12;
    hipLaunchKernelGGL(srbdqp::srbdqp_pack_u0_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, u_local, slot, items, h->cfg.horizon);
    HIP_TRY(h, hipGetLastError());
    if ((rc = rccl.all_gather(slot, u0_all, (size_t)items, /* ncclDouble */ 8, rccl_comm, st)) != 0) return fail("ncclAllGather", rc);
    return SRBDQP_OK;
}

int srbdqp_flush(srbdqp_handle* h, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (auto& sl : h->slots) {
        if (!sl.used) continue;
        if (stream && sl.st != reinterpret_cast<hipStream_t>(stream)) continue;
        if (sl.last_tail) { HIP_TRY(h, hipStreamWaitEvent(sl.st, sl.last_tail, 0)); sl.last_tail = nullptr; }   // restart passes on the slot's tail stream
        if (!sl.tail_live) continue;
        const int rc = flush_slot(h, &sl, sl.st);
        if (rc != SRBDQP_OK) return rc;
    }
    return SRBDQP_OK;
}

int srbdqp_synchronize(srbdqp_handle* h) {
    if (!h) return SRBDQP_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->cfg.device));       // (the flush below launches a kernel: a thread that drives several devices may have another one current)
    for (auto& sl : h->slots) {                    // deferred tails of the handle's own stream are part of "everything enqueued"
        if (!sl.used || sl.st != h->stream) continue;
        if (sl.last_tail) { HIP_TRY(h, hipStreamWaitEvent(sl.st, sl.last_tail, 0)); sl.last_tail = nullptr; }
        if (sl.tail_live) { const int rc = flush_slot(h, &sl, sl.st); if (rc != SRBDQP_OK) return rc; }
    }
    const int rq = aql_quiesce(h);
    if (rq != SRBDQP_OK) return rq;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SRBDQP_OK;
}

double srbdqp_last_kernel_ms(srbdqp_handle* h) {
    if (!h || !h->ev_valid) return -1.0;
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

int srbdqp_last_kernel_parts_ms(srbdqp_handle* h, double* setup_ms, double* admm_ms) {
    if (!h || !setup_ms || !admm_ms) return SRBDQP_E_INVALID;
    if (!h->ev_valid || !h->ev_mid_valid) { h->err = "the last solve was not a timed split-pipeline solve"; return SRBDQP_E_INVALID; }
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float a = -1.0f, b = -1.0f;
    HIP_TRY(h, hipEventElapsedTime(&a, h->ev0, h->ev_mid));
    HIP_TRY(h, hipEventElapsedTime(&b, h->ev_mid, h->ev1));
    *setup_ms = (double)a; *admm_ms = (double)b;
    return SRBDQP_OK;
}

}  // extern "C"

namespace {

// SRBDQP_FLAG_DEFER_TAIL on a kernel that restarts by further launches: the first pass on the caller's stream, the restart passes on the slot's own tail stream
// behind an event -- beside whatever the caller enqueues next, e.g. the next batch's first pass -- each taking the list of QPs the pass before it left at its cap
// as its dispatch order (working workgroups first; the rest of the grid leaves after one scalar load).  Outputs of the continued QPs arrive when the tail
// stream gets there; srbdqp_flush() makes the caller's stream wait for it.
int solve_deferred_passes(srbdqp_handle* h, const KArgs& a, hipStream_t lst, int maxs, int restart, int rcount) {
    auto* slot = stream_slot(h, lst);
    if (!slot) return SRBDQP_E_INVALID;
    const size_t m = 20 * (size_t)h->cfg.horizon;
    int rc = ensure_restart_buffers(h, slot, lst, (size_t)a.B, m, 3);
    if (rc != SRBDQP_OK) return rc;
    if (!slot->tail_st) {
        int least = 0, greatest = 0;
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(h, hipStreamCreateWithPriority(&slot->tail_st, hipStreamNonBlocking, greatest));
        HIP_TRY(h, hipEventCreateWithFlags(&slot->ev_main, hipEventDisableTiming));
        for (auto& r : slot->rsets) HIP_TRY(h, hipEventCreateWithFlags(&r.ev_tail, hipEventDisableTiming));
    }
    auto& set = slot->rsets[slot->rs_k++ % 3];
    if (set.ev_used) HIP_TRY(h, hipStreamWaitEvent(lst, set.ev_tail, 0));      // the set's last user (three solves ago) has finished its passes
    HIP_TRY(h, hipMemsetAsync(set.cnt, 0, 16 * sizeof(int32_t), lst));
    KArgs a1 = a;
    a1.max_iter = restart;
    a1.resid_out = set.resid;
    if (!a1.y_out) { a1.y_out = set.ybuf; a1.y_capped_only = 1; }
    if (!a1.status) a1.status = set.stbuf;
    a1.cap_list = set.list[0]; a1.cap_count = set.cnt;
    rc = launch(h, a1, lst, maxs, 1);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipEventRecord(slot->ev_main, lst));
    HIP_TRY(h, hipStreamWaitEvent(slot->tail_st, slot->ev_main, 0));
    for (int p = 1; p <= rcount; ++p) {
        const int done = p * restart, left = h->cfg.max_iter - done;
        const bool last = p >= rcount || restart >= left;
        KArgs a2 = a1;
        a2.resid_in = set.resid; a2.resid_out = last ? nullptr : set.resid;
        a2.warm_u = a1.u_out; a2.warm_y = a1.y_out;
        a2.max_iter = last ? left : restart; a2.iters_base = done;
        a2.rho_qp = (p == 1) ? a1.rho_qp : set.rhobuf[p % 2];
        a2.rho_out = last ? nullptr : set.rhobuf[(p + 1) % 2];
        a2.perm = set.list[p - 1]; a2.count_ptr = set.cnt + (p - 1);
        a2.cap_list = last ? nullptr : set.list[p]; a2.cap_count = last ? nullptr : set.cnt + p;
        rc = launch(h, a2, slot->tail_st, maxs, 2);
        if (rc != SRBDQP_OK) return rc;
        if (last) break;
    }
    HIP_TRY(h, hipEventRecord(set.ev_tail, slot->tail_st));
    set.ev_used = true;
    slot->last_tail = set.ev_tail;
    return SRBDQP_OK;
}

// common body of the device-buffer entry points; the element type of the caller's buffers is h->io_f32 ? float : double
int solve_device_impl(srbdqp_handle* h, int32_t B, const void* x0, const void* x_ref, const void* foot, const uint8_t* contact,
                      const void* pcom, const void* warm_u, const void* warm_y, void* u_out, void* x_out, void* y_out,
                      int32_t* status, int32_t* iters, void* stream) {
    if (B < 0 || (B > 0 && (!x0 || !x_ref || !foot || !contact || !u_out))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->staged_call) {
        const int rq = aql_quiesce(h);
        if (rq != SRBDQP_OK) return rq;
    }
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_args(h->cfg, a);
    a.x0 = static_cast<const double*>(x0); a.xref = static_cast<const double*>(x_ref); a.foot = static_cast<const double*>(foot);
    a.contact = contact; a.pcom = static_cast<const double*>(pcom);
    a.warm_u = static_cast<const double*>(warm_u); a.warm_y = static_cast<const double*>(warm_y);
    a.u_out = static_cast<double*>(u_out); a.x_out = static_cast<double*>(x_out); a.y_out = static_cast<double*>(y_out);
    a.status = status; a.iters = iters;
    a.B = B; a.mode = 0; a.stamps = h->stamps;
    if (h->signal_next) { a.done_flag = h->done_dev; a.done_count = h->done_count; a.done_value = h->done_seq; }
    hipStream_t lst = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
    if (h->sched_hint && B > 1 && (size_t)B <= h->sched_hint_len) {
        auto* slot = stream_slot(h, lst);
        if (!slot) return SRBDQP_E_INVALID;
        if ((size_t)B > slot->perm_cap) {
            HIP_TRY(h, hipStreamSynchronize(lst));          // a previous launch on this stream may still read the old order
            if (slot->perm) HIP_TRY(h, hipFree(slot->perm));
            slot->perm = nullptr; slot->perm_cap = 0;
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&slot->perm), sizeof(int32_t) * (size_t)B));
            slot->perm_cap = (size_t)B;
        }
        hipLaunchKernelGGL(srbdqp_schedule_kernel, dim3(1), dim3(1024), 0, lst, h->sched_hint, slot->perm, (int)B);
        a.perm = slot->perm;
    }
    int maxs = h->maxs_override ? h->maxs_override : (h->cfg.max_contacts_per_step > 0 ? h->cfg.max_contacts_per_step : 4);
    h->lazy_pending = false;
    const bool wave = uses_wave(h, maxs, B, a.stamps != nullptr, a.done_flag != nullptr);
    int rcount = 1;
    const int restart = (h->stamps || B < 1) ? 0 : restart_iter_of(h, maxs, B, wave, &rcount);
    if (restart && wave) { a.restart_every = restart; a.restart_max = rcount; }   // the one-wave kernel restarts in place
    if (restart && wave && (h->cfg.flags & SRBDQP_FLAG_DEFER_TAIL) && !a.stamps && !a.done_flag && !h->io_f32 && rcount <= srbdqp_handle::StreamSlot::kTailHist) {
        // ... or not at all: continuations deferred to the next launch on this stream (srbdqp_flush() completes them)
        auto* slot = stream_slot(h, lst);
        if (!slot) return SRBDQP_E_INVALID;
        int rc = ensure_tail_lists(h, slot, lst, (size_t)B, rcount);
        if (rc != SRBDQP_OK) return rc;
        return launch_wave_defer_any(h, a, lst, slot, maxs);
    }
    if (!restart || wave) return launch(h, a, lst, maxs);
    if ((h->cfg.flags & SRBDQP_FLAG_DEFER_TAIL) && !h->lazy_restart && !a.stamps && !a.done_flag && rcount <= 3) return solve_deferred_passes(h, a, lst, maxs, restart, rcount);

    // ---- several passes: cap the first at rho_restart_iter, re-balance rho for the QPs that reach it, continue those (up to rcount times)
    auto* slot = stream_slot(h, lst);
    if (!slot) return SRBDQP_E_INVALID;
    const size_t m = 20 * (size_t)h->cfg.horizon;
    int rc = ensure_restart_buffers(h, slot, lst, (size_t)B, m);
    if (rc != SRBDQP_OK) return rc;
    // (a handle with SRBDQP_FLAG_DEFER_TAIL whose device-buffer solves ran their passes on the tail stream: slot->resid / ybuf / stbuf / rhobuf alias set 0 of the
    //  rotation, and a staged or completion-word solve on the same stream comes through here -- it must not overwrite what a tail pass still reads)
    if (slot->rs_nsets > 1)
        for (auto& r : slot->rsets) if (r.ev_used) HIP_TRY(h, hipStreamWaitEvent(lst, r.ev_tail, 0));
    const bool lazy = h->lazy_restart;                      // staged path: the host looks at status[] before a second pass
    KArgs a1 = a;
    a1.max_iter = restart;
    a1.resid_out = slot->resid;
    if (!a1.y_out) { a1.y_out = slot->ybuf; a1.y_capped_only = 1; }
    if (!a1.status) a1.status = slot->stbuf;
    if (!lazy) { a1.done_flag = nullptr; a1.done_count = nullptr; }
    if (lazy && slot->ubuf && !h->io_f32) a1.u_dev = slot->ubuf;   // (the staged arrays are host memory: the pass behind this one reads its warm start on the device)
    rc = launch(h, a1, lst, maxs, lazy ? 0 : 1);
    if (lazy) { h->last_args = a1; h->lazy_pending = (rc == SRBDQP_OK); h->lazy_rcount = rcount; h->lazy_slot = slot; }
    if (rc != SRBDQP_OK || lazy) return rc;
    for (int p = 1; p <= rcount; ++p) {
        bool last = true;
        const int every = a1.max_iter;
        const bool is_last = p >= rcount || every >= h->cfg.max_iter - p * every;
        rc = srbdqp_restart_pass(h, a1, lst, maxs, is_last && a.done_flag != nullptr, slot, p, rcount, &last);
        if (rc != SRBDQP_OK || last) break;
    }
    return rc;
}

// common body of the host-buffer entry points (esz = sizeof the caller's element type)
int solve_host_impl(srbdqp_handle* h, int32_t B, size_t esz, const void* x0, const void* x_ref, const void* foot,
                    const uint8_t* contact, const void* pcom, const void* warm_u, const void* warm_y, void* u_out,
                    void* x_out, void* y_out, int32_t* status, int32_t* iters) {
    if (B < 0 || (B > 0 && (!x0 || !x_ref || !foot || !contact || !u_out))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    {
        const int rq = aql_quiesce(h);
        if (rq != SRBDQP_OK) return rq;
    }
    const size_t N = (size_t)h->cfg.horizon, n = 12 * N, m = 20 * N, b = (size_t)B;
    Carver sz(nullptr);
    auto carve = [&](Carver& c, char*& dx0, char*& dxr, char*& dft, uint8_t*& dct, char*& dpc, char*& dwu,
                     char*& dwy, char*& du, char*& dx, char*& dy, int32_t*& dst, int32_t*& dit) {
        dx0 = c.take<char>(b * 13 * esz); dxr = c.take<char>(b * N * 13 * esz); dft = c.take<char>(b * N * 12 * esz);
        dct = c.take<uint8_t>(b * N * 4);
        dpc = pcom ? c.take<char>(b * N * 3 * esz) : nullptr;
        dwu = warm_u ? c.take<char>(b * n * esz) : nullptr;
        dwy = warm_y ? c.take<char>(b * m * esz) : nullptr;
        du = c.take<char>(b * n * esz);
        dx = x_out ? c.take<char>(b * (N + 1) * 13 * esz) : nullptr;
        dy = y_out ? c.take<char>(b * m * esz) : nullptr;
        dst = c.take<int32_t>(b); dit = c.take<int32_t>(b);
    };
    char *dx0, *dxr, *dft, *dpc, *dwu, *dwy, *du, *dx, *dy; uint8_t* dct; int32_t *dst, *dit;
    carve(sz, dx0, dxr, dft, dct, dpc, dwu, dwy, du, dx, dy, dst, dit);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv, dx0, dxr, dft, dct, dpc, dwu, dwy, du, dx, dy, dst, dit);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(dx0, x0, b * 13 * esz, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dxr, x_ref, b * N * 13 * esz, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dft, foot, b * N * 12 * esz, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dct, contact, b * N * 4, hipMemcpyHostToDevice, st));
    if (pcom) HIP_TRY(h, hipMemcpyAsync(dpc, pcom, b * N * 3 * esz, hipMemcpyHostToDevice, st));
    if (warm_u) HIP_TRY(h, hipMemcpyAsync(dwu, warm_u, b * n * esz, hipMemcpyHostToDevice, st));
    if (warm_y) HIP_TRY(h, hipMemcpyAsync(dwy, warm_y, b * m * esz, hipMemcpyHostToDevice, st));
    if (h->cfg.max_contacts_per_step <= 0) {   // pick the kernel instantiation from the batch's own contact flags
        int worst = 0;
        for (size_t q = 0; q < b * N && worst <= 2; ++q) {
            const uint8_t* c = contact + 4 * q;
            const int cnt = (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
            if (cnt > worst) worst = cnt;
        }
        h->maxs_override = (worst <= 2) ? 2 : 4;
    }
    const int32_t* hint_keep = h->sched_hint;               // the dispatch hint belongs to the device-buffer API
    h->sched_hint = nullptr;
    rc = solve_device_impl(h, B, dx0, dxr, dft, dct, dpc, dwu, dwy, du, dx, dy, dst, dit, st);
    h->sched_hint = hint_keep;
    h->maxs_override = 0;
    if (rc != SRBDQP_OK) return rc;
    if (h->cfg.flags & SRBDQP_FLAG_DEFER_TAIL) {            // a host-buffer call returns finished results: whatever was deferred runs now
        rc = srbdqp_flush(h, st);
        if (rc != SRBDQP_OK) return rc;
    }
    HIP_TRY(h, hipMemcpyAsync(u_out, du, b * n * esz, hipMemcpyDeviceToHost, st));
    if (x_out) HIP_TRY(h, hipMemcpyAsync(x_out, dx, b * (N + 1) * 13 * esz, hipMemcpyDeviceToHost, st));
    if (y_out) HIP_TRY(h, hipMemcpyAsync(y_out, dy, b * m * esz, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, dst, b * 4, hipMemcpyDeviceToHost, st));
    if (iters) HIP_TRY(h, hipMemcpyAsync(iters, dit, b * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return SRBDQP_OK;
}

}  // namespace

extern "C" {

int srbdqp_solve_batch_device_f64(srbdqp_handle* h, int32_t B, const double* x0, const double* x_ref,
                                  const double* foot, const uint8_t* contact, const double* pcom,
                                  const double* warm_u, const double* warm_y, double* u_out, double* x_out,
                                  double* y_out, int32_t* status, int32_t* iters, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    h->io_f32 = false;
    return solve_device_impl(h, B, x0, x_ref, foot, contact, pcom, warm_u, warm_y, u_out, x_out, y_out, status, iters, stream);
}

int srbdqp_solve_batch_device_f32(srbdqp_handle* h, int32_t B, const float* x0, const float* x_ref,
                                  const float* foot, const uint8_t* contact, const float* pcom,
                                  const float* warm_u, const float* warm_y, float* u_out, float* x_out,
                                  float* y_out, int32_t* status, int32_t* iters, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    h->io_f32 = true;
    const int rc = solve_device_impl(h, B, x0, x_ref, foot, contact, pcom, warm_u, warm_y, u_out, x_out, y_out, status, iters, stream);
    h->io_f32 = false;
    return rc;
}

int srbdqp_solve_batch_f64(srbdqp_handle* h, int32_t B, const double* x0, const double* x_ref, const double* foot,
                           const uint8_t* contact, const double* pcom, const double* warm_u, const double* warm_y,
                           double* u_out, double* x_out, double* y_out, int32_t* status, int32_t* iters) {
    if (!h) return SRBDQP_E_INVALID;
    h->io_f32 = false;
    return solve_host_impl(h, B, sizeof(double), x0, x_ref, foot, contact, pcom, warm_u, warm_y, u_out, x_out, y_out, status, iters);
}

int srbdqp_solve_batch_f32(srbdqp_handle* h, int32_t B, const float* x0, const float* x_ref, const float* foot,
                           const uint8_t* contact, const float* pcom, const float* warm_u, const float* warm_y,
                           float* u_out, float* x_out, float* y_out, int32_t* status, int32_t* iters) {
    if (!h) return SRBDQP_E_INVALID;
    h->io_f32 = true;
    const int rc = solve_host_impl(h, B, sizeof(float), x0, x_ref, foot, contact, pcom, warm_u, warm_y, u_out, x_out, y_out, status, iters);
    h->io_f32 = false;
    return rc;
}

int srbdqp_assemble_f64(srbdqp_handle* h, int32_t B, const double* x0, const double* x_ref, const double* foot,
                        const uint8_t* contact, const double* pcom, double* P_out, double* q_out, double* l_out,
                        double* ub_out) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x0 || !x_ref || !foot || !contact || !P_out || !q_out || !l_out || !ub_out))) { h->err = "null pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = (size_t)h->cfg.horizon, n = 12 * N, m = 20 * N, b = (size_t)B;
    int maxs = h->cfg.max_contacts_per_step > 0 ? h->cfg.max_contacts_per_step : 4;
    if (h->cfg.max_contacts_per_step <= 0) {   // as the host-buffer solve: the instantiation follows the batch's own contact flags
        int worst = 0;
        for (size_t q = 0; q < b * N && worst <= 2; ++q) {
            const uint8_t* c = contact + 4 * q;
            const int cnt = (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
            if (cnt > worst) worst = cnt;
        }
        maxs = (worst <= 2) ? 2 : 4;
    }
    h->io_f32 = false;
    if (uses_wrench(h, maxs, B)) { h->err = "this configuration solves on the general kernel: use srbdqp_assemble_wrench_f64"; return SRBDQP_E_INVALID; }
    auto carve = [&](Carver& c, double*& dx0, double*& dxr, double*& dft, uint8_t*& dct, double*& dpc, double*& dP,
                     double*& dq, double*& dl, double*& du) {
        dx0 = c.take<double>(b * 13); dxr = c.take<double>(b * N * 13); dft = c.take<double>(b * N * 12);
        dct = c.take<uint8_t>(b * N * 4);
        dpc = pcom ? c.take<double>(b * N * 3) : nullptr;
        dP = c.take<double>(b * n * n); dq = c.take<double>(b * n); dl = c.take<double>(b * m); du = c.take<double>(b * m);
    };
    double *dx0, *dxr, *dft, *dpc, *dP, *dq, *dl, *du; uint8_t* dct;
    Carver sz(nullptr);
    carve(sz, dx0, dxr, dft, dct, dpc, dP, dq, dl, du);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv, dx0, dxr, dft, dct, dpc, dP, dq, dl, du);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(dx0, x0, b * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dxr, x_ref, b * N * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dft, foot, b * N * 12 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dct, contact, b * N * 4, hipMemcpyHostToDevice, st));
    if (pcom) HIP_TRY(h, hipMemcpyAsync(dpc, pcom, b * N * 3 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemsetAsync(dP, 0, b * n * n * 8, st));
    HIP_TRY(h, hipMemsetAsync(dq, 0, b * n * 8, st));
    HIP_TRY(h, hipMemsetAsync(dl, 0, b * m * 8, st));
    HIP_TRY(h, hipMemsetAsync(du, 0, b * m * 8, st));
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_args(h->cfg, a);
    a.x0 = dx0; a.xref = dxr; a.foot = dft; a.contact = dct; a.pcom = dpc;
    a.P_out = dP; a.q_out = dq; a.l_out = dl; a.ub_out = du;
    a.B = B; a.mode = 1;
    rc = launch(h, a, st, maxs);                 // the kernel a solve of this batch would run, stopped before its factorisation
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipGetLastError());
    // compact K, q, map -> full-size P, q in the original variable order (on the host: bookkeeping, not the hot path)
    std::vector<double> K(n * n), qc(n), mp(m);
    const double rho = h->cfg.rho, aa = 4.0 * h->cfg.mu * h->cfg.mu + h->cfg.rho_fz_scale, sc = h->cfg.force_scale;
    for (size_t q = 0; q < b; ++q) {
        HIP_TRY(h, hipMemcpyAsync(K.data(), dP + q * n * n, n * n * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(qc.data(), dq + q * n, n * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(mp.data(), dl + q * m, m * 8, hipMemcpyDeviceToHost, st));
        double nad = 0.0;
        HIP_TRY(h, hipMemcpyAsync(&nad, du + q * m, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        double* P = P_out + q * n * n;
        double* qq = q_out + q * n;
        std::fill(P, P + n * n, 0.0);
        std::fill(qq, qq + n, 0.0);
        if (nad < 0.0) { h->err = "a QP of the batch violates max_contacts_per_step"; return SRBDQP_E_INVALID; }
        const int na = (int)nad, ne = 3 * na;
        if (na == 0) { mp[12 * N] = -srbdqp::kInf; mp[12 * N + 1] = 0.0; }   // flight phase: the kernel leaves before it dumps the friction-row bounds
        for (int r = 0; r < ne; ++r) {
            const int vr = 3 * (int)mp[r / 3] + r % 3;
            qq[vr] = qc[r];
            for (int c = 0; c < ne; ++c) {
                const int vc = 3 * (int)mp[c / 3] + c % 3;
                double v = K[(size_t)r * n + c];
                if (r == c) v -= h->cfg.sigma + rho * ((r % 3 < 2) ? 2.0 : aa);
                P[(size_t)vr * n + vc] = v;
            }
        }
        // a8: rows 20 k + 5 i + j of step k, contact i.  Stance contacts: the bounds the kernel dumped (what its ADMM loop uses);
        // swing contacts were eliminated by the presolve (their rows do not exist on the device): force clamped to 0
        for (size_t k = 0; k < N; ++k)
            for (int i = 0; i < 4; ++i) {
                double* lo = l_out + q * m + 20 * k + 5 * i;
                double* hi = ub_out + q * m + 20 * k + 5 * i;
                for (int j = 0; j < 4; ++j) { lo[j] = mp[12 * N]; hi[j] = mp[12 * N + 1]; }
                lo[4] = 0.0; hi[4] = 0.0;
            }
        for (int e = 0; e < na; ++e) {
            const int gc = (int)mp[e];
            l_out[q * m + 5 * gc + 4] = mp[4 * N + e];
            ub_out[q * m + 5 * gc + 4] = mp[8 * N + e];
        }
        (void)sc;
    }
    return SRBDQP_OK;
}

int srbdqp_assemble_wrench_f64(srbdqp_handle* h, int32_t B, const double* x0, const double* x_ref, const double* foot,
                               const uint8_t* contact, const double* pcom, double* T_out, double* q_out, double* blocks_out,
                               double* goff_out) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x0 || !x_ref || !foot || !contact || !T_out || !q_out || !blocks_out || !goff_out))) { h->err = "null pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t N = (size_t)h->cfg.horizon, n = 12 * N, ng = 6 * N, b = (size_t)B;
    auto carve = [&](Carver& c, double*& dx0, double*& dxr, double*& dft, uint8_t*& dct, double*& dpc, double*& dT,
                     double*& dq, double*& dbl, double*& dgo) {
        dx0 = c.take<double>(b * 13); dxr = c.take<double>(b * N * 13); dft = c.take<double>(b * N * 12);
        dct = c.take<uint8_t>(b * N * 4);
        dpc = pcom ? c.take<double>(b * N * 3) : nullptr;
        dT = c.take<double>(b * ng * ng); dq = c.take<double>(b * n); dbl = c.take<double>(b * n * 24); dgo = c.take<double>(b * (N + 1));
    };
    double *dx0, *dxr, *dft, *dpc, *dT, *dq, *dbl, *dgo; uint8_t* dct;
    Carver sz(nullptr);
    carve(sz, dx0, dxr, dft, dct, dpc, dT, dq, dbl, dgo);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv, dx0, dxr, dft, dct, dpc, dT, dq, dbl, dgo);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(dx0, x0, b * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dxr, x_ref, b * N * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dft, foot, b * N * 12 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dct, contact, b * N * 4, hipMemcpyHostToDevice, st));
    if (pcom) HIP_TRY(h, hipMemcpyAsync(dpc, pcom, b * N * 3 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemsetAsync(dT, 0, b * ng * ng * 8, st));
    HIP_TRY(h, hipMemsetAsync(dq, 0, b * n * 8, st));
    HIP_TRY(h, hipMemsetAsync(dbl, 0, b * n * 24 * 8, st));
    HIP_TRY(h, hipMemsetAsync(dgo, 0, b * (N + 1) * 8, st));
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    fill_args(h->cfg, a);
    a.x0 = dx0; a.xref = dxr; a.foot = dft; a.contact = dct; a.pcom = dpc;
    a.P_out = dT; a.q_out = dq; a.l_out = dbl; a.ub_out = dgo;
    a.B = B; a.mode = 1;
    h->io_f32 = false;
    switch (h->cfg.horizon) {
        case 4: rc = launch_wrench<4>(h, a, st); break;
        case 8: rc = launch_wrench<8>(h, a, st); break;
        case 10: rc = launch_wrench<10>(h, a, st); break;
        case 12: rc = launch_wrench<12>(h, a, st); break;
        case 16: rc = launch_wrench<16>(h, a, st); break;
        case 20: rc = launch_wrench<20>(h, a, st); break;
        case 24: rc = launch_wrench<24>(h, a, st); break;
        default: h->err = "unsupported horizon"; return SRBDQP_E_INVALID;
    }
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(T_out, dT, b * ng * ng * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(q_out, dq, b * n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(blocks_out, dbl, b * n * 24 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(goff_out, dgo, b * (N + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return SRBDQP_OK;
}

// ---- ragged batches (BASELINE.json configs[4]): mixed horizons, one launch per horizon bucket, all in flight together ----
struct srbdqp_ragged {
    std::vector<srbdqp_handle*> hs;          // one engine (own stream) per horizon bucket
    std::vector<int> horizons;
    std::vector<hipEvent_t> ev_out;
    std::vector<char> ev_out_used;            // ev_out[i] has been recorded by an earlier call
    hipEvent_t ev_in = nullptr;
    bool ev_in_pending = false;
    int device = 0;
    int32_t *d_perm = nullptr, *d_off = nullptr, *h_perm = nullptr, *h_off = nullptr;   // device arrays + pinned mirrors
    size_t cap = 0;
    // rho restart of the long-horizon buckets: fp32 maxima of a QP's last check [cap][4], status when the caller passes none
    // [cap], duals of the first pass [row_cap][20] (the second pass warm-starts from them)
    float* d_resid = nullptr; int32_t* d_status = nullptr; double* d_y = nullptr; size_t row_cap = 0;
    double* d_rho = nullptr;                  // [2][cap]: the rho a restart pass ran its QPs with, for the pass behind it
    // SRBDQP_FLAG_DEFER_TAIL: every array above exists kSets times (set = call number mod kSets) and the restart passes of a bucket run on the bucket's own tail
    // stream behind its first pass, beside the next calls; a set is reused only behind the events that close its last user's passes
    static constexpr int kSets = 3;
    bool defer = false;
    unsigned long long call_k = 0;
    std::vector<hipStream_t> tail_st;         // per bucket
    std::vector<hipEvent_t> ev_tail;          // [kSets][nb]
    std::vector<char> ev_tail_used;           // [kSets][nb]
    std::vector<hipEvent_t> last_tail;        // per bucket: closes the passes of the last call that had any (srbdqp_ragged_flush), or null
    char* ws = nullptr; size_t ws_bytes = 0;  // host-buffer entry point: device copies of the caller's arrays
    hipStream_t stream = nullptr;             // ... and the stream its copies run on
    std::string err;
};

namespace {
std::string g_ragged_err;
#define RAG_TRY(r, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (r)->err = std::string(#call) + ": " + hipGetErrorString(e_); return SRBDQP_E_HIP; } } while (0)

int ragged_launch_bucket(srbdqp_handle* bh, const KArgs& a, hipStream_t st, bool f32) {
    bh->io_f32 = f32;
    struct Reset { srbdqp_handle* h; ~Reset() { h->io_f32 = false; } } reset_on_return{bh};
    switch (bh->cfg.horizon) {
        case 4: return launch_wrench<4>(bh, a, st);
        case 8: return launch_wrench<8>(bh, a, st);
        case 10: return launch_wrench<10>(bh, a, st);
        case 12: return launch_wrench<12>(bh, a, st);
        case 16: return launch_wrench<16>(bh, a, st);
        case 20: return launch_wrench<20>(bh, a, st);
        case 24: return launch_wrench<24>(bh, a, st);
        default: bh->err = "unsupported horizon"; return SRBDQP_E_INVALID;
    }
}
}  // namespace

int srbdqp_ragged_create(const srbdqp_config* cfg, const int32_t* horizons, int32_t n_horizons, srbdqp_ragged** out) {
    if (!cfg || !horizons || !out || n_horizons < 1 || n_horizons > 16) { g_ragged_err = "bad argument"; return SRBDQP_E_INVALID; }
    *out = nullptr;
    srbdqp_ragged* r = new (std::nothrow) srbdqp_ragged();
    if (!r) { g_ragged_err = "out of host memory"; return SRBDQP_E_NOMEM; }
    r->device = cfg->device;
    auto fail = [&](int rc, const std::string& what) { g_ragged_err = what; srbdqp_ragged_destroy(r); return rc; };
    for (int i = 0; i < n_horizons; ++i) {
        for (int j = 0; j < i; ++j) if (horizons[j] == horizons[i]) return fail(SRBDQP_E_INVALID, "duplicate horizon");
        srbdqp_config c = *cfg;
        c.horizon = horizons[i];
        c.kernel = SRBDQP_KERNEL_WRENCH;       // per-QP contact schedules are free: the general kernel at every horizon
        srbdqp_handle* h = nullptr;
        const int rc = srbdqp_create(&c, &h);
        if (rc != SRBDQP_OK) return fail(rc, std::string("bucket engine: ") + srbdqp_last_error(nullptr));
        r->hs.push_back(h);
        r->horizons.push_back(horizons[i]);
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(SRBDQP_E_HIP, "hipEventCreate");
        r->ev_out.push_back(ev);
        r->ev_out_used.push_back(0);
    }
    if (hipEventCreateWithFlags(&r->ev_in, hipEventDisableTiming) != hipSuccess) return fail(SRBDQP_E_HIP, "hipEventCreate");
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return fail(SRBDQP_E_HIP, "hipStreamCreate");
    r->defer = (cfg->flags & SRBDQP_FLAG_DEFER_TAIL) != 0;
    if (r->defer) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return fail(SRBDQP_E_HIP, "hipDeviceGetStreamPriorityRange");
        for (int i = 0; i < n_horizons; ++i) {
            hipStream_t ts = nullptr;
            if (hipStreamCreateWithPriority(&ts, hipStreamNonBlocking, greatest) != hipSuccess) return fail(SRBDQP_E_HIP, "hipStreamCreate");
            r->tail_st.push_back(ts);
            r->last_tail.push_back(nullptr);
        }
        for (int i = 0; i < srbdqp_ragged::kSets * n_horizons; ++i) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(SRBDQP_E_HIP, "hipEventCreate");
            r->ev_tail.push_back(ev);
            r->ev_tail_used.push_back(0);
        }
    }
    *out = r;
    return SRBDQP_OK;
}

int srbdqp_ragged_destroy(srbdqp_ragged* r) {
    if (!r) return SRBDQP_OK;
    (void)hipSetDevice(r->device);
    for (auto* h : r->hs) srbdqp_destroy(h);
    for (auto ts : r->tail_st) if (ts) { (void)hipStreamSynchronize(ts); (void)hipStreamDestroy(ts); }
    for (auto ev : r->ev_tail) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : r->ev_out) if (ev) (void)hipEventDestroy(ev);
    if (r->ev_in) (void)hipEventDestroy(r->ev_in);
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    if (r->d_perm) (void)hipFree(r->d_perm);
    if (r->d_off) (void)hipFree(r->d_off);
    if (r->d_resid) (void)hipFree(r->d_resid);
    if (r->d_status) (void)hipFree(r->d_status);
    if (r->d_rho) (void)hipFree(r->d_rho);
    if (r->d_y) (void)hipFree(r->d_y);
    if (r->h_perm) (void)hipHostFree(r->h_perm);
    if (r->h_off) (void)hipHostFree(r->h_off);
    if (r->ws) (void)hipFree(r->ws);
    delete r;
    return SRBDQP_OK;
}

const char* srbdqp_ragged_last_error(const srbdqp_ragged* r) { return r ? r->err.c_str() : g_ragged_err.c_str(); }

}  // extern "C"

extern "C" int srbdqp_ragged_flush(srbdqp_ragged* r, void* stream);
namespace {
// common body of the ragged device entry points; esz = element size of the caller's buffers (8 or 4)
int ragged_device_impl(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const void* x0, const void* x_ref, const void* foot,
                       const uint8_t* contact, const void* warm_u, const void* warm_y, void* u_out, void* x_out, void* y_out,
                       int32_t* status, int32_t* iters, void* stream, bool f32) {
    if (!r) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!N_per_qp || !x0 || !x_ref || !foot || !contact || !u_out))) { r->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    RAG_TRY(r, hipSetDevice(r->device));
    hipStream_t sin = stream ? reinterpret_cast<hipStream_t>(stream) : r->stream;
    const size_t nb = r->hs.size();
    if ((size_t)B > r->cap) {   // (re)allocate the index arrays: the only point that waits, and only for earlier solves of this object
        for (auto* h : r->hs) RAG_TRY(r, hipStreamSynchronize(h->stream));
        for (auto ts : r->tail_st) RAG_TRY(r, hipStreamSynchronize(ts));
        RAG_TRY(r, hipStreamSynchronize(sin));
        std::fill(r->ev_tail_used.begin(), r->ev_tail_used.end(), 0);
        if (r->d_perm) (void)hipFree(r->d_perm);
        if (r->d_off) (void)hipFree(r->d_off);
        if (r->h_perm) (void)hipHostFree(r->h_perm);
        if (r->h_off) (void)hipHostFree(r->h_off);
        r->d_perm = r->d_off = r->h_perm = r->h_off = nullptr; r->cap = 0;
        const size_t want = (size_t)B + (size_t)B / 4 + 64;
        const size_t ns = r->defer ? (size_t)srbdqp_ragged::kSets : 1;
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_perm), ns * want * 4));
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_off), ns * want * 4));
        RAG_TRY(r, hipHostMalloc(reinterpret_cast<void**>(&r->h_perm), want * 4, hipHostMallocDefault));
        RAG_TRY(r, hipHostMalloc(reinterpret_cast<void**>(&r->h_off), want * 4, hipHostMallocDefault));
        if (r->d_resid) (void)hipFree(r->d_resid);
        if (r->d_status) (void)hipFree(r->d_status);
        if (r->d_rho) (void)hipFree(r->d_rho);
        r->d_resid = nullptr; r->d_status = nullptr; r->d_rho = nullptr;
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_resid), ns * want * 16));
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_status), ns * want * 4));
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_rho), ns * 2 * want * sizeof(double)));
        r->cap = want;
        r->ev_in_pending = false;
    }
    if (r->ev_in_pending) RAG_TRY(r, hipEventSynchronize(r->ev_in));   // the previous call's index upload has left the pinned mirrors
    // bucket permutation (counting sort by horizon) and the packed row offsets
    std::vector<int> cnt(nb, 0), start(nb + 1, 0), which((size_t)B);
    long long rows = 0;
    for (int32_t b = 0; b < B; ++b) {
        int k = -1;
        for (size_t i = 0; i < nb; ++i) if (r->horizons[i] == N_per_qp[b]) { k = (int)i; break; }
        if (k < 0) { r->err = "N_per_qp holds a horizon this object was not created for"; return SRBDQP_E_INVALID; }
        which[(size_t)b] = k; ++cnt[(size_t)k];
        if (rows > 2000000000LL) { r->err = "more than 2^31 horizon rows in one call"; return SRBDQP_E_INVALID; }
        r->h_off[b] = (int32_t)rows;
        rows += N_per_qp[b];
    }
    for (size_t i = 0; i < nb; ++i) start[i + 1] = start[i] + cnt[i];
    bool any_restart = false;
    for (size_t i = 0; i < nb; ++i) any_restart = any_restart || (cnt[i] > 0 && restart_iter_of(r->hs[i], 4, cnt[i]) > 0);
    if (any_restart && (size_t)rows > r->row_cap) {   // dual buffer of the restart (waits for earlier solves of this object only)
        for (auto* h : r->hs) RAG_TRY(r, hipStreamSynchronize(h->stream));
        for (auto ts : r->tail_st) RAG_TRY(r, hipStreamSynchronize(ts));
        if (r->d_y) (void)hipFree(r->d_y);
        r->d_y = nullptr; r->row_cap = 0;
        const size_t want = (size_t)rows + (size_t)rows / 4 + 64;
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->d_y), (r->defer ? (size_t)srbdqp_ragged::kSets : 1) * want * 20 * sizeof(double)));
        r->row_cap = want;
    }
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int32_t b = 0; b < B; ++b) r->h_perm[fill[(size_t)which[(size_t)b]]++] = b;
    // the index arrays (and the restart buffers) are shared by the calls of this object: the upload below must not overtake the
    // bucket kernels of an earlier call made on ANOTHER caller stream (calls on one stream are ordered through ev_out already)
    for (size_t i = 0; i < nb; ++i) if (r->ev_out_used[i]) RAG_TRY(r, hipStreamWaitEvent(sin, r->ev_out[i], 0));
    // this call's set of the shared arrays (deferred restart passes: three in rotation; behind the passes of the set's last user)
    const size_t set = r->defer ? (size_t)(r->call_k++ % srbdqp_ragged::kSets) : 0;
    if (r->defer)
        for (size_t i = 0; i < nb; ++i)
            if (r->ev_tail_used[set * nb + i]) RAG_TRY(r, hipStreamWaitEvent(sin, r->ev_tail[set * nb + i], 0));
    int32_t* const d_off = r->d_off + set * r->cap;
    int32_t* const d_perm = r->d_perm + set * r->cap;
    float* const d_resid = r->d_resid + set * r->cap * 4;
    int32_t* const d_status = r->d_status + set * r->cap;
    double* const d_rho = r->d_rho + set * 2 * r->cap;
    double* const d_y = r->d_y ? r->d_y + set * r->row_cap * 20 : nullptr;
    RAG_TRY(r, hipMemcpyAsync(d_off, r->h_off, (size_t)B * 4, hipMemcpyHostToDevice, sin));
    RAG_TRY(r, hipMemcpyAsync(d_perm, r->h_perm, (size_t)B * 4, hipMemcpyHostToDevice, sin));
    RAG_TRY(r, hipEventRecord(r->ev_in, sin));
    r->ev_in_pending = true;
    // one launch per non-empty bucket, each on its engine's own stream behind the upload; the caller's stream then waits for all
    for (size_t i = 0; i < nb; ++i) {
        if (cnt[i] == 0) continue;
        srbdqp_handle* bh = r->hs[i];
        hipStream_t bs = bh->stream;
        RAG_TRY(r, hipStreamWaitEvent(bs, r->ev_in, 0));
        KArgs a;
        std::memset(&a, 0, sizeof(a));
        fill_args(bh->cfg, a);
        a.x0 = static_cast<const double*>(x0); a.xref = static_cast<const double*>(x_ref); a.foot = static_cast<const double*>(foot); a.contact = contact;
        a.warm_u = static_cast<const double*>(warm_u); a.warm_y = static_cast<const double*>(warm_y);
        a.u_out = static_cast<double*>(u_out); a.x_out = static_cast<double*>(x_out); a.y_out = static_cast<double*>(y_out);
        a.status = status; a.iters = iters;
        a.perm = d_perm + start[i]; a.row_off = d_off;
        a.B = cnt[i]; a.mode = 0;
        int rcount = 1;
        const int restart = restart_iter_of(bh, 4, cnt[i], false, &rcount);
        int rc;
        if (restart > 0) {   // several passes over the bucket, as srbdqp_solve_batch_* does (the later ones select their QPs in-kernel)
            KArgs a1 = a;
            a1.max_iter = restart; a1.resid_out = d_resid;
            if (!a1.y_out) { a1.y_out = d_y; a1.y_capped_only = 1; }   // (a later pass warm-starts from the duals of the pass before it)
            if (!a1.status) a1.status = d_status;
            rc = ragged_launch_bucket(bh, a1, bs, f32);
            hipStream_t ps = bs;                                          // the stream the restart passes run on
            if (r->defer && rc == SRBDQP_OK) {
                // ... the bucket's tail stream, behind its first pass: the caller's stream waits for the first pass only, the passes run beside what it enqueues next
                RAG_TRY(r, hipEventRecord(r->ev_out[i], bs));
                r->ev_out_used[i] = 1;
                RAG_TRY(r, hipStreamWaitEvent(sin, r->ev_out[i], 0));
                ps = r->tail_st[i];
                RAG_TRY(r, hipStreamWaitEvent(ps, r->ev_out[i], 0));
            }
            for (int p = 1; rc == SRBDQP_OK && p <= rcount; ++p) {        // (srbdqp_restart_pass, on the ragged object's own buffers)
                const int done = p * restart, left = bh->cfg.max_iter - done;
                const bool last = p >= rcount || restart >= left;
                KArgs a2 = a1;
                a2.resid_in = d_resid; a2.resid_out = last ? nullptr : d_resid;
                a2.warm_u = a1.u_out; a2.warm_y = a1.y_out;
                a2.max_iter = last ? left : restart; a2.iters_base = done;
                a2.rho_qp = (p == 1) ? nullptr : d_rho + (size_t)(p % 2) * r->cap;
                a2.rho_out = last ? nullptr : d_rho + (size_t)((p + 1) % 2) * r->cap;
                rc = ragged_launch_bucket(bh, a2, ps, f32);
                if (last) break;
            }
            if (r->defer && rc == SRBDQP_OK) {
                RAG_TRY(r, hipEventRecord(r->ev_tail[set * nb + i], ps));
                r->ev_tail_used[set * nb + i] = 1;
                r->last_tail[i] = r->ev_tail[set * nb + i];
                continue;                                                 // (ev_out was recorded behind the first pass)
            }
        } else {
            rc = ragged_launch_bucket(bh, a, bs, f32);
        }
        if (rc != SRBDQP_OK) { r->err = std::string("bucket N=") + std::to_string(r->horizons[i]) + ": " + bh->err; return rc; }
        RAG_TRY(r, hipEventRecord(r->ev_out[i], bs));
        r->ev_out_used[i] = 1;
        RAG_TRY(r, hipStreamWaitEvent(sin, r->ev_out[i], 0));
    }
    return SRBDQP_OK;
}

// common body of the ragged host entry points
int ragged_host_impl(srbdqp_ragged* r, int32_t B, size_t esz, const int32_t* N_per_qp, const void* x0, const void* x_ref, const void* foot,
                     const uint8_t* contact, void* u_out, void* x_out, int32_t* status, int32_t* iters) {
    if (!r) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!N_per_qp || !x0 || !x_ref || !foot || !contact || !u_out))) { r->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    RAG_TRY(r, hipSetDevice(r->device));
    size_t rows = 0;
    for (int32_t b = 0; b < B; ++b) { if (N_per_qp[b] < 1 || N_per_qp[b] > SRBDQP_MAX_HORIZON) { r->err = "bad horizon in N_per_qp"; return SRBDQP_E_INVALID; } rows += (size_t)N_per_qp[b]; }
    const size_t b = (size_t)B;
    Carver sz(nullptr);
    char *dx0, *dxr, *dft, *du, *dx; uint8_t* dct; int32_t *dst, *dit;
    auto carve = [&](Carver& c) {
        dx0 = c.take<char>(b * 13 * esz); dxr = c.take<char>(rows * 13 * esz); dft = c.take<char>(rows * 12 * esz); dct = c.take<uint8_t>(rows * 4);
        du = c.take<char>(rows * 12 * esz); dx = x_out ? c.take<char>((rows + b) * 13 * esz) : nullptr;
        dst = c.take<int32_t>(b); dit = c.take<int32_t>(b);
    };
    carve(sz);
    if (sz.off > r->ws_bytes) {
        RAG_TRY(r, hipStreamSynchronize(r->stream));
        if (r->ws) { (void)hipFree(r->ws); r->ws = nullptr; r->ws_bytes = 0; }
        RAG_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->ws), sz.off + sz.off / 4));
        r->ws_bytes = sz.off + sz.off / 4;
    }
    Carver cv(r->ws);
    carve(cv);
    hipStream_t st = r->stream;
    RAG_TRY(r, hipMemcpyAsync(dx0, x0, b * 13 * esz, hipMemcpyHostToDevice, st));
    RAG_TRY(r, hipMemcpyAsync(dxr, x_ref, rows * 13 * esz, hipMemcpyHostToDevice, st));
    RAG_TRY(r, hipMemcpyAsync(dft, foot, rows * 12 * esz, hipMemcpyHostToDevice, st));
    RAG_TRY(r, hipMemcpyAsync(dct, contact, rows * 4, hipMemcpyHostToDevice, st));
    int rc = ragged_device_impl(r, B, N_per_qp, dx0, dxr, dft, dct, nullptr, nullptr, du, dx, nullptr, dst, dit, st, esz == 4);
    if (rc != SRBDQP_OK) return rc;
    rc = srbdqp_ragged_flush(r, st);                     // (deferred restart passes: the copies below need every QP finished)
    if (rc != SRBDQP_OK) return rc;
    RAG_TRY(r, hipMemcpyAsync(u_out, du, rows * 12 * esz, hipMemcpyDeviceToHost, st));
    if (x_out) RAG_TRY(r, hipMemcpyAsync(x_out, dx, (rows + b) * 13 * esz, hipMemcpyDeviceToHost, st));
    if (status) RAG_TRY(r, hipMemcpyAsync(status, dst, b * 4, hipMemcpyDeviceToHost, st));
    if (iters) RAG_TRY(r, hipMemcpyAsync(iters, dit, b * 4, hipMemcpyDeviceToHost, st));
    RAG_TRY(r, hipStreamSynchronize(st));
    return SRBDQP_OK;
}
}  // namespace

extern "C" {

int srbdqp_ragged_flush(srbdqp_ragged* r, void* stream) {
    if (!r) return SRBDQP_E_INVALID;
    RAG_TRY(r, hipSetDevice(r->device));
    hipStream_t sin = stream ? reinterpret_cast<hipStream_t>(stream) : r->stream;
    // (the events stay: they are re-recorded by every call, waiting for a completed one costs nothing, and a caller that issued calls on two streams
    //  flushes each of them -- a flush of the other stream must still find the passes that write ITS outputs)
    for (auto& ev : r->last_tail)
        if (ev) RAG_TRY(r, hipStreamWaitEvent(sin, ev, 0));
    return SRBDQP_OK;
}

int srbdqp_solve_ragged_device_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const double* x0, const double* x_ref,
                                   const double* foot, const uint8_t* contact, double* u_out, double* x_out, int32_t* status,
                                   int32_t* iters, void* stream) {
    return ragged_device_impl(r, B, N_per_qp, x0, x_ref, foot, contact, nullptr, nullptr, u_out, x_out, nullptr, status, iters, stream, false);
}

int srbdqp_solve_ragged_device_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const float* x0, const float* x_ref,
                                   const float* foot, const uint8_t* contact, float* u_out, float* x_out, int32_t* status,
                                   int32_t* iters, void* stream) {
    return ragged_device_impl(r, B, N_per_qp, x0, x_ref, foot, contact, nullptr, nullptr, u_out, x_out, nullptr, status, iters, stream, true);
}

int srbdqp_solve_ragged_warm_device_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const double* x0, const double* x_ref,
                                        const double* foot, const uint8_t* contact, const double* warm_u, const double* warm_y,
                                        double* u_out, double* x_out, double* y_out, int32_t* status, int32_t* iters, void* stream) {
    return ragged_device_impl(r, B, N_per_qp, x0, x_ref, foot, contact, warm_u, warm_y, u_out, x_out, y_out, status, iters, stream, false);
}

int srbdqp_solve_ragged_warm_device_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const float* x0, const float* x_ref,
                                        const float* foot, const uint8_t* contact, const float* warm_u, const float* warm_y,
                                        float* u_out, float* x_out, float* y_out, int32_t* status, int32_t* iters, void* stream) {
    return ragged_device_impl(r, B, N_per_qp, x0, x_ref, foot, contact, warm_u, warm_y, u_out, x_out, y_out, status, iters, stream, true);
}

int srbdqp_solve_ragged_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const double* x0, const double* x_ref,
                            const double* foot, const uint8_t* contact, double* u_out, double* x_out, int32_t* status, int32_t* iters) {
    return ragged_host_impl(r, B, sizeof(double), N_per_qp, x0, x_ref, foot, contact, u_out, x_out, status, iters);
}

int srbdqp_solve_ragged_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp, const float* x0, const float* x_ref,
                            const float* foot, const uint8_t* contact, float* u_out, float* x_out, int32_t* status, int32_t* iters) {
    return ragged_host_impl(r, B, sizeof(float), N_per_qp, x0, x_ref, foot, contact, u_out, x_out, status, iters);
}

// ---- the steps either side of the QP (include/srbdqp_cascade.h) --------------------------------------------------
namespace {
inline unsigned elementwise_grid(long long B) {
    // grid-stride kernels: enough 256-thread workgroups to fill 256 CUs several times over, no more
    const long long want = (B + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
}
}  // namespace

int srbdqp_swing_device_f64(srbdqp_handle* h, int64_t B, const double* p_start, const double* p_final,
                            const double* z_middle, const double* progress, double final_velocity_z,
                            double first_half_share, double* pos, double* vel_z, double* acc_z, double* coeff,
                            void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!p_start || !p_final || !z_middle || !progress || !pos))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    srbdqp::SwingArgs a{p_start, p_final, z_middle, progress, pos, vel_z, acc_z, coeff, final_velocity_z, first_half_share, (long long)B};
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
    hipLaunchKernelGGL(srbdqp::srbdqp_swing_kernel, dim3(elementwise_grid(B)), dim3(256), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    h->kname = "swing_f64";
    return SRBDQP_OK;
}

int srbdqp_swing_f64(srbdqp_handle* h, int64_t B, const double* p_start, const double* p_final, const double* z_middle,
                     const double* progress, double final_velocity_z, double first_half_share, double* pos,
                     double* vel_z, double* acc_z, double* coeff) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!p_start || !p_final || !z_middle || !progress || !pos))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t b = (size_t)B;
    double *ds, *df, *dm, *dt, *dp, *dv, *da, *dc;
    auto carve = [&](Carver& c) {
        ds = c.take<double>(b * 3); df = c.take<double>(b * 3); dm = c.take<double>(b); dt = c.take<double>(b);
        dp = c.take<double>(b * 3);
        dv = vel_z ? c.take<double>(b) : nullptr; da = acc_z ? c.take<double>(b) : nullptr;
        dc = coeff ? c.take<double>(b * 7) : nullptr;
    };
    Carver sz(nullptr);
    carve(sz);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(ds, p_start, b * 24, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(df, p_final, b * 24, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dm, z_middle, b * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dt, progress, b * 8, hipMemcpyHostToDevice, st));
    rc = srbdqp_swing_device_f64(h, B, ds, df, dm, dt, final_velocity_z, first_half_share, dp, dv, da, dc, st);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(pos, dp, b * 24, hipMemcpyDeviceToHost, st));
    if (vel_z) HIP_TRY(h, hipMemcpyAsync(vel_z, dv, b * 8, hipMemcpyDeviceToHost, st));
    if (acc_z) HIP_TRY(h, hipMemcpyAsync(acc_z, da, b * 8, hipMemcpyDeviceToHost, st));
    if (coeff) HIP_TRY(h, hipMemcpyAsync(coeff, dc, b * 56, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return SRBDQP_OK;
}

int srbdqp_wbid_reference_device_f64(srbdqp_handle* h, int64_t B, const double* x_next, const double* u0,
                                     const double* foot, int32_t as_written, double* R, double* base_vel,
                                     double* base_acc, double* com_acc, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x_next || !u0 || !foot || !R || !base_vel || !base_acc || !com_acc))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    srbdqp::WbidRefArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x_next = x_next; a.u0 = u0; a.foot = foot; a.R = R; a.base_vel = base_vel; a.base_acc = base_acc; a.com_acc = com_acc;
    for (int i = 0; i < 3; ++i) a.iinv[i] = 1.0 / h->cfg.inertia[i];
    a.mass = h->cfg.mass;
    a.gravity = -9.80665;   // wbid.py:286
    a.as_written = as_written ? 1 : 0;
    a.B = (long long)B;
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
    hipLaunchKernelGGL(srbdqp::srbdqp_wbid_reference_kernel, dim3(elementwise_grid(B)), dim3(256), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    h->kname = "wbid_reference_f64";
    return SRBDQP_OK;
}

int srbdqp_wbid_reference_f64(srbdqp_handle* h, int64_t B, const double* x_next, const double* u0, const double* foot,
                              int32_t as_written, double* R, double* base_vel, double* base_acc, double* com_acc) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x_next || !u0 || !foot || !R || !base_vel || !base_acc || !com_acc))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t b = (size_t)B;
    double *dx, *du, *df, *dR, *dv, *da, *dc;
    auto carve = [&](Carver& c) {
        dx = c.take<double>(b * 13); du = c.take<double>(b * 12); df = c.take<double>(b * 12);
        dR = c.take<double>(b * 9); dv = c.take<double>(b * 6); da = c.take<double>(b * 6); dc = c.take<double>(b * 3);
    };
    Carver sz(nullptr);
    carve(sz);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(dx, x_next, b * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(du, u0, b * 12 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(df, foot, b * 12 * 8, hipMemcpyHostToDevice, st));
    rc = srbdqp_wbid_reference_device_f64(h, B, dx, du, df, as_written, dR, dv, da, dc, st);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(R, dR, b * 9 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(base_vel, dv, b * 6 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(base_acc, da, b * 6 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(com_acc, dc, b * 3 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return SRBDQP_OK;
}

int srbdqp_mpc_inputs_device_f64(srbdqp_handle* h, int64_t B, const double* x0, const double* feet, const double* stamp,
                                 const double* v_ref, const uint8_t* standing, const srbdqp_gait* gait,
                                 double* x_ref, double* foot, uint8_t* contact, double* pcom, double* landing, void* stream) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x0 || !feet || !stamp || !v_ref || !x_ref || !foot || !contact || !pcom))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (!gait || gait->struct_size != (int32_t)sizeof(srbdqp_gait) || gait->period_steps < 1 || gait->double_support_steps < 0 ||
        gait->double_support_steps > gait->period_steps) { h->err = "invalid srbdqp_gait (struct_size, 1 <= period_steps, 0 <= double_support_steps <= period_steps)"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    srbdqp::MpcInputsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x0 = x0; a.feet = feet; a.stamp = stamp; a.v_ref = v_ref; a.standing = standing;
    a.x_ref = x_ref; a.foot = foot; a.contact = contact; a.pcom = pcom; a.landing = landing;
    for (int i = 0; i < 3; ++i) a.com_target[i] = gait->com_target[i];
    a.hip_offset_y = gait->hip_offset_y; a.dt = h->cfg.dt;
    a.N = h->cfg.horizon; a.period = gait->period_steps; a.ds = gait->double_support_steps;
    a.B = (long long)B;
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
    const long long tiles = ((long long)B + 31) / 32;                   // one tile of 32 robots per workgroup pass
    const dim3 grid((unsigned)(tiles < 1 ? 1 : (tiles > 256 * 32 ? 256 * 32 : tiles)));
    switch (h->cfg.horizon) {
        case 4: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<4>, grid, dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<8>, grid, dim3(256), 0, st, a); break;
        case 10: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<10>, grid, dim3(256), 0, st, a); break;
        case 12: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<12>, grid, dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<16>, grid, dim3(256), 0, st, a); break;
        case 20: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<20>, grid, dim3(256), 0, st, a); break;
        case 24: hipLaunchKernelGGL(srbdqp::srbdqp_mpc_inputs_kernel<24>, grid, dim3(256), 0, st, a); break;
        default: h->err = "unsupported horizon"; return SRBDQP_E_INVALID;
    }
    HIP_TRY(h, hipGetLastError());
    h->kname = "mpc_inputs_f64";
    return SRBDQP_OK;
}

int srbdqp_mpc_inputs_f64(srbdqp_handle* h, int64_t B, const double* x0, const double* feet, const double* stamp,
                          const double* v_ref, const uint8_t* standing, const srbdqp_gait* gait,
                          double* x_ref, double* foot, uint8_t* contact, double* pcom, double* landing) {
    if (!h) return SRBDQP_E_INVALID;
    if (B < 0 || (B > 0 && (!x0 || !feet || !stamp || !v_ref || !x_ref || !foot || !contact || !pcom))) { h->err = "null input/output pointer"; return SRBDQP_E_INVALID; }
    if (B == 0) return SRBDQP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t b = (size_t)B, N = (size_t)h->cfg.horizon;
    double *dx0, *dfe, *dst, *dv, *dxr, *dft, *dpc, *dlp;
    uint8_t *dsd, *dct;
    auto carve = [&](Carver& c) {
        dx0 = c.take<double>(b * 13); dfe = c.take<double>(b * 12); dst = c.take<double>(b); dv = c.take<double>(b * 2);
        dsd = standing ? c.take<uint8_t>(b) : nullptr;
        dxr = c.take<double>(b * N * 13); dft = c.take<double>(b * N * 12); dct = c.take<uint8_t>(b * N * 4);
        dpc = c.take<double>(b * N * 3); dlp = landing ? c.take<double>(b * 3) : nullptr;
    };
    Carver sz(nullptr);
    carve(sz);
    int rc = ensure_ws(h, sz.off);
    if (rc != SRBDQP_OK) return rc;
    Carver cv(h->ws);
    carve(cv);
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemcpyAsync(dx0, x0, b * 13 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dfe, feet, b * 12 * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dst, stamp, b * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dv, v_ref, b * 16, hipMemcpyHostToDevice, st));
    if (standing) HIP_TRY(h, hipMemcpyAsync(dsd, standing, b, hipMemcpyHostToDevice, st));
    rc = srbdqp_mpc_inputs_device_f64(h, B, dx0, dfe, dst, dv, dsd, gait, dxr, dft, dct, dpc, dlp, st);
    if (rc != SRBDQP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(x_ref, dxr, b * N * 13 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(foot, dft, b * N * 12 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(contact, dct, b * N * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(pcom, dpc, b * N * 3 * 8, hipMemcpyDeviceToHost, st));
    if (landing) HIP_TRY(h, hipMemcpyAsync(landing, dlp, b * 3 * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return SRBDQP_OK;
}

}  // extern "C"

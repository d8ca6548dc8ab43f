// srbdqp_resident.hpp -- the batch-1 control loop without a kernel launch per step (SRBDQP_FLAG_RESIDENT).
//
// Why: a staged batch-1 solve is 44 us of kernel and ~20 us of launch, dispatch and completion around it (a flight-phase
// QP, whose kernel only rolls the state out, takes 20 us through the same call -- tools/floor_probe.py).  Here ONE
// workgroup of the 4-wave kernel stays on the device.  The host rings a doorbell word in GPU-mapped host memory; wave
// 0 polls it over PCIe, the workgroup solves the QP it finds in the staging arrays with compact_qp() -- the very source
// of srbdqp_compact_kernel: same algorithm, same iteration counts -- and publishes the completion word the host spins
// on.  No HIP call is on the per-step path.
//
// Every wave reaches an exit: the kernel leaves (a) when the host writes kResidentQuit (srbdqp_destroy,
// srbdqp_resident_stop, before any hipFree of the handle), (b) after idle_ticks of the 100 MHz clock without a request
// (default 100 ms: a crashed or finished host process never leaves a spinning workgroup behind).  (b) races with a
// request: the kernel first publishes state = exited, then looks at the doorbell once more and still serves what it
// finds; the host, seeing "exited" without its completion word, joins the stream and starts a new kernel.
#pragma once
#include "srbdqp_compact.hpp"

namespace srbdqp {

struct ResidentCmd {          // 256-byte block in the staging slab (host-mapped, fine-grained)
    int32_t doorbell;         // host -> kernel: kResidentQuit, or (sequence << 6) | request flags; a new value = a new request
    int32_t state;            // kernel -> host: 1 = polling / solving, 2 = exited (or about to, serving at most one more request)
    int32_t done;             // kernel -> host: the doorbell value of the last finished request
    int32_t pad0;
    long long t_seen, t_done; // kernel -> host: 100 MHz clock when the last request was seen / its outputs were stored
    long long c_seen, c_done; //                 the same two moments on the shader clock (s_memtime)
    int32_t pad[52];
};
constexpr int32_t kResidentQuit = -1;
// request flags (low 6 bits of the doorbell)
constexpr int32_t kResPcom = 1, kResWarm = 2, kResWantX = 4, kResWantY = 8, kResMaxs4 = 16;

// table[f]: the arguments of a batch-1 solve over the staging arrays for the request flags f = doorbell & 15 (optional
// pointers a request does not use are null; done_flag null -- completion is published here), in device memory: a
// per-request copy of the 700-byte struct edited in registers ends up in scratch.  One instantiation serves one MAXS (the two bodies in one
// kernel cost 736 bytes of scratch): a request for the other one makes the kernel leave without serving it, exactly
// like a time-out, and the host starts the matching instantiation with the request pending.  One workgroup alone on
// its CU: no occupancy to protect, so the full register file (with the launched kernel's budget of 128 the loop around
// compact_qp spills 300-600 bytes).  compact_qp is compiled a second time here; fused multiply-adds are formed
// slightly differently, so results agree with the launched kernel to rounding (~1e-9 N), not bit for bit.
template <int N, int MAXS>
__global__ __launch_bounds__(kThreads, 1) void srbdqp_resident_kernel(const KArgs* table, ResidentCmd* cmd, int32_t last_word, long long idle_ticks) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_word, s_act;
    const int t = threadIdx.x;
    if (t == 0) __hip_atomic_store(&cmd->state, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    int last = last_word;
    for (;;) {
        if (t == 0) {
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            int word, act;
            for (;;) {
                word = __hip_atomic_load(&cmd->doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (word == kResidentQuit) { act = 0; break; }
                if (word != last) { act = (((word & kResMaxs4) != 0) == (MAXS == 4)) ? 1 : 0; break; }
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) {
                    __hip_atomic_store(&cmd->state, 2, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                    word = __hip_atomic_load(&cmd->doorbell, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                    act = (word != last && word != kResidentQuit && ((word & kResMaxs4) != 0) == (MAXS == 4)) ? 2 : 0;   // a request that raced the time-out is served
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            s_word = word; s_act = act;
        }
        __syncthreads();
        const int act = s_act, word = s_word;
        if (act == 0) break;
        const long long t_seen = (long long)__builtin_amdgcn_s_memrealtime(), c_seen = (long long)__builtin_amdgcn_s_memtime();
        // the staging arrays were rewritten by the host since the last request: drop every cached copy
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        __builtin_amdgcn_s_dcache_inv();
        compact_qp<N, MAXS, false>(table[word & 15], 0, sm);
        __threadfence_system();                                // outputs before the completion word
        __syncthreads();
        if (t == 0) {
            cmd->t_seen = t_seen; cmd->c_seen = c_seen;
            cmd->c_done = (long long)__builtin_amdgcn_s_memtime();
            cmd->t_done = (long long)__builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // the two time stamps; the outputs were fenced above
            __hip_atomic_store(&cmd->done, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        last = word;
        if (act == 2) break;
    }
    if (t == 0) __hip_atomic_store(&cmd->state, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace srbdqp

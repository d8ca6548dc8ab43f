// srbdqp_admm.hpp -- cross-lane helpers of the contact-local ADMM iterations (a9): DPP swaps, ds_bpermute moves, wave
// maxima.  Everything in an ADMM iteration except x~ = K^-1 rhs is local to ONE contact point (3 variables, 5 constraint
// rows), so the kernels keep a contact on a few neighbouring lanes and couple them with these.
#pragma once
#include "srbdqp_common.hpp"

namespace srbdqp {

#ifdef SRBDQP_PROFILE_ADMM
#define ADMM_T(var) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = (long long)t_; } while (0)
#else
#define ADMM_T(var) do { } while (0)
#endif

__device__ __forceinline__ double dpp_swap1(double v) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dpp_swap2(double v) {   // value of lane ^ 2 (quad_perm [2,3,0,1])
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// value of the next / the previous lane of the WAVE (GFX9's whole-wave DPP shifts, which gfx950 still has: they cross the 16-lane rows; lane 63 / lane 0 read 0).
// tools/wave_shift_probe.hip: 30 cycles for a dependent shift against 74 for a dependent ds_bpermute on an idle CU -- and no LDS pipeline to queue for.  The
// three lanes of a contact are neighbours, so the f_z gather and the friction rows' sums of every ADMM iteration are one or two of these instead of a
// ds_bpermute round trip each.
__device__ __forceinline__ int wave_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }    // wave_shl:1
__device__ __forceinline__ int wave_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }    // wave_shr:1
__device__ __forceinline__ float wave_next(float v) { return __int_as_float(wave_next(__float_as_int(v))); }
__device__ __forceinline__ float wave_prev(float v) { return __int_as_float(wave_prev(__float_as_int(v))); }
__device__ __forceinline__ double wave_next(double v) { return __hiloint2double(wave_next(__double2hiint(v)), wave_next(__double2loint(v))); }
__device__ __forceinline__ double wave_prev(double v) { return __hiloint2double(wave_prev(__double2hiint(v)), wave_prev(__double2loint(v))); }
// f_z of the lane's contact (lanes base + {0, 1, 2} = fx, fy, fz; ax = lane - base)
template <typename T> __device__ __forceinline__ T contact_fz(T v, int ax) { const T n1 = wave_next(v), n2 = wave_next(n1); return (ax == 0) ? n2 : ((ax == 1) ? n1 : v); }
// (value of the fx lane) + (value of the fy lane), for the fz lane of the contact (other lanes: not used)
template <typename T> __device__ __forceinline__ T contact_sum_xy(T v) { const T p1 = wave_prev(v), p2 = wave_prev(p1); return p2 + p1; }

// The four 16-lane rows of a register, each broadcast to all four rows: ch[c] of lane l = v of lane 16 c + (l & 15).  gfx950's row swaps
// (v_permlane32_swap: rows 2, 3 of the first operand <-> rows 0, 1 of the second; v_permlane16_swap: odd rows of the first <-> even rows of the second) on
// copies of the value: [R0 R1 R0 R1], [R2 R3 R2 R3], then [R0 x 4], [R1 x 4], [R2 x 4], [R3 x 4] -- three swaps and no LDS operation per dword, against
// four ds_bpermute (tools/permlane_probe.hip: 96 against 112 cycles for a dependent round on an idle CU; the LDS pipeline of a CU with 8 resident QPs is the busier one).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void row_chunks(int v, int (&ch)[4]) {
    const v2u_t ab = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    const v2u_t p = __builtin_amdgcn_permlane16_swap(ab[0], ab[0], false, false);
    const v2u_t q = __builtin_amdgcn_permlane16_swap(ab[1], ab[1], false, false);
    ch[0] = (int)p[0]; ch[1] = (int)p[1]; ch[2] = (int)q[0]; ch[3] = (int)q[1];
}
__device__ __forceinline__ void row_chunks(double v, double (&ch)[4]) {
    int lo[4], hi[4];
    row_chunks(__double2loint(v), lo);
    row_chunks(__double2hiint(v), hi);
#pragma unroll
    for (int c = 0; c < 4; ++c) ch[c] = __hiloint2double(hi[c], lo[c]);
}

__device__ __forceinline__ double bperm_f64(double v, int src_lane) {   // value of lane src_lane
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// acc += k * (value of `chunk` in lane J of this lane's 16-lane row): a DP-ALU DPP operand (gfx90a+: 64-bit VOP2 instructions take
// row_newbcast), so a wave-wide broadcast mat-vec needs no LDS read per term.  Measured against the LDS form (one ds_write_b64 +
// broadcast ds_read_b128s + v_fmac_f64) on a 64 x 64 row-per-lane product, tools/dpp_probe.hip: 545 against 781 cycles per
// product for one wave alone, 666 against 1082 with 8 waves per CU -- the broadcast reads of 8 resident QPs keep the CU's LDS
// pipeline busier than its vector ALUs.
// The terms of one 16-value chunk are ONE asm block: the chunk register is read through DPP by every term, a VGPR written by a vector
// instruction must not be read through DPP by the next one (two wait states), and the hazard recogniser cannot see into inline asm.  As
// separate asm statements only the first term carried the s_nop, and nothing kept the register allocator from copying or
// rematerialising `chunk` between two of the others (a v_mov feeding straight into a DPP read: the product silently uses stale data --
// unis() in srbdqp_common.hpp documents the same hazard moving forces by 1e-2 N).  Inside one block the operands are pinned for all
// terms and one s_nop covers them (advisor finding, round 3).
#define SRBDQP_FD(A, K, J) "v_fmac_f64_dpp %" #A ", %4, %" #K " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define SRBDQP_FD4(K0, K1, K2, K3, J0, J1, J2, J3) SRBDQP_FD(0, K0, J0) SRBDQP_FD(1, K1, J1) SRBDQP_FD(2, K2, J2) SRBDQP_FD(3, K3, J3)
// acc[j & 3] += k[j] * bcast_j(ch) for j = 0 .. CNT - 1 (CNT a multiple of 4, at most 16)
template <int CNT>
__device__ __forceinline__ void fmac_row_bcast_block(double (&acc)[4], double ch, const double* k) {
    static_assert(CNT == 4 || CNT == 8 || CNT == 12 || CNT == 16, "whole groups of four terms");
    if constexpr (CNT == 4)
        asm volatile("s_nop 1\n\t" SRBDQP_FD4(5, 6, 7, 8, 0, 1, 2, 3)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(ch), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]));
    else if constexpr (CNT == 8)
        asm volatile("s_nop 1\n\t" SRBDQP_FD4(5, 6, 7, 8, 0, 1, 2, 3) SRBDQP_FD4(9, 10, 11, 12, 4, 5, 6, 7)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "v"(ch), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]));
    else if constexpr (CNT == 12)
        asm volatile("s_nop 1\n\t" SRBDQP_FD4(5, 6, 7, 8, 0, 1, 2, 3) SRBDQP_FD4(9, 10, 11, 12, 4, 5, 6, 7) SRBDQP_FD4(13, 14, 15, 16, 8, 9, 10, 11)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "v"(ch), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]), "v"(k[8]), "v"(k[9]), "v"(k[10]),
                       "v"(k[11]));
    else
        asm volatile("s_nop 1\n\t" SRBDQP_FD4(5, 6, 7, 8, 0, 1, 2, 3) SRBDQP_FD4(9, 10, 11, 12, 4, 5, 6, 7) SRBDQP_FD4(13, 14, 15, 16, 8, 9, 10, 11)
                     SRBDQP_FD4(17, 18, 19, 20, 12, 13, 14, 15)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "v"(ch), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]), "v"(k[8]), "v"(k[9]), "v"(k[10]),
                       "v"(k[11]), "v"(k[12]), "v"(k[13]), "v"(k[14]), "v"(k[15]));
}
// acc[j & 3] += kin[16 C + j] * bcast_j(ch) for j = 0 .. 15, columns < KS
template <int KS, int C>
struct RowBcastChunk {
    template <int KN>
    static __device__ __forceinline__ void run(double (&acc)[4], double ch, const double (&kin)[KN]) {
        constexpr int CNT = (KS - 16 * C) < 16 ? (KS - 16 * C) : 16;
        if constexpr (CNT > 0) fmac_row_bcast_block<CNT>(acc, ch, kin + 16 * C);
    }
};

// max over the wave of a NON-NEGATIVE value (0 is shifted in at the row edges), result valid in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max_nonneg(double v) {
    v = dpp_max_step<0x111, 0xF>(v);   // row_shr:1
    v = dpp_max_step<0x112, 0xF>(v);   // row_shr:2
    v = dpp_max_step<0x114, 0xF>(v);   // row_shr:4
    v = dpp_max_step<0x118, 0xF>(v);   // row_shr:8
    v = dpp_max_step<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
    v = dpp_max_step<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
    return v;                          // lane 63 holds the wave maximum
}

// fp32 variant: one v_max_f32 with a DPP source per step (no 64-bit lane moves).  Monotone rounding commutes with max,
// so max over float(v_i) == float(max v_i): the oracle reproduces the decision by rounding its maxima to float32.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_maxf_step(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return fmaxf(v, __int_as_float(o));
}
__device__ __forceinline__ float wave_maxf_nonneg(float v) {
    v = dpp_maxf_step<0x111, 0xF>(v);
    v = dpp_maxf_step<0x112, 0xF>(v);
    v = dpp_maxf_step<0x114, 0xF>(v);
    v = dpp_maxf_step<0x118, 0xF>(v);
    v = dpp_maxf_step<0x142, 0xA>(v);
    v = dpp_maxf_step<0x143, 0xC>(v);
    return v;                          // lane 63 holds the wave maximum
}

// Four wave maxima at once (the four fp32 maxima of a convergence check), every step ONE v_max_f32 with a DPP source operand, in place: a lane whose DPP source is
// shifted in from outside its row (bound_ctrl off) or whose row the step masks out is simply not written and keeps its value -- max(v, 0) = v for the non-negative
// values these are.  Written out by the compiler from dpp_maxf_step a step was v_mov_b32 0 + v_mov_b32_dpp + a canonicalising v_max_f32 v, v, v + the v_max_f32 itself
// and, one value at a time, an s_nop for the DPP read-after-write hazard: 96 vector instructions and 18 s_nop per check against 24 here (the four chains interleaved
// put three instructions between a write and its DPP read: no wait states needed).  Inputs must not be NaN (the callers map a NaN residual to a huge number first).
// Results valid in lane 63.
#define SRBDQP_MAX4(CTRL) "v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %2, %2, %2 " CTRL "\n\tv_max_f32_dpp %3, %3, %3 " CTRL "\n\t"
__device__ __forceinline__ void wave_maxf4_nonneg(float& v0, float& v1, float& v2, float& v3) {
    asm volatile("s_nop 1\n\t"
                 SRBDQP_MAX4("row_shr:1 row_mask:0xf bank_mask:0xf") SRBDQP_MAX4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 SRBDQP_MAX4("row_shr:4 row_mask:0xf bank_mask:0xf") SRBDQP_MAX4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 SRBDQP_MAX4("row_bcast:15 row_mask:0xa bank_mask:0xf") SRBDQP_MAX4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
}

}  // namespace srbdqp

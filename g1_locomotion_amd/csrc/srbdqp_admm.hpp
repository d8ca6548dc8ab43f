// srbdqp_admm.hpp -- contact-local ADMM iteration (a9) for the register-resident explicit inverse.
//
// Everything in an ADMM iteration except x~ = K^-1 rhs is local to ONE contact point (3 variables, 5 constraint
// rows): z~ = A x~, the projection, the dual update, A'(rho z - y) and A' nu only couple fx, fy, fz of the same
// contact.  So wave w owns contacts [N w, N w + N): lane = 2 lr + h, lr = 3 cg + ax is the wave-local variable
// (cg = contact, ax = axis), and the 6 lanes of a contact also carry its 5 constraint rows
//     (ax, h) = (0,0) (0,1) (1,0) (1,1) (2,0)  <->  rows  fx-mu fz, -fx-mu fz, fy-mu fz, -fy-mu fz, fz.
// The coupling inside a contact goes through one DPP lane swap and two ds_bpermute per vector; the only LDS
// traffic per iteration is the double-buffered rhs vector (one barrier per iteration instead of three).
// Same arithmetic as oracle/srbd_oracle.py admm_solve(), iteration for iteration.
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

__device__ __forceinline__ double bperm_f64(double v, int src_lane) {   // value of lane src_lane
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

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

// block-wide max of NV non-negative values; `slot` alternates 0/1 between consecutive calls so that one barrier
// per call is enough (the second buffer is only rewritten after every wave has passed the next call's barrier).
template <int NV>
__device__ __forceinline__ void block_max_nonneg(double (&v)[NV], double* red, int slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* buf = red + slot * 32;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const double x = wave_max_nonneg(v[q]);
        if (lane == 63) buf[wave * 8 + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = fmax(fmax(buf[q], buf[8 + q]), fmax(buf[16 + q], buf[24 + q]));
}

// row r of K^-1 owned by this thread in the contact-local mapping (or -1)
template <int N>
__device__ __forceinline__ int admm_row_of_thread(int t) {
    const int w = t >> 6, lane = t & 63, lr = lane >> 1;
    return (lr < 3 * N) ? 3 * N * w + lr : -1;
}

// Preconditions: kin = this thread's fragment of row admm_row_of_thread(t); LDS q, px0 set; barrier passed.
// rhsbuf: 2 x (n + 8) doubles.  On return xs[0..n) holds the scaled solution (after a barrier).
template <int N, class L, int CH>
__device__ int admm_loop_local(const KArgs& a, int b, double* sm, double* rhsbuf, double* xs, const double (&kin)[CH],
                               int* status_out) {
    using S = L;
    constexpr int n = Dims<N>::n, m = Dims<N>::m;
    static_assert(6 * N <= 64, "one wave carries N contacts (6 lanes each)");
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int lr = lane >> 1, h = lane & 1;
    const bool active = lr < 3 * N;
    const int cg = lr / 3, ax = lr - 3 * cg;
    const int base = 6 * cg;                       // first lane of this contact's group
    const int gc = N * w + cg;                     // global contact index = 4 k + ci
    const int r = active ? 3 * gc + ax : 0;        // variable index
    const bool has_row = active && (ax < 2 || h == 0);
    const int j = (ax < 2) ? 2 * ax + h : 4;
    const int irow = has_row ? 5 * gc + j : 0;
    const double sigma = a.sigma, alpha = a.alpha, mu = a.mu;
    double* red = sm + S::o_red;

    // row constants
    const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
    const bool on = active && sct[gc] != 0;
    double lo = 0.0, hi = 0.0, rho = a.rho;
    if (j < 4) { lo = -kInf; hi = 0.0; }
    else { lo = on ? a.fzmin_s : 0.0; hi = on ? a.fzmax_s : 0.0; rho = on ? a.rho : a.rho_eq; }
    const double irho = 1.0 / rho;
    const double sgn = (j < 4 && h == 1) ? -1.0 : 1.0;
    const double muc = (j < 4) ? mu : 0.0;
    const double rowm = has_row ? 1.0 : 0.0;       // masks lanes that carry no row

    // A' v for the variable of this lane, v = per-row values (0 on row-less lanes)
    auto At = [&](double v) -> double {
        const double other = dpp_swap1(v);
        const double s = v + other;
        const double s01 = bperm_f64(s, base + 0), s23 = bperm_f64(s, base + 2);
        const double d = (h == 0) ? v - other : other - v;
        return (ax < 2) ? d : fma(-mu, s01 + s23, s);
    };
    // (A v)_row for the row of this lane, v = per-variable values
    auto Arow = [&](double v) -> double {
        const double vf = bperm_f64(v, base + 4);
        return fma(-muc, vf, sgn * v);
    };

    // ---- initial point
    const double qv = active ? sm[S::o_q + r] : 0.0;
    double x = (active && a.warm_u) ? a.warm_u[(size_t)b * n + r] / a.s : 0.0;
    double px = active ? sm[S::o_px0 + r] : 0.0;
    double y = (has_row && a.warm_y) ? a.warm_y[(size_t)b * m + irow] : 0.0;
    double z = rowm * fmin(fmax(Arow(x), lo), hi);
    double qn[1] = {fabs(qv)};
    block_max<1>(qn, red);
    {
        const double rhs0 = sigma * x - qv + At(rowm * (rho * z - y));
        if (active && h == 0) rhsbuf[r] = rhs0;
    }
    __syncthreads();

    int status = 2, iters = a.max_iter, nchk = 0;
    long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0, seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0;
    (void)tp0; (void)tp1; (void)tp2; (void)tp3; (void)tp4; (void)seg0; (void)seg1; (void)seg2; (void)seg3;
    for (int k = 1; k <= a.max_iter; ++k) {
        ADMM_T(tp0);
        const bool check = (k % a.check_every == 0) || (k == a.max_iter);
        const double* rb = rhsbuf + ((k - 1) & 1) * (n + 8);
        double* wb = rhsbuf + (k & 1) * (n + 8);
        // ---- x~ = K^-1 rhs (row fragment x broadcast rhs), pair-reduced over the two halves
        double xt;
        {
            // rhs is read in blocks of BL ds_read_b128 that are all in flight before the FMAs that consume them
            // (the mat-vec is LDS-latency bound otherwise: the compiler keeps only ~3 loads in flight)
            constexpr int BL = 8, NV = CH / 2, NB = (NV + BL - 1) / BL;
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            const double2* rv = reinterpret_cast<const double2*>(rb + CH * h);
            double2 cur[BL], nxt[BL];
#pragma unroll
            for (int i = 0; i < BL; ++i) cur[i] = rv[i < NV ? i : 0];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (blk + 1 < NB) {
#pragma unroll
                    for (int i = 0; i < BL; ++i) { const int q = (blk + 1) * BL + i; nxt[i] = rv[q < NV ? q : 0]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BL; i += 2) {
                    const int q = blk * BL + i;
                    if (q < NV) { acc0 = fma(kin[2 * q], cur[i].x, acc0); acc1 = fma(kin[2 * q + 1], cur[i].y, acc1); }
                    if (q + 1 < NV) { acc2 = fma(kin[2 * q + 2], cur[i + 1].x, acc2); acc3 = fma(kin[2 * q + 3], cur[i + 1].y, acc3); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BL; ++i) cur[i] = nxt[i];
            }
            const double acc = (acc0 + acc1) + (acc2 + acc3);
            xt = acc + dpp_swap1(acc);
        }
        ADMM_T(tp1);
        // ---- rows of this contact: z~, nu, relaxation, projection, dual update
        const double zt = Arow(xt);
        const double nu = rowm * (rho * (zt - z) + y);
        const double zh = alpha * zt + (1.0 - alpha) * z;
        const double zn = fmin(fmax(zh + y * irho, lo), hi);
        y = rowm * (y + rho * (zh - zn));
        z = rowm * zn;
        const double wv = rho * z - y;
        // ---- variables: P x~ from the KKT identity, relaxation, next right-hand side
        const double atnu = At(nu), atw = At(wv);
        const double pxt = sigma * (x - xt) - qv - atnu;
        x = alpha * xt + (1.0 - alpha) * x;
        px = alpha * pxt + (1.0 - alpha) * px;
        if (active && h == 0) wb[r] = sigma * x - qv + atw;
        ADMM_T(tp2);
        if (check) {
            const double ax_ = Arow(x);
            const double aty = At(y);
            double rd = fabs(px + qv + aty), rp = fabs(ax_ - z);
            rd = (rd == rd) ? rd : kInf * 10.0;                  // a NaN residual must survive the max
            rp = (rp == rp) ? rp : kInf * 10.0;
            double v[4];
            v[0] = rowm * rp;
            v[1] = rowm * fmax(fabs(ax_), fabs(z));
            v[2] = active ? rd : 0.0;
            v[3] = active ? fmax(fabs(px), fabs(aty)) : 0.0;
            block_max_nonneg<4>(v, red, nchk & 1);   // its barrier also publishes wb[]
            ++nchk;
            ADMM_T(tp3); seg3 += tp3 - tp2;
            const double e_prim = a.eps_abs + a.eps_rel * v[1];
            const double e_dual = a.eps_abs + a.eps_rel * fmax(v[3], qn[0]);
            const bool bad = !(v[0] <= kInf) || !(v[2] <= kInf);
            if (bad) { status = -1; iters = k; break; }
            if (v[0] <= e_prim && v[2] <= e_dual) { status = 1; iters = k; break; }
        } else {
            __syncthreads();
            ADMM_T(tp3); seg2 += tp3 - tp2;
        }
        seg0 += tp1 - tp0; seg1 += tp2 - tp1;
    }
#ifdef SRBDQP_PROFILE_ADMM
    if (a.stamps && t == 0) { a.stamps[(size_t)b * 16 + 12] = seg0; a.stamps[(size_t)b * 16 + 13] = seg1; a.stamps[(size_t)b * 16 + 14] = seg2; a.stamps[(size_t)b * 16 + 15] = seg3; }
#endif
    if (active && h == 0) xs[r] = x;
    if (a.y_out && has_row) a.y_out[(size_t)b * m + irow] = y;
    __syncthreads();
    *status_out = status;
    return iters;
}

}  // namespace srbdqp

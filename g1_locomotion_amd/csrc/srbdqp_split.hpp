// srbdqp_split.hpp -- second kernel of the split pipeline: the ADMM iterations and the roll-out of ONE QP on ONE wave.
//
// Why split: inside the fused kernel an ADMM iteration is a chain of ~75 dependent instructions on each of 4 waves plus
// an LDS exchange and a barrier (1.45 k cycles, 4 x 115 issue slots); the set-up phases want 4 workgroups per CU and
// therefore a 128-register budget, which is what forces the K^-1 row over 4 lanes.  Here a lane owns a whole K^-1 row
// (n_eff <= 64 doubles in registers, 2 waves per SIMD), the right-hand side is exchanged inside the wave (one
// ds_write + broadcast ds_reads, no barrier), a contact is 3 lanes that carry 2 + 2 + 1 rows, and the convergence
// decision is a wave reduction: ~180 issue slots per iteration on one wave instead of 460 on four, 8 QPs in flight per
// CU instead of 4.  The price is the hand-over of K^-1 (n_eff^2 doubles) and the persistent strip through HBM
// (SplitWs, 34 KB per QP for N = 10): ~0.9 TB/s at the measured rate, far from the 8 TB/s bound.
//
// Arithmetic as oracle admm_solve() / admm_loop_compact(): same recursions (P x = c - A's), same fp32-rounded
// maxima, same pre-test rule, so iteration counts agree with the fused kernel and the oracle.
#pragma once
#include "srbdqp_compact.hpp"

namespace srbdqp {

template <int N, int MAXS>
struct SplitSmem {
    using S = CompactSmem<N, MAXS>;
    static constexpr int o_rhs = (S::o_R + 1) & ~1;        // 64 (+ padding to the row stride)
    static constexpr int o_xs = o_rhs + 72;                // n: full variable vector for the roll-out
    static constexpr int o_scr = o_xs + Dims<N>::n;        // 12N roll-out scratch
    static constexpr int o_end = o_scr + 12 * N + 2;
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
};

// The ADMM iterations of one QP on ONE wave (wave-local: no barrier inside; the caller's waves other than this one may wait at one).  sm: the persistent
// strip at [0, S::o_R); kin: the K^-1 row of this lane (zeros for lanes >= n_eff); xs_full: n doubles that receive the solution in the full variable order
// (zeros elsewhere).  Used by the one-wave kernels and by the 4-wave kernel's batch-1 instantiation (srbdqp_compact.hpp, TAIL1).
// In-kernel rho restart of the one-wave kernel (srbdqp_setup1.hpp, RST): the pass this call is, what it may run, where a pass that ends at its cap leaves
// (x in newtons, y) for the next one, and the maxima of its last full check for the re-balancing rule.
struct WaveRestart {
    int pass = 0;            // 0 = first pass (warm start from the caller's arrays, if any); > 0: from park[]
    int kcap = 0;            // iterations this pass may run
    bool more = false;       // another pass may follow: a QP that ends at kcap parks its (x, y) and stores no duals
    double* park = nullptr;  // LDS [3][64]: x (newtons), y of slot A, y of slot B, per lane
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
};

// io: the QP's own pointers (the launch's, or -- a deferred continuation -- those of the launch it came from); warm_u / warm_y: the caller's warm start or null
// (a continuation starts from rs->park, never from these).
template <int N, int MAXS, bool RST>
__device__ __forceinline__ void admm_wave_iterations(const KArgs& a, const QpIo& io, const double* warm_u, const double* warm_y, int b, double rho_b, double* sm,
                                                     const double (&kin)[SplitWs<N, MAXS>::KS], double* xs_full, int& status_out, int& iters_out, WaveRestart* rs) {
    using S = CompactSmem<N, MAXS>;
    using W = SplitWs<N, MAXS>;
    constexpr int n = Dims<N>::n, m = Dims<N>::m, KS = W::KS;
    int lane = threadIdx.x & 63;
    if constexpr (RST) asm volatile("" : "+v"(lane));                   // (inside the restart loop: an opaque value in every trip, see srbdqp_setup1.hpp)
    for (int i = lane; i < n; i += 64) xs_full[i] = 0.0;                 // (one wave: its LDS operations complete in order)
    const int* icnt = reinterpret_cast<const int*>(sm + S::o_int);
    const int* imisc = icnt + 2 * N;
    const uint8_t* act = reinterpret_cast<const uint8_t*>(imisc + 8);
    const int na = imisc[0], n_eff = 3 * na;
    const bool failed = sm[S::o_misc] != 0.0;

    // ---- lane mapping: contact cg = lane / 3 owns lanes 3 cg + {0, 1, 2} = its variables fx, fy, fz; rows: slot A of
    // the fx / fy lanes = +f - mu fz <= 0, slot B = -f - mu fz <= 0, slot A of the fz lane = the normal-force bound
    const int cg = lane / 3, ax = lane - 3 * cg;
    const bool active = cg < na;
    const int base = 3 * cg;
    const int gc = active ? act[cg] : 0;
    const bool rowA = active, rowB = active && ax < 2;
    const double sigma = a.sigma, alpha = a.alpha, oma = 1.0 - a.alpha, mu = a.mu;
    const double rho_A = (ax < 2) ? rho_b : rho_b * a.rho_fz;              // slot A of the fz lane = the normal-force row: its own penalty
    const double irhoA = 1.0 / rho_A, irhoB = 1.0 / rho_b;
    const double rhoA = rowA ? rho_A : 0.0, rhoB = rowB ? rho_b : 0.0;      // rho = 0 freezes a slot at y = z = 0
    const double loA = !rowA ? 0.0 : (ax < 2 ? -kInf : a.fzmin_s), hiA = !rowA ? 0.0 : (ax < 2 ? 0.0 : a.fzmax_s);
    const double loB = rowB ? -kInf : 0.0, hiB = 0.0;
    const double mucA = (ax < 2) ? mu : 0.0;
    const int irowA = 5 * gc + ((ax < 2) ? 2 * ax : 4), irowB = 5 * gc + 2 * ax + 1;

    // A'(wA, wB) for the variable of this lane; (A v)_slot for the two slots
    auto At = [&](double wA, double wB) -> double {
        const double ssum = wA + wB;
        const double sxy = contact_sum_xy(ssum);                     // (outside the select: a cross-lane read must run with the source lanes enabled)
        return (ax < 2) ? wA - wB : fma(-mu, sxy, wA);
    };

    int status = -1, iters = 0;
    if (!failed) {
        const double qv = active ? sm[S::o_q + lane] : 0.0;
        double x = (active && warm_u) ? warm_u[(size_t)b * n + 3 * gc + ax] / a.s : 0.0;
        double cpx = active ? sm[S::o_px0 + lane] : 0.0, spxA = 0.0, spxB = 0.0;
        double yA = (rowA && warm_y) ? warm_y[(size_t)b * m + irowA] : 0.0;
        double yB = (rowB && warm_y) ? warm_y[(size_t)b * m + irowB] : 0.0;
        int kmax = a.max_iter;
        if constexpr (RST) {
            kmax = rs->kcap;
            if (rs->pass > 0) {   // (wave-uniform) continue the pass before: the forces re-read in newtons, as a second launch would
                x = active ? rs->park[lane] / a.s : 0.0;
                yA = rowA ? rs->park[64 + lane] : 0.0;
                yB = rowB ? rs->park[128 + lane] : 0.0;
            }
        }
        const double fz0 = contact_fz(x, ax);
        double axA = rowA ? fma(-mucA, fz0, x) : 0.0, axB = rowB ? fma(-mu, fz0, -x) : 0.0;     // (A x)_slot by recursion
        double zA = fmin(fmax(axA, loA), hiA), zB = fmin(fmax(axB, loB), hiB);
        const float qnf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_maxf_nonneg((float)fabs(qv))), 63));
        double rhs = sigma * x - qv + At(rhoA * zA - yA, rhoB * zB - yB);     // (inactive lanes: 0)
        status = 2; iters = kmax;
        double e_prim_last = kInf * 1.0e10;
        bool vote_ok = true;
        int ph = 0;
        float lastv0 = 0.0f, lastv1 = 0.0f, lastv2 = 0.0f, lastv3 = 0.0f;   // maxima of the last full check (restart rule)
        for (int k = 1; k <= kmax; ++k) {
            if (++ph == a.check_every) ph = 0;
            const bool at_mark = (ph == 0);
            const bool check = (at_mark && vote_ok) || (k == kmax);
            const bool pretest = (ph == a.check_every - 1);
            // x~ = K^-1 rhs, a K^-1 row per lane.  The right-hand side never leaves the register file's neighbourhood: every 16-lane
            // row fetches the four 16-value chunks of the vector with ds_bpermute (8 instructions), then each term is one
            // v_fmac_f64 whose operand is broadcast inside the row by DPP (fmac_row_bcast).  Until round 3 this was one ds_write_b64
            // and KS / 2 broadcast ds_read_b128 per iteration: with 8 QPs resident per CU the LDS pipeline, not the vector ALU,
            // paced the iteration.
            double acc4[4] = {0.0, 0.0, 0.0, 0.0};
            {
                constexpr int NCH = (KS + 15) / 16;
                static_assert(NCH <= 4, "one K^-1 row per lane of one wave");
                double ch[4];
                row_chunks(rhs, ch);
                RowBcastChunk<KS, 0>::run(acc4, ch[0], kin);
                if constexpr (NCH > 1) RowBcastChunk<KS, 1>::run(acc4, ch[1], kin);
                if constexpr (NCH > 2) RowBcastChunk<KS, 2>::run(acc4, ch[2], kin);
                if constexpr (NCH > 3) RowBcastChunk<KS, 3>::run(acc4, ch[3], kin);
            }
            const double acc0 = acc4[0], acc1 = acc4[1], acc2 = acc4[2], acc3 = acc4[3];
            const double xt = (acc0 + acc1) + (acc2 + acc3);
            const double fzt = contact_fz(xt, ax);
            const double ztA = fma(-mucA, fzt, xt), ztB = fma(-mu, fzt, -xt);
            // rows (two slots), relaxation, projection, dual update
            const double nuA = fma(rhoA, ztA - zA, yA), nuB = fma(rhoB, ztB - zB, yB);
            const double zhA = fma(alpha, ztA, oma * zA), zhB = fma(alpha, ztB, oma * zB);
            const double znA = fmin(fmax(fma(yA, irhoA, zhA), loA), hiA), znB = fmin(fmax(fma(yB, irhoB, zhB), loB), hiB);
            yA = fma(rhoA, zhA - znA, yA); yB = fma(rhoB, zhB - znB, yB);
            zA = znA; zB = znB;
            axA = fma(alpha, ztA, oma * axA); axB = fma(alpha, ztB, oma * axB);
            spxA = fma(alpha, nuA, oma * spxA); spxB = fma(alpha, nuB, oma * spxB);
            const double atw = At(fma(rhoA, zA, -yA), fma(rhoB, zB, -yB));
            cpx = fma(alpha, fma(sigma, x - xt, -qv), oma * cpx);
            x = fma(alpha, xt, oma * x);
            rhs = fma(sigma, x, -qv) + atw;                                // (inactive lanes: 0)
            if (pretest) {
                const bool bad = (rowA && !(fabs(axA - zA) <= e_prim_last)) || (rowB && !(fabs(axB - zB) <= e_prim_last));
                vote_ok = __ballot(bad) == 0ull;
            }
            if (check) {
                const double aty = At(yA, yB), px = cpx - At(spxA, spxB);
                double rd = fabs(px + qv + aty);
                double rp = fmax(rowA ? fabs(axA - zA) : 0.0, rowB ? fabs(axB - zB) : 0.0);
                rd = (rd == rd) ? rd : kInf * 10.0;                        // a NaN residual must survive the max
                rp = (rp == rp) ? rp : kInf * 10.0;
                const double nr = fmax(rowA ? fmax(fabs(axA), fabs(zA)) : 0.0, rowB ? fmax(fabs(axB), fabs(zB)) : 0.0);
                const float v0 = (float)rp, v1 = (float)nr;
                const float v2 = active ? (float)rd : 0.0f, v3 = active ? (float)fmax(fabs(px), fabs(aty)) : 0.0f;
                float r0 = v0, r1 = v1, r2 = v2, r3 = v3;
                wave_maxf4_nonneg(r0, r1, r2, r3);
                const float m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), 63));
                const float m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), 63));
                const float m2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r2), 63));
                const float m3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r3), 63));
                const double e_prim = a.eps_abs + a.eps_rel * (double)m1;
                const double e_dual = a.eps_abs + a.eps_rel * fmax((double)m3, (double)qnf);
                e_prim_last = e_prim;
                lastv0 = m0; lastv1 = m1; lastv2 = m2; lastv3 = fmaxf(m3, qnf);
                if (!((double)m0 <= kInf) || !((double)m2 <= kInf)) { status = -1; iters = k; break; }
                if ((double)m0 <= e_prim && (double)m2 <= e_dual) { status = 1; iters = k; break; }
            }
        }
        if (a.resid_out && status == 2 && lane == 0) {
            float* ro = a.resid_out + (size_t)b * 4;
            ro[0] = lastv0; ro[1] = lastv1; ro[2] = lastv2; ro[3] = lastv3;
            if (a.cap_list) a.cap_list[atomicAdd(a.cap_count, 1)] = b;
        }
        if (status < 0) { x = 0.0; yA = 0.0; yB = 0.0; }                    // a numerical failure returns zero forces, never NaN
        if (active) xs_full[3 * gc + ax] = x;
        bool parked = false;
        if constexpr (RST) {
            rs->v[0] = lastv0; rs->v[1] = lastv1; rs->v[2] = lastv2; rs->v[3] = lastv3;
            parked = status == 2 && rs->more;
            if (parked) { rs->park[lane] = a.s * x; rs->park[64 + lane] = yA; rs->park[128 + lane] = yB; }
        }
        if (io.y_out && (!a.y_capped_only || status == 2) && !parked) {
            if (rowA) io.y_out[(size_t)b * m + irowA] = yA;
            if (rowB) io.y_out[(size_t)b * m + irowB] = yB;
        }
    }
    status_out = status; iters_out = iters;
}

// What follows the iterations of the one-wave kernels: the duals of the eliminated rows, status, iteration count, roll-out
template <int N, int MAXS>
__device__ __forceinline__ void admm_wave_finish(const KArgs& a, const QpIo& io, int b, double* sm, int status, int iters) {
    using S = CompactSmem<N, MAXS>;
    using L = SplitSmem<N, MAXS>;
    constexpr int m = Dims<N>::m;
    const int lane = threadIdx.x;
    double* xs_full = sm + L::o_xs;
    const bool failed = sm[S::o_misc] != 0.0;
    __syncthreads();
    if (io.y_out && (!a.y_capped_only || status == 2)) {   // rows of eliminated (swing) contacts, or of a failed solve: 0
        const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
        for (int i = lane; i < m; i += 64)
            if (failed || sct[i / 5] == 0) io.y_out[(size_t)b * m + i] = 0.0;
    }
    if (lane == 0) {
        if (io.status) io.status[b] = status;
        if (io.iters) io.iters[b] = iters + a.iters_base;
    }
    rollout_and_store_to<N, S, 64>(a, io.u_out, io.x_out, b, sm, xs_full, sm + L::o_scr);
}

// ADMM iterations + roll-out of one QP on one wave (the one-wave kernels).  sm: the persistent strip at [0, S::o_R) and SplitSmem's vectors behind it.
template <int N, int MAXS>
__device__ __forceinline__ void admm_wave_body(const KArgs& a, int b, double rho_b, double* sm, const double (&kin)[SplitWs<N, MAXS>::KS]) {
    int status = -1, iters = 0;
    const QpIo io = io_of(a);
    admm_wave_iterations<N, MAXS, false>(a, io, a.warm_u, a.warm_y, b, rho_b, sm, kin, sm + SplitSmem<N, MAXS>::o_xs, status, iters, nullptr);
    admm_wave_finish<N, MAXS>(a, io, b, sm, status, iters);
}

template <int N, int MAXS>
__global__ __launch_bounds__(64, 2) void srbdqp_admm_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    using S = CompactSmem<N, MAXS>;
    using W = SplitWs<N, MAXS>;
    constexpr int KS = W::KS;
    static_assert(W::supported, "one K^-1 row per lane of one wave");
    if ((int)blockIdx.x >= a.B) return;
    if (a.count_ptr && (int)blockIdx.x >= *a.count_ptr) return;   // restart pass: nothing listed for this workgroup
    const int b = SRBDQP_QP_INDEX(a);
    if (SRBDQP_RESTART_SKIP(a, b)) return;
    const double rho_b = SRBDQP_RHO_OF(a, b);
    const int lane = threadIdx.x;
    const double* ws = a.ws + (size_t)b * W::doubles;
    if (ws[S::o_misc + 1] != 0.0) { signal_done(a); return; }   // finished by the set-up kernel (no stance contact / bound)
    for (int i = lane; i < S::o_R; i += 64) sm[i] = ws[i];
    __syncthreads();
    if (a.defer_x0) {   // two-phase call: the set-up saw a predicted x0; q = q(x0_pred) + dq/dx0 (x0 - x0_pred), roll-out from x0
        const double* phi = ws + W::o_phi;
        const double* gx0 = a.x0 + (size_t)b * 13;
        const int ne = 3 * (reinterpret_cast<const int*>(sm + S::o_int) + 2 * N)[0];
        if (lane < ne) {
            double dq = 0.0;
#pragma unroll
            for (int k = 0; k < 13; ++k) dq = fma(phi[lane * 13 + k], gx0[k] - sm[S::o_x0 + k], dq);
            sm[S::o_q + lane] += dq;
        }
        __syncthreads();
        if (lane < 13) sm[S::o_x0 + lane] = gx0[lane];
        __syncthreads();
    }
    const int n_eff = 3 * (reinterpret_cast<const int*>(sm + S::o_int) + 2 * N)[0];
    // K^-1 row of this lane
    double kin[KS];
    {
        const double2* src = reinterpret_cast<const double2*>(ws + W::o_kinv + (size_t)(lane < n_eff ? lane : 0) * KS);
#pragma unroll
        for (int c = 0; c < KS / 2; ++c) {
            const double2 v = src[c];
            kin[2 * c] = (lane < n_eff) ? v.x : 0.0;
            kin[2 * c + 1] = (lane < n_eff) ? v.y : 0.0;
        }
    }

    admm_wave_body<N, MAXS>(a, b, rho_b, sm, kin);
    signal_done(a);   // two-phase staged call: completion word in host memory (no-op without a done_flag)
}

}  // namespace srbdqp

// srbdqp_compact.hpp -- kernel variant v2 ("compact"): the fp64 pipeline on the PRESOLVED QP, assembled in closed form.
//
// Presolve = fixed-variable elimination: a swing contact's force is clamped to zero by its constraint rows, so its
// 3 variables and 5 rows are dropped before anything dense is built.  With na stance contact points over the
// horizon the QP has n_eff = 3 na variables and 5 na rows (2-contact single support: n_eff = 6 N instead of 12 N):
// 4x less factor work, a 4x smaller ADMM mat-vec, and (measured on the oracle) a much shorter tail of
// slow-converging QPs, because the rho_eq-weighted clamp rows no longer ill-condition K.
//
// Template parameter MAXS = bound on stance contact points per horizon step (2 = single support classes,
// 4 = anything): it sizes LDS and the register tile slots.  A QP that violates the bound is reported with status
// SRBDQP_CONTACT_BOUND, never silently mis-solved.
//
// Compact ordering: stance contacts sorted by (step, contact index) -> e = 0..na-1; variable 3e+ax, row 5e+j.
//
// Assembly: the condensed input matrix G = Q^1/2 s B_qp is never built.  Its blocks have the SRBD structure
// theta: dt^2 (C_i - C_j) J_j, p: (i-j) dt^2/m I, omega: dt J_j, v: dt/m I (C = prefix sums of R_z'), so K = G'G + ...
// is J_e' M(j,m) J_e' per contact pair -- assembled since round 4 in the rank-6 form M(j,m) = D_m - C_j' E_m (one table row per presolved variable, every lane its own
// C-layout entries: srbdqp_common.hpp, kasm_rows / kasm_tile) -- and G'v is a suffix sum per step (see phase A in the kernel).  From K on it is v1's machinery with run-time tile counts: register-resident tiled Cholesky with
// the diagonal tiles inverted on the matrix cores, in-place W = L^-1, K^-1 = W'W.
#pragma once
#include "srbdqp_common.hpp"
#include "srbdqp_admm.hpp"
#include "srbdqp_mfma.hpp"

namespace srbdqp {

// With a 128-register budget the compiler otherwise hoists every lane-index expression of the later phases (swizzled
// tile offsets, the identity pattern of the diagonal-tile inversion) to the top of the kernel and then spills them
// across the factorisation: ~17 dwords per lane of scratch = 70 MB of HBM writes per 4096-QP launch.  Passing the lane
// indices through an empty asm at each phase boundary makes those expressions local to their phase again.
#define SRBDQP_PHASE_LOCAL(...) asm volatile("" : __VA_ARGS__)

constexpr int kStatusContactBound = -2;

template <int N, int MAXS>
struct CompactSmem {
    static constexpr int n = Dims<N>::n, m = Dims<N>::m;
    static constexpr int NCMAX = MAXS * N;                // stance contacts over the horizon
    static constexpr int nmax = 3 * NCMAX;                // compact variables
    static constexpr int NT = (nmax + 15) / 16;
    static constexpr int NTT = NT * (NT + 1) / 2;
    static constexpr int TS = (NTT + 3) / 4;
    // lanes per K^-1 row in the ADMM mat-vec: 4 when 4 nmax lanes fit the workgroup (a row fragment of nmax/4 doubles
    // keeps the kernel inside the 128-register budget of 4 workgroups per CU), else 2
#ifdef SRBDQP_FORCE_LPR2
    static constexpr int LPR = 2;
#else
    static constexpr int LPR = (4 * nmax <= kThreads) ? 4 : 2;
#endif
    static constexpr int CHMAX = 2 * ((nmax + 2 * LPR - 1) / (2 * LPR));   // columns per mat-vec part (even)
    static constexpr int up2(int v) { return (v + 1) & ~1; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // ---- persistent
    static constexpr int o_x0 = 0;
    static constexpr int o_tm = o_x0 + 14;
    static constexpr int o_J = o_tm + up2(N * 9);
    static constexpr int o_q = o_J + N * 36;              // nmax (compact gradient)
    static constexpr int o_px0 = o_q + up2(nmax);         // nmax
    static constexpr int o_red = o_px0 + up2(nmax);       // 32 (block_max: 4 waves x 8; the ADMM's fp32 maxima + flags: 18)
    static constexpr int o_ct = o_red + 32;
    static constexpr int o_misc = o_ct + up2((N * 4 + 7) / 8);
    static constexpr int o_sq = o_misc + 8;
    static constexpr int o_int = o_sq + 12;               // ints: cnt[N], rowbase[N], na, flags; bytes: act[4N]
    static constexpr int o_R = o_int + up2((2 * N + 8) / 2 + (4 * N + 7) / 8 + 1);
    // phase A (closed-form assembly: inputs, C prefix sums, error vector, the small tables of the assembly)
    static constexpr int NPAIR = N * (N + 1) / 2;
    static constexpr int o_xref = o_R;
    static constexpr int o_foot = o_xref + up2(N * 13);
    static constexpr int o_pcom = o_foot + N * 12;
    static constexpr int o_cp = o_pcom + up2(N * 3);
    static constexpr int o_eh = o_cp + up2(N * 9);        // n: Q^1/2 (A_qp x0 - x_ref)
    static constexpr int o_t1 = o_eh + n;                 // 9N: T1(m)
    static constexpr int o_t2 = o_t1 + up2(9 * N);        // 9N: T2(m)
    static constexpr int o_mt = o_t2 + up2(9 * N);        // 18N: D_m, E_m of the rank-6 assembly (srbdqp_common.hpp; the region keeps the 9 NPAIR doubles the M(j, m) table had)
    static constexpr int o_gv = o_mt + up2(18 * N);       // 9N: G'v tables
    static constexpr int o_gx = o_gv + up2(9 * N);        // n: G x^0 (warm start)
    static constexpr int o_tf = o_gx + n;                 // 6N: per-step torque / force sums of x^0
    static constexpr int o_x0c = o_tf + 6 * N;            // nmax (+16): compact warm start
    static constexpr int o_ab = o_x0c + up2(nmax) + 16;   // 16 NT rows of the K assembly's table, one per presolved variable
    static constexpr int endA = o_ab + 16 * NT * kAbStride;
    // phase B.  Every lane builds its K entries in registers from the table rows; the tile store re-uses region R.  (Until round 4 the small problems
    // assembled K contact pair by contact pair straight into a tile store behind the phase-A arrays.)
    static constexpr int o_T = o_R;
    static constexpr int endB = o_T + NTT * 256;
    // phase C
    static constexpr int o_rhs = o_R;                     // 2 x (LPR CHMAX + 8)
    static constexpr int o_xs = o_rhs + 2 * (LPR * CHMAX + 8);   // n (full variable vector, for the roll-out)
    static constexpr int endC = o_xs + n;
    static constexpr int o_end = cmax(endA, cmax(endB, endC));
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
    // workgroups per CU that LDS admits (160 KiB per CU), capped at 3: the register budget the kernel is compiled for
    // (3 only for the small problems: a 168-register budget cannot hold a K^-1 row fragment of more than 30 doubles)
    static constexpr int lds_wgs = 163840 / (int)bytes;
#ifdef SRBDQP_COMPACT_WPS   // experiments: one register budget for every instantiation
    static constexpr int waves_per_simd = SRBDQP_COMPACT_WPS;
#else
    static constexpr int waves_per_simd = (lds_wgs >= 4 && LPR == 4 && nmax <= 60) ? 4 : (lds_wgs >= 3 && nmax <= 72) ? 3 : (lds_wgs >= 2 ? 2 : 1);
#endif
};

// Lane mapping of the presolved ADMM: a contact owns 3 LPR consecutive lanes (variable ax on lanes LPR ax .. LPR ax +
// LPR - 1, each holding 1/LPR of that K^-1 row; the contact's 5 constraint rows ride on the first two lanes of each
// variable), and the na stance contacts are packed into as FEW waves as possible (64 / (3 LPR) contacts per wave): the
// mat-vec is bound by the LDS broadcast of rhs, which every participating wave reads in full.  Idle waves only join
// barriers.
template <int LPR> __device__ __forceinline__ int admm_waves_used(int na) { constexpr int C = 64 / (3 * LPR); return (na + C - 1) / C; }
template <int LPR> __device__ __forceinline__ int admm_contacts_per_wave(int na) { const int wu = admm_waves_used<LPR>(na); return (na + wu - 1) / wu; }

// The presolved ADMM (OSQP Algorithm 1 on the compact contacts), arithmetic as oracle admm_solve().  Per iteration:
//   x~ = K^-1 rhs      row fragment x rhs broadcast from LDS (all loads in flight, then the FMAs), halves summed by DPP
//   rows               the lane carrying row j gets fz~ by one ds_bpermute pair; z~, nu, relaxation, projection, dual
//   A'(rho z - y)      one DPP swap + two ds_bpermute pairs; new rhs to the OTHER LDS buffer; ONE barrier
// P x is NOT updated every iteration: with c_{k+1} = (1-a) c_k + a (sigma (x_k - x~) - q) per variable and
// s_{k+1} = (1-a) s_k + a nu_k per row, P x_k = c_k - A' s_k, so A' nu is only applied at check iterations.
// Check iterations reduce 4 maxima in fp32 with DPP and publish them on the iteration's own barrier; the decision
// is read at the top of the next iteration (no extra barrier, no extra iteration).
template <int N, class L, int CHMAX>
__device__ int admm_loop_compact(const KArgs& a, int b, double* sm, double* rhsbuf, double* xs_full,
                                 const double (&kin)[CHMAX], int na, int CH, const uint8_t* act, double rho_b, int* status_out) {
    using S = L;
    constexpr int n = Dims<N>::n, m = Dims<N>::m;
    constexpr int LPR = S::LPR;
    constexpr int RB = LPR * CHMAX + 8;                // one rhs buffer
    // chunk stride of the rhs vector: the LPR = 4 lanes of a row read their chunks with 16-byte reads CH doubles apart -- at CH = 16 (N = 10, 20 stance contacts)
    // two of the four sit on the same banks; 18 apart they do not (the 8 spare doubles of the buffer hold the extra 3 x 2)
    const int CHS = (LPR == 4 && (CH & 15) == 0 && LPR * (CH + 2) <= RB) ? CH + 2 : CH;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int lr = lane / LPR, hp = lane % LPR;        // row slot in the wave, part of the row this lane multiplies
    const int h = hp & 1;
    const bool prim = hp < 2;                          // the two lanes of a variable that carry constraint rows
    const int CPW = admm_contacts_per_wave<LPR>(na);
    const int cg = lr / 3, ax = lr - 3 * cg;
    const int e = w * CPW + cg;
    const bool active = (cg < CPW) && (e < na);
    const int base = 3 * LPR * cg;
    const int gc = active ? act[e] : 0;                // original contact index 4 k + ci
    const int r = active ? 3 * e + ax : 0;             // compact variable
    const int rpos = (r / CH) * CHS + (r % CH);        // ... and its slot in the chunked rhs vector
    const bool has_row = active && prim && (ax < 2 || h == 0);
    const int j = (ax < 2) ? 2 * ax + h : 4;
    const int irow = 5 * gc + j;                       // original row index (for warm_y / y_out)
    const double sigma = a.sigma, alpha = a.alpha, oma = 1.0 - a.alpha, mu = a.mu;
    float* redf = reinterpret_cast<float*>(sm + S::o_red);
    const double lo = (j < 4) ? -kInf : a.fzmin_s, hi = (j < 4) ? 0.0 : a.fzmax_s;
    const double rho = (j == 4) ? rho_b * a.rho_fz : rho_b, irho = 1.0 / rho;   // the normal-force row has its own penalty
    const double sgn = (j < 4 && h == 1) ? -1.0 : 1.0;
    const double muc = (j < 4) ? mu : 0.0;
    const double rowm = has_row ? 1.0 : 0.0;
    const bool wave_on = w < admm_waves_used<LPR>(na);

    // row values live on the prim lanes (0 elsewhere: rowm); the results are valid on the prim lanes and, for LPR = 4,
    // only consumed there (x, c, s of the other two lanes of a variable are never read)
    auto At = [&](double v) -> double {
        const double other = dpp_swap1(v);
        const double s = v + other;
        const double s01 = bperm_f64(s, base + 0), s23 = bperm_f64(s, base + LPR);
        const double d = (h == 0) ? v - other : other - v;
        return (ax < 2) ? d : fma(-mu, s01 + s23, s);
    };
    auto Arow = [&](double v) -> double {
        const double vf = bperm_f64(v, base + 2 * LPR);
        return fma(-muc, vf, sgn * v);
    };

    for (int i = t; i < 2 * RB; i += kThreads) rhsbuf[i] = 0.0;     // padding columns must read as 0
    for (int i = t; i < n; i += kThreads) xs_full[i] = 0.0;
    const double qv = active ? sm[S::o_q + r] : 0.0;
    double x = (active && a.warm_u) ? a.warm_u[(size_t)b * n + 3 * gc + ax] / a.s : 0.0;
    double cpx = active ? sm[S::o_px0 + r] : 0.0;       // P x = cpx - A' spx
    double spx = 0.0;
    double y = (has_row && a.warm_y) ? a.warm_y[(size_t)b * m + irow] : 0.0;
    double axr = rowm * Arow(x);                        // (A x)_row, carried by recursion
    double z = rowm * fmin(fmax(axr, lo), hi);
    double qn[1] = {fabs(qv)};
    block_max<1>(qn, sm + S::o_red);                    // its barriers also order the zero fill above
    {
        const double rhs0 = sigma * x - qv + At(rowm * (rho * z - y));
        if (active && hp == 0) rhsbuf[rpos] = rhs0;
    }
    if (t < 4) reinterpret_cast<int*>(redf + 32)[t] = 0;
    __syncthreads();

    int status = 2, iters = a.max_iter, nchk = 0;
    long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0, tp5 = 0, seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0, seg4 = 0;
    (void)tp0; (void)tp1; (void)tp2; (void)tp3; (void)tp4; (void)tp5; (void)seg0; (void)seg1; (void)seg2; (void)seg3; (void)seg4;
    bool pending = false;                               // a check's maxima are waiting in redf[(nchk - 1) & 1]
    double e_prim_last = kInf * 1.0e10;                 // e_prim of the last full check (pre-test threshold)
    int* vflag = reinterpret_cast<int*>(redf + 32);     // [4] per-wave 'some row fails the pre-test' flags
    bool vote_ok = true;
    int ph = 0;
    float lastv0 = 0.0f, lastv1 = 0.0f, lastv2 = 0.0f, lastv3 = 0.0f;   // maxima of the last full check (restart rule)
    for (int k = 1; k <= a.max_iter + 1; ++k) {
        if (pending) {   // decision of the check made at iteration k - 1 (its maxima rode on that iteration's barrier)
            const float* buf = redf + ((nchk - 1) & 1) * 16;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (double)fmaxf(fmaxf(buf[q], buf[4 + q]), fmaxf(buf[8 + q], buf[12 + q]));
            const double e_prim = a.eps_abs + a.eps_rel * v[1];
            const double e_dual = a.eps_abs + a.eps_rel * fmax(v[3], (double)(float)qn[0]);
            e_prim_last = e_prim;
            lastv0 = (float)v[0]; lastv1 = (float)v[1]; lastv2 = (float)v[2]; lastv3 = (float)fmax(v[3], (double)(float)qn[0]);
            if (!(v[0] <= kInf) || !(v[2] <= kInf)) { status = -1; iters = k - 1; break; }
            if (v[0] <= e_prim && v[2] <= e_dual) { status = 1; iters = k - 1; break; }
            pending = false;
        }
        if (k > a.max_iter) break;
        if (++ph == a.check_every) ph = 0;                  // ph = k mod check_every, without a division per iteration
        const bool at_mark = (ph == 0);
        if (at_mark) vote_ok = (vflag[0] | vflag[1] | vflag[2] | vflag[3]) == 0;   // pre-test made at iteration k - 1
        const bool check = (at_mark && vote_ok) || (k == a.max_iter);
        const bool pretest = (ph == a.check_every - 1);     // k + 1 is a mark
        const double* rb = rhsbuf + ((k - 1) & 1) * RB;
        double* wb = rhsbuf + (k & 1) * RB;
        ADMM_T(tp0);
        if (wave_on) {
            double xt;
            {
                // rhs in blocks of 8 ds_read_b128, all in flight before the FMAs that consume them (a bigger block spills:
                // kin + 16 double2 exceed the 168-VGPR budget of 3 workgroups/CU, and a spilled kin value costs a scratch
                // round trip EVERY iteration).  Columns past CH read the zero padding and meet kin = 0.
                constexpr int NV = CHMAX / 2, BL = (NV <= 8) ? NV : 8, NBMAX = (NV + BL - 1) / BL;   // 8 ds_read_b128 in flight
                double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
                const double2* rv = reinterpret_cast<const double2*>(rb + CHS * hp);
#pragma unroll
                for (int blk = 0; blk < NBMAX; ++blk) {
                    if (blk == 0 || blk * 2 * BL < CH) {                       // wave-uniform
                        double2 v[BL];
#pragma unroll
                        for (int i = 0; i < BL; ++i) v[i] = (blk * BL + i < NV) ? rv[blk * BL + i] : make_double2(0.0, 0.0);
#pragma unroll
                        for (int i = 0; i < BL; ++i) {
                            const int c0 = 2 * (blk * BL + i);
                            if (c0 + 1 < CHMAX) {
                                if (i & 1) { acc2 = fma(kin[c0], v[i].x, acc2); acc3 = fma(kin[c0 + 1], v[i].y, acc3); }
                                else { acc0 = fma(kin[c0], v[i].x, acc0); acc1 = fma(kin[c0 + 1], v[i].y, acc1); }
                            }
                        }
                    }
                }
                const double acc = (acc0 + acc1) + (acc2 + acc3);
                xt = acc + dpp_swap1(acc);
                if constexpr (LPR == 4) xt += dpp_swap2(xt);
            }
            ADMM_T(tp1);
            const double zt = Arow(xt);
            const double nu = rowm * (rho * (zt - z) + y);
            const double zh = alpha * zt + oma * z;
            const double zn = fmin(fmax(zh + y * irho, lo), hi);
            y = rowm * (y + rho * (zh - zn));
            z = rowm * zn;
            axr = rowm * (alpha * zt + oma * axr);          // A x^{k+1} = alpha A x~ + (1 - alpha) A x^k
            spx = alpha * nu + oma * spx;
            ADMM_T(tp2);
            const double atw = At(rho * z - y);
            cpx = alpha * (sigma * (x - xt) - qv) + oma * cpx;
            x = alpha * xt + oma * x;
            if (active && hp == 0) wb[rpos] = sigma * x - qv + atw;
            ADMM_T(tp3);
            if (pretest) {   // one ballot instead of a reduction: does any row still violate the last e_prim?
                const unsigned long long bad = __ballot(has_row && !(fabs(axr - z) <= e_prim_last));
                if (lane == 0) vflag[w] = (bad != 0ull) ? 1 : 0;
            }
            if (check) {
                const double aty = At(y), px = cpx - At(spx);
                double rd = fabs(px + qv + aty), rp = fabs(axr - z);
                rd = (rd == rd) ? rd : kInf * 10.0;              // a NaN residual must survive the max
                rp = (rp == rp) ? rp : kInf * 10.0;
                const float v0 = (float)(rowm * rp), v1 = (float)(rowm * fmax(fabs(axr), fabs(z)));
                const bool varlane = active && prim;            // lanes on which A'(.) is the variable's value
                const float v2 = varlane ? (float)rd : 0.0f, v3 = varlane ? (float)fmax(fabs(px), fabs(aty)) : 0.0f;
                float m0 = v0, m1 = v1, m2 = v2, m3 = v3;
                wave_maxf4_nonneg(m0, m1, m2, m3);
                if (lane == 63) {
                    float* buf = redf + (nchk & 1) * 16 + 4 * w;
                    buf[0] = m0; buf[1] = m1; buf[2] = m2; buf[3] = m3;
                }
            }
        } else {                                              // idle waves contribute zeros / 'no objection'
            if (check && lane == 63) {
                float* buf = redf + (nchk & 1) * 16 + 4 * w;
                buf[0] = 0.0f; buf[1] = 0.0f; buf[2] = 0.0f; buf[3] = 0.0f;
            }
            if (pretest && lane == 0) vflag[w] = 0;
        }
        if (check) { ++nchk; pending = true; }
        ADMM_T(tp4);
        __syncthreads();
        ADMM_T(tp5);
        seg0 += tp1 - tp0; seg1 += tp2 - tp1; seg2 += tp3 - tp2; seg3 += tp4 - tp3; seg4 += tp5 - tp4;
    }
#ifdef SRBDQP_PROFILE_ADMM
    if (a.stamps && t == 0) { long long* st = a.stamps + (size_t)b * 16; st[12] = seg0; st[13] = seg1; st[14] = seg2; st[15] = seg3; st[1] = seg4; }
#endif
    if (status < 0) { x = 0.0; y = 0.0; }                               // a numerical failure returns zero forces, never NaN
    if (active && hp == 0) xs_full[3 * gc + ax] = x;
    if (a.y_out && has_row && (!a.y_capped_only || status == 2)) a.y_out[(size_t)b * m + irow] = y;
    if (a.resid_out && status == 2 && t == 0) {
        float* ro = a.resid_out + (size_t)b * 4;
        ro[0] = lastv0; ro[1] = lastv1; ro[2] = lastv2; ro[3] = lastv3;
        if (a.cap_list) a.cap_list[atomicAdd(a.cap_count, 1)] = b;
    }
    __syncthreads();
    *status_out = status;
    return iters;
}

// Split mode (SPLIT = true): the kernel stops after K^-1 and hands the QP to srbdqp_admm_kernel (srbdqp_split.hpp)
// through a per-QP workspace in HBM: [0, o_R) = the persistent LDS strip as is, then K^-1 dense row-major
// (n_eff rows of SplitWs::KS doubles).  strip[o_misc + 1] != 0 tells the second kernel the QP is already finished.
template <int N, int MAXS>
struct SplitWs {
    using S = CompactSmem<N, MAXS>;
    static constexpr bool supported = (S::nmax <= 64);     // one K^-1 row per lane of ONE wave
    static constexpr int KS = (S::nmax + 1) & ~1;          // row stride (even: 16-byte loads)
    static constexpr int o_kinv = (S::o_R + 1) & ~1;
    static constexpr int o_phi = o_kinv + S::nmax * KS;    // two-phase call: dq/dx0, [nmax][13] (+1 pad)
    static constexpr int doubles = o_phi + ((S::nmax * 13 + 1) & ~1);
};

// Assembly dump shared by the compact and the one-wave kernels (mode 1): the reduced-KKT matrix K = P + sigma I + A' rho A
// of the PRESOLVED QP exactly as the kernel holds it before its factorisation (upper tiles, C layout -> `tiles(put)`
// enumerates (row, col, value)), written dense at P_out with row stride 12N (both triangles), the compact gradient at
// q_out, the compaction map (compact contact -> original contact 4 k + i) at l_out and na at ub_out[0]; and the bounds of the
// constraint rows exactly as the ADMM loops of these kernels form them (admm_loop_compact / admm_wave_body: friction rows
// (-kInf, 0], normal-force row [fzmin_s, fzmax_s] from the same KArgs fields): l_out[4N + e], l_out[8N + e] = bounds of the
// normal-force row of compact contact e, l_out[12N], l_out[12N + 1] = bounds of a friction row.
template <int N, class Tiles>
__device__ __forceinline__ void dump_presolved(const KArgs& a, int b, int n_eff, int na, const double* qc, const uint8_t* act, Tiles&& tiles) {
    constexpr int n = 12 * N, m = 20 * N;
    double* P = a.P_out + (size_t)b * n * n;
    tiles([&](int r, int c, double v) {
        if (r < n_eff && c < n_eff) { P[(size_t)r * n + c] = v; P[(size_t)c * n + r] = v; }
    });
    for (int c = threadIdx.x; c < n_eff; c += blockDim.x) a.q_out[(size_t)b * n + c] = qc[c];
    for (int e = threadIdx.x; e < na; e += blockDim.x) {
        a.l_out[(size_t)b * m + e] = (double)act[e];
        const int j = 4;                                                   // row 4 of a contact: the normal-force row
        a.l_out[(size_t)b * m + 4 * N + e] = (j < 4) ? -kInf : a.fzmin_s;   // (the expressions of admm_loop_compact())
        a.l_out[(size_t)b * m + 8 * N + e] = (j < 4) ? 0.0 : a.fzmax_s;
    }
    if (threadIdx.x == 0) {
        a.ub_out[(size_t)b * m] = (double)na;
        const int j = 0;                                                   // rows 0..3: the friction pyramid
        a.l_out[(size_t)b * m + 12 * N] = (j < 4) ? -kInf : a.fzmin_s;
        a.l_out[(size_t)b * m + 12 * N + 1] = (j < 4) ? 0.0 : a.fzmax_s;
    }
}

// One QP (index b) on one 256-thread workgroup; sm = the workgroup's dynamic LDS (CompactSmem<N, MAXS>::bytes).
// Every exit is workgroup-uniform and leaves no state in LDS that a later call would rely on.
template <int N, int MAXS> struct SplitWs;
struct WaveRestart;
template <int N, int MAXS, bool RST>
__device__ __forceinline__ void admm_wave_iterations(const KArgs& a, const QpIo& io, const double* warm_u, const double* warm_y, int b, double rho_b, double* sm,
                                                     const double (&kin)[SplitWs<N, MAXS>::KS], double* xs_full, int& status_out, int& iters_out,
                                                     WaveRestart* rs);   // srbdqp_split.hpp

// TAIL1 (the staged batch-1 instantiation, compiled for one workgroup's worth of registers): the four waves set the problem up, then wave 0 alone runs the
// one-wave iteration (a K^-1 row per lane, no LDS operation and no barrier in the loop: 0.36 us per iteration against 0.52 for the 4-wave loop at batch 1,
// tools/batch1_kernel_probe.py) while the others wait at the barrier in front of the roll-out, which all four share again.
// roll-out, stores and the completion word of a QP whose status / iteration count thread 0 has already stored; CS (the batch-1 instantiations): with the checksum of
// the outputs instead of a system-scope fence when the launch asks for it (KArgs::done_cs, srbdqp_common.hpp signal_done_checksum)
template <int N, class S, bool CS>
__device__ __forceinline__ void rollout_store_signal(const KArgs& a, int b, double* sm, int status, int iters_total) {
    if constexpr (CS) {
        if (a.done_cs) {
            unsigned long long cs = 0;
            rollout_and_store_to<N, S, kThreads, true>(a, a.u_out, a.x_out, b, sm, sm + S::o_xs, sm + S::o_rhs, &cs);
            if (threadIdx.x == 0) cs ^= done_cs_pack(status, iters_total);
            signal_done_checksum<kThreads>(a.done_flag, a.done_value, cs);
            return;
        }
    }
    rollout_and_store<N, S>(a, b, sm, sm + S::o_xs, sm + S::o_rhs);
    signal_done(a);
}

template <int N, int MAXS, bool SPLIT = false, bool DUMP = false, bool TAIL1 = false>
__device__ __forceinline__ void compact_qp(const KArgs& a, const int b, double* sm) {
    using S = CompactSmem<N, MAXS>;
    constexpr int n = Dims<N>::n, m = Dims<N>::m;
    constexpr int TS = S::TS, CHMAX = S::CHMAX;
    static_assert(S::NT <= 8, "W phase assumes at most two tiles per wave per block row");
    static_assert(2 * S::nmax <= kThreads, "two threads per compact column / row");
    static_assert(4 * N <= 128, "the presolve compacts the 4N contact flags with at most two wave-wide ballots");
    static_assert((S::o_R % 2) == 0 && (S::o_rhs % 2) == 0, "16-byte alignment");
    const double rho_b = SRBDQP_RHO_OF(a, b);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    int mcol = lane & 15, kq = lane >> 4;                   // re-laundered per phase, see SRBDQP_PHASE_LOCAL
    double* T = sm + S::o_T;
    int* icnt = reinterpret_cast<int*>(sm + S::o_int);          // cnt[i] = stance contacts in steps 0..i
    int* imisc = icnt + 2 * N;                                  // [0] na, [1] bound violated
    uint8_t* act = reinterpret_cast<uint8_t*>(imisc + 8);       // compact contact -> original contact

    // ================= phase A =================
    SRBDQP_STAMP(a, b, 0);
#ifndef SRBDQP_PROFILE_ADMM
    if (a.stamps && t == 0) a.stamps[(size_t)b * 16 + 12] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    load_and_linearise<N, S>(a, b, sm);
    if constexpr (4 * N <= 64) {
        if (t < 64) {   // presolve: compact the stance contacts (wave 0)
            const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
            const bool flag = (t < 4 * N) && sct[t < 4 * N ? t : 0] != 0;
            const unsigned long long bal = __ballot(flag);
            if (flag) act[__popcll(bal & ((1ull << t) - 1ull))] = (uint8_t)t;
            if (t < N) icnt[t] = __popcll(bal & ((4 * (t + 1) >= 64) ? ~0ull : ((1ull << (4 * (t + 1))) - 1ull)));
            if (t == 0) {
                imisc[0] = __popcll(bal);
                sm[S::o_misc] = 0.0;
                sm[S::o_misc + 1] = 0.0;
            }
        }
        __syncthreads();
    } else {            // 64 < 4N <= 128 flags: one ballot in each of waves 0 and 1, joined through LDS
        unsigned long long* bals = reinterpret_cast<unsigned long long*>(imisc + 2);
        const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
        const bool flag = (t < 4 * N) && sct[t < 4 * N ? t : 0] != 0;
        if (t < 128) {
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) bals[w] = bal;
        }
        __syncthreads();
        if (t < 128) {
            const unsigned long long b0 = bals[0], b1 = bals[1];
            const unsigned long long below = (1ull << lane) - 1ull;
            if (flag) act[(w ? __popcll(b0) : 0) + __popcll((w ? b1 : b0) & below)] = (uint8_t)t;
            if (t < N) {
                const int end = 4 * (t + 1);
                icnt[t] = (end < 64) ? __popcll(b0 & ((1ull << end) - 1ull))
                        : (end == 64) ? __popcll(b0)
                        : __popcll(b0) + __popcll(b1 & ((end - 64 >= 64) ? ~0ull : ((1ull << (end - 64)) - 1ull)));
            }
            if (t == 0) {
                imisc[0] = __popcll(b0) + __popcll(b1);
                sm[S::o_misc] = 0.0;
                sm[S::o_misc + 1] = 0.0;
            }
        }
        __syncthreads();
    }
    if (t == 0) {   // the per-step bound check
        int viol = 0;
        for (int i = 0; i < N; ++i) {
            const int ci = icnt[i] - (i ? icnt[i - 1] : 0);
            viol |= (ci > MAXS);
        }
        imisc[1] = viol;
    }
    __syncthreads();
    const int na = imisc[0];
    const int n_eff = 3 * na;
    if (imisc[1] != 0 || na == 0) {   // bound violated (status -2) or nothing to solve (all forces 0)
        if constexpr (DUMP) { if (t == 0) a.ub_out[(size_t)b * m] = (imisc[1] != 0) ? -1.0 : 0.0; return; }   // assembly dump: nothing to show
        for (int c = t; c < n; c += kThreads) sm[S::o_xs + c] = 0.0;
        if (a.y_out) for (int i = t; i < m; i += kThreads) a.y_out[(size_t)b * m + i] = 0.0;
        if (t == 0) {
            if (a.status) a.status[b] = (imisc[1] != 0) ? kStatusContactBound : 1;
            if (a.iters) a.iters[b] = 0;
        }
        __syncthreads();
        if constexpr (TAIL1) { rollout_store_signal<N, S, true>(a, b, sm, (imisc[1] != 0) ? kStatusContactBound : 1, 0); return; }
        rollout_and_store<N, S>(a, b, sm, sm + S::o_xs, sm + S::o_rhs);
        if constexpr (SPLIT) { if (t == 0) a.ws[(size_t)b * SplitWs<N, MAXS>::doubles + S::o_misc + 1] = 1.0; }
        signal_done(a);
        return;
    }
    SRBDQP_STAMP(a, b, 1);
    // ---- a6 + a7 in closed form.  G = Q^1/2 s B_qp is never materialised: with D_im = C_i - C_m the blocks of B_qp are
    // theta: dt^2 D_ij J_j, p: (i-j) dt^2/m I, omega: dt J_j, v: dt/m I (bqp_entry()), so for contacts e (step j) and
    // e' (step m >= j)
    //   (G'G)[e,a][e',a'] = s^2 ( J_e[:,a]' M(j,m) J_e'[:,a'] + [a = a'] (w_p[a] dt^4/m^2 Sp(j,m) + w_v[a] dt^2/m^2 (N-m)) ),
    //   M(j,m) = dt^4 (T2(m) + (C_m - C_j)' W_th T1(m)) + (N-m) dt^2 W_om,   T1(m) = sum_{i>=m} D_im,
    //   T2(m) = sum_{i>=m} D_im' W_th D_im,   Sp(j,m) = sum_{i>=m} (i-j)(i-m),
    // and G'v for a stacked vector v is a suffix sum per step followed by one 3-vector product per contact.  The
    // differences D are formed before anything is multiplied, as in the dense products of the oracle (no cancellation).
    const double dt = a.dt, dt2 = a.dt * a.dt, dtm = a.dt * a.inv_mass, dt2m = dt2 * a.inv_mass;
    double* T1 = sm + S::o_t1;
    double* T2 = sm + S::o_t2;
    double* DE = sm + S::o_mt;
    double* GV = sm + S::o_gv;
    const double* CP = sm + S::o_cp;
    const double* SQ = sm + S::o_sq;
    for (int k = t; k < n; k += kThreads) {   // Q^1/2 (A_qp x0 - x_ref), all 12 N rows
        const int i = k / 12, kk = k - 12 * i;
        sm[S::o_eh + k] = SQ[kk] * (free_response<N, S>(a, sm, i, kk) - sm[S::o_xref + i * 13 + kk]);
    }
    if (t < 9 * N) {
        const int mm = t / 9, pq = t - 9 * mm, p = pq / 3, q = pq - 3 * p;
        const double* Cm = CP + mm * 9;
        const double w0 = SQ[0] * SQ[0], w1 = SQ[1] * SQ[1], w2 = SQ[2] * SQ[2];
        const double m0p = Cm[p], m1p = Cm[3 + p], m2p = Cm[6 + p], m0q = Cm[q], m1q = Cm[3 + q], m2q = Cm[6 + q], mpq = Cm[pq];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {   // unrolled with a predicate: all the LDS reads are in flight together
            const double* Ci = CP + i * 9;
            const double on = (i >= mm) ? 1.0 : 0.0;
            const double d0p = Ci[p] - m0p, d1p = Ci[3 + p] - m1p, d2p = Ci[6 + p] - m2p;
            const double d0q = Ci[q] - m0q, d1q = Ci[3 + q] - m1q, d2q = Ci[6 + q] - m2q;
            s1 = fma(on, Ci[pq] - mpq, s1);
            s2 = fma(on, (w0 * d0p) * d0q + (w1 * d1p) * d1q + (w2 * d2p) * d2q, s2);
        }
        T1[t] = s1;
        T2[t] = s2;
    }
    __syncthreads();
    // G'v tables: per step j the 3-vector g (theta + omega rows) and the per-axis sums of the p and v rows
    auto gt_tables = [&](const double* vec) {
        // torque entries (O(N) loop of 7 LDS reads) on the first wave(s), force entries (1 read) on the following ones
        constexpr int GT_HL = 64 * ((3 * N + 63) / 64);
        static_assert(GT_HL + 6 * N <= kThreads, "one pass over the G'v tables");
        if (t < 3 * N) {
            const int j = t / 3, comp = t - 3 * j;
            const double* Cj = CP + j * 9;
            double acc = 0.0;
            const double c0 = Cj[comp], c1 = Cj[3 + comp], c2 = Cj[6 + comp];
            const double q0 = SQ[0] * dt2, q1 = SQ[1] * dt2, q2 = SQ[2] * dt2, qw = SQ[6 + comp] * dt;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double* Ci = CP + i * 9;
                const double* v = vec + 12 * i;
                const double on = (i >= j) ? 1.0 : 0.0;
                acc = fma(on, (Ci[comp] - c0) * (q0 * v[0]) + (Ci[3 + comp] - c1) * (q1 * v[1]) + (Ci[6 + comp] - c2) * (q2 * v[2]) + qw * v[6 + comp], acc);
            }
            GV[9 * j + comp] = acc;
        } else if (t >= GT_HL && t < GT_HL + 6 * N) {
            const int e = t - GT_HL, j = e / 6, comp = 3 + (e - 6 * j);
            const int kk = (comp < 6) ? comp : 3 + comp;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double wgt = (i >= j) ? ((comp < 6) ? (double)(i - j) : 1.0) : 0.0;
                acc = fma(wgt, vec[12 * i + kk], acc);
            }
            GV[9 * j + comp] = acc;
        }
    };
    auto gt_eval = [&](int c) -> double {   // (G'v)[c] for compact variable c, from the tables
        const int e = c / 3, ax = c - 3 * e, gc = act[e], j = gc >> 2;
        const double* J = sm + S::o_J + j * 36 + 3 * (gc & 3) + ax;
        const double* g = GV + 9 * j;
        return a.s * (J[0] * g[0] + J[12] * g[1] + J[24] * g[2] + SQ[3 + ax] * dt2m * g[3 + ax] + SQ[9 + ax] * dtm * g[6 + ax]);
    };
    gt_tables(sm + S::o_eh);
    de_tables<N>(CP, T1, T2, SQ, dt2, DE, t, kThreads);
    __syncthreads();
    SRBDQP_STAMP(a, b, 2);
    for (int c = t; c < n_eff; c += kThreads) sm[S::o_q + c] = gt_eval(c);
    if (a.warm_u) {   // P x^0 = G'(G x^0) + R s^2 x^0 on the compact variables
        double* TF = sm + S::o_tf;
        for (int c = t; c < n_eff; c += kThreads)
            sm[S::o_x0c + c] = a.warm_u[(size_t)b * n + 3 * act[c / 3] + (c % 3)] / a.s;
        __syncthreads();
        if (t < 6 * N) {   // per step: tau_j = sum_e J_e x_e (3), f_j = sum_e x_e (3)
            const int j = t / 6, comp = t - 6 * j;
            double acc = 0.0;
            for (int e = (j ? icnt[j - 1] : 0); e < icnt[j]; ++e) {
                const double* x = sm + S::o_x0c + 3 * e;
                if (comp < 3) {
                    const double* J = sm + S::o_J + j * 36 + comp * 12 + 3 * (act[e] & 3);
                    acc += J[0] * x[0] + J[1] * x[1] + J[2] * x[2];
                } else {
                    acc += x[comp - 3];
                }
            }
            TF[t] = acc;
        }
        __syncthreads();
        for (int k = t; k < n; k += kThreads) {   // G x^0, row kk of step i
            const int i = k / 12, kk = k - 12 * i;
            const double acc = gx_row<N>(CP, TF, i, kk, dt, dt2, dtm, dt2m);
            sm[S::o_gx + k] = SQ[kk] * a.s * acc;
        }
        __syncthreads();
        gt_tables(sm + S::o_gx);
        __syncthreads();
        for (int c = t; c < n_eff; c += kThreads) sm[S::o_px0 + c] = gt_eval(c) + a.rs2 * sm[S::o_x0c + c];
    } else {
        for (int c = t; c < n_eff; c += kThreads) sm[S::o_px0 + c] = 0.0;
    }
    SRBDQP_STAMP(a, b, 3);

    // run-time tile geometry
    const int NT = (n_eff + 15) >> 4;
    const int NTT = (NT * (NT + 1)) >> 1;
    int ta[TS], tb[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int id = 4 * s + w;
        int bb = 0;
        while (((bb + 1) * (bb + 2)) / 2 <= id) ++bb;
        tb[s] = (id < NTT) ? bb : -1;
        ta[s] = (id < NTT) ? id - (bb * (bb + 1)) / 2 : -1;
    }

    // ================= phase H: K = G'G + R s^2 + sigma + A' rho A, entry by entry into the C-layout register tiles ====
    v4d acc[TS];
    {   // the rank-6 form (srbdqp_common.hpp): a table row per presolved variable, then every lane its own C-layout entries
        double* AB = sm + S::o_ab;
        kasm_rows<N>(sm + S::o_J, CP, DE, SQ, act, n_eff, a.s, dt2m, dtm, AB, t, kThreads, 16 * NT);
        __syncthreads();
        const double dgxy = a.rs2 + a.sigma + 2.0 * rho_b, dgz = a.rs2 + a.sigma + (4.0 * a.mu * a.mu + a.rho_fz) * rho_b;
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (ta[s] >= 0) {
                double o[4];
                kasm_tile(AB, ta[s], tb[s], mcol, kq, n_eff, dgxy, dgz, o);
                acc[s] = (v4d){o[0], o[1], o[2], o[3]};
            }
        }
    }
    SRBDQP_STAMP(a, b, 4);
    __syncthreads();
    SRBDQP_STAMP(a, b, 5);
    if constexpr (DUMP) {   // assembly dump (srbdqp_assemble_f64): K of the presolved QP, its gradient, the compaction map
        dump_presolved<N>(a, b, n_eff, na, sm + S::o_q, act, [&](auto&& put) {
#pragma unroll
            for (int s = 0; s < TS; ++s)
                if (ta[s] >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (16 * ta[s] + kq + 4 * q <= 16 * tb[s] + mcol) put(16 * ta[s] + kq + 4 * q, 16 * tb[s] + mcol, acc[s][q]);   // (the upper triangle: put() mirrors it)
                }
        });
        return;
    }

    // ================= phase F =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
    for (int j = 0; j < NT; ++j) {
        double* Djj = T + tile_id(j, j) * 256;
        {   // the wave that holds the diagonal tile inverts it in registers (blocked, on the matrix cores)
            bool mine = false;
            v4d d = acc[0];
#pragma unroll
            for (int s = 0; s < TS; ++s)
                if (ta[s] == j && tb[s] == j) { mine = true; d = acc[s]; }
            if (mine) {
                int lane_j = lane;
                SRBDQP_PHASE_LOCAL("+v"(lane_j));           // keeps the lane predicates of the inversion inside this block
                bool ok;
                if constexpr (TAIL1) {   // (batch 1, a workgroup's worth of registers: one column per lane, DPP multiply-adds -- 2.9 k against 3.3 k cycles a tile, srbdqp_mfma.hpp)
                    store_tile<false>(Djj, d, lane_j);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wave: its LDS operations complete in order
                    diag16_invert_dpp_tiles<double, double>(Djj, Djj, lane_j, ok);
                } else {
                    const v4d winv = diag16_invert_mfma(d, lane_j, ok);
                    store_tile<true>(Djj, winv, lane_j);
                }
                if (!ok && lane == 0) sm[S::o_misc] = 1.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            if (ta[s] == j && tb[s] > j) {
                v4d o = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    double av = Djj[mcol * 16 + (k ^ mcol)];
                    av = (k <= mcol) ? av : 0.0;
                    o = mfma_f64(av, acc[s][r], o);
                }
                acc[s] = o;
                store_tile<false>(T + tile_id(j, tb[s]) * 256, o, lane);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            if (ta[s] > j) {
                const double* Ua = T + tile_id(j, ta[s]) * 256;
                const double* Ub = T + tile_id(j, tb[s]) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    acc[s] = mfma_f64(-Ua[k * 16 + mcol], Ub[k * 16 + mcol], acc[s]);
                }
            }
        }
    }
    __syncthreads();
    SRBDQP_STAMP(a, b, 6);

    // ================= phase W =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
    for (int i = 1; i < NT; ++i) {
        v4d res[2];
        const double* Dii = T + tile_id(i, i) * 256;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = w + 4 * q;
            res[q] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (j < i) {
                v4d o = (v4d){0.0, 0.0, 0.0, 0.0};
                {
                    const double* Uji = T + tile_id(j, i) * 256;
                    const double* Djj = T + tile_id(j, j) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        double bv = Djj[k * 16 + (mcol ^ k)];
                        bv = (mcol <= k) ? bv : 0.0;
                        o = mfma_f64(Uji[k * 16 + mcol], bv, o);
                    }
                }
                for (int k2 = j + 1; k2 < i; ++k2) {
                    const double* Uki = T + tile_id(k2, i) * 256;
                    const double* Wkj = T + tile_id(j, k2) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        o = mfma_f64(Uki[k * 16 + mcol], Wkj[k * 16 + mcol], o);
                    }
                }
                v4d o2 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    double av = Dii[mcol * 16 + (k ^ mcol)];
                    av = (k <= mcol) ? -av : 0.0;
                    o2 = mfma_f64(av, o[r], o2);
                }
                res[q] = o2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = w + 4 * q;
            if (j < i) store_tile<false>(T + tile_id(j, i) * 256, res[q], lane);
        }
        __syncthreads();
    }
    SRBDQP_STAMP(a, b, 7);

    // ================= phase I =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (ta[s] >= 0) {
            const int ia = ta[s], ib = tb[s];
            const double* Dbb = T + tile_id(ib, ib) * 256;
            {
                const double* Wba = T + tile_id(ia, ib) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    double dv = Dbb[k * 16 + (mcol ^ k)];
                    dv = (mcol <= k) ? dv : 0.0;
                    const double av = (ia < ib) ? Wba[k * 16 + mcol] : dv;
                    acc[s] = mfma_f64(av, dv, acc[s]);
                }
            }
            for (int i = ib + 1; i < NT; ++i) {
                const double* Wia = T + tile_id(ia, i) * 256;
                const double* Wib = T + tile_id(ib, i) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    acc[s] = mfma_f64(Wia[k * 16 + mcol], Wib[k * 16 + mcol], acc[s]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TS; ++s)
        if (ta[s] >= 0) store_tile<true>(T + tile_id(ta[s], tb[s]) * 256, acc[s], lane);
    __syncthreads();
    SRBDQP_STAMP(a, b, 8);

    if constexpr (SPLIT) {   // hand the QP over: persistent strip + dense K^-1 (coalesced stores)
        using W = SplitWs<N, MAXS>;
        double* ws = a.ws + (size_t)b * W::doubles;
        for (int i = t; i < S::o_R; i += kThreads) ws[i] = sm[i];
        for (int idx = t; idx < n_eff * W::KS; idx += kThreads) {
            const int rr = idx / W::KS, c = idx - rr * W::KS;
            const int cs = (c < n_eff) ? c : 0;
            const int lo = (rr <= cs) ? rr : cs, hi = (rr <= cs) ? cs : rr;
            const int row = lo & 15, col = hi & 15;
            const double v = T[tile_id(lo >> 4, hi >> 4) * 256 + row * 16 + (col ^ row)];
            ws[W::o_kinv + idx] = (c < n_eff) ? v : 0.0;
        }
        return;
    } else {

    if constexpr (TAIL1) {
        using W1 = SplitWs<N, MAXS>;
        constexpr int KS1 = W1::KS;
        // K^-1 from the swizzled upper-triangular tile store to plain rows behind the kernel's own LDS (KS1 x (KS1 + 2) doubles, part of the launch's dynamic
        // size): every wave moves 16 columns of every row, then wave 0 reads its lanes' rows with 16-byte loads.  (One wave pulling 64 entries per lane through
        // the tile indexing took 5.1 k cycles of a 98 k-cycle solve.)
        constexpr int RS1 = KS1 + 2;
        static_assert(KS1 % 4 == 0 && KS1 <= 64, "TAIL1: a quarter of the columns per wave, one row per lane");
        double* full = sm + S::o_end;
        {
            const bool rowok = lane < n_eff;
            const int rr = rowok ? lane : 0;
            constexpr int CW = KS1 / 4;                                 // columns per wave
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) {
                const int c = CW * w + cc;
                const bool ok = rowok && (c < n_eff);
                const int cs = ok ? c : 0;
                const int lo = (rr <= cs) ? rr : cs, hi = (rr <= cs) ? cs : rr;
                const int row = lo & 15, col = hi & 15;
                const double v = T[tile_id(lo >> 4, hi >> 4) * 256 + row * 16 + (col ^ row)];
                full[lane * RS1 + c] = ok ? v : 0.0;
            }
        }
        const bool failed = sm[S::o_misc] != 0.0;
        __syncthreads();   // tiles are dead; region R becomes the solution vector and the roll-out's scratch
        double kin1[KS1];
        if (w == 0) {
            const double2* src = reinterpret_cast<const double2*>(full + lane * RS1);
#pragma unroll
            for (int c = 0; c < KS1 / 2; ++c) { const double2 v = src[c]; kin1[2 * c] = v.x; kin1[2 * c + 1] = v.y; }
        }
        SRBDQP_STAMP(a, b, 9);
        int status = -1, iters = 0;
        int* fin = reinterpret_cast<int*>(sm + S::o_red);
        if (!failed) {
            if (w == 0) {
                admm_wave_iterations<N, MAXS, false>(a, io_of(a), a.warm_u, a.warm_y, b, rho_b, sm, kin1, sm + S::o_xs, status, iters, nullptr);
                if (lane == 0) { fin[0] = status; fin[1] = iters; }
            }
            __syncthreads();
            status = fin[0]; iters = fin[1];
        } else {
            for (int c = t; c < n; c += kThreads) sm[S::o_xs + c] = 0.0;
            __syncthreads();
        }
        if (a.y_out && (!a.y_capped_only || status == 2)) {
            const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
            for (int i = t; i < m; i += kThreads)
                if (failed || sct[i / 5] == 0) a.y_out[(size_t)b * m + i] = 0.0;
        }
        if (t == 0) {
            if (a.status) a.status[b] = status;
            if (a.iters) a.iters[b] = iters + a.iters_base;
        }
        SRBDQP_STAMP(a, b, 10);
        rollout_store_signal<N, S, true>(a, b, sm, status, iters + a.iters_base);
        SRBDQP_STAMP(a, b, 11);
        if (a.stamps && t == 0) a.stamps[(size_t)b * 16 + 13] = (long long)__builtin_amdgcn_s_memrealtime();
        return;
    }
    // K^-1 row fragments in the compact contact-local mapping
    constexpr int LPR = S::LPR;
    const int CH = 2 * ((n_eff + 2 * LPR - 1) / (2 * LPR));
    double kin[CHMAX];
    {
        const int lr = lane / LPR, h = lane % LPR;
        const int CPW = admm_contacts_per_wave<LPR>(na);
        const int cg = lr / 3, e = w * CPW + cg;
        const bool rowok = (cg < CPW) && (e < na);
        const int rr = rowok ? 3 * e + (lr - 3 * cg) : 0;
#pragma unroll
        for (int cc = 0; cc < CHMAX; ++cc) {
            const int c = CH * h + cc;
            const bool ok = rowok && (cc < CH) && (c < n_eff);
            const int cs = ok ? c : 0;
            const int lo = (rr <= cs) ? rr : cs, hi = (rr <= cs) ? cs : rr;
            const int row = lo & 15, col = hi & 15;
            const double v = T[tile_id(lo >> 4, hi >> 4) * 256 + row * 16 + (col ^ row)];
            kin[cc] = ok ? v : 0.0;
        }
    }
    const bool failed = sm[S::o_misc] != 0.0;
    __syncthreads();   // tiles are dead; region R becomes the ADMM vectors

    SRBDQP_STAMP(a, b, 9);
    int status = -1, iters = 0;
    if (!failed) iters = admm_loop_compact<N, S, CHMAX>(a, b, sm, sm + S::o_rhs, sm + S::o_xs, kin, na, CH, act, rho_b, &status);
    else {
        for (int c = t; c < n; c += kThreads) sm[S::o_xs + c] = 0.0;
        __syncthreads();
    }
    if (a.y_out && (!a.y_capped_only || status == 2)) {   // rows of eliminated (swing) contacts, or of a failed solve: 0
        const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
        for (int i = t; i < m; i += kThreads)
            if (failed || sct[i / 5] == 0) a.y_out[(size_t)b * m + i] = 0.0;
    }
    if (t == 0) {
        if (a.status) a.status[b] = status;
        if (a.iters) a.iters[b] = iters + a.iters_base;
    }
    SRBDQP_STAMP(a, b, 10);
    rollout_and_store<N, S>(a, b, sm, sm + S::o_xs, sm + S::o_rhs);
    signal_done(a);
    SRBDQP_STAMP(a, b, 11);
#ifndef SRBDQP_PROFILE_ADMM
    if (a.stamps && t == 0) a.stamps[(size_t)b * 16 + 13] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    }   // !SPLIT
}

template <int N, int MAXS, bool SPLIT = false, bool DUMP = false, bool TAIL1 = false>
__global__ __launch_bounds__(kThreads, (TAIL1 ? 1 : CompactSmem<N, MAXS>::waves_per_simd)) void srbdqp_compact_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= a.B) return;
    const int b = SRBDQP_QP_INDEX(a);
    if ((a.count_ptr && (int)blockIdx.x >= *a.count_ptr) || SRBDQP_RESTART_SKIP(a, b)) {   // restart pass: not this workgroup's
        signal_done(a);
        return;
    }
    compact_qp<N, MAXS, SPLIT, DUMP, TAIL1>(a, b, sm);
}

// ... the batch-1 instantiation with the QP's inputs in the kernel-argument segment (StagedIn, srbdqp_common.hpp)
template <int N, int MAXS>
__global__ __launch_bounds__(kThreads, 1) void srbdqp_compact_kernel_in(KArgs a, StagedIn<N> in) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    (void)in;                                            // (read through staged_in_base(): a.inline_in is set)
    compact_qp<N, MAXS, false, false, true>(a, 0, sm);
}

template <int N, int MAXS>
struct CompactTraits {
    static constexpr bool supported = (3 * MAXS * N <= 128);
    static constexpr size_t lds_bytes = CompactSmem<N, MAXS>::bytes;
    static constexpr size_t lds_bytes_tail1 = lds_bytes + (size_t)64 * 66 * sizeof(double);   // + K^-1 as plain rows (TAIL1)
};

}  // namespace srbdqp

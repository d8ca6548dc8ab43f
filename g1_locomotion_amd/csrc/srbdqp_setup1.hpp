// srbdqp_setup1.hpp -- set-up kernel of the split pipeline with ONE WAVE PER QP (first kernel; srbdqp_admm_kernel is
// the second).
//
// The 4-wave set-up kernel is latency-bound (61 % of its wave time parked on s_waitcnt / barriers) and holds only 4 QPs
// per CU, because a QP owns 4 waves x 118 registers and 38 KB of LDS (mostly the tile store).  Since the assembly is
// closed form (srbdqp_compact.hpp) nothing needs 256 threads: here the 10 upper 16x16 tiles of K are 80 registers of one
// wave in MFMA C layout and the whole factorisation stays in that wave's registers --
//   F  right-looking Cholesky K = U'U: diagonal tile inverted on the matrix cores (diag16_invert_mfma), panel
//      U_jb = L_jj^-1 K_jb (the A operand L_jj^-1 goes through a 2 KB wave-private LDS tile to be transposed),
//      trailing update K_ab -= U_ja' U_jb with both operands straight from registers (register r of a C-layout tile is
//      the A operand of K-step r of a product that contracts over the tile's row index, and the B operand as well);
//   W  L^-1 block row by block row, in place over the U tiles (W_ij takes the slot of U_ji);
//   I  K^-1 = W'W tile by tile, stored to the hand-over workspace as it is produced --
// with no barrier anywhere (a workgroup is one wave) and 3x the QPs in flight per CU.  Produces exactly the workspace
// (persistent strip + dense K^-1) the 4-wave set-up kernel produces; everything else is shared with it.
#pragma once
#include "srbdqp_split.hpp"

namespace srbdqp {

// phase stamps of the wave kernel only in diagnostic builds (tools/build_variant.sh -DSRBDQP_WAVE_STAMPS): the stamp
// code costs registers even when the stamp buffer is null
#ifdef SRBDQP_WAVE_STAMPS
#define WSTAMP(a, b, idx) SRBDQP_STAMP(a, b, idx)
#define WSTAMP_RT(a, b, idx) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)(b) * 16 + (idx)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(a, b, idx) do { } while (0)
#define WSTAMP_RT(a, b, idx) do { } while (0)
#endif

// LDS of one wave: [0, S::o_R) the persistent strip with the offsets of CompactSmem (the ADMM kernel reads it back from
// the workspace), then the phase-A arrays with their lifetimes shared: inputs -> error vector / warm-start vectors ->
// the transpose tile of the factorisation.
template <int N, int MAXS>
struct Setup1Smem {
    using S = CompactSmem<N, MAXS>;
    static constexpr int n = Dims<N>::n;
    static constexpr int up2(int v) { return (v + 1) & ~1; }
    static constexpr int o_x0 = S::o_x0, o_tm = S::o_tm, o_J = S::o_J;   // strip offsets used by the shared helpers
    static constexpr int o_cp = S::o_R;                    // 9N   prefix sums C_k
    static constexpr int o_mt = o_cp + up2(9 * N);         // 18N  D_m, E_m of the rank-6 assembly (srbdqp_common.hpp, de_tables; until round 4: the 9 NPAIR doubles of M(j, m))
    static constexpr int o_ab = o_mt + 18 * N;             // the assembly's table rows, one per presolved variable: over everything below, which is dead by then
    static constexpr int o_gv = o_mt + up2(9 * S::NPAIR);  // 9N   G'v tables
    static constexpr int o_t1 = o_gv + up2(9 * N);         // 9N   T1(m)          | later: x0c (nmax + 16), tf (6N)
    static constexpr int o_t2 = o_t1 + up2(9 * N);         // 9N   T2(m)
    static constexpr int o_x0c = o_t1;
    static constexpr int o_tf = o_x0c + up2(S::nmax) + 16;
    static constexpr int endT = (o_t2 + up2(9 * N) > o_tf + 6 * N) ? o_t2 + up2(9 * N) : o_tf + 6 * N;
    static constexpr int o_xref = up2(endT);               // 13N  inputs          | later: the 16 x 16 transpose tile
    static constexpr int o_foot = o_xref + up2(N * 13);    // 12N                  | later: eh (n), then G x^0 (n)
    static constexpr int o_pcom = o_foot + N * 12;         // 3N
    static constexpr int o_eh = o_foot;
    static constexpr int o_gx = o_foot;
    static constexpr int o_scr = o_xref;
    static constexpr int endIn = o_pcom + up2(N * 3);
    static constexpr int endIn2 = (endIn > o_scr + 258) ? endIn : o_scr + 258;   // the tile + the spare slot of the K assembly
    static constexpr int kStgRow = 18;                    // row stride of the K^-1 staging tiles (16 + 2: the 16-byte row reads of 16 lanes hit 64 different banks)
    static constexpr int kStgTile = 16 * kStgRow;
    static constexpr int endStg = (endIn2 > o_cp + S::NT * kStgTile) ? endIn2 : o_cp + S::NT * kStgTile;   // K^-1 block-column staging (wave kernel)
    static constexpr int endAb = o_ab + 16 * S::NT * kAbStride;
    static constexpr int o_end = endStg > endAb ? endStg : endAb;
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
    static constexpr bool supported = SplitWs<N, MAXS>::supported && S::NT <= 4;
    static_assert(n + 6 * N <= o_end - o_eh || true, "");
};

// FUSED = false: first kernel of the split pipeline (stores the persistent strip + K^-1 to the hand-over workspace).
// FUSED = true : the whole solve on this wave ("wave" kernel): the K^-1 tiles are turned into one row per lane through
//                the 2 KB transpose tile, then the ADMM iterations and the roll-out of srbdqp_split.hpp follow in place --
//                nothing but the inputs and the outputs touches HBM.
// PHI (split only): also store dq/dx0 for the two-phase call (srbdqp_prepare_staged_f64): q is affine in x0, so 13 more
//                passes of the gradient tables with x0 = e_k and x_ref = 0 give its 13 columns.
// RST (fused only): the rho restart IN PLACE.  A QP that has not converged after a.restart_every iterations re-balances its rho (OSQP's rule from the maxima of its
//                last check, each time from the rho of the pass that ended), repeats the part of the set-up that depends on rho -- the tables of the K assembly,
//                K, its factorisation and inverse, P x of the point it continues from; the inputs, J and the gradient stay -- and continues from its own (x, y), at most a.restart_max times; the
//                cap a.max_iter is on the total.  What a second launch over the capped QPs does for the other kernels (srbdqp.hip), without the launches: 4 % of
//                the configs[1] QPs pass the first mark, 1 % the second.  The body is compiled TWICE for this: the first pass as straight-line code (RP = 1: the
//                code of the kernel without restart), the continued passes as a loop behind it (RP = 2) -- ONE copy inside a loop has hipcc hoist the body's
//                lane-index arithmetic, fp64 constants and literal materialisations in front of the loop and keep them in registers through the whole body
//                (256 VGPRs + 228 bytes of scratch against 224 + 0, -7 % on every QP); this way only the restarted QPs run the loop's code.
//                The (x, y) of a pass wait in 1.5 KB of LDS behind the kernel's own.  oracle: SrbdParams.rho_restart_iter / rho_restart_count.
// One pass of the body.  RP: 0 = the whole solve of a kernel without restart; 1 = first pass of a restart kernel; 2 = a continued pass in the SAME workgroup (the
// strip is in place); 3 = a continued pass in ANOTHER workgroup (deferred tails, srbdqp_wave_defer_kernel: the strip is rebuilt from the inputs, (x, y) of the
// pass before were put into park[] by the caller).  io: the QP's own pointers (RP = 3: those of the launch it came from).  Returns true if another pass follows
// (rho_b, rs_pass, rs_done updated; (x, y) in park[]).
template <int N, int MAXS, bool FUSED, bool DUMP, bool PHI, int RP>
__device__ __forceinline__ bool setup1_pass(const KArgs& a, const QpIo& io, const int b, double* sm, double& rho_b, int& rs_pass, int& rs_done) {
    using S = CompactSmem<N, MAXS>;
    using W = SplitWs<N, MAXS>;
    using L1 = Setup1Smem<N, MAXS>;
    constexpr int n = Dims<N>::n, m = Dims<N>::m, NT = S::NT;
    int lane = threadIdx.x;
    if constexpr (RP == 2) asm volatile("" : "+v"(lane));   // (inside the loop of the continued passes: an opaque value in every trip, or its arithmetic is hoisted too)
    [[maybe_unused]] double* const park = sm + L1::o_end;   // RP > 0: [3][64] behind the kernel's own LDS
    int mcol = lane & 15, kq = lane >> 4;
    double* ws = FUSED ? nullptr : a.ws + (size_t)b * W::doubles;
    int* icnt = reinterpret_cast<int*>(sm + S::o_int);
    int* imisc = icnt + 2 * N;
    uint8_t* act = reinterpret_cast<uint8_t*>(imisc + 8);
    uint8_t* sct = reinterpret_cast<uint8_t*>(sm + S::o_ct);
    const double* SQ = sm + S::o_sq;
    const double* CP = sm + L1::o_cp;

    // ================= load + linearise (a5) =================
    WSTAMP(a, b, 0);
    WSTAMP_RT(a, b, 12);
    if constexpr (RP == 2) {   // the strip (contact lists, T, J, x0, q) is in place; the prefix sums were overwritten by the K^-1 staging
        if (lane < 9) {
            double acc = 0.0;
            for (int k = 0; k < N; ++k) { acc += sm[S::o_tm + k * 9 + lane]; sm[L1::o_cp + k * 9 + lane] = acc; }
        }
        __syncthreads();
    } else {
        const double* gx0 = io.x0 + (size_t)b * 13;
        const double* gxr = io.xref + (size_t)b * N * 13;
        const double* gft = io.foot + (size_t)b * N * 12;
        const uint8_t* gct = io.contact + (size_t)b * N * 4;
        if (lane < 13) sm[S::o_x0 + lane] = gx0[lane];
        if (lane >= 32 && lane < 44) sm[S::o_sq + lane - 32] = a.sqrtq[lane - 32];
        for (int i = lane; i < N * 13; i += 64) sm[L1::o_xref + i] = gxr[i];
        for (int i = lane; i < N * 12; i += 64) sm[L1::o_foot + i] = gft[i];
        if (lane < N * 4) sct[lane] = gct[lane] ? 1 : 0;
        if (io.pcom) {
            const double* gpc = io.pcom + (size_t)b * N * 3;
            if (lane < N * 3) sm[L1::o_pcom + lane] = gpc[lane];
        }
        __syncthreads();
        if (!io.pcom && lane < N * 3) sm[L1::o_pcom + lane] = sm[L1::o_xref + (lane / 3) * 13 + 3 + (lane % 3)];
        if (lane < N) {   // Rz(yaw_k)'
            double sn, cs;
            sincos(sm[L1::o_xref + lane * 13 + 2], &sn, &cs);
            double* T = sm + S::o_tm + lane * 9;
            T[0] = cs;  T[1] = sn;  T[2] = 0.0;
            T[3] = -sn; T[4] = cs;  T[5] = 0.0;
            T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
        }
        {   // presolve: compact the stance contacts
            const bool flag = (lane < 4 * N) && sct[lane < 4 * N ? lane : 0] != 0;
            const unsigned long long bal = __ballot(flag);
            if (flag) act[__popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)lane;
            if (lane < N) icnt[lane] = __popcll(bal & ((4 * (lane + 1) >= 64) ? ~0ull : ((1ull << (4 * (lane + 1))) - 1ull)));
            if (lane == 0) {
                imisc[0] = __popcll(bal);
                sm[S::o_misc] = 0.0;
                sm[S::o_misc + 1] = 0.0;
            }
        }
        __syncthreads();
        if (lane < 9) {   // prefix sums C_k = sum_{l<=k} T_l
            double acc = 0.0;
            for (int k = 0; k < N; ++k) { acc += sm[S::o_tm + k * 9 + lane]; sm[L1::o_cp + k * 9 + lane] = acc; }
        }
        if (lane == 16) {   // per-step bound check
            int viol = 0;
            for (int i = 0; i < N; ++i) viol |= ((icnt[i] - (i ? icnt[i - 1] : 0)) > MAXS);
            imisc[1] = viol;
        }
        for (int tt = lane; tt < N * 12; tt += 64) {   // J_k[:, 3 ci + ax] = Iw^-1 * skew(r)[:, ax]
            const int k = tt / 12, cc = tt % 12, ci = cc / 3, ax = cc % 3;
            const double cs = sm[S::o_tm + k * 9 + 0], sn = sm[S::o_tm + k * 9 + 1];
            const double i0 = a.iinv[0], i1 = a.iinv[1], i2 = a.iinv[2];
            const double w00 = cs * cs * i0 + sn * sn * i1, w01 = cs * sn * (i0 - i1), w11 = sn * sn * i0 + cs * cs * i1;
            const double rx = sm[L1::o_foot + k * 12 + 3 * ci + 0] - sm[L1::o_pcom + k * 3 + 0];
            const double ry = sm[L1::o_foot + k * 12 + 3 * ci + 1] - sm[L1::o_pcom + k * 3 + 1];
            const double rz = sm[L1::o_foot + k * 12 + 3 * ci + 2] - sm[L1::o_pcom + k * 3 + 2];
            // column ax of skew(r) as selects (a three-way if chain on this value was miscompiled in srbdqp_wrench.hpp)
            const double s0 = (ax == 0) ? 0.0 : ((ax == 1) ? -rz : ry);
            const double s1 = (ax == 0) ? rz : ((ax == 1) ? 0.0 : -rx);
            const double s2 = (ax == 0) ? -ry : ((ax == 1) ? rx : 0.0);
            double* J = sm + S::o_J + k * 36;
            J[0 * 12 + cc] = w00 * s0 + w01 * s1;
            J[1 * 12 + cc] = w01 * s0 + w11 * s1;
            J[2 * 12 + cc] = i2 * s2;
        }
        __syncthreads();
    }
    const int na = imisc[0];
    const int n_eff = 3 * na;
    double* xs0 = sm + L1::o_mt;                             // scratch vectors of the early exit (n + 12N doubles; M is not built)
    static_assert(L1::o_end - L1::o_mt >= n + 12 * N, "early-exit scratch");
    if (RP != 2 && (imisc[1] != 0 || na == 0)) {   // bound violated (status -2) or nothing to solve (all forces 0): finished here
        if constexpr (DUMP) { if (lane == 0) a.ub_out[(size_t)b * m] = (imisc[1] != 0) ? -1.0 : 0.0; return false; }   // assembly dump: nothing to show
        for (int c = lane; c < n; c += 64) xs0[c] = 0.0;
        if (io.y_out) for (int i = lane; i < m; i += 64) io.y_out[(size_t)b * m + i] = 0.0;
        if (lane == 0) {
            if (io.status) io.status[b] = (imisc[1] != 0) ? kStatusContactBound : 1;
            if (io.iters) io.iters[b] = a.iters_base;
            if constexpr (!FUSED) ws[S::o_misc + 1] = 1.0;
        }
        __syncthreads();
        rollout_and_store_to<N, S, 64>(a, io.u_out, io.x_out, b, sm, xs0, xs0 + n);
        return false;
    }

    WSTAMP(a, b, 1);
    // ================= closed-form tables, gradient, warm-start P x^0 (see srbdqp_compact.hpp, phase A) =================
    const double dt = a.dt, dt2 = a.dt * a.dt, dtm = a.dt * a.inv_mass, dt2m = dt2 * a.inv_mass;
    double* T1 = sm + L1::o_t1;
    double* T2 = sm + L1::o_t2;
    double* DE = sm + L1::o_mt;
    double* GV = sm + L1::o_gv;
    if constexpr (RP != 2) {
        for (int k = lane; k < n; k += 64) {
            const int i = k / 12, kk = k - 12 * i;
            sm[L1::o_eh + k] = SQ[kk] * (free_response<N, L1>(a, sm, i, kk) - sm[L1::o_xref + i * 13 + kk]);
        }
    }
    // T2(m) is symmetric: one lane per (m, p <= q) -- 6 N entries, one round of the wave at N = 10 instead of two -- which also forms
    // both T1 entries of its pair
    for (int tt = lane; tt < 6 * N; tt += 64) {
        const int mm = tt / 6, u = tt - 6 * mm;
        const int p = (u < 3) ? 0 : ((u < 5) ? 1 : 2), q = (u < 3) ? u : ((u < 5) ? u - 2 : 2);   // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        const int pq = 3 * p + q, qp = 3 * q + p;
        const double* Cm = CP + mm * 9;
        const double w0 = SQ[0] * SQ[0], w1 = SQ[1] * SQ[1], w2 = SQ[2] * SQ[2];
        double s1 = 0.0, s1t = 0.0, s2 = 0.0;
        // (all N steps, the ones before m adding exact zeros: a loop from m has a trip count per lane, is not unrolled, and every trip waits for its own eight LDS
        // reads -- ten round trips; this way the reads of all trips are in flight together)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double* Ci = CP + i * 9;
            const bool on = i >= mm;
            const double a1 = Ci[pq] - Cm[pq], a1t = Ci[qp] - Cm[qp];
            const double d0p = Ci[p] - Cm[p], d1p = Ci[3 + p] - Cm[3 + p], d2p = Ci[6 + p] - Cm[6 + p];
            const double d0q = Ci[q] - Cm[q], d1q = Ci[3 + q] - Cm[3 + q], d2q = Ci[6 + q] - Cm[6 + q];
            const double a2 = (w0 * d0p) * d0q + (w1 * d1p) * d1q + (w2 * d2p) * d2q;
            s1 += on ? a1 : 0.0;
            s1t += on ? a1t : 0.0;
            s2 += on ? a2 : 0.0;
        }
        T1[9 * mm + pq] = s1;
        T1[9 * mm + qp] = s1t;
        T2[9 * mm + pq] = s2;
        T2[9 * mm + qp] = s2;
    }
    __syncthreads();
    auto gt_tables = [&](const double* vec) {
        // the 3 torque entries of a step (an O(N) loop of 7 LDS reads) first, then the 6 force entries (1 read), the latter starting on
        // the lanes the former leave free: entry by entry every round of the wave ran all three loops (N = 10: 2 x 3 loops -> 1 + 2)
        for (int e = lane; e < 3 * N; e += 64) {
            const int j = e / 3, comp = e - 3 * j;
            const double* Cj = CP + j * 9;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {   // (steps before j add exact zeros: see T1 / T2 above)
                const double* Ci = CP + i * 9;
                const double* v = vec + 12 * i;
                const double term = dt2 * ((Ci[comp] - Cj[comp]) * (SQ[0] * v[0]) + (Ci[3 + comp] - Cj[3 + comp]) * (SQ[1] * v[1]) +
                                           (Ci[6 + comp] - Cj[6 + comp]) * (SQ[2] * v[2])) + dt * (SQ[6 + comp] * v[6 + comp]);
                acc += (i >= j) ? term : 0.0;
            }
            GV[9 * j + comp] = acc;
        }
        constexpr int HL = (3 * N) % 64;
        for (int e = lane - HL; e < 6 * N; e += 64) {
            if (e < 0) continue;
            const int j = e / 6, comp = 3 + (e - 6 * j);
            double acc = 0.0;
            const int off = (comp < 6) ? comp : 3 + comp;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double v = vec[12 * i + off];
                const double term = (comp < 6) ? (double)(i - j) * v : v;
                acc += (i >= j) ? term : 0.0;
            }
            GV[9 * j + comp] = acc;
        }
    };
    auto gt_eval = [&](int c) -> double {
        const int e = c / 3, ax = c - 3 * e, gc = act[e], j = gc >> 2;
        const double* J = sm + S::o_J + j * 36 + 3 * (gc & 3) + ax;
        const double* g = GV + 9 * j;
        return a.s * (J[0] * g[0] + J[12] * g[1] + J[24] * g[2] + SQ[3 + ax] * dt2m * g[3 + ax] + SQ[9 + ax] * dtm * g[6 + ax]);
    };
    if constexpr (RP != 2) gt_tables(sm + L1::o_eh);
    de_tables<N>(CP, T1, T2, SQ, dt2, DE, lane, 64);
    __syncthreads();
    WSTAMP(a, b, 2);
    if constexpr (RP != 2) for (int c = lane; c < n_eff; c += 64) sm[S::o_q + c] = gt_eval(c);      // (a continued pass: q stays)
    if (RP >= 2 || a.warm_u) {   // P x^0 through the tables; a continued pass: of the x the pass before left (in newtons, one per lane), as a second launch would
        double* TF = sm + L1::o_tf;
        for (int c = lane; c < n_eff; c += 64)
            sm[L1::o_x0c + c] = (RP >= 2 ? park[c] : a.warm_u[(size_t)b * n + 3 * act[c / 3] + (c % 3)]) / a.s;
        __syncthreads();
        for (int tt = lane; tt < 6 * N; tt += 64) {
            const int j = tt / 6, comp = tt - 6 * j;
            double acc = 0.0;
            for (int e = (j ? icnt[j - 1] : 0); e < icnt[j]; ++e) {
                const double* x = sm + L1::o_x0c + 3 * e;
                if (comp < 3) {
                    const double* J = sm + S::o_J + j * 36 + comp * 12 + 3 * (act[e] & 3);
                    acc += J[0] * x[0] + J[1] * x[1] + J[2] * x[2];
                } else {
                    acc += x[comp - 3];
                }
            }
            TF[tt] = acc;
        }
        __syncthreads();
        for (int k = lane; k < n; k += 64) {
            const int i = k / 12, kk = k - 12 * i;
            const double acc = gx_row<N>(CP, TF, i, kk, dt, dt2, dtm, dt2m);
            sm[L1::o_gx + k] = SQ[kk] * a.s * acc;
        }
        __syncthreads();
        gt_tables(sm + L1::o_gx);
        __syncthreads();
        for (int c = lane; c < n_eff; c += 64) sm[S::o_px0 + c] = gt_eval(c) + a.rs2 * sm[L1::o_x0c + c];
    } else {
        for (int c = lane; c < n_eff; c += 64) sm[S::o_px0 + c] = 0.0;
    }

    if constexpr (PHI) {
        double* phi = ws + W::o_phi;
        // the x0 the gradient above was formed with (NOT re-read from the staging array afterwards: the host may already have
        // written the measured state there while this kernel runs)
        const double x0_keep = (lane < 13) ? sm[S::o_x0 + lane] : 0.0;
        for (int kc = 0; kc < 13; ++kc) {
            __syncthreads();
            if (lane < 13) sm[S::o_x0 + lane] = (lane == kc) ? 1.0 : 0.0;
            __syncthreads();
            for (int k = lane; k < n; k += 64) {
                const int i = k / 12, kk = k - 12 * i;
                sm[L1::o_eh + k] = SQ[kk] * free_response<N, L1>(a, sm, i, kk);
            }
            __syncthreads();
            gt_tables(sm + L1::o_eh);
            __syncthreads();
            for (int c = lane; c < n_eff; c += 64) phi[c * 13 + kc] = gt_eval(c);
        }
        __syncthreads();
        if (lane < 13) sm[S::o_x0 + lane] = x0_keep;
        __syncthreads();
    }
    WSTAMP(a, b, 3);
    // ================= K = G'G + R s^2 + sigma + A' rho A: the rank-6 form, every lane its own C-layout entries (srbdqp_common.hpp) ====================
    v4d Kt[NT][NT];
    __syncthreads();   // the inputs / error vector / warm-start vectors under the table rows are dead
    {
        double* AB = sm + L1::o_ab;
        kasm_rows<N>(sm + S::o_J, CP, DE, SQ, act, n_eff, a.s, dt2m, dtm, AB, lane, 64, 16 * NT);
        __syncthreads();
        typedef double d2 __attribute__((ext_vector_type(2)));
        const double dgxy = uni(a.rs2 + a.sigma + 2.0 * rho_b), dgz = uni(a.rs2 + a.sigma + (4.0 * a.mu * a.mu + a.rho_fz) * rho_b);
        // same axis <=> r = c (mod 3); r = 16 ta + kq + 4 q = ta + q + kq, c = 16 tb + mcol = tb + mcol (mod 3): per lane ONE residue decides, per entry a compile-time one
        const int d3 = (mcol - kq + 3) % 3;
        const double ind[3] = {d3 == 0 ? 1.0 : 0.0, d3 == 1 ? 1.0 : 0.0, d3 == 2 ? 1.0 : 0.0};
        double eqd[4];                                       // the lane's entry q of a diagonal tile is ON the diagonal
#pragma unroll
        for (int q = 0; q < 4; ++q) eqd[q] = (kq + 4 * q == mcol) ? 1.0 : 0.0;
        double Bv[NT][8], dcol[NT];
#pragma unroll
        for (int tb = 0; tb < NT; ++tb) {
            const int c = 16 * tb + mcol;
            const d2* row = reinterpret_cast<const d2*>(AB + kAbStride * c + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const d2 v = row[i]; Bv[tb][2 * i] = v[0]; Bv[tb][2 * i + 1] = v[1]; }
            dcol[tb] = (c < n_eff) ? (((tb + mcol) % 3 < 2) ? dgxy : dgz) : 1.0;       // (c mod 3 = the axis; padding -> identity)
        }
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * ta + kq + 4 * q;
                const d2* row = reinterpret_cast<const d2*>(AB + kAbStride * r);
                double Av[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const d2 v = row[i]; Av[2 * i] = v[0]; Av[2 * i + 1] = v[1]; }
#pragma unroll
                for (int tb = ta; tb < NT; ++tb) {
                    double v = Av[0] * Bv[tb][0];
#pragma unroll
                    for (int i = 1; i < 6; ++i) v = fma(Av[i], Bv[tb][i], v);
                    const double ft = fma(Av[7], Bv[tb][7], Av[6] * Bv[tb][6]);
                    v = fma(ind[((ta + q - tb) % 3 + 3) % 3], ft, v);
                    if (ta == tb) v = fma(eqd[q], dcol[tb], v);
                    Kt[ta][tb][q] = v;
                }
            }
        }
    }
    __syncthreads();   // the phase-A arrays are dead; the scratch tile is used from here on
    WSTAMP(a, b, 4);
    WSTAMP(a, b, 5);
    if constexpr (DUMP) {   // assembly dump (srbdqp_assemble_f64), see dump_presolved()
        dump_presolved<N>(a, b, n_eff, na, sm + S::o_q, act, [&](auto&& put) {
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = ta; tb < NT; ++tb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (16 * ta + kq + 4 * q <= 16 * tb + mcol) put(16 * ta + kq + 4 * q, 16 * tb + mcol, Kt[ta][tb][q]);   // (the upper triangle: put() mirrors it)
        });
        return false;
    }

    // A-operand form of a C-layout tile X (operand[r] at lane (i, k') = X[i][4r + k']): through the wave-private tile
    double* scr = sm + L1::o_scr;
    auto a_operand = [&](const v4d& x, double (&op)[4]) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = kq + 4 * q;
            scr[row * 16 + (mcol ^ row)] = x[q];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) op[r] = scr[mcol * 16 + ((4 * r + kq) ^ mcol)];
        asm volatile("" ::: "memory");
    };
    const v4d zero4 = (v4d){0.0, 0.0, 0.0, 0.0};

    // ================= F: K = U'U, tiles (j, b > j) become U_jb, tile (j, j) becomes L_jj^-1 =================
    bool all_ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bool ok;
        v4d w = diag16_invert_mfma(Kt[j][j], lane, ok);
        all_ok = all_ok && ok;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (mcol <= kq + 4 * q) ? w[q] : 0.0;     // exact zeros above the diagonal
        Kt[j][j] = w;
        if (j + 1 < NT) {
            double wa[4];
            a_operand(w, wa);
#pragma unroll
            for (int bb = j + 1; bb < NT; ++bb) {   // panel: U_jb = L_jj^-1 K_jb
                v4d o = zero4;
#pragma unroll
                for (int r = 0; r < 4; ++r) o = mfma_f64(wa[r], Kt[j][bb][r], o);
                Kt[j][bb] = o;
            }
#pragma unroll
            for (int aa = j + 1; aa < NT; ++aa)      // trailing: K_ab -= U_ja' U_jb
#pragma unroll
                for (int bb = aa; bb < NT; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Kt[aa][bb] = mfma_f64(-Kt[j][aa][r], Kt[j][bb][r], Kt[aa][bb]);
        }
    }
    if (!all_ok && lane == 0) sm[S::o_misc] = 1.0;
    WSTAMP(a, b, 6);

    // ================= W = L^-1, block row by block row, W_ij (i > j) into the slot of U_ji =================
#pragma unroll
    for (int i = 1; i < NT; ++i) {
        double wa[4];
        a_operand(Kt[i][i], wa);                             // W_ii as the left factor
#pragma unroll
        for (int j = 0; j < i; ++j) {
            v4d sacc = zero4;                                // sum_{k=j}^{i-1} L_ik W_kj, L_ik = U_ki'
#pragma unroll
            for (int k = j; k < i; ++k) {
                const v4d& wkj = (k == j) ? Kt[j][j] : Kt[j][k];
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc = mfma_f64(Kt[k][i][r], wkj[r], sacc);
            }
            v4d o = zero4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o = mfma_f64(-wa[r], sacc[r], o);
            Kt[j][i] = o;                                    // U_ji is not needed any more (j' > j use U_j'i only)
        }
    }

    WSTAMP(a, b, 7);
    // ================= I: K^-1 = W'W tile by tile; split: stored as produced, fused: one row per lane =================
    double kin[W::KS];
    if constexpr (!FUSED) {
        for (int i = lane; i < S::o_R; i += 64) ws[i] = sm[i];
        double* kinv = ws + W::o_kinv;
#pragma unroll
        for (int aa = 0; aa < NT; ++aa) {
#pragma unroll
            for (int bb = aa; bb < NT; ++bb) {
                v4d o = zero4;                               // sum_{k >= b} W_ka' W_kb
#pragma unroll
                for (int k = bb; k < NT; ++k) {
                    const v4d& wka = (k == aa) ? Kt[aa][aa] : Kt[aa][k];
                    const v4d& wkb = (k == bb) ? Kt[bb][bb] : Kt[bb][k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = mfma_f64(wka[r], wkb[r], o);
                }
                const int col = 16 * bb + mcol;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * aa + kq + 4 * q;
                    if (row < n_eff && col < W::KS) kinv[row * W::KS + col] = (col < n_eff) ? o[q] : 0.0;
                    if (aa != bb && col < n_eff && row < W::KS) kinv[col * W::KS + row] = (row < n_eff) ? o[q] : 0.0;
                }
            }
        }
    } else {
        // one K^-1 row per lane: for every block column cb all NT blocks (g, cb) are produced directly in C layout
        // ((g, cb) with g > cb is the same product with the operands swapped -- no transposes), staged as NT row-major
        // tiles in the dead phase-A arrays, and lane (g, i) pulls row i of tile g with 16-byte reads (rows 18 doubles apart: with 16 the sixteen lanes of a
        // read pass hit two banks, an 8-way conflict on every one of the 32 reads)
        double* stg = sm + L1::o_cp;
        static_assert(L1::o_end - L1::o_cp >= NT * L1::kStgTile, "staging of one block column of K^-1");
        const int grp = lane >> 4;
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                v4d o = zero4;                               // sum_{k >= max(g, cb)} W_kg' W_k,cb
#pragma unroll
                for (int k = (g > cb ? g : cb); k < NT; ++k) {
                    const v4d& wkg = (k == g) ? Kt[g][g] : Kt[g][k];
                    const v4d& wkc = (k == cb) ? Kt[cb][cb] : Kt[cb][k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = mfma_f64(wkg[r], wkc[r], o);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) stg[g * L1::kStgTile + (kq + 4 * q) * L1::kStgRow + mcol] = o[q];
            }
            asm volatile("" ::: "memory");
            const double2* rowp = reinterpret_cast<const double2*>(stg + grp * L1::kStgTile + mcol * L1::kStgRow);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                if (16 * cb + 2 * h + 1 < W::KS + 1) {
                    const double2 v = rowp[h];
                    if (16 * cb + 2 * h < W::KS) kin[16 * cb + 2 * h] = v.x;
                    if (16 * cb + 2 * h + 1 < W::KS) kin[16 * cb + 2 * h + 1] = v.y;
                }
            }
            asm volatile("" ::: "memory");
        }
    }
    if constexpr (FUSED) {
#pragma unroll
        for (int c = 0; c < W::KS; ++c) kin[c] = (lane < n_eff && c < n_eff) ? kin[c] : 0.0;
        __syncthreads();
        WSTAMP(a, b, 8);
        WSTAMP(a, b, 9);
        static_assert(L1::o_end >= SplitSmem<N, MAXS>::o_end, "the ADMM body's vectors live in this kernel's LDS");
        if constexpr (RP == 0) {
            admm_wave_body<N, MAXS>(a, b, rho_b, sm, kin);
        } else {
            WaveRestart rs;
            const int left = a.max_iter - rs_done;                                   // the cap is on the total
            rs.pass = rs_pass;
            rs.more = rs_pass < a.restart_max && a.restart_every < left;
            rs.kcap = rs.more ? a.restart_every : left;
            rs.park = park;
            int status = -1, iters = 0;
            admm_wave_iterations<N, MAXS, true>(a, io, RP == 1 ? a.warm_u : nullptr, RP == 1 ? a.warm_y : nullptr, b, rho_b, sm, kin, sm + SplitSmem<N, MAXS>::o_xs,
                                                status, iters, &rs);
            rs_done += iters;
            if (status == 2 && rs.more) {   // (wave-uniform) at the mark, not converged: another pass
                rho_b = restart_rho_of(rho_b, rs.v);
                ++rs_pass;
                __syncthreads();
                return true;
            }
            admm_wave_finish<N, MAXS>(a, io, b, sm, status, rs_done);
        }
        WSTAMP(a, b, 10);
        WSTAMP(a, b, 11);
        WSTAMP_RT(a, b, 13);
    }
    return false;
}

template <int N, int MAXS, bool FUSED, bool DUMP = false, bool PHI = false, bool RST = false>
__global__ __launch_bounds__(64, 2) void srbdqp_setup1_kernel(KArgs a) {
    static_assert(!(PHI && FUSED), "dq/dx0 is a hand-over of the split pipeline");
    static_assert(!RST || (FUSED && !DUMP), "the restart in place belongs to the fused solve");
    static_assert(Setup1Smem<N, MAXS>::supported, "one wave holds all tiles: at most 4 x 4 tiles");
    static_assert(4 * N <= 64, "one ballot compacts the contact flags");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= a.B) return;
    if (a.count_ptr && (int)blockIdx.x >= *a.count_ptr) return;
    const int b = SRBDQP_QP_INDEX(a);
    if (SRBDQP_RESTART_SKIP(a, b)) return;
    double rho_b = SRBDQP_RHO_OF(a, b);
    int rs_pass = 0, rs_done = 0;
    const QpIo io = io_of(a);
    if constexpr (!RST) {
        setup1_pass<N, MAXS, FUSED, DUMP, PHI, 0>(a, io, b, sm, rho_b, rs_pass, rs_done);
    } else {
        // (round 5: the continued passes as three straight copies instead of a loop -- a loop around the body has the compiler hoist the body's index arithmetic and
        //  constants in front of it and keep them live through it: 256 VGPRs + 52 bytes of scratch at N = 10.  Exact for rho_restart_count <= 3; with more
        //  re-balancings allowed the third continued pass runs on to the iteration cap)
        bool again = setup1_pass<N, MAXS, FUSED, DUMP, PHI, 1>(a, io, b, sm, rho_b, rs_pass, rs_done);
        if (!again) return;
        again = setup1_pass<N, MAXS, FUSED, DUMP, PHI, 2>(a, io, b, sm, rho_b, rs_pass, rs_done);
        if (!again) return;
        again = setup1_pass<N, MAXS, FUSED, DUMP, PHI, 2>(a, io, b, sm, rho_b, rs_pass, rs_done);
        if (!again) return;
        KArgs al = a;
        al.restart_max = 0;
        setup1_pass<N, MAXS, FUSED, DUMP, PHI, 2>(al, io, b, sm, rho_b, rs_pass, rs_done);
    }
}

// ---- deferred tails (SRBDQP_FLAG_DEFER_TAIL) --------------------------------------------------------------------------------------------------------------
// A launch of the restart kernel above lasts as long as its slowest QP: up to three set-ups and 250 iterations on one wave (0.24 ms alone on the device, when
// the bulk of 4096 QPs is done after 0.12 ms) -- which is why the step rate of round 3 leaned on a longest-first dispatch hint and on a second stream.  Here no
// workgroup runs more than ONE pass: a QP that reaches a restart mark unconverged parks (x, y), re-balances its rho and appends itself -- with the pointers
// of its own launch -- to a list in HBM; the first tail_wgs workgroups of the NEXT launch on the same stream (another batch, other buffers) each take one record,
// rebuild the strip from the QP's inputs and run its next pass.  Same passes, same arithmetic, same statuses and iteration counts as the restart in place; the
// outputs of a deferred QP arrive one or two launches later (status[] says SRBDQP_PENDING until then; srbdqp_flush() runs what is left).  No loop around the
// body anywhere: both arms are straight-line code.
__device__ __forceinline__ unsigned long long unis_u64(unsigned long long v) {
    int lo, hi;
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"((int)(unsigned)v), "v"((int)(unsigned)(v >> 32)));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
template <typename T> __device__ __forceinline__ T* unis_ptr(T* p) { return reinterpret_cast<T*>(unis_u64(reinterpret_cast<unsigned long long>(p))); }

// a free record of the list this launch appends to, or null: the list is full (round 5: the lists are sized for a quarter of a launch's QPs, not for all of them
// three times over; the count may run past tail_cap -- the reader clamps it -- and a QP that finds no room runs its remaining passes in place)
__device__ __forceinline__ double* tail_claim(const KArgs& a) {
    int slot = 0;
    if (threadIdx.x == 0) slot = atomicAdd(a.tail_cnt + a.tail_iout, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= a.tail_cap) return nullptr;
    return reinterpret_cast<double*>(a.tail_lists) + ((size_t)a.tail_iout * (size_t)a.tail_cap + (size_t)slot) * kTailRecDoubles;
}

__device__ __forceinline__ bool tail_export(const KArgs& a, const QpIo& io, int b, double rho_b, int rs_pass, int rs_done, const double* park) {
    const int lane = threadIdx.x;
    double* rec = tail_claim(a);
    if (!rec) return false;
    if (lane == 0) {
        TailRecHead* hd = reinterpret_cast<TailRecHead*>(rec);
        hd->b = b; hd->pass = rs_pass; hd->done = rs_done; hd->pad = 0; hd->rho = rho_b; hd->io = io;
        if (io.status) io.status[b] = kStatusPending;
    }
    rec[16 + lane] = park[lane]; rec[80 + lane] = park[64 + lane]; rec[144 + lane] = park[128 + lane];
    return true;
}

// FLUSH: the continuations alone (srbdqp_flush: a launch with no QPs of its own; a kernel of its own name, so that profiles keep the two apart).  Nothing comes
// behind a flush that could pick a record up, so here a workgroup runs ALL the passes its QP has left, the later ones in place as the restart kernel does (the
// loop's hoisted registers cost this instantiation only): one flush launch instead of rho_restart_count of them.
template <int N, int MAXS, bool FLUSH = false>
__global__ __launch_bounds__(64, 2) void srbdqp_wave_defer_kernel(KArgs a) {
    static_assert(Setup1Smem<N, MAXS>::supported && 4 * N <= 64, "the one-wave kernel's limits");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    using L1 = Setup1Smem<N, MAXS>;
    const int lane = threadIdx.x;
    const int T = a.tail_wgs;
    double* const park = sm + L1::o_end;
    // the passes a QP has left, in place: three straight copies of the body (see below), in each arm on its own -- the arms share nothing, so that the first passes
    // (96 % of the workgroups) run the code they ran before there was an overflow path
    auto in_place = [&](const QpIo& io, const int b, double rho_b, int rs_pass, int rs_done) __attribute__((always_inline)) {
        bool again = setup1_pass<N, MAXS, true, false, false, 2>(a, io, b, sm, rho_b, rs_pass, rs_done);
        if (!again) return;
        again = setup1_pass<N, MAXS, true, false, false, 2>(a, io, b, sm, rho_b, rs_pass, rs_done);
        if (!again) return;
        KArgs al = a;
        al.restart_max = 0;
        setup1_pass<N, MAXS, true, false, false, 2>(al, io, b, sm, rho_b, rs_pass, rs_done);
    };
    if ((int)blockIdx.x < T) {
        // ---- a continuation of an earlier launch (first in the grid: these are the long ones)
        const int i = blockIdx.x;
        int cnt = a.tail_cnt[a.tail_iin];
        cnt = cnt < a.tail_cap ? cnt : a.tail_cap;                                          // (claims past the end of a full list found no room: tail_claim)
        if (i == 0 && lane == 0) a.tail_cnt[a.tail_izero] = 0;                             // the list the NEXT launch appends to
        // (cnt <= T: the host launches one tail workgroup per record the list can hold -- a bound it knows, srbdqp.hip launch_wave_defer)
        const double* lin = reinterpret_cast<const double*>(a.tail_lists) + (size_t)a.tail_iin * (size_t)a.tail_cap * kTailRecDoubles;
        if (i >= cnt) return;
        const double* rec = lin + (size_t)i * kTailRecDoubles;
        const TailRecHead* hd = reinterpret_cast<const TailRecHead*>(rec);
        QpIo io;
        io.x0 = unis_ptr(hd->io.x0); io.xref = unis_ptr(hd->io.xref); io.foot = unis_ptr(hd->io.foot); io.pcom = unis_ptr(hd->io.pcom);
        io.contact = unis_ptr(hd->io.contact); io.u_out = unis_ptr(hd->io.u_out); io.x_out = unis_ptr(hd->io.x_out); io.y_out = unis_ptr(hd->io.y_out);
        io.status = unis_ptr(hd->io.status); io.iters = unis_ptr(hd->io.iters);
        const int b = __builtin_amdgcn_readfirstlane(hd->b);
        int rs_pass = __builtin_amdgcn_readfirstlane(hd->pass), rs_done = __builtin_amdgcn_readfirstlane(hd->done);
        double rho_b = unis(hd->rho);
        park[lane] = rec[16 + lane]; park[64 + lane] = rec[80 + lane]; park[128 + lane] = rec[144 + lane];
        __syncthreads();
        if (!setup1_pass<N, MAXS, true, false, false, 3>(a, io, b, sm, rho_b, rs_pass, rs_done)) return;
        // at a mark, unconverged: on to the next launch on the stream -- or, when the list has no room left (or nothing comes behind a flush), the passes it has left
        // in place, as the restart kernel runs them (the strip is in place by now).  Same passes, same arithmetic either way.
        if constexpr (!FLUSH) { if (tail_export(a, io, b, rho_b, rs_pass, rs_done, park)) return; }
        in_place(io, b, rho_b, rs_pass, rs_done);
    } else if constexpr (!FLUSH) {
        // ---- a QP of this launch: its first pass
        const int wg = (int)blockIdx.x - T;
        if (wg >= a.B) return;
        const int b = a.perm ? a.perm[wg] : wg;
        double rho_b = a.rho_qp ? a.rho_qp[b] : a.rho;
        int rs_pass = 0, rs_done = 0;
        const QpIo io = io_of(a);
        if (!setup1_pass<N, MAXS, true, false, false, 1>(a, io, b, sm, rho_b, rs_pass, rs_done)) return;
        if (tail_export(a, io, b, rho_b, rs_pass, rs_done, park)) return;
        // The list is full (rare: the lists hold a quarter of a launch).  In place, but NOT as a loop: a loop around the body has the compiler hoist the body's index
        // arithmetic and constants in front of it and keep them live through it (256 VGPRs + 72 bytes of scratch for the whole kernel) -- three straight copies
        // instead, which cover rho_restart_count <= 3 exactly (values above 3 mean 3: srbdqp.h).
        in_place(io, b, rho_b, rs_pass, rs_done);
    }
}

}  // namespace srbdqp

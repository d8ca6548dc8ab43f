// srbdqp_mfma.hpp -- kernel variant v1 ("mfma"): the condensed-QP hot path on the fp64 matrix cores.
//
// One 256-thread workgroup (4 wave64) per QP, 79 KB of LDS (two workgroups per CU), no HBM traffic between the
// input load and the output store.  All dense n x n work is 16x16 tiles on v_mfma_f64_16x16x4_f64:
//
//   H  : K = G'G (+ R s^2 + sigma + A' rho A on the diagonal), G = Q^1/2 s B_qp packed block-lower in LDS; the 36
//        upper tiles (n = 120 -> 8x8 tiles) accumulate in registers, 9 tiles per wave.          [a7, MFMA contraction]
//   F  : right-looking Cholesky K = U'U with the trailing tiles resident in registers; per block column one wave
//        factors and inverts the 16x16 diagonal tile (diag16_invert_mfma: blocked 4x4, on the matrix cores), the
//        panel is L_jj^-1 K_jb and the trailing update K_ab -= U_ja' U_jb.                          [a9 factor]
//   W  : W = L^-1 row by row, in place over U (W_ij = -W_ii sum_k L_ik W_kj).
//   I  : K^-1 = W'W, tiles swizzled into LDS, then every thread pulls its row fragment of K^-1 into registers.
//   ADMM: shared admm_loop() -- the two triangular solves of each iteration are applied as one mat-vec with the
//        explicit inverse (depth-1 instead of depth-n substitution).                              [a9 iterations]
//
// Register tiles are in the MFMA C/D layout (lane l: column l&15, rows (l>>4)+4r, r = 0..3).  Register r of a
// C-layout tile is exactly the B operand of K-step r of a product that contracts over the tile's ROW index, so
// chains of tile products (panel, W rows) never leave the register file.  LDS tiles are row-major 16x16; tiles that
// are read both along rows and along columns (the diagonal inverses, the final K^-1) are XOR-swizzled
// (col ^ row), which makes both access directions bank-conflict free.
#pragma once
#include "srbdqp_common.hpp"
#include "srbdqp_admm.hpp"

namespace srbdqp {

constexpr bool kMfmaReady = true;

#ifdef SRBDQP_PROFILE_F
#define F_T(var) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = (long long)t_; } while (0)
#else
#define F_T(var) do { } while (0)
#endif

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <int N>
struct MfmaSmem {
    static constexpr int n = Dims<N>::n, m = Dims<N>::m;
    static constexpr int NT = (n + 15) / 16;              // tiles per side
    static constexpr int NTT = NT * (NT + 1) / 2;         // upper tiles
    static constexpr int TS = (NTT + 3) / 4;              // tile slots per wave
    static constexpr int up2(int v) { return (v + 1) & ~1; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // ---- persistent across all phases
    static constexpr int o_x0 = 0;                        // 13 (+1)
    static constexpr int o_tm = o_x0 + 14;                // N*9
    static constexpr int o_J = o_tm + up2(N * 9);         // N*36
    static constexpr int o_q = o_J + N * 36;              // n
    static constexpr int o_px0 = o_q + n;                 // n   P x^0 (warm start)
    static constexpr int o_red = o_px0 + n;               // 64
    static constexpr int o_ct = o_red + 64;               // contact flags (bytes)
    static constexpr int o_misc = o_ct + up2((N * 4 + 7) / 8);   // [0] = numerical-failure flag
    static constexpr int o_sq = o_misc + 8;               // 12  sqrt(q_diag)
    static constexpr int o_R = o_sq + 12;                 // ---- the big region, re-used phase by phase
    // phase A (assembly): G + inputs
    static constexpr int o_G = o_R;
    static constexpr int o_xref = o_G + Dims<N>::gtot;
    static constexpr int o_foot = o_xref + up2(N * 13);
    static constexpr int o_pcom = o_foot + N * 12;
    static constexpr int o_cp = o_pcom + up2(N * 3);
    static constexpr int o_eh = o_cp + up2(N * 9);
    static constexpr int o_gx = o_eh + n;                 // n   G x^0 (warm start)
    static constexpr int endA = o_gx + n + 16;            // +16: masked operand reads may run past the last G row
    // phase B (factor / inverse): tiles
    static constexpr int o_T = o_R;
    static constexpr int endB = o_T + NTT * 256;
    // phase C (ADMM): double-buffered rhs + the solution vector
    static constexpr int o_rhs = o_R;                     // 2 x (n + 8)
    static constexpr int o_xs = o_rhs + 2 * (n + 8);      // n
    static constexpr int endC = o_xs + n;
    static constexpr int o_end = cmax(endA, cmax(endB, endC));
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
};

template <int N>
struct MfmaTraits {
    static constexpr bool supported = (12 * N <= 128);
    static constexpr size_t lds_bytes = MfmaSmem<N>::bytes;
    static constexpr const char* name = (N == 10) ? "mfma_f64_n10" : (N == 8) ? "mfma_f64_n8" : (N == 4) ? "mfma_f64_n4" : "mfma_f64";
};

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// index of upper tile (a, b), a <= b
__device__ __forceinline__ int tile_id(int a, int b) { return (b * (b + 1)) / 2 + a; }

// C-layout register tile -> LDS tile (row-major; swizzled: col ^ row)
template <bool SWZ>
__device__ __forceinline__ void store_tile(double* tile, const v4d& v, int lane) {
    const int col = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = g + 4 * r;
        tile[row * 16 + (SWZ ? (col ^ row) : col)] = v[r];
    }
}

// 1/sqrt(d) to full double precision: v_rsq_f64 seed + two Newton steps (d is a positive, normal pivot).
__device__ __forceinline__ double fast_rsqrt(double d) {
    double r = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d * r;
    r = fma(r, fma(-h, r, 0.5), r);
    h = 0.5 * d * r;
    return fma(r, fma(-h, r, 0.5), r);
}

// ---------------------------------------------------------------------------------------------------------
// 16x16 diagonal tile, blocked on the matrix cores: S (SPD, C-layout register tile) -> W = L^-1 (C layout), S = L L'.
// One wave, nothing leaves the register file.  Row block kb (4 rows) of a C-layout tile is register kb; kept as the
// UPPER factor U = L', register kb is at once the A operand (A[i][k] = L[i][4kb+k]) and the B operand
// (B[k][j] = U[4kb+k][j]) of the rank-4 products, so per block the work is
//   - 10 v_readlane pairs broadcast the 4x4 diagonal block; every lane factors it and inverts the factor (4 rsqrt),
//   - U_kb = L_kk^-1 S_kb              one MFMA (L_kk^-1 embedded in rows 4kb..4kb+3 of the A operand),
//   - S   -= U_kb' U_kb                one MFMA (trailing update of the whole tile),
//   - W_kb = L_kk^-1 R_kb, R -= L[:,kb] W_kb   two MFMAs (block forward substitution L W = I riding along),
// 14 MFMAs and ~350 other instructions per tile (a row-by-row LDL' elimination with v_readlane broadcasts, the first
// version of this routine, took ~1000 and twice the time).
// ---------------------------------------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void diag16_block(v4d& s, v4d& rr, v4d& w, double& pmin, int lane) {
    const int col = lane & 15, k = lane >> 4, li = col & 3;
    const double sk = s[KB];
    // diagonal block D[g][c] = S[4KB+g][4KB+c] sits in lane 16 g + 4KB + c of register KB
    const double d00 = readlane_f64(sk, 4 * KB), d01 = readlane_f64(sk, 4 * KB + 1), d02 = readlane_f64(sk, 4 * KB + 2), d03 = readlane_f64(sk, 4 * KB + 3);
    const double d11 = readlane_f64(sk, 16 + 4 * KB + 1), d12 = readlane_f64(sk, 16 + 4 * KB + 2), d13 = readlane_f64(sk, 16 + 4 * KB + 3);
    const double d22 = readlane_f64(sk, 32 + 4 * KB + 2), d23 = readlane_f64(sk, 32 + 4 * KB + 3);
    const double d33 = readlane_f64(sk, 48 + 4 * KB + 3);
    // D = R'R (R upper), X = R^-1
    const double x00 = fast_rsqrt(d00);
    const double r01 = d01 * x00, r02 = d02 * x00, r03 = d03 * x00;
    const double p1 = fma(-r01, r01, d11);
    const double x11 = fast_rsqrt(p1);
    const double r12 = fma(-r01, r02, d12) * x11, r13 = fma(-r01, r03, d13) * x11;
    const double p2 = fma(-r12, r12, fma(-r02, r02, d22));
    const double x22 = fast_rsqrt(p2);
    const double r23 = fma(-r12, r13, fma(-r02, r03, d23)) * x22;
    const double p3 = fma(-r23, r23, fma(-r13, r13, fma(-r03, r03, d33)));
    const double x33 = fast_rsqrt(p3);
    pmin = (d00 > 0.0 && p1 > 0.0 && p2 > 0.0 && p3 > 0.0) ? pmin : -1.0;   // (a NaN pivot fails the comparison too)
    const double x01 = -x00 * r01 * x11;
    const double x12 = -x11 * r12 * x22;
    const double x23 = -x22 * r23 * x33;
    const double x02 = -fma(x00, r02, x01 * r12) * x22;
    const double x13 = -fma(x11, r13, x12 * r23) * x33;
    const double x03 = -fma(x00, r03, fma(x01, r13, x02 * r23)) * x33;
    // A operand: lane (i = col, k): L_kk^-1[i - 4KB][k] = X[k][i - 4KB] for rows i of this block, k <= i - 4KB; else 0
    const double c0 = x00;                                              // li = 0: k = 0
    const double c1 = (k == 0) ? x01 : x11;                             // li = 1: k = 0, 1
    const double c2 = (k == 0) ? x02 : (k == 1) ? x12 : x22;            // li = 2
    const double c3 = (k == 0) ? x03 : (k == 1) ? x13 : (k == 2) ? x23 : x33;
    double a = (li == 0) ? c0 : (li == 1) ? c1 : (li == 2) ? c2 : c3;
    a = ((col >> 2) == KB && k <= li) ? a : 0.0;
    const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
    double u = mfma_f64(a, sk, zero)[KB];                               // U_kb = L_kk^-1 S_kb
    u = (col >= 4 * KB) ? u : 0.0;                                      // columns left of the block: exact zeros
    const double wk = mfma_f64(a, rr[KB], zero)[KB];                    // W_kb = L_kk^-1 R_kb
    w[KB] = wk;
    if constexpr (KB < 3) {
        s = mfma_f64(-u, u, s);                                         // S -= U_kb' U_kb
        const double am = (col >= 4 * (KB + 1)) ? u : 0.0;              // L[i][4KB + k] for the rows below the block
        rr = mfma_f64(-am, wk, rr);                                     // R -= L[:, kb] W_kb
        diag16_block<KB + 1>(s, rr, w, pmin, lane);
    }
}

// returns W = L^-1 in C layout (entries above the diagonal are rounding-level garbage: consumers mask them);
// ok = every pivot positive
__device__ __forceinline__ v4d diag16_invert_mfma(v4d s, int lane, bool& ok) {
    const int col = lane & 15, g = lane >> 4;
    v4d rr, w;
#pragma unroll
    for (int r = 0; r < 4; ++r) { rr[r] = (g + 4 * r == col) ? 1.0 : 0.0; w[r] = 0.0; }
    double pmin = 1.0e300;
    diag16_block<0>(s, rr, w, pmin, lane);
    ok = pmin > 0.0;
    return w;
}


template <int N>
__global__ __launch_bounds__(kThreads, 2) void srbdqp_mfma_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    using S = MfmaSmem<N>;
    constexpr int n = Dims<N>::n, m = Dims<N>::m, CH = n / 2;
    constexpr int NT = S::NT, NTT = S::NTT, TS = S::TS;
    static_assert(n <= 128, "mfma variant: two threads per row of K^-1, n <= 128");
    static_assert(NT <= 8, "W phase assumes at most two tiles per wave per block row");
    static_assert((S::o_rhs % 2) == 0 && (CH % 2) == 0 && (S::o_R % 2) == 0, "16-byte alignment");
    if ((int)blockIdx.x >= a.B) return;
    const int b = SRBDQP_QP_INDEX(a);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int mcol = lane & 15, kq = lane >> 4;       // MFMA operand coordinates of this lane
    double* G = sm + S::o_G;
    double* T = sm + S::o_T;

    // ================= phase A: linearise, condense, gradient =================
    SRBDQP_STAMP(a, b, 0);
    if (a.stamps && t == 0) a.stamps[(size_t)b * 16 + 12] = (long long)__builtin_amdgcn_s_memrealtime();
    load_and_linearise<N, S>(a, b, sm);
    SRBDQP_STAMP(a, b, 1);
    build_G<N, S>(a, sm, G);
    if (t == 0) sm[S::o_misc] = 0.0;
    __syncthreads();
    SRBDQP_STAMP(a, b, 2);
    build_gradient<N, S>(a, sm, G);
    SRBDQP_STAMP(a, b, 3);
    if (a.warm_u) {   // P x^0 = G'(G x^0) + R s^2 x^0 while G is still around
        for (int k = t; k < n; k += kThreads) {
            const int i = k / 12, kk = k - 12 * i, len = 12 * (i + 1);
            const double* row = G + g_row_off(i, kk);
            double acc = 0.0;
            for (int c = 0; c < len; ++c) acc = fma(row[c], a.warm_u[(size_t)b * n + c] / a.s, acc);
            sm[S::o_gx + k] = acc;
        }
        __syncthreads();
        for (int c = t; c < n; c += kThreads) {
            double acc = a.rs2 * (a.warm_u[(size_t)b * n + c] / a.s);
            for (int k = 12 * (c / 12); k < n; ++k) {
                const int i = k / 12, kk = k - 12 * i;
                acc = fma(G[g_row_off(i, kk) + c], sm[S::o_gx + k], acc);
            }
            sm[S::o_px0 + c] = acc;
        }
    } else {
        for (int c = t; c < n; c += kThreads) sm[S::o_px0 + c] = 0.0;
    }

    // tile slots of this wave: slot s <-> upper tile id 4 s + w
    int ta[TS], tb[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        const int id = 4 * s + w;
        int bb = 0;
        while (((bb + 1) * (bb + 2)) / 2 <= id) ++bb;
        tb[s] = (id < NTT) ? bb : -1;
        ta[s] = (id < NTT) ? id - (bb * (bb + 1)) / 2 : -1;
    }

    // ================= phase H: K tiles = G'G on the fp64 matrix cores =================
    v4d acc[TS];
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (ta[s] >= 0) {
            const int ca = 16 * ta[s] + mcol, cb = 16 * tb[s] + mcol;
            for (int i = (16 * tb[s]) / 12; i < N; ++i) {
                const int len = 12 * (i + 1);
                const double* base = G + 72 * i * (i + 1) + len * kq;
#pragma unroll
                for (int kk0 = 0; kk0 < 12; kk0 += 4) {
                    const double* row = base + len * kk0;
                    double av = row[ca], bv = row[cb];
                    av = (ca < len) ? av : 0.0;
                    bv = (cb < len) ? bv : 0.0;
                    acc[s] = mfma_f64(av, bv, acc[s]);
                }
            }
            if (ta[s] == tb[s]) {   // diagonal tile: + R s^2 + sigma + A' rho A; padding rows/cols -> identity
                const int var = 16 * ta[s] + mcol;
                const double dv = (var < n) ? a.rs2 + a.sigma + rho_diag<N, S>(a, sm, var < n ? var : 0) : 1.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (kq + 4 * r == mcol) acc[s][r] += dv;
            }
        }
    }
    SRBDQP_STAMP(a, b, 4);
    __syncthreads();   // G is dead from here on; region R becomes the tile store
    SRBDQP_STAMP(a, b, 5);

    // ================= phase F: tiled right-looking Cholesky K = U'U, trailing tiles in registers =================
    long long fs0 = 0, fs1 = 0, fs2 = 0, fs3 = 0, ft0 = 0, ft1 = 0, ft2 = 0, ft3 = 0, ft4 = 0;
    (void)fs0; (void)fs1; (void)fs2; (void)fs3; (void)ft0; (void)ft1; (void)ft2; (void)ft3; (void)ft4;
    for (int j = 0; j < NT; ++j) {
        double* Djj = T + tile_id(j, j) * 256;
        F_T(ft0);
#pragma unroll
        for (int s = 0; s < TS; ++s)
            if (ta[s] == j && tb[s] == j) store_tile<false>(Djj, acc[s], lane);
        __syncthreads();
        F_T(ft1);
        if (w == (j & 3)) {   // load the tile back in C layout, invert it on the matrix cores, store L_jj^-1 swizzled
            v4d d;
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = Djj[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
            bool ok;
            const v4d winv = diag16_invert_mfma(d, lane, ok);
            store_tile<true>(Djj, winv, lane);
            if (!ok && lane == 0) sm[S::o_misc] = 1.0;
        }
        __syncthreads();
        F_T(ft2);
        // panel: U_jb = L_jj^-1 K_jb
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            if (ta[s] == j && tb[s] > j) {
                v4d o = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    double av = Djj[mcol * 16 + (k ^ mcol)];      // L_jj^-1[m][k], k <= m
                    av = (k <= mcol) ? av : 0.0;
                    o = mfma_f64(av, acc[s][r], o);
                }
                acc[s] = o;
                store_tile<false>(T + tile_id(j, tb[s]) * 256, o, lane);
            }
        }
        __syncthreads();
        F_T(ft3);
        // trailing update: K_ab -= U_ja' U_jb
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
        F_T(ft4);
        fs0 += ft1 - ft0; fs1 += ft2 - ft1; fs2 += ft3 - ft2; fs3 += ft4 - ft3;
    }
#ifdef SRBDQP_PROFILE_F
    if (a.stamps && t == 0) { a.stamps[(size_t)b * 16 + 12] = fs0; a.stamps[(size_t)b * 16 + 13] = fs1; a.stamps[(size_t)b * 16 + 14] = fs2; a.stamps[(size_t)b * 16 + 15] = fs3; }
#endif
    __syncthreads();

    SRBDQP_STAMP(a, b, 6);
    // ================= phase W: W = L^-1 (L = U'), block row by block row, in place over U =================
    for (int i = 1; i < NT; ++i) {
        v4d res[2];
        const double* Dii = T + tile_id(i, i) * 256;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = w + 4 * q;
            res[q] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (j < i) {
                v4d o = (v4d){0.0, 0.0, 0.0, 0.0};
                {   // k = j:  L_ij W_jj
                    const double* Uji = T + tile_id(j, i) * 256;
                    const double* Djj = T + tile_id(j, j) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        double bv = Djj[k * 16 + (mcol ^ k)];     // W_jj[k][n], n <= k
                        bv = (mcol <= k) ? bv : 0.0;
                        o = mfma_f64(Uji[k * 16 + mcol], bv, o);
                    }
                }
                for (int k2 = j + 1; k2 < i; ++k2) {   // L_ik W_kj
                    const double* Uki = T + tile_id(k2, i) * 256;
                    const double* Wkj = T + tile_id(j, k2) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        o = mfma_f64(Uki[k * 16 + mcol], Wkj[k * 16 + mcol], o);
                    }
                }
                v4d o2 = (v4d){0.0, 0.0, 0.0, 0.0};     // W_ij = -W_ii * (sum)
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
    // ================= phase I: K^-1 = W'W =================
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (ta[s] >= 0) {
            const int ia = ta[s], ib = tb[s];
            const double* Dbb = T + tile_id(ib, ib) * 256;
            {   // i = ib
                const double* Wba = T + tile_id(ia, ib) * 256;     // only read when ia < ib
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    double dv = Dbb[k * 16 + (mcol ^ k)];          // W_bb[k][.], lower
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
    // row fragment of K^-1 for the ADMM mat-vec: thread (row rr, half h) owns K^-1[rr][CH h .. CH h + CH - 1];
    // rows are dealt to lanes contact by contact (srbdqp_admm.hpp)
    const int rown = admm_row_of_thread<N>(t), h = t & 1;
    double kin[CH];
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const int c = CH * h + cc;
        const int rr = (rown >= 0) ? rown : 0;            // keep the address of idle threads inside the tile store
        const int lo = (rr <= c) ? rr : c, hi = (rr <= c) ? c : rr;
        const int row = lo & 15, col = hi & 15;
        const double v = T[tile_id(lo >> 4, hi >> 4) * 256 + row * 16 + (col ^ row)];
        kin[cc] = (rown >= 0) ? v : 0.0;
    }
    const bool failed = sm[S::o_misc] != 0.0;
    __syncthreads();   // tiles are dead; region R becomes the ADMM vectors

    SRBDQP_STAMP(a, b, 9);
    int status = -1, iters = 0;
    if (!failed) iters = admm_loop_local<N, S, CH>(a, b, sm, sm + S::o_rhs, sm + S::o_xs, kin, &status);
    else {
        for (int c = t; c < n; c += kThreads) sm[S::o_xs + c] = 0.0;
        if (a.y_out) for (int i = t; i < m; i += kThreads) a.y_out[(size_t)b * m + i] = 0.0;
        __syncthreads();
    }
    if (t == 0) {
        if (a.status) a.status[b] = status;
        if (a.iters) a.iters[b] = iters;
    }
    SRBDQP_STAMP(a, b, 10);
    rollout_and_store<N, S>(a, b, sm, sm + S::o_xs, sm + S::o_rhs);
    SRBDQP_STAMP(a, b, 11);
#if !defined(SRBDQP_PROFILE_F) && !defined(SRBDQP_PROFILE_ADMM)
    if (a.stamps && t == 0) a.stamps[(size_t)b * 16 + 13] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace srbdqp

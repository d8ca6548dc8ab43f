// srbdqp_mfma.hpp -- the fp64 matrix-core building blocks shared by every kernel of the engine (the round-1 kernel "v1"
// that introduced them -- all 12N variables, packed G + GᵀG on the matrix cores -- is retired; its phases live on in
// srbdqp_compact.hpp / srbdqp_setup1.hpp / srbdqp_wrench.hpp on the presolved problems).
//
// All dense n x n work is 16x16 tiles on v_mfma_f64_16x16x4_f64:
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

// 1/sqrt(d) of a positive, normal pivot: v_rsq_f64 seed + ONE Newton step.  Measured on gfx950 (tools/rsq_probe.hip, 4 M random
// doubles over 40 binades): seed 5.2e-8 relative, one step 4.1e-15, two steps 1.4e-16 -- the second step (four more dependent
// fp64 instructions per pivot, 16 pivots per diagonal tile, on the critical path of every factorisation) bought nothing the
// ADMM can see: K^-1 is applied to right-hand sides that are themselves only converged to 1e-6.
__device__ __forceinline__ double fast_rsqrt(double d) {
    const double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d * r;
    return fma(r, fma(-h, r, 0.5), r);
}

// ... with the second Newton step (1.4e-16 relative): the 6 x 6 wrench blocks E of the general kernel, whose inverse enters T and
// V at full weight (the assembly parity tests hold both to 1e-11).  Still a fifth of the instructions of 1.0 / sqrt(d).
__device__ __forceinline__ double fast_rsqrt2(double d) {
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
//   - U_kb = L_kk^-1 S_kb              one MFMA (L_kk^-1 repeated in every row block of the A operand; accumulator register kb is the result),
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
    a = (k <= li) ? a : 0.0;   // (the same 4 x 4 block in every row block of the operand: only accumulator register KB -- rows 4 KB .. 4 KB + 3 -- of the two products is read)
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


// ---------------------------------------------------------------------------------------------------------
// The same inversion without the matrix cores and without v_readlane: one COLUMN per lane.  Lane c of every 16-lane DPP row holds column c of S
// (16 registers S_i = S[i][c]) and of R (R_i, the identity at the start).  Pivot k: d = S[k][k] is a row_newbcast of register S_k, u = S_k / sqrt(d)
// is row k of L' (own lane: L[c][k]), W_k = R_k / sqrt(d) is row k of L^-1, and every row below takes ONE DPP multiply-add per matrix,
//     S_i -= L[i][k] u,   R_i -= L[i][k] W_k     with L[i][k] = u in lane i = the instruction's row_newbcast:i source operand,
// 240 v_fmac_f64_dpp + 16 x 9 instructions of pivot chain (the next pivot's chain depends on the first multiply-add of a pivot only: the
// scheduler runs it beside the rest).  The blocked routine above is as many instructions but spends half its time waiting on dependent MFMAs and
// on v_readlane hazards: 3309 cycles a tile alone on its SIMD against 2888 here, 3571 against 2961 with two waves per SIMD (tools/diag_probe.hip; a DPP
// multiply-add of doubles issues every 8 cycles from one wave alone).  In and out through a 16 x 16 LDS tile of the caller's (C layout -> columns; W back
// row-major, XOR-swizzled as the consumers read it); the four DPP rows work redundantly.  Used by the general kernel and by the 4-wave kernel's batch-1
// instantiation (N = 10 double support 14.75 -> 14.99 M QP/s, configs[4] 2.83 -> 2.88 M, the batch-1 call -0.8 us); NOT by the one-wave kernel: its 64 live
// registers beside that kernel's ten register tiles end in scratch memory (256 VGPRs + 88 bytes; configs[1] at 35 fixed iterations 1.775 -> 1.852 ms per 65,536 QPs).
// ---------------------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void fmac_newbcast(double& acc, double nu, double m) {   // acc += nu[lane I of the row] * m
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(nu), "v"(m), "n"(I));
}
template <int K>
__device__ __forceinline__ double mov_newbcast(double v) {   // (the source may come out of one of the asm statements above, which the hazard recognizer does not see into: the two wait states by hand)
    double d;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
    return d;
}
template <int K, int I>
struct Diag16Rows {
    static __device__ __forceinline__ void run(double (&S)[16], double (&R)[16], double nu, double u, double wk) {
        if constexpr (I < 16) {
            fmac_newbcast<I>(S[I], nu, u);
            fmac_newbcast<I>(R[I], nu, wk);
            Diag16Rows<K, I + 1>::run(S, R, nu, u, wk);
        }
    }
};
// pivots K .. KEND - 1 (KEND = 16: the whole tile; the low-latency general kernel runs the tile in two halves around a workgroup barrier)
template <int K, int KEND = 16>
__device__ __forceinline__ void diag16_pivot(double (&S)[16], double (&R)[16], bool& ok) {
    const double d = mov_newbcast<K>(S[K]);
    ok = ok && (d > 0.0);                         // (a NaN pivot fails the comparison too)
    const double r = fast_rsqrt(d);
    const double u = S[K] * r;
    const double wk = R[K] * r;
    R[K] = wk;
    if constexpr (K < 15) {
        double nu;
        asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(nu) : "v"(S[K]), "v"(-r));       // -u, two wait states before its first DPP read
        Diag16Rows<K, K + 1>::run(S, R, nu, u, wk);
    }
    if constexpr (K + 1 < KEND) diag16_pivot<K + 1, KEND>(S, R, ok);
}
// tile: 256 doubles of LDS owned by this wave; on return it holds W = L^-1 row-major with the column XOR-swizzled (tile[row * 16 + (col ^ row)]), exact zeros
// above the diagonal; the return value is the same W in C layout.
__device__ __forceinline__ v4d diag16_invert_dpp(v4d s, int lane, bool& ok, double* tile) {
    const int col = lane & 15, g = lane >> 4;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(g + 4 * r) * 16 + col] = s[r];
    asm volatile("" ::: "memory");
    double S[16], R[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = tile[i * 16 + col]; R[i] = (i == col) ? 1.0 : 0.0; }     // (row i of the upper triangle; below the diagonal the lanes carry values nobody reads)
    asm volatile("" ::: "memory");
    ok = true;
    diag16_pivot<0>(S, R, ok);
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[i * 16 + (col ^ i)] = R[i];
    }
    asm volatile("" ::: "memory");
    v4d w;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int row = g + 4 * q; w[q] = tile[row * 16 + (col ^ row)]; }
    asm volatile("" ::: "memory");
    return w;
}
// ... from a row-major 16 x 16 tile `in` (float or double, not swizzled; this wave's own writes, complete) into the swizzled tile `out` (may be the same memory when
// the types agree): the form the tile phases use -- the inverse of a diagonal tile is only ever read back from LDS.
template <typename TI, typename TO>
__device__ __forceinline__ void diag16_invert_dpp_tiles(const TI* in, TO* out, int lane, bool& ok) {
    const int col = lane & 15, g = lane >> 4;
    double S[16], R[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = (double)in[i * 16 + col]; R[i] = (i == col) ? 1.0 : 0.0; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the reads are done before `out` -- possibly the same tile -- is written)
    ok = true;
    diag16_pivot<0>(S, R, ok);
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i * 16 + (col ^ i)] = (TO)R[i];
    }
    asm volatile("" ::: "memory");
}

}  // namespace srbdqp

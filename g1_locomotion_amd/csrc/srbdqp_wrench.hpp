// srbdqp_wrench.hpp -- the general kernel ("wrench"): any contact pattern, horizons up to 24, fp64 or fp32 iterations.
//
// Presolve in two steps.  (1) Swing contacts are eliminated as in srbdqp_compact.hpp.  (2) A step's 3c stance-force
// variables act on the body only through the 6-vector wrench g = W u, W = [J_e ... ; I I ...] (angular acceleration
// I_w^-1 sum r x f, total force), and A' rho A is diagonal per contact (rho diag(2, 2, 4 mu^2 + rho_fz): the normal-force row has its own penalty), so the reduced-KKT matrix of
// the ADMM is  K = D + Y' S Y  with D DIAGONAL and S the Hessian in the per-step coordinates g_j: the 6 wrench
// coordinates when the step has >= 3 stance contacts (Y_j = W_j), the 3c force variables themselves otherwise (Y_j = I).
// Woodbury twice:
//     K^-1 = Bd + V' T^-1 V,   T = S + E^-1,  E = Y D^-1 Y' (6x6 per step),  V = E^-1 Y D^-1,  Bd = D^-1 - D^-1 Y' V.
// Only T (n_g x n_g, n_g <= 6N) is assembled, factored and inverted: full double support at N = 20 (BASELINE.json
// configs[2]) is a 120 x 120 problem instead of 240 x 240 -- 8x less factor work, 3x less work per ADMM iteration -- and
// T is far better conditioned than K (8e4 instead of 1e8 on that config).  Bd and V are 12 x 12 and 6 x 12 per step and
// live in the registers of the step's lanes.  oracle/srbd_oracle.py wrench_reduce() restates this in NumPy.
//
// Mapping: 12 lanes per horizon step, 5 steps per wave (lanes 60..63 idle), NW = ceil(N / 5) waves per QP.  Lane ul of a
// step group is at once force variable (contact ul / 3, axis ul % 3) -- a contact = 3 lanes carrying 2 + 2 + 1 rows, as in
// srbdqp_split.hpp -- and half h = ul % 2 of row ul / 2 of T^-1 for that step.  Per ADMM iteration the 12 right-hand
// sides of a step are exchanged inside the wave (no barrier), v = V w crosses the workgroup through a double-buffered
// LDS vector (ONE barrier per iteration), t = T^-1 v is a register mat-vec (a half row per lane), and
// x~ = x_q + Bd w + V' t is again local to the step.
//
// Precision (template parameters R, TT): the assembly (tables, E^-1, the entries of T) is always fp64.  R = float keeps
// T^-1, V, Bd and the iterates in fp32 and reads / writes fp32 buffers.  TT = the type of the 16 x 16 tiles T is stored,
// factored (MFMA 16x16x4 f64 or f32) and inverted in: double everywhere, or float for the fp32 calls' QPs whose steps all
// have 0 or >= 3 stance contacts (cond(T) ~ 5e4; half the LDS, three workgroups per CU at N = 20) -- x_q then gets one
// refinement step with the fp64 residual K x_q + q, and V / Bd are formed after the factorisation from the triangle of
// E^-1 kept in LDS.  What makes fp32 iterations converge at all is the split
// x~ = x_q + K^-1 (sigma x + A'(rho z - y)) with x_q = -K^-1 q computed ONCE in fp64: q is O(1e4) in the scaled
// variables and would otherwise drown the O(1) iterate in the mat-vec's cancellation; and the dual residual is
// tracked as c = P x + q (recursion c~ = sigma (x - x~) - A' nu), which never sees q either.  R = double runs the same
// recursions (the oracle twin is admm_solve_split()).
//
// The reference for all of this is the absent submodule g1_mpc; conventions from
// g1_mujoco_sim/src/run_simulation.py:73-111.
#pragma once
#include <type_traits>
#include "srbdqp_common.hpp"
#include "srbdqp_admm.hpp"
#include "srbdqp_mfma.hpp"

namespace srbdqp {

#ifndef SRBDQP_PHASE_LOCAL
#define SRBDQP_PHASE_LOCAL(...) asm volatile("" : __VA_ARGS__)
#endif

// TB = bytes per tile element: 8 (fp64 tiles) or 4 (fp32 tiles, double-support QPs of the fp32 path)
// SPW = horizon steps per wave: 5 (60 of 64 lanes busy)
// XW  = extra waves of the workgroup that take part in the set-up only (tables, T assembly, the tile phases F / W / I) and end
//       before x_q and the iterations: the low-latency instantiation of the staged batch-1 path (N = 10: 2 + 2 waves).  The
//       tile phases want as many waves as there are tiles; the iteration wants as FEW waves as hold its lanes -- every wave
//       reads the whole vector v from LDS for its T^-1 rows, so with the steps spread over 4 waves (3 per wave, tried first) the
//       broadcast reads alone took ~550 of an iteration's ~1950 cycles (tools/wrench_stamps_staged.py, -DSRBDQP_PROFILE_WADMM)
template <int N, int TB = 8, int SPW = 5, int XW = 0>
struct WrenchSmem {
    static_assert(SPW >= 1 && SPW <= 5, "12 lanes per step");
    static constexpr int n = 12 * N, m = 20 * N;
    static constexpr int NWS = (N + SPW - 1) / SPW;       // waves that carry horizon steps (and run x_q, the iterations, the roll-out)
    static constexpr int NW = NWS + XW;                   // waves of the workgroup (tile phases)
    static constexpr int BT = 64 * NW;
    static constexpr int LT = 64 * NWS;                   // threads alive behind the set-up
    static constexpr int NG = 6 * N;                      // upper bound of n_g
    static constexpr int NT = (NG + 15) / 16;
    static constexpr int NTT = NT * (NT + 1) / 2;
    static constexpr int TS = (NTT + NW - 1) / NW;        // tile slots per wave
    static constexpr int WQ = (NT + NW - 1) / NW;         // W phase: tiles of a block row per wave
    static constexpr int CHMAX = 2 * ((NG + 3) / 4);      // columns per half row of T^-1 (even)
    static constexpr int NPAIR = N * (N + 1) / 2;
    static constexpr int up2(int v) { return (v + 1) & ~1; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // ---- persistent strip
    static constexpr int o_x0 = 0;                        // 13 (+1)
    static constexpr int o_tm = o_x0 + 14;                // 9N   Rz' per step
    static constexpr int o_J = o_tm + up2(N * 9);         // 36N  I_w^-1 [r]x per step
    static constexpr int o_red = o_J + N * 36;            // 32   reductions / check maxima / vote flags
    static constexpr int o_ct = o_red + 32;               // 4N bytes of contact flags
    static constexpr int o_misc = o_ct + up2((N * 4 + 7) / 8);   // [0] numerical failure
    static constexpr int o_sq = o_misc + 4;               // 12   sqrt(q_diag)
    static constexpr int o_int = o_sq + 12;               // ints: gsz[N], goff[N + 1], n_g, na, wrench flag[N]
    static constexpr int o_R = o_int + up2((3 * N + 6) / 2 + 1);
    // ---- phase C (ADMM vectors; element type R <= 8 bytes, offsets in doubles)
    static constexpr int VB = 2 * CHMAX + 8;              // one v buffer (elements)
    static constexpr int o_wb = o_R;                      // NW x 64   right-hand sides, wave private
    static constexpr int o_tb = o_wb + 64 * NW;           // NW x 32   t of the wave's steps, wave private
    static constexpr int o_vb = o_tb + 32 * NW;           // 2 x VB    v = V w, double buffered
    static constexpr int o_xs = o_vb + 2 * VB;            // n         full solution (scaled) for the roll-out
    static constexpr int o_scr = o_xs + n;                // 12N       roll-out scratch
    static constexpr int endC = o_scr + n + 2;
    // ---- phase A (closed-form assembly)
    static constexpr int o_xref = o_R;                    // 13N
    static constexpr int o_foot = o_xref + up2(N * 13);   // 12N
    static constexpr int o_pcom = o_foot + N * 12;        // 3N
    static constexpr int o_cp = o_pcom + up2(N * 3);      // 9N   prefix sums of Rz'
    static constexpr int o_eh = o_cp + up2(N * 9);        // 12N  Q^1/2 (A_qp x0 - x_ref)
    static constexpr int o_t1 = o_eh + n;                 // 9N
    static constexpr int o_t2 = o_t1 + up2(9 * N);        // 9N
    static constexpr int o_mt = o_t2 + up2(9 * N);        // 18N: D_m, E_m (de_tables); the region keeps the 9 NPAIR doubles the M(j, m) table had -- the T assembly's rows follow in it (o_ab)
    static constexpr int o_gv = o_mt + up2(9 * NPAIR);    // 9N   G'v tables
    // 12N  G x^0 (P x of the warm start, of the refinement).  fp32 tiles: both uses come behind the tile phases (WARM_LATE), so the array lives
    // in the dead tile region then (behind the parked V rows, in front of the G'v tables) and takes no room in phase A
    static constexpr bool GX_LATE = TB == 4 && (up2(endC) + 7 * n <= o_gv);
    static constexpr int o_gx = GX_LATE ? up2(endC) + 6 * n : o_gv + up2(9 * N);
    static constexpr int o_tf = GX_LATE ? o_gv + up2(9 * N) : o_gx + n;   // 6N
    static constexpr int o_x0c = o_eh;                    // 12N  warm start in the scaled variables (the gradient is done with o_eh by then)
    static constexpr int o_ab = o_mt + 18 * N;            // (see below)
    static constexpr int o_bb = o_ab + 8 * 16 * NT;       // ... and, where they fit in front of the 6-vectors without growing the layout (N >= 16), the columns' [D_m z_ang, E_m z_ang] rows
    static constexpr bool BROWS = o_bb + 6 * 16 * NT <= o_tf + 6 * N;
    static constexpr int o_zt = cmax(o_tf + 6 * N, o_ab + 8 * 16 * NT);   // 6 NG: wrench-space 6-vector of every g coordinate (N = 4: behind the T assembly's rows)
    static constexpr int o_ei = o_zt + 6 * NG;            // 36N  E^-1 per step
    // rank-6 T assembly (round 4): D_m, E_m (18N doubles, srbdqp_common.hpp de_tables) at o_mt where the 9 NPAIR doubles of M(j, m) were, and behind them -- over
    // the rest of that table, the G'v tables and the warm-start vectors, all dead by then -- one 8-double row per g coordinate [z_ang(3), -C_j z_ang(3), z_lin, 4096 j + 36 j + 6 l]
    static_assert(o_ab + 8 * 16 * NT <= o_zt, "the T assembly's rows fit between the D / E tables and the 6-vectors");
    static constexpr int o_gs = o_ei + 36 * N;            // NG bytes: step of every g coordinate
    static constexpr int endA = o_gs + up2((NG + 7) / 8);
    // ---- phase B (tiles)
    static constexpr int o_T = o_R;
    static constexpr int o_ws = o_T + NTT * 256 * TB / 8;   // fp32 tiles: one 16 x 16 scratch tile per wave (operand re-layout)
    // fp64 iterations with a half row longer than 60 (N = 24): its last KTAIL entries per lane, entry-major [KTAIL][BT] (the tiles
    // are dead by then and their region is far larger)
#ifndef SRBDQP_WRENCH_KTAIL36
#define SRBDQP_WRENCH_KTAIL36 12
#endif
#ifndef SRBDQP_WRENCH_VLDS
#define SRBDQP_WRENCH_VLDS 1
#endif
#ifndef SRBDQP_WRENCH_JLDS
#define SRBDQP_WRENCH_JLDS 1
#endif
#ifndef SRBDQP_WRENCH_KREG64
#define SRBDQP_WRENCH_KREG64 56   // entries of the fp64 half row kept in registers when it is longer than 60 (N = 24 mixed gait: 56 -> 1.06 M QP/s with 2 reloads from scratch left in the iteration, 48 -> 1.03 M with none, 40 -> 1.01 M)
#endif
    static constexpr int KTAIL = (TB == 8 && CHMAX > 60) ? CHMAX - SRBDQP_WRENCH_KREG64 : ((TB == 8 && CHMAX == 36) ? SRBDQP_WRENCH_KTAIL36 : 0);   // N = 12: 3 waves per SIMD
    static constexpr int o_kt = up2(endC);
    static constexpr int o_vl = o_kt + KTAIL * BT;        // VL: row and column of V per lane, entry-major [12][BT]
    static constexpr int endC2 = o_vl + ((TB == 8 && SRBDQP_WRENCH_VLDS && CHMAX <= 36) ? 12 * BT : 0);
    static constexpr int wgs_of(int doubles) { return 163840 / (((doubles * 8 + 1279) / 1280) * 1280); }   // (LDS is handed out in blocks of 320 dwords on gfx950)
    // fp32 tiles: the lower triangle of E^-1 per step (21 N doubles: the very values that went into T -- V and Bd formed from a
    // rounded copy break the Woodbury identity 100 times worse than rounding V and Bd themselves), written in phase E and kept
    // through phases F / W / I behind the tiles AND behind the phase-A arrays; the rows of V and Bd are formed from it afterwards
    // (round 5: the same late formation for the fp64-tile instantiations wherever the triangle fits WITHOUT costing a workgroup per CU -- their V / Bd rows were
    //  the 116 - 312 bytes per lane these kernels kept in scratch memory across the tile phases: 5 x the algorithmic HBM traffic on the N = 12 bucket of configs[4])
    static constexpr int o_e4x = cmax(o_ws, endA);
    static constexpr bool E4_FITS = TB == 8 && XW == 0 && wgs_of(cmax(cmax(endA, endC2), o_e4x + 21 * N)) == wgs_of(cmax(cmax(endA, endC2), o_ws));
    static constexpr int o_e4 = TB == 4 ? cmax(o_ws + NW * 128, endA) : (E4_FITS ? o_e4x : o_ws);
    // ... and the gradient of every variable (fp32 tiles: 168 registers per lane -- it waited in scratch memory from the tables to x_q)
    // (only where it costs no occupancy: the fp32-tile kernels are compiled for 3 waves per SIMD = 12 waves per CU)
    static constexpr int o_qv = o_e4 + ((TB == 4 || E4_FITS) ? 21 * N : 0);
    static constexpr bool QV_PARK = TB == 4 && (wgs_of(cmax(endA, o_qv + n / 2)) * NW >= 12 || wgs_of(cmax(endA, o_qv + n / 2)) == wgs_of(cmax(endA, o_qv)));
    static constexpr int endB = o_qv + (QV_PARK ? n / 2 : 0);   // (n floats: the leading fp32 part; the remainder -- one register -- stays with the lane)
    // fp32 tiles, long horizons: the lane's fp64 row and column of V wait in the dead tile region while x_q and its refinement run
    // (entry-major [6][n] each, indexed by the lane's variable; two free regions: behind the ADMM vectors up to the G'v tables the
    // refinement still needs, and the 6-vectors + E^-1 blocks of the assembly) -- they were the larger half of the kernel's spills
#ifndef SRBDQP_WRENCH_VPARK
#define SRBDQP_WRENCH_VPARK 1
#endif
    static constexpr int o_vpr = up2(endC);
    static constexpr int o_vpc = o_zt;
    static constexpr bool VPARK = TB == 4 && SRBDQP_WRENCH_VPARK && (o_gv - o_vpr >= 6 * n) && (o_vpc + 6 * n <= o_e4);
    static_assert(!GX_LATE || o_gx == o_vpr + 6 * n, "fp32 tiles: G x^0 right behind the parked V rows");
    static constexpr int o_pre = cmax(endA, cmax(endB, endC2));   // low-latency instantiation: the scratch tile of the diagonal tile that is inverted beside the assembly (the tile store still holds tables then)
    static constexpr int o_hand = o_pre + (XW > 0 ? 256 : 0);   // ... and, with the tile phases pipelined (XW = 2, below), the two tiles assembled for wave 0 by waves 1 and 3
    static constexpr int o_abx = o_hand + (XW == 2 ? 512 : 0);  // ... and the T assembly's row / column tables, built by a set-up helper beside phase E (they overlay live tables in their batch-kernel place)
    static constexpr int o_bbx = o_abx + (XW == 2 ? 8 * 16 * NT : 0);
    static constexpr int o_end = o_bbx + (XW == 2 ? 6 * 16 * NT : 0);
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
    static constexpr int lds_wgs = wgs_of(o_end);
};

// diagnostic builds (-DSRBDQP_PROFILE_WADMM): s_memtime stamps inside the ADMM iteration of the general kernel, summed per
// segment by thread 0 and written to the second row of the stamp buffer of a B = 1 solve (tools/wrench_stamps_staged.py prints them)
// diagnostic builds (-DSRBDQP_XQ_STAMPS): stamps 10 / 11 / 12 inside "fragments + x_q" (tools/wrench_stamps.py prints them): behind the half rows, behind x_q, behind G'(G x_q)
#ifdef SRBDQP_XQ_STAMPS
#define XQSTAMP(a, b, i) SRBDQP_STAMP(a, b, i)
#else
#define XQSTAMP(a, b, i) do { } while (0)
#endif
#ifdef SRBDQP_PROFILE_WADMM
#define WADMM_T(i) do { if (wadmm_t) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); wadmm_t[i] = (long long)t_; } } while (0)
#define WADMM_DECL long long wadmm_tt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wadmm_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long* wadmm_t = wadmm_tt
#define WADMM_ARGS , wadmm_t
#define WADMM_PARAMS , long long* wadmm_t = nullptr
#else
#define WADMM_T(i) do { } while (0)
#define WADMM_DECL do { } while (0)
#define WADMM_ARGS
#define WADMM_PARAMS
#endif

// ---- small helpers on the iteration type ----------------------------------------------------------------------------
__device__ __forceinline__ float dpp_swap1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float bperm(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
__device__ __forceinline__ double bperm(double v, int src_lane) { return bperm_f64(v, src_lane); }
__device__ __forceinline__ float rmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double rmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double rmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float rabs(float a) { return fabsf(a); }
__device__ __forceinline__ double rabs(double a) { return fabs(a); }

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4d mma16(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f mma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// row of accumulator register q in the MFMA C/D layout: fp64 16x16x4: (lane >> 4) + 4 q; fp32 16x16x4: 4 (lane >> 4) + q
template <typename TT> __device__ __forceinline__ int crow(int kq, int q) { return (sizeof(TT) == 8) ? kq + 4 * q : 4 * kq + q; }
template <typename TT, bool SWZ, typename V4>
__device__ __forceinline__ void store_tile_t(TT* tile, const V4& v, int lane) {
    const int col = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = crow<TT>(g, q);
        tile[row * 16 + (SWZ ? (col ^ row) : col)] = v[q];
    }
}

// workgroup-wide max of one non-negative value (NW waves); two barriers
template <int NW>
__device__ __forceinline__ double wg_max1(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double x = wave_max_nonneg(v);
    if (lane == 63) red[wave] = x;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmax(r, red[w]);
    __syncthreads();
    return r;
}

// x~-part of one application of K^-1 = Bd + V' T^-1 V on the step-group mapping; all exchange through LDS.
//   wv      this lane's right-hand side (0 on inactive lanes)
//   returns (K^-1 w) for the variable of this lane
// Used once in fp64 for x_q and every iteration in R.  `vb` is the v buffer to publish into; a workgroup barrier sits
// between the publication of v and its use.  hook() runs right after that barrier (the ADMM loop reads the convergence
// decision there).
// bj: the block-diagonal part Bd of K^-1, in one of two forms.
//   BDN = 12: the lane's row of Bd over its step's 12 variables, explicit (formed in fp64 at set-up) -- the fp32 iterations:
//             the implicit form below subtracts two nearly equal fp32 numbers every iteration and left 0.3 % of configs[2]
//             short of the 2e-6 tolerance (0.06 % with the explicit rows);
//   BDN = 4:  {J[0..2][u], 1 / D_u}: Bd w = D^-1 (w - Y' v) from the step's own v = V w -- the fp64 iterations: 8 registers
//             instead of 24, 6 multiply-adds instead of 12, and only the lane's half of the step's right-hand sides is read.
// KREG < CHMAX (fp64 iterations at N = 24): the last CHMAX - KREG entries of the lane's T^-1 half row are read from LDS
// (ktail[(c - KREG) kts], lane-contiguous per entry) instead of registers -- a 5-wave workgroup puts two waves on one SIMD,
// so a wave has 256 registers, and the 72-double half row + V + state spilled 8 values per iteration to scratch memory.
// VL (fp64 iterations of the small instantiations, 3 waves per SIMD): the lane's row and column of V are read from LDS
// (vlds[i kts] = vrow[i], vlds[(6 + i) kts] = vcol[i], lane-contiguous per entry) instead of 24 registers.
// WIDE (the low-latency instantiation: one workgroup's worth of registers): every broadcast read of the T^-1 product in flight
// at once instead of blocks of four (each block was one more LDS round trip in the iteration's chain), and the 6-term
// products on two accumulators.
template <typename R, int CHMAX, int KREG = CHMAX, bool VL = false, bool WIDE = false, bool JL = false, typename KT, int BDN, class Hook>
__device__ __forceinline__ R apply_kinv(R wv, R* wbw, R* tbw, R* vb, int lane, int sg, int ul, bool active_g, int Rrow, int CH,
                                        const KT (&kin)[CHMAX], const R (&vrow_)[6], const R (&vcol_)[6], const R (&bj)[BDN], int vsoff, int vssel,
                                        Hook&& hook, const R* ktail = nullptr, int kts = 0, const R* vlds = nullptr, const R* vldc = nullptr, const double* jl = nullptr WADMM_PARAMS) {
    auto KIN = [&](int c) -> R { return (c < KREG) ? (R)kin[c] : ktail[(c - KREG) * kts]; };
    R vrow[6], vcol[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { vrow[i] = VL ? vlds[i * kts] : vrow_[i]; }
    static_assert(BDN == 4 || BDN == 12, "Bd: implicit (4) or explicit row (12)");
    typedef R R4 __attribute__((ext_vector_type(4)));
    typedef R R2 __attribute__((ext_vector_type(2)));
    const int h = ul & 1;
    wbw[lane] = wv;
    asm volatile("" ::: "memory");
    R xb;
    if constexpr (BDN == 4) {
        R wg[6];                                            // the lane's half (h) of its step's 12 right-hand sides
        {
            const R* src = wbw + 12 * sg + 6 * h;           // 8-byte aligned for float (24 B), 16-byte for double (48 B)
            const R2* s2 = reinterpret_cast<const R2*>(src);
#pragma unroll
            for (int i = 0; i < 3; ++i) { const R2 v = s2[i]; wg[2 * i] = v[0]; wg[2 * i + 1] = v[1]; }
        }
        R vp = R(0);
        if constexpr (WIDE) {
            R vq = R(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) { vp = fma(vrow[2 * i], wg[2 * i], vp); vq = fma(vrow[2 * i + 1], wg[2 * i + 1], vq); }
            vp += vq;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) vp = fma(vrow[i], wg[i], vp);
        }
        const R v = vp + dpp_swap1(vp);
        if (active_g && h == 0) vb[Rrow] = v;
        asm volatile("" ::: "memory");
        // the g rows of the lane's own step were written by this wave just above (LDS operations of one wave complete in
        // order): no barrier needed.  Y' column of the variable: [J[:, u]; e_ax] on a wrench step, the unit vector of its g
        // row on a force-variable step (where the whole term is w - w = 0).
        const R* vs = vb + vsoff;
        if constexpr (JL) {   // (the iterations of the fp64 N = 12 instantiation: +1.5 %; N = 8: -2 %, N = 10: nothing)
              // ... J[:, u] from the persistent strip (jl[0], jl[12], jl[24]) instead of three more registers: they were reloaded from scratch
            xb = bj[3] * (wv - fma(jl[0], vs[0], fma(jl[12], vs[1], fma(jl[24], vs[2], vb[vssel]))));   // memory every iteration (a force-variable step has bj[3] = 0: its J does not matter)
        } else {
            xb = bj[3] * (wv - fma(bj[0], vs[0], fma(bj[1], vs[1], fma(bj[2], vs[2], vb[vssel]))));
        }
    } else {
        R wg[12];
        {
            const R* src = wbw + 12 * sg;                   // 12 sg elements: 16-byte aligned for float (48 B) and double (96 B)
            if constexpr (sizeof(R) == 4) {
                const R4* s4 = reinterpret_cast<const R4*>(src);
#pragma unroll
                for (int i = 0; i < 3; ++i) { const R4 v = s4[i]; wg[4 * i] = v[0]; wg[4 * i + 1] = v[1]; wg[4 * i + 2] = v[2]; wg[4 * i + 3] = v[3]; }
            } else {
                const R2* s2 = reinterpret_cast<const R2*>(src);
#pragma unroll
                for (int i = 0; i < 6; ++i) { const R2 v = s2[i]; wg[2 * i] = v[0]; wg[2 * i + 1] = v[1]; }
            }
        }
        // (2-vectors = packed fp32 instructions; in fp64 they only lengthen the dependency chains: N = 24 ran 12 % slower)
        R vp;
        if constexpr (sizeof(R) == 4) {
            R2 vp2 = (R2){R(0), R(0)};
#pragma unroll
            for (int i = 0; i < 3; ++i)
                vp2 = __builtin_elementwise_fma((R2){vrow[2 * i], vrow[2 * i + 1]}, h ? (R2){wg[6 + 2 * i], wg[7 + 2 * i]} : (R2){wg[2 * i], wg[2 * i + 1]}, vp2);
            vp = vp2[0] + vp2[1];
        } else {
            vp = R(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) vp = fma(vrow[i], h ? wg[6 + i] : wg[i], vp);
        }
        const R v = vp + dpp_swap1(vp);
        if (active_g && h == 0) vb[Rrow] = v;
        if constexpr (sizeof(R) == 4) {
            R2 xb2 = (R2){R(0), R(0)};
#pragma unroll
            for (int i = 0; i < 6; ++i) xb2 = __builtin_elementwise_fma((R2){bj[2 * i], bj[2 * i + 1]}, (R2){wg[2 * i], wg[2 * i + 1]}, xb2);
            xb = xb2[0] + xb2[1];
        } else {
            xb = R(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) xb = fma(bj[i], wg[i], xb);
        }
    }
    WADMM_T(1);
    __syncthreads();
    WADMM_T(2);
    hook();
    R tp;
    {
        R acc0 = R(0), acc1 = R(0), acc2 = R(0), acc3 = R(0);
        const R* src = vb + CH * h;                         // CH even; VB multiple of 8: 16-byte aligned for double, 8-byte for float
        if constexpr (sizeof(R) == 4) {
            // columns past CH read the zero padding of the buffer and meet kin = 0
            // packed fp32 multiply-adds (v_pk_fma_f32: two lanes' worth of FMA per issue slot)
            R2 a01 = (R2){R(0), R(0)}, a23 = (R2){R(0), R(0)};
            if ((CH & 3) == 0 && (CHMAX & 3) == 0) {            // wave-uniform: the second half starts 16-byte aligned
                const R4* s4 = reinterpret_cast<const R4*>(src);
#pragma unroll
                for (int c = 0; c < CHMAX / 4; ++c) {
                    const R4 vv = s4[c];
                    a01 = __builtin_elementwise_fma((R2){(R)kin[4 * c], (R)kin[4 * c + 1]}, (R2){vv[0], vv[1]}, a01);
                    a23 = __builtin_elementwise_fma((R2){(R)kin[4 * c + 2], (R)kin[4 * c + 3]}, (R2){vv[2], vv[3]}, a23);
                }
            } else {
                const R2* s2 = reinterpret_cast<const R2*>(src);
#pragma unroll
                for (int c = 0; c < CHMAX / 2; ++c) {
                    const R2 vv = s2[c];
                    if (c & 1) a23 = __builtin_elementwise_fma((R2){(R)kin[2 * c], (R)kin[2 * c + 1]}, vv, a23);
                    else a01 = __builtin_elementwise_fma((R2){(R)kin[2 * c], (R)kin[2 * c + 1]}, vv, a01);
                }
            }
            acc0 = a01[0]; acc1 = a01[1]; acc2 = a23[0]; acc3 = a23[1];
        } else {
            const R2* s2 = reinterpret_cast<const R2*>(src);
            constexpr int NV = CHMAX / 2, BL = (WIDE || NV <= 4) ? NV : ((sizeof(KT) < sizeof(R)) ? 2 : 4), NB = (NV + BL - 1) / BL;   // (fp64 on an fp32 row -- x_q of the fp32-tile kernel, 168 registers: two reads in flight)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (blk == 0 || blk * 2 * BL < CH) {
                    R2 vv[BL];
#pragma unroll
                    for (int i = 0; i < BL; ++i) vv[i] = (blk * BL + i < NV) ? s2[blk * BL + i] : (R2){R(0), R(0)};
#pragma unroll
                    for (int i = 0; i < BL; ++i) {
                        const int c0 = 2 * (blk * BL + i);
                        if (c0 + 1 < CHMAX) {
                            if (i & 1) { acc2 = fma(KIN(c0), vv[i][0], acc2); acc3 = fma(KIN(c0 + 1), vv[i][1], acc3); }
                            else { acc0 = fma(KIN(c0), vv[i][0], acc0); acc1 = fma(KIN(c0 + 1), vv[i][1], acc1); }
                        }
                    }
                }
            }
        }
        tp = (acc0 + acc1) + (acc2 + acc3);
    }
    const R tv = tp + dpp_swap1(tp);
    WADMM_T(3);
    if (h == 0) tbw[6 * sg + (ul >> 1)] = active_g ? tv : R(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 6; ++i) { vcol[i] = VL ? (vldc ? vldc[i * kts] : vlds[(6 + i) * kts]) : vcol_[i]; }
    R xt;
    {
        const R* src = tbw + 6 * sg;                        // 6 sg elements: 8-byte aligned (float), 16-byte (double)
        const R2* s2 = reinterpret_cast<const R2*>(src);
        if constexpr (sizeof(R) == 4) {
            R2 x2 = (R2){xb, R(0)};
#pragma unroll
            for (int i = 0; i < 3; ++i) x2 = __builtin_elementwise_fma((R2){vcol[2 * i], vcol[2 * i + 1]}, s2[i], x2);
            xt = x2[0] + x2[1];
        } else {
            xt = xb;
            if constexpr (WIDE) {
                R xu = R(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) { const R2 vv = s2[i]; xt = fma(vcol[2 * i], vv[0], xt); xu = fma(vcol[2 * i + 1], vv[1], xu); }
                xt += xu;
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) { const R2 vv = s2[i]; xt = fma(vcol[2 * i], vv[0], xt); xt = fma(vcol[2 * i + 1], vv[1], xt); }
            }
        }
    }
    asm volatile("" ::: "memory");
    return xt;
}

// ---- the tile phases of the low-latency instantiation as ONE static pipeline on four waves (round 5) -------------------------------------------------------
// A batch-1 solve has the CU to itself: what counts is the CHAIN of the factorisation -- invert diagonal tile j, form the one panel tile (j, j + 1), take its
// square off diagonal tile j + 1, invert that -- not the work.  Phases F, W, I one after the other were 29 k cycles of a 119 k solve (F 18.8 k: per block
// column an inversion, a barrier, the panels, a barrier and the owner's share of the trailing updates; then W 5.0 k and I 5.2 k, block row by block row with two
// barriers each).  Here wave j inverts diagonal tile (j, j) and owns the tile above it, (j - 1, j), so a block column costs the chain
//     t1(J)    the panel U_J,J+1 = L_JJ^-1 T_J,J+1 and D_J+1 -= U'U straight from the owner's registers (an accumulator register is at once the A and the
//              B operand of that product), beside the other waves' panels                                                              -> barrier B
//     t2(J)    the inversion of diagonal tile J + 1, in two halves around barrier M                                                     -> barrier A
// and everything else rides under the inversions, on the three waves that are not inverting: the trailing updates of column J and the closing product of
// block row J of W = L^-1 in front of M (res = -W_JJ o, o formed one column earlier); o of block row J + 1 and term J of T^-1 = W'W behind it.  A 16 x 16 x 16
// product is four v_mfma_f64_16x16x4 of 64 cycles each on the wave's SIMD, so the schedule is STATIC -- one straight-line block per (wave, column), no loop, no
// runtime tile tables: the loads of a block's products are in flight together and the MFMAs run back to back -- and balanced: at most four products per wave
// behind an M, against the ~1.6 k cycles the second half of an inversion lasts.  (A first version with a loop over the columns and run-time ownership tables
// spent 500 - 800 cycles per product and left 1 - 1.3 k cycles per column exposed: F + W + I 23.5 k; tools/latp_stamps.py.)
//   F tiles (accumulators):   wave 0: (0,0) (0,2) (1,3)    wave 1: (1,1) (0,1)    wave 2: (2,2) (1,2) (0,3)    wave 3: (3,3) (2,3)
//   W tiles (o -> res):       row 1: (0,1) wave 0;   row 2: (0,2) wave 0, (1,2) wave 1;   row 3: (2,3) wave 0, (1,3) wave 1, (0,3) wave 2
//   T^-1 tiles (accumulators): wave 0: (0,0) (0,2) (1,2)    wave 1: (0,1) (1,1)    wave 2: (2,2) (3,3)    wave 3: (0,3) (1,3) (2,3)
// NTC = number of 16 x 16 block columns (the schedule of fewer columns is the same one without the tiles that do not exist).  Same products, same operand order
// as the phases of the batch instantiations: the tiles of T^-1 agree to rounding.  The slots of the tile store hold, in turn, U_ab, W_b<-a and T^-1_ab.
#ifndef SRBDQP_LATP_SPLIT
#define SRBDQP_LATP_SPLIT 5
#endif
#ifdef SRBDQP_LATP_STAMPS
#define ESTAMP(a, i) do { if ((a).stamps && (a).B == 1 && threadIdx.x == 0) (a).stamps[16 * 5 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ESTAMP(a, i) do { } while (0)
#endif
#ifdef SRBDQP_LATP_STAMPS   // (diagnostic builds: when wave w arrives at a join of the set-up -- rows 1 + w of the stamp buffer of a B = 1 solve)
#define WAVE_ARRIVE(a, w, lane, i) do { if ((a).stamps && (a).B == 1 && (lane) == 0 && (w) < 4) (a).stamps[16 * (1 + (w)) + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WAVE_ARRIVE(a, w, lane, i) do { } while (0)
#endif
#ifdef SRBDQP_LATP_STAMPS   // diagnostic builds: when each wave ARRIVES at the barriers of the pipeline (rows 1 + wave of the stamp buffer of a B = 1 solve; tools/latp_stamps.py)
#define LBAR(i) do { if (stamps && lane == 0) stamps[16 * (1 + W) + (i)] = (long long)__builtin_amdgcn_s_memtime(); __syncthreads(); } while (0)
#else
#define LBAR(i) __syncthreads()
#endif
template <int W, int NTC>
__device__ __forceinline__ void latp_run(double* T, const double* D0, v4d& acc0, v4d& acc1, v4d& acc2, double* misc, const int lane, long long* stamps) {
    (void)stamps;
    constexpr int KSPLIT = SRBDQP_LATP_SPLIT;
    const int mcol = lane & 15, kq = lane >> 4;
    const v4d zero4 = (v4d){0.0, 0.0, 0.0, 0.0};
    auto TL = [&](int a_, int b_) __attribute__((always_inline)) -> double* { return T + tile_id(a_, b_) * 256; };
    // o = sign * Wjj * reg   (Wjj = L_jj^-1 from its swizzled slot; reg = a C-layout register tile: register r is the B operand of K-step r)      panel, closing product of a W tile
    auto p_dinv_reg = [&](const double* D, const v4d& reg, const double sign) __attribute__((always_inline)) -> v4d {
        v4d o = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kq + 4 * r;
            double av = D[mcol * 16 + (k ^ mcol)];
            av = (k <= mcol) ? sign * av : 0.0;
            o = mma16(av, reg[r], o);
        }
        return o;
    };
    // acc += sign * A' B, both row-major tiles of the store                                                                                    trailing update, W'W term, U W term
    auto p_at_b = [&](v4d& acc, const double* A, const double* B, const double sign) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int k = 4 * r + kq; acc = mma16(sign * A[k * 16 + mcol], B[k * 16 + mcol], acc); }
    };
    // acc += A' Wjj (A row-major tile, or null: Wjj' Wjj)                                                                                      first product of a W tile, last-row term of T^-1
    auto p_at_d = [&](v4d& acc, const double* A, const double* D) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * r + kq;
            double dv = D[k * 16 + (mcol ^ k)];
            dv = (mcol <= k) ? dv : 0.0;
            const double av = A ? A[k * 16 + mcol] : dv;
            acc = mma16(av, dv, acc);
        }
    };
    auto p_reg_sq = [&](v4d& acc, const v4d& reg) __attribute__((always_inline)) {      // acc -= reg' reg from registers
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mma16(-reg[r], reg[r], acc);
    };
    double dS[16], dR[16];
    bool dok = true;
    auto inv_first = [&](double* slot) __attribute__((always_inline)) {     // the raw diagonal tile is in its slot (this wave's own store, complete behind barrier B)
        const int col = lane & 15;
#pragma unroll
        for (int i = 0; i < 16; ++i) { dS[i] = slot[i * 16 + col]; dR[i] = (i == col) ? 1.0 : 0.0; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        diag16_pivot<0, KSPLIT>(dS, dR, dok);
    };
    auto inv_second = [&](double* slot) __attribute__((always_inline)) {
        diag16_pivot<KSPLIT, 16>(dS, dR, dok);
        const int col = lane & 15;
        if ((lane >> 4) == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) slot[i * 16 + (col ^ i)] = dR[i];
        }
        if (!dok && lane == 0) misc[0] = 1.0;
    };
    // panel of this wave's super-diagonal tile, its square off the wave's diagonal tile, the diagonal tile into its slot for the inversion
    auto chain_t1 = [&](int j) __attribute__((always_inline)) {
        acc1 = p_dinv_reg(j == 0 ? D0 : TL(j, j), acc1, 1.0);
        store_tile_t<double, false>(TL(j, j + 1), acc1, lane);
        p_reg_sq(acc0, acc1);
        store_tile_t<double, false>(TL(j + 1, j + 1), acc0, lane);
    };
    v4d i0 = zero4, i1 = zero4, i2 = zero4, wo = zero4;
    // (no barrier here: the inverse of tile (0, 0) -- D0 -- sits in wave 0's scratch tile since before the barrier that ended the assembly, and is read there)
    // ================================ column 0
    if constexpr (NTC > 1) {
        if constexpr (W == 0 && NTC > 2) { acc1 = p_dinv_reg(D0, acc1, 1.0); store_tile_t<double, false>(TL(0, 2), acc1, lane); }
        if constexpr (W == 1) chain_t1(0);
        if constexpr (W == 2 && NTC > 3) { acc2 = p_dinv_reg(D0, acc2, 1.0); store_tile_t<double, false>(TL(0, 3), acc2, lane); }
        LBAR(1);                                                 // B0
    }
    if constexpr (NTC > 1 && W == 1) inv_first(TL(1, 1));
    else {
        if constexpr (W == 0 && NTC > 3) p_at_b(acc2, TL(0, 1), TL(0, 3), -1.0);
        if constexpr (W == 2 && NTC > 2) { p_at_b(acc1, TL(0, 1), TL(0, 2), -1.0); p_at_b(acc0, TL(0, 2), TL(0, 2), -1.0); }
        if constexpr (W == 3 && NTC > 3) { p_at_b(acc1, TL(0, 2), TL(0, 3), -1.0); p_at_b(acc0, TL(0, 3), TL(0, 3), -1.0); }
    }
    LBAR(2);                                                     // M0
    if constexpr (NTC > 1 && W == 1) inv_second(TL(1, 1));
    else if constexpr (W == 0) {
        if constexpr (NTC > 1) p_at_d(wo, TL(0, 1), D0);
        p_at_d(i0, nullptr, D0);
    }
    LBAR(3);                                                     // A1
    // ================================ column 1
    if constexpr (NTC > 1) {
        if constexpr (NTC > 2) {
            if constexpr (W == 0 && NTC > 3) { acc2 = p_dinv_reg(TL(1, 1), acc2, 1.0); store_tile_t<double, false>(TL(1, 3), acc2, lane); }
            if constexpr (W == 2) chain_t1(1);
            LBAR(4);                                             // B1
        }
        if constexpr (NTC > 2 && W == 2) inv_first(TL(2, 2));
        else {
            if constexpr (W == 0) { const v4d res = p_dinv_reg(TL(1, 1), wo, -1.0); store_tile_t<double, false>(TL(0, 1), res, lane); }
            if constexpr (W == 3 && NTC > 3) { p_at_b(acc1, TL(1, 2), TL(1, 3), -1.0); p_at_b(acc0, TL(1, 3), TL(1, 3), -1.0); }
        }
        LBAR(5);                                                 // M1
        if constexpr (NTC > 2 && W == 2) inv_second(TL(2, 2));
        else {
            if constexpr (W == 0) {
                if constexpr (NTC > 2) { wo = zero4; p_at_d(wo, TL(0, 2), D0); p_at_b(wo, TL(1, 2), TL(0, 1), 1.0); }
                p_at_b(i0, TL(0, 1), TL(0, 1), 1.0);
            }
            if constexpr (W == 1) {
                if constexpr (NTC > 2) p_at_d(wo, TL(1, 2), TL(1, 1));
                p_at_d(i0, TL(0, 1), TL(1, 1));
                p_at_d(i1, nullptr, TL(1, 1));
            }
        }
        LBAR(6);                                                 // A2
    }
    // ================================ column 2
    if constexpr (NTC > 2) {
        if constexpr (NTC > 3) {
            if constexpr (W == 3) chain_t1(2);
            LBAR(7);                                             // B2
        }
        if constexpr (NTC > 3 && W == 3) inv_first(TL(3, 3));
        else {
            if constexpr (W == 0) { const v4d res = p_dinv_reg(TL(2, 2), wo, -1.0); store_tile_t<double, false>(TL(0, 2), res, lane); }
            if constexpr (W == 1) { const v4d res = p_dinv_reg(TL(2, 2), wo, -1.0); store_tile_t<double, false>(TL(1, 2), res, lane); }
        }
        LBAR(8);                                                 // M2
        if constexpr (NTC > 3 && W == 3) inv_second(TL(3, 3));
        else {
            if constexpr (W == 0) {
                if constexpr (NTC > 3) { wo = zero4; p_at_d(wo, TL(2, 3), TL(2, 2)); }
                p_at_b(i0, TL(0, 2), TL(0, 2), 1.0);
                p_at_d(i1, TL(0, 2), TL(2, 2));
                p_at_d(i2, TL(1, 2), TL(2, 2));
            }
            if constexpr (W == 1) {
                if constexpr (NTC > 3) { wo = zero4; p_at_d(wo, TL(1, 3), TL(1, 1)); p_at_b(wo, TL(2, 3), TL(1, 2), 1.0); }
                p_at_b(i0, TL(0, 2), TL(1, 2), 1.0);
                p_at_b(i1, TL(1, 2), TL(1, 2), 1.0);
            }
            if constexpr (W == 2) {
                if constexpr (NTC > 3) { p_at_d(wo, TL(0, 3), D0); p_at_b(wo, TL(1, 3), TL(0, 1), 1.0); p_at_b(wo, TL(2, 3), TL(0, 2), 1.0); }
                p_at_d(i0, nullptr, TL(2, 2));
            }
        }
        LBAR(9);                                                 // A3
    }
    // ================================ column 3
    if constexpr (NTC > 3) {
        if constexpr (W == 0) { const v4d res = p_dinv_reg(TL(3, 3), wo, -1.0); store_tile_t<double, false>(TL(2, 3), res, lane); }
        if constexpr (W == 1) { const v4d res = p_dinv_reg(TL(3, 3), wo, -1.0); store_tile_t<double, false>(TL(1, 3), res, lane); }
        if constexpr (W == 2) { const v4d res = p_dinv_reg(TL(3, 3), wo, -1.0); store_tile_t<double, false>(TL(0, 3), res, lane); }
        LBAR(10);                                                 // M3
        if constexpr (W == 0) { p_at_b(i0, TL(0, 3), TL(0, 3), 1.0); p_at_b(i1, TL(0, 3), TL(2, 3), 1.0); p_at_b(i2, TL(1, 3), TL(2, 3), 1.0); }
        if constexpr (W == 1) { p_at_b(i0, TL(0, 3), TL(1, 3), 1.0); p_at_b(i1, TL(1, 3), TL(1, 3), 1.0); }
        if constexpr (W == 2) { p_at_b(i0, TL(2, 3), TL(2, 3), 1.0); p_at_d(i1, nullptr, TL(3, 3)); }
        if constexpr (W == 3) { p_at_d(i0, TL(0, 3), TL(3, 3)); p_at_d(i1, TL(1, 3), TL(3, 3)); p_at_d(i2, TL(2, 3), TL(3, 3)); }
        LBAR(11);                                                 // A4: every read of W is done
    }
    // ================================ T^-1 into the store (swizzled, as the half rows are read)
    if constexpr (W == 0) {
        store_tile_t<double, true>(T, i0, lane);
        if constexpr (NTC > 2) { store_tile_t<double, true>(TL(0, 2), i1, lane); store_tile_t<double, true>(TL(1, 2), i2, lane); }
    }
    if constexpr (W == 1 && NTC > 1) { store_tile_t<double, true>(TL(0, 1), i0, lane); store_tile_t<double, true>(TL(1, 1), i1, lane); }
    if constexpr (W == 2 && NTC > 2) { store_tile_t<double, true>(TL(2, 2), i0, lane); if constexpr (NTC > 3) store_tile_t<double, true>(TL(3, 3), i1, lane); }
    if constexpr (W == 3 && NTC > 3) { store_tile_t<double, true>(TL(0, 3), i0, lane); store_tile_t<double, true>(TL(1, 3), i1, lane); store_tile_t<double, true>(TL(2, 3), i2, lane); }
    LBAR(12);
}
template <int NTC>
__device__ __forceinline__ void latp_dispatch(const int w, double* T, const double* D0, v4d& acc0, v4d& acc1, v4d& acc2, double* misc, const int lane, long long* stamps) {
    if (w == 0) latp_run<0, NTC>(T, D0, acc0, acc1, acc2, misc, lane, stamps);
    else if (w == 1) latp_run<1, NTC>(T, D0, acc0, acc1, acc2, misc, lane, stamps);
    else if (w == 2) latp_run<2, NTC>(T, D0, acc0, acc1, acc2, misc, lane, stamps);
    else latp_run<3, NTC>(T, D0, acc0, acc1, acc2, misc, lane, stamps);
}

// One QP (index b) on one workgroup of NW waves.  TIO = element type of the caller's buffers, R = iteration type.
template <int N, typename R, typename TIO, int MODE, typename TT = double, int SPW = 5, int XW = 0>
__device__ __forceinline__ void wrench_qp(const KArgs& a, const int b, double* sm) {
    using S = WrenchSmem<N, (int)sizeof(TT), SPW, XW>;
    typedef TT v4t __attribute__((ext_vector_type(4)));
    static_assert(sizeof(TT) == 8 || (sizeof(R) == 4 && MODE == 0), "fp32 tiles belong to the fp32 path");
    constexpr int n = S::n, m = S::m, NW = S::NW, NWS = S::NWS, BT = S::BT, LT = S::LT, TS = S::TS, CHMAX = S::CHMAX;
    static_assert(XW == 0 || (sizeof(TT) == 8 && MODE == 0), "extra set-up waves: fp64 tiles, solve mode");
    static_assert((S::o_R % 2) == 0 && (S::o_wb % 2) == 0 && (S::o_tb % 2) == 0 && (S::o_vb % 2) == 0, "16-byte alignment");
    static_assert(S::NT <= 2 * NW || S::WQ >= 1, "");
    const double rho_b = unis(SRBDQP_RHO_OF(a, b));   // (per-QP values are wave-uniform: scalar registers, see uni())
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr bool CSUM = XW > 0 && sizeof(TIO) == 8 && MODE == 0 && S::LT <= 256;   // the low-latency instantiations: the completion word may carry a checksum (KArgs::done_cs)
    unsigned long long cs_host = 0;                                  // XOR of the 64-bit patterns this thread stores for the host
    int mcol = lane & 15, kq = lane >> 4;
    TT* T = reinterpret_cast<TT*>(sm + S::o_T);
    int* igsz = reinterpret_cast<int*>(sm + S::o_int);            // gsz[N]
    int* igoff = igsz + N;                                        // goff[N + 1]
    int* imisc = igoff + N + 1;                                   // [0] n_g, [1] na
    int* iwr = imisc + 2;                                         // [N] 1 = the step uses its 6 wrench coordinates
    uint8_t* sct = reinterpret_cast<uint8_t*>(sm + S::o_ct);
    uint8_t* gstep = reinterpret_cast<uint8_t*>(sm + S::o_gs);
    const double* SQ = sm + S::o_sq;
    const double* CP = sm + S::o_cp;
    const size_t row0 = a.row_off ? (size_t)a.row_off[b] : (size_t)b * N;   // first horizon row of this QP in the step-major arrays
    const TIO* gwu = reinterpret_cast<const TIO*>(a.warm_u) + row0 * 12;
    const TIO* gwy = reinterpret_cast<const TIO*>(a.warm_y) + row0 * 20;

    // ================= load (coalesced, one batch of loads) + linearise (a5) =================
    SRBDQP_STAMP(a, b, 0);
    {
        const char* kin_ = (sizeof(TIO) == 8) ? staged_in_base<N>(a) : nullptr;     // (one staged QP: its inputs in the kernel-argument segment)
        const TIO* gx0 = kin_ ? reinterpret_cast<const TIO*>(kin_ + offsetof(StagedIn<N>, x0)) : reinterpret_cast<const TIO*>(a.x0) + (size_t)b * 13;
        const TIO* gxr = kin_ ? reinterpret_cast<const TIO*>(kin_ + offsetof(StagedIn<N>, xref)) : reinterpret_cast<const TIO*>(a.xref) + row0 * 13;
        const TIO* gft = kin_ ? reinterpret_cast<const TIO*>(kin_ + offsetof(StagedIn<N>, foot)) : reinterpret_cast<const TIO*>(a.foot) + row0 * 12;
        const uint8_t* gct = kin_ ? reinterpret_cast<const uint8_t*>(kin_ + offsetof(StagedIn<N>, contact)) : a.contact + row0 * 4;
        constexpr int RX = (N * 13 + BT - 1) / BT, RF = (N * 12 + BT - 1) / BT;
        static_assert(N * 4 <= BT && N * 3 <= BT, "one thread per contact flag / pcom entry");
        const TIO* gpc = a.pcom ? (kin_ ? reinterpret_cast<const TIO*>(kin_ + offsetof(StagedIn<N>, pcom)) : reinterpret_cast<const TIO*>(a.pcom) + (size_t)b * N * 3) : gx0;
        const TIO v_x0 = gx0[t < 13 ? t : 0];
        const uint8_t v_ct = gct[t < N * 4 ? t : 0];
        const TIO v_pc = gpc[(a.pcom && t < N * 3) ? t : 0];
        TIO v_xr[RX], v_ft[RF];
#pragma unroll
        for (int r = 0; r < RX; ++r) { const int i = t + r * BT; v_xr[r] = gxr[i < N * 13 ? i : 0]; }
#pragma unroll
        for (int r = 0; r < RF; ++r) { const int i = t + r * BT; v_ft[r] = gft[i < N * 12 ? i : 0]; }
        if (t < 13) sm[S::o_x0 + t] = (double)v_x0;
        if (t >= 32 && t < 44) sm[S::o_sq + t - 32] = a.sqrtq[t - 32];
#pragma unroll
        for (int r = 0; r < RX; ++r) { const int i = t + r * BT; if (i < N * 13) sm[S::o_xref + i] = (double)v_xr[r]; }
#pragma unroll
        for (int r = 0; r < RF; ++r) { const int i = t + r * BT; if (i < N * 12) sm[S::o_foot + i] = (double)v_ft[r]; }
        if (t < N * 4) sct[t] = v_ct ? 1 : 0;
        if (a.pcom && t < N * 3) sm[S::o_pcom + t] = (double)v_pc;
        if (t == 0) { sm[S::o_misc] = 0.0; sm[S::o_misc + 1] = 0.0; }
        __syncthreads();
        if (!a.pcom && t < N * 3) sm[S::o_pcom + t] = sm[S::o_xref + (t / 3) * 13 + 3 + (t % 3)];
        if (t < N) {
            double sn, cs;
            sincos(sm[S::o_xref + t * 13 + 2], &sn, &cs);
            double* Tm = sm + S::o_tm + t * 9;
            Tm[0] = cs;  Tm[1] = sn;  Tm[2] = 0.0;
            Tm[3] = -sn; Tm[4] = cs;  Tm[5] = 0.0;
            Tm[6] = 0.0; Tm[7] = 0.0; Tm[8] = 1.0;
        }
        if (t >= BT - 64 && t < BT - 64 + N) {
            // presolve bookkeeping: g coordinates per step.  One lane per step (last wave; the first one is in sincos): every lane
            // walks the flag words of all steps -- N independent LDS reads instead of a serial chain of N dependent ones on one
            // thread (15 k cycles of the 45 k of this phase at N = 20 with three QPs per CU, tools/wrench_stamps.py sub-stamps)
            const int k = t - (BT - 64);
            const uint32_t* cw = reinterpret_cast<const uint32_t*>(sct);          // 4 flags (0 / 1) per step
            int off = 0, tot = 0, na = 0, allw = 1, gk = 0, wk = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int c = __popc(cw[j]);
                const int g = (c >= 3) ? 6 : 3 * c;
                off += (j < k) ? g : 0;
                tot += g; na += c;
                allw &= (c >= 3) ? 1 : 0;
                if (j == k) { gk = g; wk = (c >= 3) ? 1 : 0; }
            }
            igsz[k] = gk; igoff[k] = off; iwr[k] = wk;
            for (int r = 0; r < gk; ++r) gstep[off + r] = (uint8_t)k;
            if (k == N - 1) { igoff[N] = tot; imisc[0] = tot; imisc[1] = na; iwr[N] = allw; }   // iwr[N]: every step in wrench coordinates
        }
        __syncthreads();
        if (t < 9) {
            double acc = 0.0;
            for (int k = 0; k < N; ++k) { acc += sm[S::o_tm + k * 9 + t]; sm[S::o_cp + k * 9 + t] = acc; }
        }
        static_assert(N * 12 <= BT, "one thread per entry of a J row");
        if (t < N * 12) {   // J_k[:, 3 ci + ax] = Iw^-1 skew(r)[:, ax]
            const int i = t;
            const int k = i / 12, cc = i - 12 * k, ci = cc / 3, ax = cc - 3 * ci;
            const double cs = sm[S::o_tm + k * 9 + 0], sn = sm[S::o_tm + k * 9 + 1];
            const double i0 = a.iinv[0], i1 = a.iinv[1], i2 = a.iinv[2];
            const double w00 = cs * cs * i0 + sn * sn * i1, w01 = cs * sn * (i0 - i1), w11 = sn * sn * i0 + cs * cs * i1;
            const double rx = sm[S::o_foot + k * 12 + 3 * ci + 0] - sm[S::o_pcom + k * 3 + 0];
            const double ry = sm[S::o_foot + k * 12 + 3 * ci + 1] - sm[S::o_pcom + k * 3 + 1];
            const double rz = sm[S::o_foot + k * 12 + 3 * ci + 2] - sm[S::o_pcom + k * 3 + 2];
            // column ax of skew(r), as selects: hipcc (ROCm 7.2) lowers the three-way if / else-if / else on this 16-bit
            // value as a switch and loses the "s1 = -rx" of the last arm for some instantiations (seen in the ISA at N = 12)
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
    const int n_g = __builtin_amdgcn_readfirstlane(imisc[0]);
    const int na = __builtin_amdgcn_readfirstlane(imisc[1]);

    // ---- lane roles (plain variables: derived a second time behind the factorisation in the fp32-tile kernel, see REROLE below)
    int sg, ul, js, ci, ax, rl, h, f0, f1, f2, f3, gsj, Rrow, uvar, cbase, irowA, irowB;
    bool stepok, wrench, active_u, active_g;
    auto lane_roles = [&](const int ln) __attribute__((always_inline)) {
        sg = ln / 12; ul = ln - 12 * sg;
        const int jstep = SPW * w + sg;
        stepok = (sg < SPW) && (jstep < N) && (XW == 0 || w < NWS);
        js = stepok ? jstep : 0;
        ci = ul / 3; ax = ul - 3 * ci;
        rl = ul >> 1; h = ul & 1;
        f0 = sct[4 * js]; f1 = sct[4 * js + 1]; f2 = sct[4 * js + 2]; f3 = sct[4 * js + 3];
        wrench = (f0 + f1 + f2 + f3) >= 3;
        active_u = stepok && sct[4 * js + ci] != 0;
        gsj = stepok ? igsz[js] : 0;
        active_g = rl < gsj;
        Rrow = igoff[js] + (active_g ? rl : 0);
        uvar = 12 * js + ul;                                         // index in the full 12N vector
        cbase = 12 * sg + 3 * ci;                                    // first lane of this contact in the wave
        irowA = 20 * js + 5 * ci + ((ax < 2) ? 2 * ax : 4); irowB = 20 * js + 5 * ci + 2 * ax + 1;
    };
    lane_roles(lane);

    if constexpr (MODE == 1) { if (na == 0) return; }   // assembly dump of an empty problem: all zeros (the host cleared the buffers)
    if (na == 0) {   // nothing to solve: all forces 0
        for (int c = t; c < n; c += BT) sm[S::o_xs + c] = 0.0;
        if (a.y_out) for (int i = t; i < m; i += BT) reinterpret_cast<TIO*>(a.y_out)[row0 * 20 + i] = TIO(0);
        if (t == 0) { if (a.status) a.status[b] = 1; if (a.iters) a.iters[b] = 0; cs_host = done_cs_pack(1, 0); }
        __syncthreads();
    } else {

    // ================= tables of the closed-form assembly (a6 + a7; srbdqp_compact.hpp has the derivation) =================
    const double dt = a.dt, dt2 = unis(a.dt * a.dt), dtm = unis(a.dt * a.inv_mass), dt2m = unis(dt2 * a.inv_mass);
    double* T1 = sm + S::o_t1;
    double* T2 = sm + S::o_t2;
    double* MT = sm + S::o_mt;
    double* GV = sm + S::o_gv;
    // Low-latency instantiation (XW = 2): the tables are the two set-up helpers' work -- wave NWS the error vector and the G'v
    // tables behind it, wave NWS + 1 T1 / T2 and D_m / E_m behind them, each chain inside ONE wave (its LDS operations complete in
    // order: no barrier between the two halves) -- while the step waves compute E, V, Bd, which need none of it; one barrier
    // joins the three.  Serially (every wave on every table, then E) the two phases took 6.4 k + 8.0 k cycles of a batch-1 solve.
    constexpr bool TSPLIT = (XW == 2);
    constexpr int TUNR = (XW > 0 && N <= 10) ? N : 4;                   // table loops over the steps: fully unrolled in the low-latency instantiation (every LDS read of a chain in flight at once)
    const bool tab_a = !TSPLIT || w == NWS, tab_b = !TSPLIT || w == NWS + 1;
    const int tt = TSPLIT ? lane : t;
    constexpr int TSTR = TSPLIT ? 64 : BT;
    if (tab_a)
    for (int k = tt; k < n; k += TSTR) {
        const int i = k / 12, kk = k - 12 * i;
        sm[S::o_eh + k] = SQ[kk] * (free_response<N, S>(a, sm, i, kk) - sm[S::o_xref + i * 13 + kk]);
    }
    if (tab_b)
    for (int idx = tt; idx < 9 * N; idx += TSTR) {
        const int mm = idx / 9, pq = idx - 9 * mm, p = pq / 3, q = pq - 3 * p;
        const double* Cm = CP + mm * 9;
        const double w0 = SQ[0] * SQ[0], w1 = SQ[1] * SQ[1], w2 = SQ[2] * SQ[2];
        const double m0p = Cm[p], m1p = Cm[3 + p], m2p = Cm[6 + p], m0q = Cm[q], m1q = Cm[3 + q], m2q = Cm[6 + q], mpq = Cm[pq];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll TUNR
        for (int i = 0; i < N; ++i) {
            const double* Ci = CP + i * 9;
            const double on = (i >= mm) ? 1.0 : 0.0;
            const double d0p = Ci[p] - m0p, d1p = Ci[3 + p] - m1p, d2p = Ci[6 + p] - m2p;
            const double d0q = Ci[q] - m0q, d1q = Ci[3 + q] - m1q, d2q = Ci[6 + q] - m2q;
            s1 = fma(on, Ci[pq] - mpq, s1);
            s2 = fma(on, (w0 * d0p) * d0q + (w1 * d1p) * d1q + (w2 * d2p) * d2q, s2);
        }
        T1[idx] = s1;
        T2[idx] = s2;
    }
    if constexpr (TSPLIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-local hand-over)
    else __syncthreads();
    // the 3 torque entries of a step are an O(N) loop of 7 LDS reads, the 6 force entries one of 1 read: the two kinds sit on
    // different waves where the workgroup has more than one (entry by entry over all threads every wave ran both loops)
    constexpr int GT_HL = (64 * ((3 * N + 63) / 64) + 6 * N <= BT) ? 64 * ((3 * N + 63) / 64) : 3 * N;
    static_assert(GT_HL + 6 * N <= BT, "one pass over the G'v tables");
    [[maybe_unused]] auto gt_tables_wave = [&](const double* vec) {   // the same tables on one wave (XW = 2: helper wave NWS)
        for (int e0 = lane; e0 < 9 * N; e0 += 64) {
            if (e0 < 3 * N) {
                const int j = e0 / 3, comp = e0 - 3 * j;
                const double* Cj = CP + j * 9;
                double acc = 0.0;
                const double c0 = Cj[comp], c1 = Cj[3 + comp], c2 = Cj[6 + comp];
                const double q0 = SQ[0] * dt2, q1 = SQ[1] * dt2, q2 = SQ[2] * dt2, qw = SQ[6 + comp] * dt;
#pragma unroll TUNR
                for (int i = 0; i < N; ++i) {
                    const double* Ci = CP + i * 9;
                    const double* v = vec + 12 * i;
                    const double on = (i >= j) ? 1.0 : 0.0;
                    acc = fma(on, (Ci[comp] - c0) * (q0 * v[0]) + (Ci[3 + comp] - c1) * (q1 * v[1]) + (Ci[6 + comp] - c2) * (q2 * v[2]) + qw * v[6 + comp], acc);
                }
                GV[9 * j + comp] = acc;
            } else {
                const int e = e0 - 3 * N, j = e / 6, comp = 3 + (e - 6 * j);
                const int kk = (comp < 6) ? comp : 3 + comp;
                double acc = 0.0;
#pragma unroll TUNR
                for (int i = 0; i < N; ++i) {
                    const double wgt = (i >= j) ? ((comp < 6) ? (double)(i - j) : 1.0) : 0.0;
                    acc = fma(wgt, vec[12 * i + kk], acc);
                }
                GV[9 * j + comp] = acc;
            }
        }
    };
    auto gt_tables = [&](const double* vec) {
        if (t < 3 * N) {
            const int j = t / 3, comp = t - 3 * j;
            const double* Cj = CP + j * 9;
            double acc = 0.0;
            const double c0 = Cj[comp], c1 = Cj[3 + comp], c2 = Cj[6 + comp];
            const double q0 = SQ[0] * dt2, q1 = SQ[1] * dt2, q2 = SQ[2] * dt2, qw = SQ[6 + comp] * dt;
#pragma unroll 4
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
#pragma unroll 4
            for (int i = 0; i < N; ++i) {
                const double wgt = (i >= j) ? ((comp < 6) ? (double)(i - j) : 1.0) : 0.0;
                acc = fma(wgt, vec[12 * i + kk], acc);
            }
            GV[9 * j + comp] = acc;
        }
    };
    // (G'v)[u] for this lane's force variable, from the tables
    auto gt_eval_u = [&]() -> double {
        int axl = ax, ull = ul;                                      // (laundered: the factors of the three calls are not worth a register -- in scratch memory -- between them)
        if constexpr (sizeof(TT) == 4) asm volatile("" : "+v"(axl), "+v"(ull));
        const double* J = sm + S::o_J + js * 36 + ull;
        const double* g = GV + 9 * js;
        return a.s * (J[0] * g[0] + J[12] * g[1] + J[24] * g[2] + SQ[3 + axl] * dt2m * g[3 + axl] + SQ[9 + axl] * dtm * g[6 + axl]);
    };
    if constexpr (TSPLIT) {
        // ... and, behind its tables, each helper builds one of the two tables of the T assembly (phase H), one g coordinate per lane, HERE: the wrench-space 6-vector
        // z of a coordinate follows from the contact flags and J alone (phase E writes the same values to ZT for the other users), and D_m / E_m are wave NWS + 1's
        // own (complete behind its wait) -- so phase H starts with its tiles instead of a table pass and a barrier, and takes a column's [D_m z, E_m z] from the
        // table instead of forming it per tile slot (round 5: T assembly 7.4 k -> 6.0 k cycles; both helpers still reach the join before the step waves)
        static_assert(16 * S::NT <= 64, "one g coordinate per lane of a helper wave");
        if (tab_a) gt_tables_wave(sm + S::o_eh);
        if (tab_b) de_tables<N>(CP, T1, T2, SQ, dt2, MT, lane, 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((tab_a || tab_b) && lane < 16 * S::NT) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const int r = lane;
            const bool on = r < n_g;
            const int rr = on ? r : 0, j = gstep[rr], l = rr - igoff[j];
            const int g0 = sct[4 * j], g1 = sct[4 * j + 1], g2 = sct[4 * j + 2], g3 = sct[4 * j + 3];
            const bool wr = (g0 + g1 + g2 + g3) >= 3;
            const int want = l / 3, lm = l - 3 * want;
            int cc = -1, seen = 0;
            if (g0) { if (seen == want && cc < 0) cc = 0; ++seen; }
            if (g1) { if (seen == want && cc < 0) cc = 1; ++seen; }
            if (g2) { if (seen == want && cc < 0) cc = 2; ++seen; }
            if (g3) { if (seen == want && cc < 0) cc = 3; ++seen; }
            const int ug = (cc >= 0 ? 3 * cc : 0) + lm;
            const double* Jj = sm + S::o_J + j * 36;
            const double z0 = wr ? ((l == 0) ? 1.0 : 0.0) : Jj[ug], z1 = wr ? ((l == 1) ? 1.0 : 0.0) : Jj[12 + ug], z2 = wr ? ((l == 2) ? 1.0 : 0.0) : Jj[24 + ug];
            if (tab_a) {        // rows: [z_ang, -C_j z_ang, z_lin[(row index) mod 3], (step, offset of the row in the step's E^-1) as one exact integer]
                const double gl = wr ? ((l >= 3) ? 1.0 : 0.0) : 1.0;
                const double* C = CP + 9 * j;
                const double code = on ? (double)(4096 * j + 36 * j + 6 * l) : -4096.0;
                d2* row = reinterpret_cast<d2*>(sm + S::o_abx + 8 * r);
                row[0] = on ? (d2){z0, z1} : (d2){0.0, 0.0};
                row[1] = on ? (d2){z2, -(C[0] * z0 + C[1] * z1 + C[2] * z2)} : (d2){0.0, 0.0};
                row[2] = on ? (d2){-(C[3] * z0 + C[4] * z1 + C[5] * z2), -(C[6] * z0 + C[7] * z1 + C[8] * z2)} : (d2){0.0, 0.0};
                row[3] = (d2){on ? gl : 0.0, code};
            } else {            // columns: s^2 [D_m z_ang, E_m z_ang]
                const double s2e = a.s * a.s;
                const double* D = MT + 18 * j;
                d2* brow = reinterpret_cast<d2*>(sm + S::o_bbx + 6 * r);
                double bv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) bv[i] = on ? s2e * (D[3 * i] * z0 + D[3 * i + 1] * z1 + D[3 * i + 2] * z2) : 0.0;
                brow[0] = (d2){bv[0], bv[1]}; brow[1] = (d2){bv[2], bv[3]}; brow[2] = (d2){bv[4], bv[5]};
            }
        }
    } else {
        gt_tables(sm + S::o_eh);
        de_tables<N>(CP, T1, T2, SQ, dt2, MT, t, BT);
        __syncthreads();
    }
    SRBDQP_STAMP(a, b, 1);
    double qv = 0.0;                                                 // gradient of this lane's variable (needs the G'v tables)
#ifdef SRBDQP_WRENCH_DEBUG
    if constexpr (MODE == 1) {   // raw LDS image after the tables (diagnostic builds only)
        double* out = a.P_out + (size_t)b * (S::NG * S::NG);
        for (int i = t; i < S::NG * S::NG && i < S::o_end; i += BT) out[i] = sm[i];
        return;
    }
#endif
    // G'(G x) for the vector x at o_x0c (u space, swing variables 0; visible to the workgroup on entry): this lane's entry
    auto gtg_of_x0c = [&]() -> double {
        double* TF = sm + S::o_tf;
        for (int idx = t; idx < 6 * N; idx += BT) {   // per step: tau_j = J_j x_j (3), f_j = sum of contact forces (3)
            const int j = idx / 6, comp = idx - 6 * j;
            const double* x = sm + S::o_x0c + 12 * j;
            double acc = 0.0;
            if (comp < 3) {
                const double* J = sm + S::o_J + j * 36 + comp * 12;
                for (int c = 0; c < 12; ++c) acc += J[c] * x[c];
            } else {
                acc = x[comp - 3] + x[comp] + x[comp + 3] + x[comp + 6];
            }
            TF[idx] = acc;
        }
        __syncthreads();
        for (int k = t; k < n; k += BT) {   // G x^0, row kk of step i
            const int i = k / 12, kk = k - 12 * i;
            const double acc = gx_row<N>(CP, TF, i, kk, dt, dt2, dtm, dt2m);
            sm[S::o_gx + k] = SQ[kk] * a.s * acc;
        }
        __syncthreads();
        gt_tables(sm + S::o_gx);
        __syncthreads();
        return active_u ? gt_eval_u() : 0.0;
    };
    double px0 = 0.0, x_init = 0.0;
    // fp32 tiles: P x^0 of a warm start is formed behind the refinement of x_q (which rebuilds the tables it needs anyway) instead of
    // waiting in two registers -- in scratch memory at the 168-register budget -- across the tile phases
    constexpr bool WARM_LATE = sizeof(TT) == 4;
    auto gradient_and_warm_start = [&]() __attribute__((always_inline)) {
        qv = active_u ? gt_eval_u() : 0.0;
        if (!WARM_LATE && a.warm_u) {   // P x^0 = G'(G x^0) + R s^2 x^0
            x_init = active_u ? (double)gwu[uvar] / a.s : 0.0;
            if (stepok) sm[S::o_x0c + uvar] = x_init;
            __syncthreads();
            const double gtg = gtg_of_x0c();
            px0 = active_u ? gtg + a.rs2 * x_init : 0.0;
        }
    };
    if constexpr (!TSPLIT) gradient_and_warm_start();   // (XW = 2: behind the barrier that joins the tables and E)
    [[maybe_unused]] float qv_lo = 0.0f;
    if constexpr (S::QV_PARK) {   // own entry: read back by the same lane in front of x_q
        const float qh = (float)qv;
        qv_lo = (float)(qv - (double)qh);
        asm volatile("" : "+v"(qv_lo));                              // (formed HERE: left to itself the compiler keeps qv -- in scratch memory -- and forms the remainder at its use)
        if (stepok) reinterpret_cast<float*>(sm + S::o_qv)[uvar] = qh;
    }

    // ================= per-step wrench blocks: E^-1, V, Bd (registers of the step's lanes) =================
    const double dxy = unis(a.rs2 + a.sigma + 2.0 * rho_b), dz = unis(a.rs2 + a.sigma + (4.0 * a.mu * a.mu + a.rho_fz) * rho_b);
    const double idxy = unis(1.0 / dxy), idz = unis(1.0 / dz);
    // fp32 tiles (3 workgroups per CU, 168 registers): the rows / columns of V and Bd are formed AFTER the factorisation, from
    // the triangle of E^-1 kept in LDS behind the tiles, and held in fp32 from then on.  Formed here they waited in scratch
    // memory across phases F / W / I: 10 GB of HBM traffic per 65,536-QP launch against 0.29 GB of inputs and outputs
    // (rocprofv3 FETCH_SIZE / WRITE_SIZE, round 2).
    // (round 5: fp64 tiles too, for the instantiations compiled at 3 waves per SIMD -- N <= 12 in fp64, every fp32 instantiation on fp64 tiles -- where the
    //  triangle fits the LDS the workgroup has anyway, WrenchSmem::E4_FITS)
    constexpr bool VBD_LATE = sizeof(TT) == 4 || (S::E4_FITS && MODE == 0 && (CHMAX <= 36 || sizeof(R) == 4));
    typedef double VS;   // (x_q and its refinement need V and Bd in fp64: rounded to fp32 the refinement contracts 10 x slower)
    VS vrow[6], vcol[6];
    double bjv[4];                                                   // bjv: J[:, u] of the lane's variable and 1 / D_u (apply_kinv)
    constexpr bool BD_EXPLICIT = (sizeof(R) == 4) || (MODE == 1);     // fp32 iterations and the assembly dump: Bd rows (apply_kinv)
#ifndef SRBDQP_WRENCH_BD_LAST
#define SRBDQP_WRENCH_BD_LAST 1
#endif
    constexpr bool BD_LAST = SRBDQP_WRENCH_BD_LAST && VBD_LATE && BD_EXPLICIT;   // ... formed after x_q and its refinement (form_bd)
    // explicit Bd rows: fp32 from the start in the fp32-tile kernel (only the iterations use them there; x_q and its refinement
    // run in fp64, where the implicit form is exact enough) -- 12 registers instead of 24 next to the T^-1 row
    typedef typename std::conditional<(sizeof(TT) == 4 || (VBD_LATE && sizeof(R) == 4)), float, double>::type BS;
    [[maybe_unused]] BS bdrow[12];
    int bsel = 0;                                                    // g row (within the step) of the unit part of Y'[:, u]
    int before_ci = 0;                                               // stance contacts of the step before ci
    int ug_id = 0;                                                   // force-variable step: variable (0..11) of g row rl
    auto lane_roles2 = [&]() __attribute__((always_inline)) {
        before_ci = (ci > 0 ? f0 : 0) + (ci > 1 ? f1 : 0) + (ci > 2 ? f2 : 0);
        const int want = rl / 3;
        int cc = -1, seen = 0;
        if (f0) { if (seen == want && cc < 0) cc = 0; ++seen; }
        if (f1) { if (seen == want && cc < 0) cc = 1; ++seen; }
        if (f2) { if (seen == want && cc < 0) cc = 2; ++seen; }
        if (f3) { if (seen == want && cc < 0) cc = 3; ++seen; }
        ug_id = (cc >= 0 ? 3 * cc : 0) + (rl % 3);
    };
    lane_roles2();
    // registers of V and Bd of a wrench step from er = row rl of E^-1 and yv = E^-1 omega_u, omega_u = [J[:, ul]; e_ax]
    // (the callers form the two from the register matrix in phase E, from the LDS triangle for the late formation; E^-1
    // itself must not be captured here: a by-reference capture turns its select chains into an indexed scratch array)
    auto form_vbd = [&](const double (&er)[6], const double (&yv)[6]) __attribute__((always_inline)) {
        const double* Jj = sm + S::o_J + js * 36;
        const int fl[4] = {f0, f1, f2, f3};                        // (selects, not factors: the flags as doubles were kept -- spilled -- across the whole kernel)
        // V[rl][6 h + i] = wgt (er[0..2] . J[:, u'] + er[3 + a'])
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int u2 = 6 * h + i;
            const int flc = (u2 / 3 == 0) ? fl[0] : (u2 / 3 == 1) ? fl[1] : (u2 / 3 == 2) ? fl[2] : fl[3];
            const double wgt = flc ? (((i % 3) < 2) ? idxy : idz) : 0.0;
            vrow[i] = (VS)(wgt * (er[0] * Jj[u2] + er[1] * Jj[12 + u2] + er[2] * Jj[24 + u2] + er[3 + (i % 3)]));
        }
        const double wu = active_u ? ((ax < 2) ? idxy : idz) : 0.0;
        const double j0 = Jj[ul], j1 = Jj[12 + ul], j2 = Jj[24 + ul];
#pragma unroll
        for (int r = 0; r < 6; ++r) vcol[r] = (VS)(wu * yv[r]);
        bjv[0] = j0; bjv[1] = j1; bjv[2] = j2; bjv[3] = wu; bsel = 3 + ax;
        if constexpr (BD_EXPLICIT && !BD_LAST) {
#pragma unroll
            for (int u2 = 0; u2 < 12; ++u2) {
                const double wgt2 = fl[u2 / 3] ? (((u2 % 3) < 2) ? idxy : idz) : 0.0;
                const double dotv = yv[0] * Jj[u2] + yv[1] * Jj[12 + u2] + yv[2] * Jj[24 + u2] + yv[3 + (u2 % 3)];
                bdrow[u2] = (BS)(((u2 == ul) ? wu : 0.0) - wu * wgt2 * dotv);   // D^-1 - D^-1 Y' V (static index: select chain)
            }
        }
    };
    // the explicit Bd row alone (fp32 tiles: formed after x_q and its refinement, which use the implicit form -- 12 registers
    // less across those two fp64 applications of K^-1)
    [[maybe_unused]] auto form_bd = [&](const double (&yv)[6]) __attribute__((always_inline)) {
        const double* Jj = sm + S::o_J + js * 36;
        const int fl[4] = {f0, f1, f2, f3};
        const double wu = active_u ? ((ax < 2) ? idxy : idz) : 0.0;
#pragma unroll
        for (int u2 = 0; u2 < 12; ++u2) {
            const double wgt2 = fl[u2 / 3] ? (((u2 % 3) < 2) ? idxy : idz) : 0.0;
            const double dotv = yv[0] * Jj[u2] + yv[1] * Jj[12 + u2] + yv[2] * Jj[24 + u2] + yv[3 + (u2 % 3)];
            bdrow[u2] = (BS)(((u2 == ul) ? wu : 0.0) - wu * wgt2 * dotv);
        }
    };
    // ... of a force-variable step: V = the selection of the stance variables, Bd = 0
    auto form_vbd_identity = [&]() __attribute__((always_inline)) {
        const int rank = 3 * before_ci + ax;                         // g row of this lane's variable
#pragma unroll
        for (int i = 0; i < 6; ++i) vrow[i] = (active_g && (6 * h + i == ug_id)) ? VS(1) : VS(0);
#pragma unroll
        for (int r = 0; r < 6; ++r) vcol[r] = (active_u && r == rank) ? VS(1) : VS(0);
        bjv[0] = 0.0; bjv[1] = 0.0; bjv[2] = 0.0; bjv[3] = 0.0; bsel = active_u ? rank : 0;
        if constexpr (BD_EXPLICIT && !BD_LAST) {
#pragma unroll
            for (int u2 = 0; u2 < 12; ++u2) bdrow[u2] = BS(0);
        }
    };
    if (!TSPLIT || w < NWS) {
        const double* Jj = sm + S::o_J + js * 36;
        const int fl[4] = {f0, f1, f2, f3};
        double* ZT = sm + S::o_zt;
        double* EI = sm + S::o_ei + 36 * js;
        ESTAMP(a, 0);
        if (wrench) {
            // E = W D^-1 W' (6 x 6, SPD): every lane of the step forms and inverts it redundantly in registers
            double Em[6][6];
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = 0; q < 6; ++q) Em[p][q] = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int a2 = 0; a2 < 3; ++a2) {
                    const int u2 = 3 * c + a2;
                    const double wgt = fl[c] ? ((a2 < 2) ? idxy : idz) : 0.0;
                    const double j0 = Jj[u2], j1 = Jj[12 + u2], j2 = Jj[24 + u2];
                    const double jw[3] = {j0 * wgt, j1 * wgt, j2 * wgt};
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        Em[p][0] = fma(jw[p], j0, Em[p][0]); Em[p][1] = fma(jw[p], j1, Em[p][1]); Em[p][2] = fma(jw[p], j2, Em[p][2]);
                        Em[p][3 + a2] += jw[p];
                    }
                    Em[3 + a2][3 + a2] += wgt;
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) Em[3 + q][p] = Em[p][3 + q];
            // (the J entries are read again below rather than kept: 36 doubles per lane across this phase were spilled)
            asm volatile("" ::: "memory");
            ESTAMP(a, 1);
            // E^-1 through its structure (round 5): E = [A B; B' G] with G = S D^-1 S' DIAGONAL (S = [I I I I]: the force rows of W), so
            //     E^-1 = [ Sc^-1, -Sc^-1 B G^-1 ; sym, G^-1 + G^-1 B' Sc^-1 B G^-1 ],   Sc = A - B G^-1 B'   (3 x 3, SPD)
            // -- a 3 x 3 Cholesky and three reciprocals instead of the 6 x 6 Cholesky + triangular inverse + L^-T L^-1 every lane of the step ran redundantly
            // (three dependent rsqrt chains instead of six, ~120 instead of ~270 fp64 instructions: 7.8 k -> cycles of the batch-1 set-up were this block).
            double Ei[6][6];
            bool okE = true;
            {
                // G = n_c diag(1 / d_xy, 1 / d_xy, 1 / d_z) with n_c = 3 or 4 stance contacts (a wrench step): its inverse without a division
                const double inc = ((f0 + f1 + f2 + f3) == 4) ? 0.25 : (1.0 / 3.0);
                const double ig[3] = {dxy * inc, dxy * inc, dz * inc};
                double BG[3][3];                                         // B G^-1
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) BG[p][q] = Em[p][3 + q] * ig[q];
                double Sc[3][3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = p; q < 3; ++q) {
                        double v = Em[p][q];
#pragma unroll
                        for (int k2 = 0; k2 < 3; ++k2) v = fma(-BG[p][k2], Em[q][3 + k2], v);
                        Sc[p][q] = v;
                    }
                // Sc = L L' (lower), Li = L^-1
                auto rs = [&](double d) -> double {
                    if constexpr (XW > 0) return (d > 0.0) ? fast_rsqrt2(d) : 0.0;
                    else return 1.0 / sqrt(d);
                };
                okE = okE && (Sc[0][0] > 0.0);
                const double r0 = rs(Sc[0][0]);
                const double l10 = Sc[0][1] * r0, l20 = Sc[0][2] * r0;
                const double d1 = fma(-l10, l10, Sc[1][1]);
                okE = okE && (d1 > 0.0);
                const double r1 = rs(d1);
                const double l21 = fma(-l20, l10, Sc[1][2]) * r1;
                const double d2 = fma(-l21, l21, fma(-l20, l20, Sc[2][2]));
                okE = okE && (d2 > 0.0);
                const double r2 = rs(d2);
                const double m10 = -l10 * r0 * r1;                       // Li[1][0]
                const double m21 = -l21 * r1 * r2;                       // Li[2][1]
                const double m20 = -(l20 * r0 + l21 * m10) * r2;         // Li[2][0]
                double Si[3][3];                                         // Sc^-1 = Li' Li
                Si[0][0] = fma(r0, r0, fma(m10, m10, m20 * m20)); Si[0][1] = fma(m10, r1, m20 * m21); Si[0][2] = m20 * r2;
                Si[1][1] = fma(r1, r1, m21 * m21); Si[1][2] = m21 * r2; Si[2][2] = r2 * r2;
                Si[1][0] = Si[0][1]; Si[2][0] = Si[0][2]; Si[2][1] = Si[1][2];
                double C12[3][3];                                        // -Sc^-1 B G^-1
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) C12[p][q] = -(Si[p][0] * BG[0][q] + Si[p][1] * BG[1][q] + Si[p][2] * BG[2][q]);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) { Ei[p][q] = Si[p][q]; Ei[p][3 + q] = C12[p][q]; Ei[3 + q][p] = C12[p][q]; }
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = p; q < 3; ++q) {
                        double v = (p == q) ? ig[p] : 0.0;                 // G^-1 - (B G^-1)' C12
#pragma unroll
                        for (int k2 = 0; k2 < 3; ++k2) v = fma(-BG[k2][p], C12[k2][q], v);
                        Ei[3 + p][3 + q] = v; Ei[3 + q][3 + p] = v;
                    }
            }
            ESTAMP(a, 2);
            if (!okE && stepok && ul == 0) sm[S::o_misc] = 1.0;   // 4 collinear contact points: E singular
            // E^-1 goes to LDS WHOLE, from the first lane of the step, and every lane reads what it needs back (its row; one column entry per row) behind a
            // wave-local wait -- the 12 lanes of a step sit in one wave.  Until round 5 each lane picked row rl and column 3 + ax out of its register copy with
            // select chains: hipcc turned them into divergent branches around the entries' own arithmetic (37 exec-masked blocks, each run by every wave for every
            // case): 5.0 k of the 7.8 k cycles of this phase at batch 1.
            if (stepok && ul == 0) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2* EI2 = reinterpret_cast<d2*>(EI);                   // (36 js doubles: 16-byte aligned)
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) EI2[3 * r + c2] = (d2){Ei[r][2 * c2], Ei[r][2 * c2 + 1]};
                if constexpr (VBD_LATE) {
                    double* E4 = sm + S::o_e4 + 21 * js;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) E4[(r * (r + 1)) / 2 + c] = Ei[r][c];
                }
            }
            if (stepok) {
#pragma unroll
                for (int i = 0; i < 3; ++i) ZT[6 * Rrow + 3 * h + i] = (3 * h + i == rl) ? 1.0 : 0.0;
            }
            if constexpr (!VBD_LATE) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (one wave: its LDS operations complete in order)
                const double j0 = Jj[ul], j1 = Jj[12 + ul], j2 = Jj[24 + ul];
                double er[6], yv[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) er[c] = EI[6 * rl + c];
#pragma unroll
                for (int r = 0; r < 6; ++r) yv[r] = Ei[r][0] * j0 + Ei[r][1] * j1 + Ei[r][2] * j2 + EI[6 * r + 3 + ax];
                form_vbd(er, yv);
            }
            ESTAMP(a, 3);
        } else {
            // identity coordinates: g row r <-> the r-th stance force variable of the step
            const int ug = ug_id;
            if constexpr (!VBD_LATE) form_vbd_identity();
            if (stepok && active_g) {
                const double dd = ((rl % 3) < 2) ? dxy : dz;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    EI[6 * rl + 3 * h + i] = (3 * h + i == rl) ? dd : 0.0;
                    ZT[6 * Rrow + 3 * h + i] = h ? (((rl % 3) == i) ? 1.0 : 0.0) : Jj[12 * i + ug];
                }
            }
        }
    }
    WAVE_ARRIVE(a, w, lane, 13);
    __syncthreads();
    if constexpr (TSPLIT) gradient_and_warm_start();
    if constexpr (MODE == 1) { if (sm[S::o_misc] != 0.0) { if (t == 0) a.ub_out[(size_t)b * (N + 1) + N] = -1.0; return; } }
    if (sm[S::o_misc] != 0.0) {   // degenerate contact geometry: report, return zero forces
        for (int c = t; c < n; c += BT) sm[S::o_xs + c] = 0.0;
        if (a.y_out) for (int i = t; i < m; i += BT) reinterpret_cast<TIO*>(a.y_out)[row0 * 20 + i] = TIO(0);
        if (t == 0) { if (a.status) a.status[b] = -1; if (a.iters) a.iters[b] = 0; cs_host = done_cs_pack(-1, 0); }
        __syncthreads();
    } else {
    SRBDQP_STAMP(a, b, 2);

    // ================= phase H: T = S + E^-1 entry by entry into the C-layout register tiles =================
    const int NT = (n_g + 15) >> 4;
    const int NTT = (NT * (NT + 1)) >> 1;
    // Low-latency instantiation on four waves (N = 8, 10: two step waves + two set-up helpers), round 5: the tile phases F / W / I as ONE pipeline around the
    // chain of diagonal-tile inversions (below).  Wave j owns diagonal tile (j, j) AND the tile above it, (j - 1, j) -- so the next diagonal tile is updated from
    // its owner's registers and inverted without waiting for anybody -- ; wave 0 also owns (0, 2) and (1, 3), wave 2 (0, 3).  In the assembly wave 0 holds (0, 0)
    // only (it inverts it beside the other waves' assembly): its two other tiles are assembled by waves 1 and 3 in their free slot and handed over through LDS.
    constexpr bool LATP = XW == 2 && NW == 4 && sizeof(TT) == 8 && MODE == 0;
    static_assert(!LATP || (TS <= 3 && S::NT <= 4), "pipelined tile phases: at most 4 x 4 tiles, three slots per wave");
    constexpr int TSL = LATP ? 3 : TS;                                  // tile slots per wave in these phases
    int ta[TSL], tb[TSL];
#pragma unroll
    for (int s = 0; s < TSL; ++s) {
        if constexpr (LATP) {
            int a_ = -1, b_ = -1;
            if (s == 0) { a_ = w; b_ = w; }
            else if (s == 1) { if (w >= 1) { a_ = w - 1; b_ = w; } }
            else { if (w == 1) { a_ = 0; b_ = 2; } else if (w == 2) { a_ = 0; b_ = 3; } else if (w == 3) { a_ = 1; b_ = 3; } }
            const bool on = b_ >= 0 && b_ < NT;
            ta[s] = on ? a_ : -1; tb[s] = on ? b_ : -1;
        } else {
        // (low-latency instantiation: wave 0 owns tile (0, 0) ONLY -- it inverts it beside the other waves' assembly, below -- and the other NW - 1 waves
        // share the rest; with 10 tiles on 4 waves nobody holds more than before)
        constexpr bool LATK = XW > 0 && N <= 10;                       // (N = 24 has set-up helper waves too, in its batch instantiation)
        static_assert(!LATK || (NW - 1) * TS >= S::NTT - 1, "low-latency instantiation: tiles on NW - 1 waves");
        const int id = LATK ? ((w == 0) ? (s == 0 ? 0 : NTT) : 1 + (NW - 1) * s + (w - 1)) : NW * s + w;
        int bb = 0;
        while (((bb + 1) * (bb + 2)) / 2 <= id) ++bb;
        tb[s] = (id < NTT) ? bb : -1;
        ta[s] = (id < NTT) ? id - (bb * (bb + 1)) / 2 : -1;
        }
    }
    v4t acc[TSL];
    {
        // The rank-6 form of srbdqp_common.hpp in the g coordinates: S(r, c) = z_r' [M(j, m), same(j, m)] z_c with M(j, m) = D_m - C_j' E_m, so with a row
        // [z_ang, -C_j z_ang, z_lin] per g coordinate (built once, below) and [D_m z_ang, E_m z_ang, f0, f1] of the lane's column (formed per tile slot) an entry
        // is 6 + 2 multiply-adds and one E^-1 look-up.  The entry-by-entry loops this replaces spent most of their time on three levels of dependent index reads
        // per entry (the step of a coordinate, its offset, the step pair's table) and, for a pair of force variables, 21 LDS reads.
        const double s2 = a.s * a.s, dt4m2 = dt2m * dt2m, dt2m2 = dtm * dtm;
        const double* ZT = sm + S::o_zt;
        const double* DE = MT;
        constexpr bool ROWS_EARLY = TSPLIT;                              // (built by the helper wave beside phase E, above)
        constexpr bool BROWS_ = S::BROWS || ROWS_EARLY;
        double* AB = sm + (ROWS_EARLY ? S::o_abx : S::o_ab);
        [[maybe_unused]] const double* BB = sm + (ROWS_EARLY ? S::o_bbx : S::o_bb);
        typedef double d2 __attribute__((ext_vector_type(2)));
        if constexpr (!ROWS_EARLY) {
        for (int r = t; r < 16 * S::NT; r += BT) {
            const bool on = r < n_g;
            const int rr = on ? r : 0, j = gstep[rr], l = rr - igoff[j];
            const double* z = ZT + 6 * rr;
            const double* C = CP + 9 * j;
            const double z0 = z[0], z1 = z[1], z2 = z[2];
            const int rm3 = rr - 3 * ((rr * 0xAAAB) >> 17);
            const double gl = (rm3 == 0) ? z[3] : (rm3 == 1) ? z[4] : z[5];
            const double code = on ? (double)(4096 * j + 36 * j + 6 * l) : -4096.0;     // (step, offset of the coordinate's row in the step's E^-1) as one exact integer
            d2* row = reinterpret_cast<d2*>(AB + 8 * r);
            row[0] = on ? (d2){z0, z1} : (d2){0.0, 0.0};
            row[1] = on ? (d2){z2, -(C[0] * z0 + C[1] * z1 + C[2] * z2)} : (d2){0.0, 0.0};
            row[2] = on ? (d2){-(C[3] * z0 + C[4] * z1 + C[5] * z2), -(C[6] * z0 + C[7] * z1 + C[8] * z2)} : (d2){0.0, 0.0};
            row[3] = (d2){on ? gl : 0.0, code};
            if constexpr (S::BROWS) {   // the column side of the same coordinate, formed once here instead of once per tile slot by every lane of the column
                const double* D = DE + 18 * j;
                d2* brow = reinterpret_cast<d2*>(sm + S::o_bb + 6 * r);
                double bv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) bv[i] = on ? s2 * (D[3 * i] * z0 + D[3 * i + 1] * z1 + D[3 * i + 2] * z2) : 0.0;
                brow[0] = (d2){bv[0], bv[1]}; brow[1] = (d2){bv[2], bv[3]}; brow[2] = (d2){bv[4], bv[5]};
            }
        }
        __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < TSL; ++s) {
            acc[s] = (v4t){TT(0), TT(0), TT(0), TT(0)};
            if (ta[s] >= 0) {
                const int c = 16 * tb[s] + mcol;
                const bool cin = c < n_g;
                const int cc = cin ? c : 0;
                int mm, l2;
                double Bv[6], gc;
                const int cm3 = cc - 3 * ((cc * 0xAAAB) >> 17);
                if constexpr (BROWS_) {
                    const d2 g67 = reinterpret_cast<const d2*>(AB + 8 * cc)[3];          // the coordinate's own row: z_lin and (step, offset) code
                    const int cwc = (int)g67[1];
                    mm = cwc >> 12;
                    l2 = (((cwc & 4095) - 36 * mm) * 0xAAAB) >> 18;                         // (36 j + 6 l - 36 j) / 6
                    gc = s2 * g67[0];
                    const d2* brow = reinterpret_cast<const d2*>(BB + 6 * cc);
                    const d2 b01 = brow[0], b23 = brow[1], b45 = brow[2];
                    Bv[0] = b01[0]; Bv[1] = b01[1]; Bv[2] = b23[0]; Bv[3] = b23[1]; Bv[4] = b45[0]; Bv[5] = b45[1];
                } else {
                    mm = gstep[cc]; l2 = cc - igoff[mm];
                    const double* zc = ZT + 6 * cc;
                    const double* D = DE + 18 * mm;
                    const double c0 = zc[0], c1 = zc[1], c2 = zc[2];
#pragma unroll
                    for (int i = 0; i < 6; ++i) Bv[i] = s2 * (D[3 * i] * c0 + D[3 * i + 1] * c1 + D[3 * i + 2] * c2);      // D_m z_ang (3), E_m z_ang (3)
                    gc = s2 * ((cm3 == 0) ? zc[3] : (cm3 == 1) ? zc[4] : zc[5]);
                }
                const int Ls = N - mm;
                const double al = (double)(((Ls - 1) * Ls * (2 * Ls - 1)) / 6), be = (double)(((Ls - 1) * Ls) / 2);
                const double fa = SQ[3 + cm3] * SQ[3 + cm3] * dt4m2, fb = SQ[9 + cm3] * SQ[9 + cm3] * dt2m2;
                const double f0 = gc * (fa * (al + (double)mm * be) + fb * (double)Ls), f1 = gc * fa * be;
                const double* EI = sm + S::o_ei + l2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * ta[s] + crow<TT>(kq, q);
                    const d2* row = reinterpret_cast<const d2*>(AB + 8 * r);
                    const d2 a01 = row[0], a23 = row[1], a45 = row[2], a67 = row[3];
                    const int cw = (int)a67[1], cj = cw >> 12, ce = cw & 4095;
                    double v = a01[0] * Bv[0];
                    v = fma(a01[1], Bv[1], v); v = fma(a23[0], Bv[2], v); v = fma(a23[1], Bv[3], v); v = fma(a45[0], Bv[4], v); v = fma(a45[1], Bv[5], v);
                    const double ft = a67[0] * fma(-(double)cj, f1, f0);
                    const int d = r + 3 * 1024 - c, dm3 = d - 3 * ((d * 0xAAAB) >> 17);
                    v += (dm3 == 0) ? ft : 0.0;                                   // same axis
                    const bool same_step = cj == mm;
                    const double ei = EI[same_step ? ce : 0];
                    v += same_step ? ei : 0.0;
                    acc[s][q] = (TT)((r < n_g && cin) ? v : ((r == c) ? 1.0 : 0.0));   // padding -> identity
                }
            }
        }
    }
    // Low-latency instantiation: tile (0, 0) needs nothing from the other tiles, so its owner inverts it HERE, while the other waves still assemble theirs (it has one
    // tile less to assemble, above): the first of the factorisation's serial diagonal inversions (3.3 k cycles) leaves the chain.  The inverse waits in registers:
    // the tile store overlays the tables the other waves are still reading.
    [[maybe_unused]] v4d winv0 = (v4d){0.0, 0.0, 0.0, 0.0};
    [[maybe_unused]] bool pre0 = false;
    if constexpr (XW > 0 && N <= 10 && sizeof(TT) == 8 && MODE == 0) {
        if (ta[0] == 0 && tb[0] == 0) {                              // (wave-uniform)
            bool ok0;
            winv0 = diag16_invert_dpp(acc[0], lane, ok0, sm + S::o_pre);
            if (!ok0 && lane == 0) sm[S::o_misc] = 1.0;
            pre0 = true;
        }
    }
    if constexpr (LATP) {   // the two tiles assembled for wave 0
        if (w == 1 && ta[2] >= 0) store_tile_t<TT, false>(sm + S::o_hand, acc[2], lane);
        if (w == 3 && ta[2] >= 0) store_tile_t<TT, false>(sm + S::o_hand + 256, acc[2], lane);
    }
    SRBDQP_STAMP(a, b, 3);
    WAVE_ARRIVE(a, w, lane, 14);
    __syncthreads();
    if constexpr (MODE == 1) {   // assembly dump (tests): T dense [NG][NG], then q[12N], V rows / Bd rows per lane, goff
        double* out = a.P_out + (size_t)b * (S::NG * S::NG);
#pragma unroll
        for (int s = 0; s < TSL; ++s) {
            if (ta[s] >= 0) {
                const int c = 16 * tb[s] + mcol;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * ta[s] + kq + 4 * q;
                    if (r < S::NG && c < S::NG && r <= c) {          // (the upper triangle: the entries of a diagonal tile below it are never formed)
                        const double v = (r < n_g && c < n_g) ? acc[s][q] : 0.0;
                        out[r * S::NG + c] = v;
                        out[c * S::NG + r] = v;
                    }
                }
            }
        }
        if (stepok) {
            a.q_out[(size_t)b * n + uvar] = qv;
            double* vo = a.l_out + (size_t)b * (24 * n) + 24 * uvar;   // per force variable: Bd row (12), V column (6), V half row (6)
#pragma unroll
            for (int i = 0; i < 12; ++i) vo[i] = bdrow[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) { vo[12 + i] = vcol[i]; vo[18 + i] = vrow[i]; }
        }
        if (t <= N) a.ub_out[(size_t)b * (N + 1) + t] = (double)igoff[t];
        return;
    } else {

    if constexpr (LATP) {
    // ================= phases F / W / I as one static pipeline around the chain of diagonal-tile inversions (latp_run, above) =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
    if (w == 0) {   // wave 0 takes over the two tiles waves 1 and 3 assembled for it; its own tile (0, 0) is inverted already
        const bool on1 = NT > 2, on2 = NT > 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = kq + 4 * q;
            acc[1][q] = on1 ? sm[S::o_hand + row * 16 + mcol] : 0.0;
            acc[2][q] = on2 ? sm[S::o_hand + 256 + row * 16 + mcol] : 0.0;
        }
    }
    {
        double* misc = sm + S::o_misc;
        long long* stp = a.stamps;
        const double* D0 = sm + S::o_pre;           // L_00^-1, swizzled, where diag16_invert_dpp left it (the scratch tile lies outside the tile store and the tables)
        if (S::NT >= 4 && NT >= 4) latp_dispatch<(S::NT >= 4 ? 4 : 3)>(w, T, D0, acc[0], acc[1], acc[2], misc, lane, stp);       // (wave-uniform: every wave of the workgroup takes the same arm, with its barriers)
        else if (NT == 3) latp_dispatch<3>(w, T, D0, acc[0], acc[1], acc[2], misc, lane, stp);
        else if (NT == 2) latp_dispatch<2>(w, T, D0, acc[0], acc[1], acc[2], misc, lane, stp);
        else latp_dispatch<1>(w, T, D0, acc[0], acc[1], acc[2], misc, lane, stp);
    }
    SRBDQP_STAMP(a, b, 4);
    SRBDQP_STAMP(a, b, 5);
    SRBDQP_STAMP(a, b, 6);
    } else {
    // ================= phase F: tiled right-looking Cholesky T = U'U, trailing tiles in registers =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
    for (int j = 0; j < NT; ++j) {
        TT* Djj = T + tile_id(j, j) * 256;
        {
            bool mine = false;
            v4t d = acc[0];
#pragma unroll
            for (int s = 0; s < TSL; ++s)
                if (ta[s] == j && tb[s] == j) { mine = true; d = acc[s]; }
            if (mine) {
                int lane_j = lane;
                SRBDQP_PHASE_LOCAL("+v"(lane_j));
                bool ok = true;
                if (XW > 0 && j == 0 && pre0) {                      // (inverted beside the assembly: the inverse waited in registers)
                    const int cj = lane_j & 15, gj = lane_j >> 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int row = gj + 4 * q; Djj[row * 16 + (cj ^ row)] = (TT)winv0[q]; }
                } else if constexpr (sizeof(TT) == 8) {
                    // one column per lane, DPP multiply-adds (srbdqp_mfma.hpp): through the tile's own slot of the store, which receives the inverse
                    store_tile_t<TT, false>(Djj, d, lane_j);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // one wave: its LDS operations complete in order
                    diag16_invert_dpp_tiles<TT, TT>(Djj, Djj, lane_j, ok);
                } else {   // fp32 tiles: the accumulator through the wave's scratch tile; the 16 x 16 inverse is formed in fp64 and stored in fp32
                    float* wsx = reinterpret_cast<float*>(sm + S::o_ws) + 256 * w;
                    store_tile_t<float, false>(wsx, d, lane_j);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    diag16_invert_dpp_tiles<float, TT>(wsx, Djj, lane_j, ok);
                }
                if (!ok && lane == 0) sm[S::o_misc] = 1.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TSL; ++s) {
            if (ta[s] == j && tb[s] > j) {
                v4t o = (v4t){TT(0), TT(0), TT(0), TT(0)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = crow<TT>(kq, r);           // accumulator register r = the B operand of this k index
                    TT av = Djj[mcol * 16 + (k ^ mcol)];
                    av = (k <= mcol) ? av : TT(0);
                    o = mma16(av, acc[s][r], o);
                }
                acc[s] = o;
                store_tile_t<TT, false>(T + tile_id(j, tb[s]) * 256, o, lane);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TSL; ++s) {
            if (ta[s] > j) {
                const TT* Ua = T + tile_id(j, ta[s]) * 256;
                const TT* Ub = T + tile_id(j, tb[s]) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    acc[s] = mma16(-Ua[k * 16 + mcol], Ub[k * 16 + mcol], acc[s]);
                }
            }
        }
    }
    __syncthreads();
    SRBDQP_STAMP(a, b, 4);

    // ================= phase W: W = L^-1 block row by block row, in place over U =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
    for (int i = 1; i < NT; ++i) {
        v4t res[S::WQ];
        const TT* Dii = T + tile_id(i, i) * 256;
#pragma unroll
        for (int q = 0; q < S::WQ; ++q) {
            const int j = w + NW * q;
            res[q] = (v4t){TT(0), TT(0), TT(0), TT(0)};
            if (j < i) {
                v4t o = (v4t){TT(0), TT(0), TT(0), TT(0)};
                {
                    const TT* Uji = T + tile_id(j, i) * 256;
                    const TT* Djj = T + tile_id(j, j) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        TT bv = Djj[k * 16 + (mcol ^ k)];
                        bv = (mcol <= k) ? bv : TT(0);
                        o = mma16(Uji[k * 16 + mcol], bv, o);
                    }
                }
                for (int k2 = j + 1; k2 < i; ++k2) {
                    const TT* Uki = T + tile_id(k2, i) * 256;
                    const TT* Wkj = T + tile_id(j, k2) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + kq;
                        o = mma16(Uki[k * 16 + mcol], Wkj[k * 16 + mcol], o);
                    }
                }
                v4t o2 = (v4t){TT(0), TT(0), TT(0), TT(0)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = crow<TT>(kq, r);
                    TT av = Dii[mcol * 16 + (k ^ mcol)];
                    av = (k <= mcol) ? -av : TT(0);
                    o2 = mma16(av, o[r], o2);
                }
                res[q] = o2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < S::WQ; ++q) {
            const int j = w + NW * q;
            if (j < i) store_tile_t<TT, false>(T + tile_id(j, i) * 256, res[q], lane);
        }
        __syncthreads();
    }
    SRBDQP_STAMP(a, b, 5);

    // ================= phase I: T^-1 = W'W =================
    SRBDQP_PHASE_LOCAL("+v"(mcol), "+v"(kq));
#pragma unroll
    for (int s = 0; s < TSL; ++s) {
        acc[s] = (v4t){TT(0), TT(0), TT(0), TT(0)};
        if (ta[s] >= 0) {
            const int ia = ta[s], ib = tb[s];
            const TT* Dbb = T + tile_id(ib, ib) * 256;
            {
                const TT* Wba = T + tile_id(ia, ib) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    TT dv = Dbb[k * 16 + (mcol ^ k)];
                    dv = (mcol <= k) ? dv : TT(0);
                    const TT av = (ia < ib) ? Wba[k * 16 + mcol] : dv;
                    acc[s] = mma16(av, dv, acc[s]);
                }
            }
            for (int i = ib + 1; i < NT; ++i) {
                const TT* Wia = T + tile_id(ia, i) * 256;
                const TT* Wib = T + tile_id(ib, i) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * r + kq;
                    acc[s] = mma16(Wia[k * 16 + mcol], Wib[k * 16 + mcol], acc[s]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TSL; ++s)
        if (ta[s] >= 0) store_tile_t<TT, true>(T + tile_id(ta[s], tb[s]) * 256, acc[s], lane);
    __syncthreads();
    SRBDQP_STAMP(a, b, 6);
    }   // !LATP

#ifndef SRBDQP_WRENCH_REROLE
#define SRBDQP_WRENCH_REROLE 1
#endif
    if constexpr (SRBDQP_WRENCH_REROLE && (VBD_LATE || SRBDQP_WRENCH_REROLE > 1)) {
        // the lane roles again, from a lane index the compiler cannot trace: the first set is dead from the last use in phase E on
        // instead of waiting in scratch memory across phases F / W / I (24 registers of the 168; the flags come from LDS again)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        lane_roles(ln);
        lane_roles2();
    }
    constexpr bool VPARK = VBD_LATE && S::VPARK;
    auto late_vbd = [&]() __attribute__((always_inline)) {   // rows / columns of V and Bd now that the accumulator tiles are gone (see phase E)
        if (wrench) {
            const double* E4 = sm + S::o_e4 + 21 * js;
            auto tri = [&](int r, int c) -> double { const int hi = r > c ? r : c, lo = r > c ? c : r; return E4[(hi * (hi + 1)) / 2 + lo]; };
            const double* Jj = sm + S::o_J + js * 36;
            const double j0 = Jj[ul], j1 = Jj[12 + ul], j2 = Jj[24 + ul];
            double er[6], yv[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) er[c] = tri(rl, c);
#pragma unroll
            for (int r = 0; r < 6; ++r) yv[r] = tri(r, 0) * j0 + tri(r, 1) * j1 + tri(r, 2) * j2 + tri(r, 3 + ax);
            form_vbd(er, yv);
        } else {
            form_vbd_identity();
        }
    };
    if constexpr (VBD_LATE && !VPARK) {
        late_vbd();
        asm volatile("" ::: "memory");   // keep the reads of the T^-1 rows below this block (their 60 registers)
    }
    // ================= half rows of T^-1 (in the tiles' type), then x_q = -K^-1 q accumulated in fp64 =================
    const int CH = 2 * ((n_g + 3) / 4);
    TT kin64[CHMAX];
    {
        const int rr = active_g ? Rrow : 0;
#pragma unroll
        for (int cc = 0; cc < CHMAX; ++cc) {
            const int c = CH * h + cc;
            const bool ok = active_g && (cc < CH) && (c < n_g);
            const int cs = ok ? c : 0;
            const int lo = (rr <= cs) ? rr : cs, hi = (rr <= cs) ? cs : rr;
            const int row = lo & 15, col = hi & 15;
            const TT v = T[tile_id(lo >> 4, hi >> 4) * 256 + row * 16 + (col ^ row)];
            kin64[cc] = ok ? v : TT(0);
        }
    }
    XQSTAMP(a, b, 10);
    const bool failed = sm[S::o_misc] != 0.0;
    const int vsoff = stepok ? igoff[js] : 0;                         // the step's own v = V w inside the v buffer
    __syncthreads();   // tiles are dead; region R becomes the ADMM vectors
    if constexpr (XW > 0) { if (w >= NWS) __builtin_amdgcn_endpgm(); }   // the set-up helpers end here (s_barrier counts live waves only)
    [[maybe_unused]] const double* vpr = sm + S::o_vpr + (stepok ? uvar : 0);
    [[maybe_unused]] const double* vpc = sm + S::o_vpc + (stepok ? uvar : 0);
    if constexpr (VPARK) {   // V formed only now, and parked in the dead tile region until the iterations start
        late_vbd();
        if (stepok) {
            double* pr = sm + S::o_vpr + uvar;
            double* pc = sm + S::o_vpc + uvar;
#pragma unroll
            for (int i = 0; i < 6; ++i) { pr[i * n] = vrow[i]; pc[i * n] = vcol[i]; }   // own entries only: no barrier needed
        }
    }
    const int vssel = vsoff + bsel;
    constexpr int KREG = (sizeof(R) == 8 && S::KTAIL > 0) ? CHMAX - S::KTAIL : CHMAX;
    [[maybe_unused]] const R* ktail = nullptr;
    if constexpr (KREG < CHMAX) {
        R* kt = reinterpret_cast<R*>(sm + S::o_kt) + t;
#pragma unroll
        for (int cc = KREG; cc < CHMAX; ++cc) kt[(cc - KREG) * LT] = (R)kin64[cc];
        ktail = kt;                                           // (published by the barriers in front of its first use)
    }
    int status = -1, iters = 0;
    R x = R(0), yA = R(0), yB = R(0);
    if (!failed) {
        double xq;
        if constexpr (S::QV_PARK) qv = stepok ? (double)reinterpret_cast<const float*>(sm + S::o_qv)[uvar] + (double)qv_lo : 0.0;
        {
            double* wbw = sm + S::o_wb + 64 * w;
            double* tbw = sm + S::o_tb + 32 * w;
            double* vb = sm + S::o_vb;
            for (int i = t; i < 2 * S::VB; i += LT) vb[i] = 0.0;
            __syncthreads();
            double vrd[6], vcd[6];
            [[maybe_unused]] double bdd[12];
#pragma unroll
            for (int i = 0; i < 6; ++i) { vrd[i] = VPARK ? 0.0 : (double)vrow[i]; vcd[i] = VPARK ? 0.0 : (double)vcol[i]; }
            if constexpr (VPARK) {
                xq = apply_kinv<double, CHMAX, CHMAX, true>(-qv, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bjv, vsoff, vssel, [] {}, nullptr, n, vpr, vpc);
            } else if constexpr (BD_EXPLICIT && !VBD_LATE) {
#pragma unroll
                for (int i = 0; i < 12; ++i) bdd[i] = (double)bdrow[i];
                xq = apply_kinv<double, CHMAX>(-qv, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bdd, vsoff, vssel, [] {});
            } else if constexpr (KREG < CHMAX) {
                xq = apply_kinv<double, CHMAX, KREG>(-qv, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bjv, vsoff, vssel, [] {}, ktail, LT);
            } else {
                xq = apply_kinv<double, CHMAX>(-qv, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bjv, vsoff, vssel, [] {});
            }
            xq = active_u ? xq : 0.0;
            __syncthreads();
            XQSTAMP(a, b, 11);
            if constexpr (sizeof(TT) == 4) {
                // fp32 tiles: T^-1 is good to ~1e-4 only.  One step of iterative refinement with the fp64 residual
                // r = K x_q + q = G'(G x_q) + D x_q + q (closed form, as the warm start's P x^0) takes the large gradient
                // out of that error: what is left is 1e-4 of the iteration's own right-hand sides (oracle: admm_solve_split).
                if (t < 9) {   // the prefix sums of Rz' lay under the tiles
                    double accp = 0.0;
                    for (int k = 0; k < N; ++k) { accp += sm[S::o_tm + k * 9 + t]; sm[S::o_cp + k * 9 + t] = accp; }
                }
                if (stepok) sm[S::o_x0c + uvar] = xq;
                __syncthreads();
                const double gtg = gtg_of_x0c();
                const double dl = (ax < 2) ? dxy : dz;
                const double rres = active_u ? fma(dl, xq, gtg + qv) : 0.0;
                for (int i = t; i < 2 * S::VB; i += LT) vb[i] = 0.0;
                __syncthreads();
                XQSTAMP(a, b, 12);
                double dxq;
                if constexpr (VPARK) dxq = apply_kinv<double, CHMAX, CHMAX, true>(-rres, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bjv, vsoff, vssel, [] {}, nullptr, n, vpr, vpc);
                else dxq = apply_kinv<double, CHMAX>(-rres, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin64, vrd, vcd, bjv, vsoff, vssel, [] {});
                xq = active_u ? xq + dxq : 0.0;
                __syncthreads();
            }
            if constexpr (WARM_LATE) {
                if (a.warm_u) {
                    if constexpr (sizeof(R) == 4) {   // x_q is only used in the iteration type from here on: one register instead of two across the tables
                        R xq_r = (R)xq;
                        asm volatile("" : "+v"(xq_r));
                        xq = (double)xq_r;
                    }
                    x_init = active_u ? (double)gwu[uvar] / a.s : 0.0;
                    if (stepok) sm[S::o_x0c + uvar] = x_init;
                    __syncthreads();
                    const double gtg = gtg_of_x0c();
                    px0 = active_u ? gtg + a.rs2 * x_init : 0.0;
                    __syncthreads();
                }
            }
        }
        SRBDQP_STAMP(a, b, 7);

        if constexpr (BD_LAST) {
            if (wrench) {
                const double* E4 = sm + S::o_e4 + 21 * js;
                auto tri = [&](int r, int c) -> double { const int hi = r > c ? r : c, lo = r > c ? c : r; return E4[(hi * (hi + 1)) / 2 + lo]; };
                const double* Jj = sm + S::o_J + js * 36;
                const double j0 = Jj[ul], j1 = Jj[12 + ul], j2 = Jj[24 + ul];
                double yv[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) yv[r] = tri(r, 0) * j0 + tri(r, 1) * j1 + tri(r, 2) * j2 + tri(r, 3 + ax);
                form_bd(yv);
            } else {
#pragma unroll
                for (int u2 = 0; u2 < 12; ++u2) bdrow[u2] = BS(0);
            }
        }
        // ================= ADMM iterations (a9) in R =================
        R* wbw = reinterpret_cast<R*>(sm + S::o_wb) + 64 * w;
        R* tbw = reinterpret_cast<R*>(sm + S::o_tb) + 32 * w;
        R* vbuf = reinterpret_cast<R*>(sm + S::o_vb);
        float* redf = reinterpret_cast<float*>(sm + S::o_red);          // [2][4 NW] check maxima
        int* vflag = reinterpret_cast<int*>(redf + 8 * NWS);            // [NWS] pre-test votes
        for (int i = t; i < 2 * S::VB; i += LT) vbuf[i] = R(0);
        if (t < NWS) vflag[t] = 0;
        constexpr int BDN = BD_EXPLICIT ? 12 : 4;
        R kin[CHMAX], vr[6], vc[6], bd[BDN];
#pragma unroll
        for (int cc = 0; cc < CHMAX; ++cc) kin[cc] = (R)kin64[cc];
#pragma unroll
        for (int i = 0; i < 6; ++i) { vr[i] = VPARK ? (R)vpr[i * n] : (R)vrow[i]; vc[i] = VPARK ? (R)vpc[i * n] : (R)vcol[i]; }
        constexpr bool VL = SRBDQP_WRENCH_VLDS && sizeof(R) == 8 && sizeof(TT) == 8 && CHMAX <= 36 && XW == 0;   // (the low-latency instantiation has the registers)
        [[maybe_unused]] const R* vlds = nullptr;
        [[maybe_unused]] const double* jlds = sm + S::o_J + js * 36 + ul;
        if constexpr (VL) {
            R* vt = reinterpret_cast<R*>(sm + S::o_vl) + t;
#pragma unroll
            for (int i = 0; i < 6; ++i) { vt[i * LT] = vr[i]; vt[(6 + i) * LT] = vc[i]; }
            vlds = vt;                                        // own entries only: no barrier needed
        }
#pragma unroll
        for (int i = 0; i < BDN; ++i) bd[i] = BD_EXPLICIT ? (R)bdrow[i] : (R)bjv[i < 4 ? i : 0];
        const R xqr = (R)xq;
        const bool rowA = active_u, rowB = active_u && ax < 2;
        const R sigma = uni((R)a.sigma), alpha = uni((R)a.alpha), oma = uni((R)(1.0 - a.alpha)), mu = uni((R)a.mu), irho = uni((R)(1.0 / rho_b));
        const R irhoz = uni((R)(1.0 / (rho_b * a.rho_fz)));                  // slot A of the fz lane = the normal-force row: its own penalty
        const R rhoA = rowA ? ((ax < 2) ? (R)rho_b : (R)(rho_b * a.rho_fz)) : R(0), rhoB = rowB ? (R)rho_b : R(0);
        const R loA = !rowA ? R(0) : (ax < 2 ? (R)-kInf : (R)a.fzmin_s), hiA = !rowA ? R(0) : (ax < 2 ? R(0) : (R)a.fzmax_s);
        const R loB = rowB ? (R)-kInf : R(0), hiB = R(0);
        const R mucA = (ax < 2) ? mu : R(0);
        auto At = [&](R wA, R wB) -> R {
            const R ssum = wA + wB;
            const R sxy = contact_sum_xy(ssum);                       // (the lanes of a contact are neighbours: wave shifts, srbdqp_admm.hpp; outside the select:
            return (ax < 2) ? wA - wB : fma(-mu, sxy, wA);            //  a cross-lane read must run with the source lanes enabled)
        };
        x = (R)x_init;
        R cpx = (R)(px0 + qv), spxA = R(0), spxB = R(0);             // c = P x + q = cpx - A' spx
        yA = (rowA && a.warm_y) ? (R)gwy[irowA] : R(0);
        yB = (rowB && a.warm_y) ? (R)gwy[irowB] : R(0);
        const R fz0 = contact_fz(x, ax);
        R axA = rowA ? fma(-mucA, fz0, x) : R(0), axB = rowB ? fma(-mu, fz0, -x) : R(0);
        R zA = rmin(rmax(axA, loA), hiA), zB = rmin(rmax(axB, loB), hiB);
        const float qnf = uni((float)wg_max1<NWS>(fabs(qv), sm + S::o_red + 24));
        R wv = fma(sigma, x, At(fma(rhoA, zA, -yA), fma(rhoB, zB, -yB)));
        __syncthreads();
        // fp32 iterations cannot certify residuals below ~2e-6 (1 + norm): the maxima themselves carry a few ulp of noise
        const double eps_a = unis((sizeof(R) == 4) ? fmax(a.eps_abs, 2.0e-6) : a.eps_abs), eps_r = unis((sizeof(R) == 4) ? fmax(a.eps_rel, 2.0e-6) : a.eps_rel);
        status = 2; iters = a.max_iter;
        int nchk = 0, ph = 0;
        bool pending = false, vote_ok = true, done = false;
        double e_prim_last = kInf * 1.0e10;
        float lastv0 = 0.0f, lastv1 = 0.0f, lastv2 = 0.0f, lastv3 = 0.0f;   // maxima of the last full check (restart rule)
        WADMM_DECL;
        // the low-latency instantiation runs alone on its CU: what its loop costs depends on where its first instruction falls in a 64-byte fetch
        // line (1467 ... 1494 cycles per iteration over five placements, profiles/r05_loop_alignment.txt), so the placement is pinned
        if constexpr (XW == 2) asm volatile(".p2align 6");
        for (int k = 1; k <= a.max_iter + 1 && !done; ++k) {
            WADMM_T(0);
            R* vb = vbuf + (k & 1) * S::VB;
            const R kw = apply_kinv<R, CHMAX, KREG, VL, (XW > 0 && CHMAX <= 36), (VL && sizeof(R) == 8 && CHMAX > 30 && SRBDQP_WRENCH_JLDS != 0)>(wv, wbw, tbw, vb, lane, sg, ul, active_g, Rrow, CH, kin, vr, vc, bd, vsoff, vssel, [&] {
                if (pending) {   // decision of the check made at iteration k - 1 (its maxima were published by this barrier)
                    const float* buf = redf + ((nchk - 1) & 1) * 4 * NWS;
                    float v0 = buf[0], v1 = buf[1], v2 = buf[2], v3 = buf[3];
#pragma unroll
                    for (int q = 1; q < NWS; ++q) { v0 = fmaxf(v0, buf[4 * q]); v1 = fmaxf(v1, buf[4 * q + 1]); v2 = fmaxf(v2, buf[4 * q + 2]); v3 = fmaxf(v3, buf[4 * q + 3]); }
                    v0 = uni(v0); v1 = uni(v1); v2 = uni(v2); v3 = uni(v3);
                    const double e_prim = uni(eps_a + eps_r * (double)v1);
                    const double e_dual = eps_a + eps_r * fmax((double)v3, (double)qnf);
                    e_prim_last = e_prim;
                    lastv0 = v0; lastv1 = v1; lastv2 = v2; lastv3 = fmaxf(v3, qnf);
                    if (!((double)v0 <= kInf) || !((double)v2 <= kInf)) { status = -1; iters = k - 1; done = true; }
                    else if ((double)v0 <= e_prim && (double)v2 <= e_dual) { status = 1; iters = k - 1; done = true; }
                    pending = false;
                }
                if (++ph == a.check_every) ph = 0;
                if (ph == 0) {
                    int vsum = 0;
#pragma unroll
                    for (int q = 0; q < NWS; ++q) vsum |= vflag[q];
                    vote_ok = (vsum == 0);
                }
            }, ktail, LT, vlds, nullptr, jlds WADMM_ARGS);
            WADMM_T(4);
            if (done || k > a.max_iter) break;
            const bool check = ((ph == 0) && vote_ok) || (k == a.max_iter);
            const bool pretest = (ph == a.check_every - 1);
            const R xt = active_u ? xqr + kw : R(0);
            const R fzt = contact_fz(xt, ax);
            WADMM_T(5);
            R atw;
            if constexpr (sizeof(R) == 4) {
                // rows A and B of the contact's cone block as one 2-vector: packed fp32 instructions in the fp32 kernels
                typedef R R2v __attribute__((ext_vector_type(2)));
                const R2v alpha2 = (R2v){alpha, alpha}, oma2 = (R2v){oma, oma}, rho2 = (R2v){rhoA, rhoB};
                const R2v z2 = (R2v){zA, zB}, y2 = (R2v){yA, yB};
                const R2v zt2 = __builtin_elementwise_fma((R2v){-mucA, -mu}, (R2v){fzt, fzt}, (R2v){xt, -xt});
                const R2v nu2 = __builtin_elementwise_fma(rho2, zt2 - z2, y2);
                const R2v zh2 = __builtin_elementwise_fma(alpha2, zt2, oma2 * z2);
                const R2v zc2 = __builtin_elementwise_fma(y2, (R2v){(ax < 2) ? irho : irhoz, irho}, zh2);
                const R znA = rmin(rmax(zc2[0], loA), hiA), znB = rmin(rmax(zc2[1], loB), hiB);
                const R2v zn2 = (R2v){znA, znB};
                const R2v yn2 = __builtin_elementwise_fma(rho2, zh2 - zn2, y2);
                yA = yn2[0]; yB = yn2[1];
                zA = znA; zB = znB;
                const R2v ax2 = __builtin_elementwise_fma(alpha2, zt2, oma2 * (R2v){axA, axB});
                axA = ax2[0]; axB = ax2[1];
                const R2v spx2 = __builtin_elementwise_fma(alpha2, nu2, oma2 * (R2v){spxA, spxB});
                spxA = spx2[0]; spxB = spx2[1];
                const R2v aw2 = __builtin_elementwise_fma(rho2, zn2, -yn2);
                atw = At(aw2[0], aw2[1]);
            } else {
                const R ztA = fma(-mucA, fzt, xt), ztB = fma(-mu, fzt, -xt);
                const R nuA = fma(rhoA, ztA - zA, yA), nuB = fma(rhoB, ztB - zB, yB);
                const R zhA = fma(alpha, ztA, oma * zA), zhB = fma(alpha, ztB, oma * zB);
                R znA, znB;
                if constexpr (VL) {   // (168 registers: the row classes as selects on scalar bounds -- eight registers of per-lane bounds less in the loop, which reloaded four values per iteration)
                    const R vA = fma(yA, (ax < 2) ? irho : irhoz, zhA), vB = fma(yB, irho, zhB);
                    znB = rowB ? rmin(vB, R(0)) : R(0);
                    znA = rowA ? ((ax < 2) ? rmin(vA, R(0)) : rmin(rmax(vA, (R)a.fzmin_s), (R)a.fzmax_s)) : R(0);
                } else {
                    znA = rmin(rmax(fma(yA, (ax < 2) ? irho : irhoz, zhA), loA), hiA); znB = rmin(rmax(fma(yB, irho, zhB), loB), hiB);
                }
                yA = fma(rhoA, zhA - znA, yA); yB = fma(rhoB, zhB - znB, yB);
                zA = znA; zB = znB;
                axA = fma(alpha, ztA, oma * axA); axB = fma(alpha, ztB, oma * axB);
                spxA = fma(alpha, nuA, oma * spxA); spxB = fma(alpha, nuB, oma * spxB);
                atw = At(fma(rhoA, zA, -yA), fma(rhoB, zB, -yB));
            }
            cpx = fma(alpha, sigma * (x - xt), oma * cpx);
            x = fma(alpha, xt, oma * x);
            wv = fma(sigma, x, atw);
            WADMM_T(6);
            if (pretest) {
                const bool bad = (rowA && !((double)rabs(axA - zA) <= e_prim_last)) || (rowB && !((double)rabs(axB - zB) <= e_prim_last));
                const unsigned long long bal = __ballot(bad);
                if (lane == 0) vflag[w] = (bal != 0ull) ? 1 : 0;
            }
            if (check) {
                const R aty = At(yA, yB), cc = cpx - At(spxA, spxB);      // cc = P x + q
                R rd = rabs(cc + aty);
                R rp = rmax(rowA ? rabs(axA - zA) : R(0), rowB ? rabs(axB - zB) : R(0));
                rd = (rd == rd) ? rd : (R)(kInf * 10.0);
                rp = (rp == rp) ? rp : (R)(kInf * 10.0);
                const R nr = rmax(rowA ? rmax(rabs(axA), rabs(zA)) : R(0), rowB ? rmax(rabs(axB), rabs(zB)) : R(0));
                const float v0 = (float)rp, v1 = (float)nr;
                const float v2 = active_u ? (float)rd : 0.0f, v3 = active_u ? (float)rmax(rabs(cc - (R)qv), rabs(aty)) : 0.0f;
                float m0 = v0, m1 = v1, m2 = v2, m3 = v3;
                wave_maxf4_nonneg(m0, m1, m2, m3);
                if (lane == 63) {
                    float* buf = redf + (nchk & 1) * 4 * NWS + 4 * w;
                    buf[0] = m0; buf[1] = m1; buf[2] = m2; buf[3] = m3;
                }
                ++nchk; pending = true;
            }
#ifdef SRBDQP_PROFILE_WADMM
            WADMM_T(7);
            for (int q_ = 0; q_ < 7; ++q_) wadmm_s[q_] += wadmm_t[q_ + 1] - wadmm_t[q_];
#endif
        }
#ifdef SRBDQP_PROFILE_WADMM
        if (a.stamps && t == 0 && a.B == 1) { for (int q_ = 0; q_ < 7; ++q_) a.stamps[16 + q_] = wadmm_s[q_]; a.stamps[16 + 7] = nchk; }   // (B = 1 probes: second row of the stamp buffer)
#endif
        if (status < 0) { x = R(0); yA = R(0); yB = R(0); }
        if (a.resid_out && status == 2 && t == 0) {   // for the rho restart (second launch over the capped QPs, srbdqp.hip)
            float* ro = a.resid_out + (size_t)b * 4;
            ro[0] = lastv0; ro[1] = lastv1; ro[2] = lastv2; ro[3] = lastv3;
            if (a.cap_list) a.cap_list[atomicAdd(a.cap_count, 1)] = b;
        }
    }
    SRBDQP_STAMP(a, b, 8);
    __syncthreads();
    if constexpr (SRBDQP_WRENCH_REROLE && VBD_LATE) {   // the roles the stores below need, derived again instead of waiting in scratch memory across the iterations
        int ln = lane;
        asm volatile("" : "+v"(ln));
        lane_roles(ln);
    }
    if (stepok) sm[S::o_xs + uvar] = (double)x;
    if (a.y_out && (!a.y_capped_only || status == 2)) {
        TIO* yo = reinterpret_cast<TIO*>(a.y_out) + row0 * 20;
        if (stepok) {
            const bool on = active_u;
            yo[irowA] = on ? (TIO)yA : TIO(0);
            if (ax < 2) yo[irowB] = on ? (TIO)yB : TIO(0);
        }
    }
    if (t == 0) {
        if (a.status) a.status[b] = status;
        if (a.iters) a.iters[b] = iters + a.iters_base;
        cs_host = done_cs_pack(status, iters + a.iters_base);
    }
    __syncthreads();
    }   // MODE
    }   // !degenerate
    }   // na > 0

    // ================= roll-out (a10) and stores =================
    if constexpr (XW > 0) { if (w >= NWS) __builtin_amdgcn_endpgm(); }   // (early-exit paths: the set-up helpers are still here)
    {
        const double* uh = sm + S::o_xs;
        double* scratch = sm + S::o_scr;
        const size_t row0 = a.row_off ? (size_t)a.row_off[b] : (size_t)b * N;
        TIO* uo = reinterpret_cast<TIO*>(a.u_out) + row0 * 12;
        ESTAMP(a, 4);
        for (int c = t; c < n; c += LT) {
            const TIO v = (TIO)(a.s * uh[c]);
            uo[c] = v;
            if constexpr (CSUM) cs_host ^= (unsigned long long)__double_as_longlong((double)v);
        }
        ESTAMP(a, 5);
        if constexpr (sizeof(TIO) == 8) { if (a.u_dev) for (int c = t; c < n; c += LT) a.u_dev[row0 * 12 + c] = a.s * uh[c]; }
        // (the completion word's address and value in scalar registers now: left to the end, their loads from the argument segment are two more round trips)
        int32_t* dflag = a.done_flag;
        int32_t dval = a.done_value, dcs = a.done_cs;
        if constexpr (CSUM) asm volatile("" : "+s"(dflag), "+s"(dval), "+s"(dcs));
        if (a.x_out) {
            const double* x0 = sm + S::o_x0;
            double* sj = scratch + 6 * N;
            // the rows of x, laid out as they go out, where u_hat was (dead behind the first barrier) -- every stage below writes its entries in place and the
            // store loop is a plain copy.  (Until round 5 the store loop formed Euler angles and positions entry by entry inside a five-way branch over
            // (k, component) -- 4.1 k of the roll-out's 6.5 k cycles at batch 1 --, then gathered them from three arrays through a divergent address select.)
            double* xrow = sm + S::o_xs;
            static_assert(13 * (N + 1) <= n + 6 * N, "the rows of x end in front of the per-step sums");
            for (int idx = t; idx < 6 * N; idx += LT) {
                const int j = idx / 6, comp = idx % 6;
                const double* u = uh + 12 * j;
                double s;
                if (comp < 3) {
                    const double* J = sm + S::o_J + j * 36 + comp * 12;
                    s = 0.0;
                    for (int c = 0; c < 12; ++c) s += J[c] * u[c];
                } else {
                    const int ax2 = comp - 3;
                    s = (u[ax2] + u[3 + ax2] + u[6 + ax2] + u[9 + ax2]) * a.inv_mass;
                }
                sj[idx] = s;
            }
            __syncthreads();
            ESTAMP(a, 6);
            // angular and linear velocities of the steps 1 .. N; row 0 (= x0) and the gravity column from the far end of the workgroup
            for (int idx = t; idx < 6 * N; idx += LT) {
                const int k = idx / 6 + 1, comp = idx % 6;
                double acc2 = 0.0;   // (all N steps, the later ones adding exact zeros: the reads of all trips in flight together, srbdqp_common.hpp rollout_and_store_to)
                if constexpr (N <= 12) {
#pragma unroll
                    for (int j = 0; j < N; ++j) { const double sv = sj[6 * j + comp]; acc2 += (j < k) ? sv : 0.0; }
                } else {   // (long horizons: k grows with the wave, and the waves of the early steps are done after a few trips)
                    for (int j = 0; j < k; ++j) acc2 += sj[6 * j + comp];
                }
                double v = x0[6 + comp] + a.dt * a.s * acc2;
                if (comp == 5) v += (double)k * a.dt * x0[12];
                xrow[13 * k + 6 + comp] = v;
            }
            for (int e = LT - 1 - t; e < 13 + N; e += LT) xrow[e < 13 ? e : 13 * (e - 12) + 12] = x0[e < 13 ? e : 12];
            __syncthreads();
            ESTAMP(a, 7);
            // Euler angles and CoM positions of the steps 1 .. N as ONE more prefix stage (6 N entries, a lane each, the reads of every trip in flight; the two
            // kinds of entries on different waves where the workgroup has more than one: no divergent arms)
            constexpr int PA_OFF = (64 * ((3 * N + 63) / 64) + 3 * N <= LT) ? 64 * ((3 * N + 63) / 64) : 3 * N;
            for (int tt = t; tt < PA_OFF + 3 * N; tt += LT) {
                if (tt < 3 * N || tt >= PA_OFF) {
                    const bool posn = tt >= PA_OFF;
                    const int e = posn ? tt - PA_OFF : tt;
                    const int k = e / 3 + 1, c3 = e % 3;
                    double acc2 = 0.0;
                    if (posn) {                                              // position: p_0 + dt (v_0 + ... + v_{k-1})
                        if constexpr (N <= 12) {
#pragma unroll
                            for (int l = 0; l < N; ++l) { const double sv = xrow[13 * l + 9 + c3]; acc2 += (l < k) ? sv : 0.0; }
                        } else {
                            for (int l = 0; l < k; ++l) acc2 += xrow[13 * l + 9 + c3];
                        }
                    } else {                                                 // Euler angles: theta_0 + dt sum_{l < k} Rz(psi_l)' omega_l
                        auto term_of = [&](int l) {
                            const double* Tm = sm + S::o_tm + l * 9 + c3 * 3;
                            const double* wv2 = xrow + 13 * l + 6;
                            return Tm[0] * wv2[0] + Tm[1] * wv2[1] + Tm[2] * wv2[2];
                        };
                        if constexpr (N <= 12) {
#pragma unroll
                            for (int l = 0; l < N; ++l) { const double term = term_of(l); acc2 += (l < k) ? term : 0.0; }
                        } else {
                            for (int l = 0; l < k; ++l) acc2 += term_of(l);
                        }
                    }
                    const int comp = c3 + (posn ? 3 : 0);
                    xrow[13 * k + comp] = x0[comp] + a.dt * acc2;
                }
            }
            __syncthreads();
            TIO* xo = reinterpret_cast<TIO*>(a.x_out) + (row0 + (size_t)b) * 13;      // N + 1 rows per QP
            for (int idx = t; idx < 13 * (N + 1); idx += LT) {
                const TIO v = (TIO)xrow[idx];
                xo[idx] = v;
                if constexpr (CSUM) cs_host ^= (unsigned long long)__double_as_longlong((double)v);
            }
        }
        // the one staged QP: the completion word with the checksum of what went to the host, no fence (srbdqp_common.hpp signal_done_checksum)
        if constexpr (CSUM) { if (dcs) signal_done_checksum<LT>(dflag, dval, cs_host); }
    }
    SRBDQP_STAMP(a, b, 9);
}

template <int N, typename R, typename TIO, int MODE, int WPS, typename TT = double, int SPW = 5, int XW = 0>
__global__ __launch_bounds__((WrenchSmem<N, 8, SPW, XW>::BT), WPS) void srbdqp_wrench_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= a.B) return;
    if (a.tile_sel) {   // fp32 calls: this launch takes the QPs with (1) / without (2) wrench coordinates only
        const uint32_t* cf = reinterpret_cast<const uint32_t*>(a.contact + (size_t)SRBDQP_QP_INDEX(a) * (N * 4));   // 4 flags per step
        const int lane = threadIdx.x & 63;
        const uint32_t v = cf[lane < N ? lane : 0];
        const int nc = ((v & 0xffu) ? 1 : 0) + ((v & 0xff00u) ? 1 : 0) + ((v & 0xff0000u) ? 1 : 0) + ((v & 0xff000000u) ? 1 : 0);
        const bool wrench_only = __ballot(lane < N && (nc == 1 || nc == 2)) == 0ull;
        if (wrench_only != (a.tile_sel == 1)) return;
    }
    // restart pass: the workgroup of a QP that did not end at the cap leaves at once; no loop over QPs here -- any loop around the body makes hipcc
    // hoist the body's lane-index expressions out of it and spill them (750 bytes of scratch per lane at N = 20)
    if ((!a.count_ptr || (int)blockIdx.x < *a.count_ptr) && !SRBDQP_RESTART_SKIP(a, SRBDQP_QP_INDEX(a)))
        wrench_qp<N, R, TIO, MODE, TT, SPW, XW>(a, SRBDQP_QP_INDEX(a), sm);
    signal_done(a);   // staged path: every workgroup of the launch reports once, with or without work (done_cs is for the *_in kernels only)
}

// ... the low-latency instantiation with the QP's inputs in the kernel-argument segment (StagedIn, srbdqp_common.hpp): one staged QP, first pass
template <int N, int XW>
__global__ __launch_bounds__((WrenchSmem<N, 8, 5, XW>::BT), 1) void srbdqp_wrench_kernel_in(KArgs a, StagedIn<N> in) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    (void)in;                                            // (read through staged_in_base(): a.inline_in is set)
    wrench_qp<N, double, double, 0, double, 5, XW>(a, 0, sm);
    if (!(XW > 0 && a.done_cs)) signal_done(a);
}

}  // namespace srbdqp

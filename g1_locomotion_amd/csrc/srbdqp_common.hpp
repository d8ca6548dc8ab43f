// srbdqp_common.hpp -- device-side building blocks shared by every kernel variant of the SRBD QP hot path.
//
// One workgroup (256 threads = 4 wave64) solves one QP; everything between the input load and the output
// store lives in LDS / registers.  The phases mirror SURVEY.md section 8(a):
//   a5 linearise  -> load_and_linearise()      a6 condense -> build_G()  (G = Q^1/2 * s * B_qp, packed block-lower)
//   a7 gradient   -> build_gradient()          a8 bounds   -> row_bounds()
//   a9 ADMM       -> admm_loop()               a10 rollout -> rollout_and_store()
// The reference implementation of these steps is the absent submodule g1_mpc (see oracle/srbd_oracle.py header);
// the conventions come from its call sites g1_mujoco_sim/src/run_simulation.py:73-111.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srbdqp {

constexpr int kThreads = 256;
constexpr double kInf = 1.0e30;

// Kernel arguments (passed by value).
struct KArgs {
    const double* x0;        // [B][13]
    const double* xref;      // [B][N][13]
    const double* foot;      // [B][N][12]
    const double* pcom;      // [B][N][3] or null
    const double* warm_u;    // [B][12N] or null (newtons)
    const double* warm_y;    // [B][20N] or null
    const uint8_t* contact;  // [B][N][4]
    double* u_out;           // [B][N][12]
    double* x_out;           // [B][N+1][13] or null
    double* y_out;           // [B][20N] or null
    int32_t* status;         // [B] or null
    int32_t* iters;          // [B] or null
    double* P_out;           // assemble mode: [B][n][n]
    double* q_out;           // [B][n]
    double* l_out;           // [B][m]
    double* ub_out;          // [B][m]
    const int32_t* perm;     // optional dispatch order: workgroup i solves QP perm[i] (longest-first scheduling), or null
    long long* stamps;       // diagnostic: [B][16] s_memtime stamps of the phase boundaries, or null
    double* ws;              // split mode: per-QP workspace between the set-up kernel and the ADMM kernel, or null
    // rho re-balancing (one OSQP-style restart of the QPs that reach the cap of the first pass, see srbdqp.hip)
    const double* rho_qp;    // second pass: rho of QP b, or null (= rho)
    float* resid_out;        // first pass: [B][4] fp32 maxima (r_prim, n_prim, r_dual, n_dual) of the last check of a QP
                             //   that ends at the cap, or null
    const int32_t* count_ptr;   // second pass: number of valid entries of perm[]; workgroups beyond it exit at once
    int32_t iters_base;      // second pass: iterations of the first pass, added to iters[] on output
    int32_t qp_span;         // host only: number of QP slots the per-QP workspaces must hold (second pass: original B)
    int32_t* done_flag;      // low-latency completion: GPU-mapped host word that receives done_value once every QP of the
    int32_t* done_count;     //   launch has stored its outputs (done_count: device counter of finished workgroups), or null
    int32_t done_value;
    int32_t B;
    int32_t mode;            // 0 = solve, 1 = assemble only
    int32_t max_iter, check_every;
    double dt, inv_mass, iinv[3], mu, fzmin_s, fzmax_s, s;
    double sqrtq[12];        // sqrt(q_diag[0..11])
    double rs2;              // r_diag * s^2
    double rho, rho_eq, sigma, alpha, eps_abs, eps_rel;
};

// QP index of this workgroup
#define SRBDQP_QP_INDEX(a) ((a).perm ? (a).perm[blockIdx.x] : (int)blockIdx.x)

// diagnostic phase stamp (thread 0 only; leaves the kernel through a buffer nothing else reads)
#define SRBDQP_STAMP(a, b, idx) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)(b) * 16 + (idx)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

// Low-latency completion signal (srbdqp_solve_staged_f64): after a workgroup's outputs are stored, make them visible
// to the host and let the last workgroup of the launch publish done_value in host memory; the host spins on that word
// instead of waiting for the stream's completion interrupt.
__device__ __forceinline__ void signal_done(const KArgs& a) {
    if (!a.done_flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (a.B > 1) {
            last = (atomicAdd(a.done_count, 1) == a.B - 1);
            if (last) *a.done_count = 0;
        }
        // relaxed: every thread's outputs were made visible by its own system-scope fence before the barrier
        if (last) __hip_atomic_store(a.done_flag, a.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int N>
struct Dims {
    static constexpr int n = 12 * N;             // decision variables
    static constexpr int m = 20 * N;             // constraint rows
    static constexpr int gtot = 72 * N * (N + 1);  // packed G entries: sum_i 12 rows * 12(i+1) cols
    static constexpr int ME = (m + kThreads - 1) / kThreads;   // constraint rows per thread
    static constexpr int VE = (n + kThreads - 1) / kThreads;   // variables per thread
};

// offset of packed G row k = 12 i + kk (length 12 (i+1))
__device__ __forceinline__ int g_row_off(int i, int kk) { return 72 * i * (i + 1) + 12 * kk * (i + 1); }

// LDS carve shared by all variants (doubles).  Everything 16-byte aligned.
template <int N>
struct Smem {
    static constexpr int n = Dims<N>::n, m = Dims<N>::m;
    static constexpr int up2(int v) { return (v + 1) & ~1; }
    static constexpr int o_x0 = 0;                          // 13 (+1)
    static constexpr int o_xref = o_x0 + 14;                // N*13
    static constexpr int o_foot = o_xref + up2(N * 13);     // N*12
    static constexpr int o_pcom = o_foot + N * 12;          // N*3
    static constexpr int o_tm = o_pcom + up2(N * 3);        // N*9   Rz' per step
    static constexpr int o_cp = o_tm + up2(N * 9);          // N*9   prefix sums of Rz'
    static constexpr int o_J = o_cp + up2(N * 9);           // N*36  Iw^-1 [r]x per step
    static constexpr int o_eh = o_J + N * 36;               // n     Q^1/2 (A_qp x0 - x_ref)
    static constexpr int o_q = o_eh + n;                    // n     gradient
    static constexpr int o_rhs = o_q + n;                   // n (+8 pad)
    static constexpr int o_xt = o_rhs + n + 8;              // n
    static constexpr int o_w = o_xt + n;                    // m     rho z - y   /  y on check iterations
    static constexpr int o_nu = o_w + m;                    // m     rho (zt - z) + y
    static constexpr int o_xs = o_nu + m;                   // n     x on check iterations / omega,v in rollout
    static constexpr int o_ys = o_xs + n;                   // m     y on check iterations
    static constexpr int o_red = o_ys + m;                  // 8*8   block reductions
    static constexpr int o_ct = o_red + 64;                 // N*4 bytes of contact flags (as doubles: N/2)
    static constexpr int o_misc = o_ct + up2((N * 4 + 7) / 8);
    static constexpr int o_sq = o_misc + 8;                 // 12    sqrt(q_diag) (lane-indexed reads must not hit the kernarg segment)
    static constexpr int o_end = o_sq + 12;
};

// ---------------------------------------------------------------------------------------------------------
// a5: load inputs (coalesced), linearise every horizon step
// ---------------------------------------------------------------------------------------------------------
template <int N, class L>
__device__ void load_and_linearise(const KArgs& a, int b, double* sm) {
    using S = L;
    const int t = threadIdx.x;
    const double* gx0 = a.x0 + (size_t)b * 13;
    const double* gxr = a.xref + (size_t)b * N * 13;
    const double* gft = a.foot + (size_t)b * N * 12;
    const uint8_t* gct = a.contact + (size_t)b * N * 4;
    uint8_t* sct = reinterpret_cast<uint8_t*>(sm + S::o_ct);
    // Every global load is issued before the first LDS store, branch-free (clamped indices), so that they travel as
    // ONE batch: on the staged batch-1 path the inputs sit in GPU-mapped host memory and each dependent round is a PCIe
    // round trip of 1.2 us (tools/pcie_probe.hip) -- five conditional load -> store blocks cost five of them.
    constexpr int RX = (N * 13 + kThreads - 1) / kThreads, RF = (N * 12 + kThreads - 1) / kThreads;
    static_assert(N * 4 <= kThreads, "one thread per contact flag");
    const double* gpc = a.pcom ? a.pcom + (size_t)b * N * 3 : gx0;          // !pcom: a valid address, value unused
    const double v_x0 = gx0[t < 13 ? t : 0];
    const uint8_t v_ct = gct[t < N * 4 ? t : 0];
    const double v_pc = gpc[(a.pcom && t < N * 3) ? t : 0];
    double v_xr[RX], v_ft[RF];
#pragma unroll
    for (int r = 0; r < RX; ++r) { const int i = t + r * kThreads; v_xr[r] = gxr[i < N * 13 ? i : 0]; }
#pragma unroll
    for (int r = 0; r < RF; ++r) { const int i = t + r * kThreads; v_ft[r] = gft[i < N * 12 ? i : 0]; }
    if (t < 13) sm[S::o_x0 + t] = v_x0;
    if (t >= 32 && t < 44) sm[S::o_sq + t - 32] = a.sqrtq[t - 32];
#pragma unroll
    for (int r = 0; r < RX; ++r) { const int i = t + r * kThreads; if (i < N * 13) sm[S::o_xref + i] = v_xr[r]; }
#pragma unroll
    for (int r = 0; r < RF; ++r) { const int i = t + r * kThreads; if (i < N * 12) sm[S::o_foot + i] = v_ft[r]; }
    if (t < N * 4) sct[t] = v_ct ? 1 : 0;
    if (a.pcom && t < N * 3) sm[S::o_pcom + t] = v_pc;
    __syncthreads();
    if (!a.pcom && t < N * 3) sm[S::o_pcom + t] = sm[S::o_xref + (t / 3) * 13 + 3 + (t % 3)];
    if (t < N) {   // Rz(yaw_k)'
        double sn, cs;
        sincos(sm[S::o_xref + t * 13 + 2], &sn, &cs);
        double* T = sm + S::o_tm + t * 9;
        T[0] = cs;  T[1] = sn;  T[2] = 0.0;
        T[3] = -sn; T[4] = cs;  T[5] = 0.0;
        T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
    }
    __syncthreads();
    if (t < 9) {   // prefix sums C_k = sum_{l<=k} T_l
        double acc = 0.0;
        for (int k = 0; k < N; ++k) { acc += sm[S::o_tm + k * 9 + t]; sm[S::o_cp + k * 9 + t] = acc; }
    }
    if (t < N * 12) {   // J_k[:, 3 ci + ax] = Iw^-1 * skew(r)[:, ax]
        const int k = t / 12, cc = t % 12, ci = cc / 3, ax = cc % 3;
        const double cs = sm[S::o_tm + k * 9 + 0], sn = sm[S::o_tm + k * 9 + 1];
        const double i0 = a.iinv[0], i1 = a.iinv[1], i2 = a.iinv[2];
        const double w00 = cs * cs * i0 + sn * sn * i1, w01 = cs * sn * (i0 - i1), w11 = sn * sn * i0 + cs * cs * i1;
        const double rx = sm[S::o_foot + k * 12 + 3 * ci + 0] - sm[S::o_pcom + k * 3 + 0];
        const double ry = sm[S::o_foot + k * 12 + 3 * ci + 1] - sm[S::o_pcom + k * 3 + 1];
        const double rz = sm[S::o_foot + k * 12 + 3 * ci + 2] - sm[S::o_pcom + k * 3 + 2];
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

// one entry of the (unweighted) condensed input matrix: row kk of block (i, j), column cc; j <= i
template <int N, class L>
__device__ __forceinline__ double bqp_entry(const KArgs& a, const double* sm, int i, int kk, int j, int cc) {
    using S = L;
    const int ax = cc % 3;
    const double* J = sm + S::o_J + j * 36;
    if (kk < 3) {
        const double* Ci = sm + S::o_cp + i * 9 + kk * 3;
        const double* Cj = sm + S::o_cp + j * 9 + kk * 3;
        const double v = (Ci[0] - Cj[0]) * J[cc] + (Ci[1] - Cj[1]) * J[12 + cc] + (Ci[2] - Cj[2]) * J[24 + cc];
        return a.dt * a.dt * v;
    } else if (kk < 6) {
        return (kk - 3 == ax) ? (double)(i - j) * a.dt * a.dt * a.inv_mass : 0.0;
    } else if (kk < 9) {
        return a.dt * J[(kk - 6) * 12 + cc];
    } else {
        return (kk - 9 == ax) ? a.dt * a.inv_mass : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// a6: condensation.  G = Q^1/2 * s * B_qp without the (all-zero) gravity rows, packed block-lower-triangular:
// row k = 12 i + kk holds columns 0 .. 12 (i+1) - 1 at g_row_off(i, kk).  One entry per thread and step, block by
// block (block = 12 x 12 entries of B_qp(i, j), j <= i); every LDS load is unconditional and the four row classes
// (theta / p / omega / v) are selected arithmetically, so a wave never diverges.
// ---------------------------------------------------------------------------------------------------------
template <int N, class L>
__device__ void build_G(const KArgs& a, double* sm, double* G) {
    using S = L;
    constexpr int NBLK = N * (N + 1) / 2;
    const int t = threadIdx.x;
    // block table: blk -> (i, j), j <= i  (kept in the reduction scratch; done with it before anyone reduces)
    int* tab = reinterpret_cast<int*>(sm + S::o_red);
    if (t < NBLK) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        tab[t] = (i << 8) | (t - i * (i + 1) / 2);
    }
    __syncthreads();
    // thread = (block group g of 3, row-pair type, rr, cc): type 0 writes the theta row rr and the omega row rr of
    // a block, type 1 the p row rr and the v row rr; all index arithmetic is hoisted out of the block loop
    const int g = t / 72, u = t - 72 * g;
    const int type = u / 36, v = u - 36 * type;
    const int rr = v / 12, cc = v - 12 * rr, ax = cc % 3;
    const double dt = a.dt, dt2 = a.dt * a.dt, dtm = a.dt * a.inv_mass, dt2m = a.dt * a.dt * a.inv_mass;
    const double wA = sm[S::o_sq + (type ? 3 : 0) + rr] * a.s;      // weight of the first row of the pair
    const double wB = sm[S::o_sq + (type ? 9 : 6) + rr] * a.s;      // ... of the second
    const int rowA = (type ? 3 : 0) + rr, rowB = (type ? 9 : 6) + rr;
    if (g < 3) {
        for (int blk = g; blk < NBLK; blk += 3) {
            const int ij = tab[blk], i = ij >> 8, j = ij & 255;
            const int len = 12 * (i + 1);
            double* dst = G + 72 * i * (i + 1) + 12 * j + cc;
            double vA, vB;
            if (type == 0) {
                const double* J = sm + S::o_J + j * 36 + cc;
                const double* Ci = sm + S::o_cp + i * 9 + rr * 3;
                const double* Cj = sm + S::o_cp + j * 9 + rr * 3;
                const double j0 = J[0], j1 = J[12], j2 = J[24];
                vA = dt2 * ((Ci[0] - Cj[0]) * j0 + (Ci[1] - Cj[1]) * j1 + (Ci[2] - Cj[2]) * j2);
                vB = dt * ((rr == 0) ? j0 : (rr == 1) ? j1 : j2);
            } else {
                vA = (rr == ax) ? (double)(i - j) * dt2m : 0.0;
                vB = (rr == ax) ? dtm : 0.0;
            }
            dst[len * rowA] = wA * vA;
            dst[len * rowB] = wB * vB;
        }
    }
}

// free response (A_qp x0) entry for predicted state x_{i+1}, dynamic component kk (0..11)
template <int N, class L>
__device__ __forceinline__ double free_response(const KArgs& a, const double* sm, int i, int kk) {
    using S = L;
    const double* x0 = sm + S::o_x0;
    if (kk < 3) {
        const double* C = sm + S::o_cp + i * 9 + kk * 3;
        return x0[kk] + a.dt * (C[0] * x0[6] + C[1] * x0[7] + C[2] * x0[8]);
    } else if (kk < 6) {
        double v = x0[kk] + (double)(i + 1) * a.dt * x0[kk + 6];
        if (kk == 5) v += a.dt * a.dt * x0[12] * (double)((i * (i + 1)) / 2);
        return v;
    } else if (kk < 9) {
        return x0[kk];
    } else {
        double v = x0[kk];
        if (kk == 11) v += (double)(i + 1) * a.dt * x0[12];
        return v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// a7 (gradient half): q = G' * Q^1/2 (A_qp x0 - x_ref).  Two threads per column (even / odd rows of every block row).
// ---------------------------------------------------------------------------------------------------------
template <int N, class L>
__device__ void build_gradient(const KArgs& a, double* sm, const double* G) {
    using S = L;
    constexpr int n = Dims<N>::n;
    static_assert(2 * n <= kThreads, "two threads per column");
    const int t = threadIdx.x;
    for (int k = t; k < n; k += kThreads) {
        const int i = k / 12, kk = k - 12 * i;
        sm[S::o_eh + k] = sm[S::o_sq + kk] * (free_response<N, L>(a, sm, i, kk) - sm[S::o_xref + i * 13 + kk]);
    }
    __syncthreads();
    const int c = t >> 1, h = t & 1;
    double acc = 0.0;
    if (c < n) {
        for (int i = c / 12; i < N; ++i) {
            const int len = 12 * (i + 1);
            const double* col = G + 72 * i * (i + 1) + len * h + c;
            const double* e = sm + S::o_eh + 12 * i + h;
            double p0 = col[0] * e[0], p1 = col[2 * len] * e[2], p2 = col[4 * len] * e[4];
            p0 = fma(col[6 * len], e[6], p0);
            p1 = fma(col[8 * len], e[8], p1);
            p2 = fma(col[10 * len], e[10], p2);
            acc += (p0 + p1) + p2;
        }
    }
    {   // sum the two halves (lanes 2c, 2c+1)
        int lo = __double2loint(acc), hi = __double2hiint(acc);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
        acc += __hiloint2double(hi, lo);
    }
    if (c < n && h == 0) sm[S::o_q + c] = acc;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// a8: friction-cone / normal-force rows.  Row r = 20 k + 5 ci + j.
// ---------------------------------------------------------------------------------------------------------
struct RowInfo { int v0; int j; double lo, hi, rho; };

template <int N, class L>
__device__ __forceinline__ RowInfo row_info(const KArgs& a, const double* sm, int r) {
    using S = L;
    const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
    RowInfo ri;
    const int k = r / 20, rr = r - 20 * k, ci = rr / 5;
    ri.j = rr - 5 * ci;
    ri.v0 = 12 * k + 3 * ci;
    const bool on = sct[k * 4 + ci] != 0;
    if (ri.j < 4) { ri.lo = -kInf; ri.hi = 0.0; ri.rho = a.rho; }
    else { ri.lo = on ? a.fzmin_s : 0.0; ri.hi = on ? a.fzmax_s : 0.0; ri.rho = on ? a.rho : a.rho_eq; }
    return ri;
}

// (A v)_r from a vector v in LDS
__device__ __forceinline__ double apply_A_row(const double* v, const RowInfo& ri, double mu) {
    const double fz = v[ri.v0 + 2];
    switch (ri.j) {
        case 0: return v[ri.v0] - mu * fz;
        case 1: return -v[ri.v0] - mu * fz;
        case 2: return v[ri.v0 + 1] - mu * fz;
        case 3: return -v[ri.v0 + 1] - mu * fz;
        default: return fz;
    }
}

// (A' w)_c from a row vector w in LDS
__device__ __forceinline__ double apply_At_col(const double* w, int c, double mu) {
    const int k = c / 12, cc = c - 12 * k, ci = cc / 3, ax = cc - 3 * ci;
    const double* p = w + 20 * k + 5 * ci;
    if (ax == 0) return p[0] - p[1];
    if (ax == 1) return p[2] - p[3];
    return -mu * (p[0] + p[1] + p[2] + p[3]) + p[4];
}

// diagonal of A' diag(rho) A for variable c
template <int N, class L>
__device__ __forceinline__ double rho_diag(const KArgs& a, const double* sm, int c) {
    using S = L;
    const uint8_t* sct = reinterpret_cast<const uint8_t*>(sm + S::o_ct);
    const int k = c / 12, cc = c - 12 * k, ci = cc / 3, ax = cc - 3 * ci;
    if (ax < 2) return 2.0 * a.rho;
    const double r5 = sct[k * 4 + ci] ? a.rho : a.rho_eq;
    return 4.0 * a.mu * a.mu * a.rho + r5;
}

// block-wide max of up to NV values per thread; result broadcast to all threads.  2 barriers.
template <int NV>
__device__ __forceinline__ void block_max(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double x = v[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x = fmax(x, __shfl_xor(x, off));
        if (lane == 0) red[wave * 8 + q] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = fmax(fmax(red[q], red[8 + q]), fmax(red[16 + q], red[24 + q]));
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// a10: roll the linear model forward with the optimal forces and store u (newtons), x horizon, status.
// xs (LDS, n doubles) holds the scaled solution u_hat.
// ---------------------------------------------------------------------------------------------------------
template <int N, class L, int BT = kThreads>
__device__ void rollout_and_store(const KArgs& a, int b, double* sm, const double* uh, double* scratch /* >= 12N */) {
    using S = L;
    constexpr int n = Dims<N>::n;
    const int t = threadIdx.x;
    for (int c = t; c < n; c += BT) a.u_out[(size_t)b * n + c] = a.s * uh[c];
    if (!a.x_out) return;
    const double* x0 = sm + S::o_x0;
    // phase A1: per-step angular / linear acceleration sums  s_j = J_j u_j (3), sum_contacts u_j / m (3)  -- one step
    // per thread (60 threads x 12 products; a thread per (k, comp) summing all steps j < k itself took 10 x as long, 4 us
    // of a batch-1 solve)
    double* sj = scratch + 6 * N;
    for (int idx = t; idx < 6 * N; idx += BT) {
        const int j = idx / 6, comp = idx % 6;
        const double* u = uh + 12 * j;
        double s;
        if (comp < 3) {
            const double* J = sm + S::o_J + j * 36 + comp * 12;
            s = 0.0;
            for (int c = 0; c < 12; ++c) s += J[c] * u[c];
        } else {
            const int ax = comp - 3;
            s = (u[ax] + u[3 + ax] + u[6 + ax] + u[9 + ax]) * a.inv_mass;
        }
        sj[idx] = s;
    }
    __syncthreads();
    // phase A2: omega_k, v_k for k = 1..N  (scratch[(k-1)*6 + comp]); same summation order as before
    for (int idx = t; idx < 6 * N; idx += BT) {
        const int k = idx / 6 + 1, comp = idx % 6;
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc += sj[6 * j + comp];
        double v = x0[6 + comp] + a.dt * a.s * acc;
        if (comp == 5) v += (double)k * a.dt * x0[12];
        scratch[idx] = v;
    }
    __syncthreads();
    double* xo = a.x_out + (size_t)b * (N + 1) * 13;
    for (int idx = t; idx < 13 * (N + 1); idx += BT) {
        const int k = idx / 13, comp = idx % 13;
        double v;
        if (k == 0) v = x0[comp];
        else if (comp == 12) v = x0[12];
        else if (comp >= 6) v = scratch[(k - 1) * 6 + comp - 6];
        else if (comp >= 3) {   // p_k = p_0 + dt sum_{l<k} v_l
            double acc = x0[6 + comp];
            for (int l = 1; l < k; ++l) acc += scratch[(l - 1) * 6 + comp];
            v = x0[comp] + a.dt * acc;
        } else {                // theta_k = theta_0 + dt sum_{l<k} T_l omega_l
            double acc = 0.0;
            for (int l = 0; l < k; ++l) {
                const double* T = sm + S::o_tm + l * 9 + comp * 3;
                const double* w = (l == 0) ? (x0 + 6) : (scratch + (l - 1) * 6);
                acc += T[0] * w[0] + T[1] * w[1] + T[2] * w[2];
            }
            v = x0[comp] + a.dt * acc;
        }
        xo[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// a9: ADMM (OSQP Algorithm 1, reduced KKT form) with the explicit inverse K^-1 held in registers:
// thread (r = t>>1, h = t&1) owns K^-1[r][CH*h .. CH*h+CH-1] in kin[].  Mirrors oracle admm_solve() operation
// for operation.  Preconditions: xs = x^0 (scaled), xt = P x^0, q set, and a barrier has passed.
// On return xs holds the scaled solution, ys the dual; returns the iteration count, *status_out the status.
// ---------------------------------------------------------------------------------------------------------
template <int N, class L, int CH>
__device__ int admm_loop(const KArgs& a, int b, double* sm, const double (&kin)[CH], int* status_out) {
    using S = L;
    constexpr int n = Dims<N>::n, m = Dims<N>::m, ME = Dims<N>::ME, VE = Dims<N>::VE;
    const int t = threadIdx.x;
    const int r = t >> 1, h = t & 1;
    double* rhs = sm + S::o_rhs;
    double* xt = sm + S::o_xt;
    double* w = sm + S::o_w;
    double* nu = sm + S::o_nu;
    double* xs = sm + S::o_xs;
    double* ys = sm + S::o_ys;
    double* red = sm + S::o_red;
    const double sigma = a.sigma, alpha = a.alpha, mu = a.mu;

    double x[VE], px[VE], qv[VE];
    double z[ME], y[ME];
    RowInfo ri[ME];
#pragma unroll
    for (int e = 0; e < ME; ++e) {
        const int i = t + e * kThreads;
        ri[e] = row_info<N, L>(a, sm, i < m ? i : 0);
    }
    double qn[1] = {0.0};
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int c = t + e * kThreads;
        x[e] = 0.0; px[e] = 0.0; qv[e] = 0.0;
        if (c < n) { qv[e] = sm[S::o_q + c]; x[e] = xs[c]; px[e] = xt[c]; }
        qn[0] = fmax(qn[0], fabs(qv[e]));
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) {
        const int i = t + e * kThreads;
        z[e] = 0.0; y[e] = 0.0;
        if (i < m) {
            if (a.warm_y) y[e] = a.warm_y[(size_t)b * m + i];
            const double ax = apply_A_row(xs, ri[e], mu);
            z[e] = fmin(fmax(ax, ri[e].lo), ri[e].hi);
            w[i] = ri[e].rho * z[e] - y[e];
        }
    }
    block_max<1>(qn, red);   // also orders the w[] writes before the reads below
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int c = t + e * kThreads;
        if (c < n) rhs[c] = sigma * x[e] - qv[e] + apply_At_col(w, c, mu);
    }
    __syncthreads();

    int status = 2, iters = a.max_iter;   // SRBDQP_MAX_ITER
    for (int k = 1; k <= a.max_iter; ++k) {
        const bool check = (k % a.check_every == 0) || (k == a.max_iter);
        // ---- x~ = K^-1 rhs
        {
            double acc0 = 0.0, acc1 = 0.0;
            const double2* rv = reinterpret_cast<const double2*>(rhs + CH * h);
#pragma unroll
            for (int cc = 0; cc < CH; cc += 2) {
                const double2 v = rv[cc >> 1];
                acc0 = fma(kin[cc], v.x, acc0);
                acc1 = fma(kin[cc + 1], v.y, acc1);
            }
            double acc = acc0 + acc1;
            acc += __shfl_xor(acc, 1);
            if (h == 0 && r < n) xt[r] = acc;
        }
        __syncthreads();
        // ---- constraint rows: z~, nu, relaxation, projection, dual update
#pragma unroll
        for (int e = 0; e < ME; ++e) {
            const int i = t + e * kThreads;
            if (i < m) {
                const double rho = ri[e].rho;
                const double zt = apply_A_row(xt, ri[e], mu);
                nu[i] = rho * (zt - z[e]) + y[e];
                const double zh = alpha * zt + (1.0 - alpha) * z[e];
                const double zn = fmin(fmax(zh + y[e] / rho, ri[e].lo), ri[e].hi);
                y[e] = y[e] + rho * (zh - zn);
                z[e] = zn;
                w[i] = rho * zn - y[e];
                if (check) ys[i] = y[e];
            }
        }
        __syncthreads();
        // ---- variables: P x~ from the KKT identity, relaxation, next right-hand side
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int c = t + e * kThreads;
            if (c < n) {
                const double xtc = xt[c];
                const double pxt = sigma * (x[e] - xtc) - qv[e] - apply_At_col(nu, c, mu);
                x[e] = alpha * xtc + (1.0 - alpha) * x[e];
                px[e] = alpha * pxt + (1.0 - alpha) * px[e];
                rhs[c] = sigma * x[e] - qv[e] + apply_At_col(w, c, mu);
                if (check) xs[c] = x[e];
            }
        }
        __syncthreads();
        if (check) {
            double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // r_prim, |Ax|, |z|  /  r_dual, max(|Px|,|A'y|)
#pragma unroll
            for (int e = 0; e < ME; ++e) {
                const int i = t + e * kThreads;
                if (i < m) {
                    const double ax = apply_A_row(xs, ri[e], mu);
                    v[0] = fmax(v[0], fabs(ax - z[e]));
                    v[1] = fmax(v[1], fmax(fabs(ax), fabs(z[e])));
                }
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const int c = t + e * kThreads;
                if (c < n) {
                    const double aty = apply_At_col(ys, c, mu);
                    const double rd = fabs(px[e] + qv[e] + aty);
                    // propagate NaN/Inf through the max (fmax would drop a NaN)
                    v[2] = (rd != rd) ? rd : fmax(v[2], rd);
                    v[3] = fmax(v[3], fmax(fabs(px[e]), fabs(aty)));
                }
            }
            v[4] = (v[2] != v[2] || v[0] != v[0]) ? 1.0 : 0.0;
            block_max<5>(v, red);
            const double e_prim = a.eps_abs + a.eps_rel * v[1];
            const double e_dual = a.eps_abs + a.eps_rel * fmax(v[3], qn[0]);
            const bool bad = (v[4] != 0.0) || !(fabs(v[0]) <= kInf) || !(fabs(v[2]) <= kInf);
            if (bad) { status = -1; iters = k; break; }
            if (v[0] <= e_prim && v[2] <= e_dual) { status = 1; iters = k; break; }
        }
    }
    // duals for warm starts (ys is current whenever we leave through a check iteration, which is always)
    if (a.y_out) {
#pragma unroll
        for (int e = 0; e < ME; ++e) { const int i = t + e * kThreads; if (i < m) a.y_out[(size_t)b * m + i] = y[e]; }
    }
    *status_out = status;
    return iters;
}

}  // namespace srbdqp

// srbdqp_common.hpp -- device-side building blocks shared by every kernel variant of the SRBD QP hot path.
//
// Everything between the input load and the output store lives in LDS / registers.  The phases mirror SURVEY.md
// section 8(a): a5 linearise -> load_and_linearise(); a6 + a7 condensation, Hessian, gradient in closed form and a8 the
// friction-cone rows -> the kernels themselves (srbdqp_compact.hpp, srbdqp_setup1.hpp, srbdqp_wrench.hpp); a9 ADMM ->
// their loops; a10 rollout -> rollout_and_store().
// The reference implementation of these steps is the absent submodule g1_mpc (see oracle/srbd_oracle.py header);
// the conventions come from its call sites g1_mujoco_sim/src/run_simulation.py:73-111.
#pragma once
#include <cstddef>
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
    int defer_x0;            // split pipeline, second kernel: the set-up ran ahead of the state estimate (two-phase call) -- patch the
                             //   gradient with dq/dx0 (x0 - x0 of the set-up) and use this x0 for the roll-out
    int tile_sel;            // general kernel, fp32 calls: 0 = every QP; 1 = only QPs whose steps all have 0 or >= 3 stance
                             //   contacts (the fp32-tile launch); 2 = only the others (the fp64-tile launch)
    const int32_t* row_off;  // ragged batches (general kernel): first horizon row of QP b in the packed [sum N][.] arrays, or
                             //   null (= b N); x_out of QP b then starts at row row_off[b] + b
    long long* stamps;       // diagnostic: [B][16] s_memtime stamps of the phase boundaries, or null
    double* ws;              // split mode: per-QP workspace between the set-up kernel and the ADMM kernel, or null
    // rho re-balancing (one OSQP-style restart of the QPs that reach the cap of the first pass, see srbdqp.hip)
    const double* rho_qp;    // rho of QP b, or null (= rho); in a restart pass: the rho the pass BEFORE it ran QP b with
    double* rho_out;         // restart pass that another one may follow: [B] receives the rho this pass runs QP b with (never the array rho_qp points to), or null
    const float* resid_in;   // second pass: the first pass's resid_out.  Workgroup of QP b leaves at once unless status[b] is
                             //   SRBDQP_MAX_ITER, and re-balances its rho from resid_in[b] (restart_rho_of) -- no list, no
                             //   selection kernel between the passes
    float* resid_out;        // first pass: [B][4] fp32 maxima (r_prim, n_prim, r_dual, n_dual) of the last check of a QP
                             //   that ends at the cap, or null
    const int32_t* count_ptr;   // second pass: number of valid entries of perm[]; workgroups beyond it exit at once
    int32_t* cap_list;       // a pass that another one may follow: QP b appends itself here when it ends at its cap (cap_list[atomicAdd(cap_count, 1)] = b) -- the
    int32_t* cap_count;      //   next pass takes the list as its perm[] / count_ptr, so that its working workgroups come first in the grid; or null
    int32_t iters_base;      // second pass: iterations of the first pass, added to iters[] on output
    int32_t restart_every;   // one-wave kernel, restart in place (srbdqp_setup1.hpp RST): re-balance rho every this many iterations ...
    int32_t restart_max;     //   ... at most this many times
    int32_t inline_in;       // the *_in kernels: the inputs of the one staged QP follow KArgs in the kernel-argument segment (StagedIn, below)
    int32_t y_capped_only;   // first pass, y_out = the engine's own buffer (the caller asked for no duals): only a QP that ends at the cap stores them
                             //   (20 N values per QP for every QP was a quarter of the HBM traffic of a configs[2] solve)
    int32_t qp_span;         // host only: number of QP slots the per-QP workspaces must hold (second pass: original B)
    double* u_dev;           // staged first pass that a restart pass may follow: a second copy of u [B][N][12] in DEVICE memory -- the pass behind it warm-starts from
                             //   there instead of reading the staging array back over PCIe (2.4 us of the two-launch calls that make the p99), or null
    // deferred tails (SRBDQP_FLAG_DEFER_TAIL, srbdqp_wave_defer_kernel): a QP that reaches a restart mark unconverged is not continued by its own
    // workgroup but appended -- (x, y), its re-balanced rho, its own input / output pointers -- to a list in HBM, and one of the first tail_wgs workgroups
    // of the NEXT launch on the same stream runs its next pass.  Three lists per launch stream in rotation: this launch reads tail_cnt[tail_iin] records
    // from list tail_iin, appends to list tail_iout and zeroes the count of list tail_izero (the one the next launch appends to).
    char* tail_lists;        // [3][tail_cap] records of kTailRecDoubles doubles, or null (no deferral)
    int32_t* tail_cnt;       // [3] record counts
    int32_t tail_cap, tail_wgs, tail_iin, tail_iout, tail_izero;
    int32_t done_cs;         // the *_in kernels: publish the completion word WITH the checksum of the outputs and without a system-scope fence (signal_done_checksum)
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
    double rho_fz;           // penalty of a stance contact's normal-force row relative to rho (srbdqp_config.rho_fz_scale, resolved)
};

// The inputs of ONE staged QP passed by value behind KArgs (the batch-1 kernels): the kernel-argument segment is fast memory for the GPU -- a first-touch read
// costs ~430 cycles there against ~2900 from the GPU-mapped pinned staging arrays (tools/kernarg_probe.hip) --, and the runtime copies the 2.4 KB at launch for
// less than the round trip it saves.  The input stage of the kernel reads from its own argument segment then (staged_in_base()).
template <int N>
struct StagedIn {
    double x0[13];
    double xref[13 * N];
    double foot[12 * N];
    double pcom[3 * N];
    uint8_t contact[4 * N];
    uint8_t pad[(8 - (4 * N) % 8) % 8];
};
// base of the StagedIn behind KArgs in this kernel's argument segment, or null (KArgs::inline_in: set by the host for the *_in kernels only).  No copy of KArgs
// is made for this: a modified copy passed on by reference lands on the stack (496 bytes of scratch per lane, every field access a scratch load: +8 us per call).
template <int N>
__device__ __forceinline__ const char* staged_in_base(const KArgs& a) {
    static_assert(sizeof(KArgs) % 8 == 0, "StagedIn follows KArgs without padding");
    if constexpr (sizeof(KArgs) + sizeof(StagedIn<N>) > 4096) return nullptr;        // (the long horizons have no *_in kernel: the segment holds 4 KB)
    else return a.inline_in ? (const char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(KArgs) : nullptr;
}

// The per-QP pointers of a solve.  The kernels of a launch read them from their KArgs; a deferred continuation (TailRec) carries those of the launch its QP
// came from, which may have been another batch in other buffers.
struct QpIo {
    const double *x0, *xref, *foot, *pcom;
    const uint8_t* contact;
    double *u_out, *x_out, *y_out;
    int32_t *status, *iters;
};
__device__ __forceinline__ QpIo io_of(const KArgs& a) { return QpIo{a.x0, a.xref, a.foot, a.pcom, a.contact, a.u_out, a.x_out, a.y_out, a.status, a.iters}; }
// One deferred continuation in HBM: header (16 doubles) + the parked x (newtons), y of slot A, y of slot B of the one-wave ADMM's 64 lanes
struct TailRecHead {
    int32_t b, pass, done, pad;      // QP index in ITS launch's arrays, passes completed, iterations completed
    double rho;                      // re-balanced rho of the pass to run
    QpIo io;
};
static_assert(sizeof(TailRecHead) <= 16 * sizeof(double), "the header of a tail record is 16 doubles");
constexpr int kTailRecDoubles = 16 + 3 * 64;
constexpr int kStatusPending = 0;    // status[] of a QP whose continuation is deferred (SRBDQP_PENDING)

// QP index of this workgroup
#define SRBDQP_QP_INDEX(a) ((a).perm ? (a).perm[blockIdx.x] : (int)blockIdx.x)

// rho of the second pass from the fp32 maxima of the first pass's last check (OSQP's rule, oracle restart_rho()):
// rho sqrt((r_prim / n_prim) / (r_dual / n_dual)), clipped to [rho / 10, 5 rho]
__device__ __forceinline__ double restart_rho_of(double rho0, const float* r) {
    const double num = (double)r[0] / fmax((double)r[1], 1e-30), den = (double)r[2] / fmax((double)r[3], 1e-30);
    double r1 = rho0;
    if (num > 0.0 && den > 0.0 && num <= 1.0e30 && den <= 1.0e30) r1 = fmin(fmax(rho0 * sqrt(num / den), rho0 * 0.1), rho0 * 5.0);
    return r1;
}
// is QP b this workgroup's to solve in a second pass (resid_in set)?  Uniform over the workgroup.
#define SRBDQP_RESTART_SKIP(a, b) ((a).resid_in && (a).status[(b)] != 2)
// rho of QP b in this launch: the caller's (rho_qp or rho); in a restart pass re-balanced from the maxima the pass before left, EACH TIME FROM THE RHO OF THAT PASS
// (rho_qp = what it recorded; the first pass's rho is the caller's).  A pass that may be followed by another records its rho in rho_out.
__device__ __forceinline__ double rho_of_qp(const KArgs& a, int b) {
    const double r0 = a.rho_qp ? a.rho_qp[b] : a.rho;
    const double r = a.resid_in ? restart_rho_of(r0, a.resid_in + (size_t)b * 4) : r0;
    if (a.rho_out && threadIdx.x == 0) a.rho_out[b] = r;
    return r;
}
#define SRBDQP_RHO_OF(a, b) rho_of_qp((a), (b))

// a wave-uniform value, moved to scalar registers (the float constants of the iteration are converted from doubles by the
// vector ALU and would otherwise each hold a vector register for the whole loop)
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uni(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ... and cut loose from the vector register it was computed in: the compiler otherwise keeps using that copy (a vector operand needs no
// constant-bus slot), i.e. holds a vector register -- or a scratch slot -- for a uniform constant across the whole kernel
__device__ __forceinline__ double unis(double v) {
    // (the builtin is folded away when the compiler can prove its operand uniform -- and the value then stays where the vector ALU left it.  s_nop: a lane read
    // right behind the fp64 instruction that wrote the register needs a wait state, and the hazard recogniser does not look into inline asm -- without it the
    // scalar got the register's PREVIOUS contents now and then: the quotient before its v_div_fixup, forces 1e-2 N off)
    int lo, hi;
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(v)), "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}

// ---- rank-6 form of the closed-form Hessian (round 4) ---------------------------------------------------------------------------------------------------------
// M(j, m) = dt^4 (T2(m) + (C_m - C_j)' W T1(m)) + (N - m) dt^2 W_w of the closed-form assembly (derivation: srbdqp_compact.hpp, phase A; until round 4 a table of one
// matrix per step pair, mt_tables) is D_m - C_j' E_m with  D_m = dt^4 (T2(m) + C_m' W T1(m)) + (N - m) dt^2 W_w,  E_m = dt^4 W T1(m)  -- one 3 x 3 pair per STEP instead of one
// matrix per step PAIR -- so the torque part of the entry of variables r (contact e1 of step j, axis x) and c (contact e2 of step m >= j, axis y),
//     [J_e1' M(j, m) J_e2]_xy = J_e1[:, x] . (D_m J_e2)[:, y] - (C_j J_e1)[:, x] . (E_m J_e2)[:, y],
// is the dot product of a 6-vector of r with a 6-vector of c, and the force part (x == y): fa_x (alpha_m + (m - j) beta_m) + fb_x (N - m) = f0_c - j f1_c.  One table
// row per presolved variable (kasm_rows: [a(6), g, -g j | s^2 b(6), f0, f1], g = 0 on padding rows), and every lane forms its own C-layout entries from the row of
// its r and the row of its c: 6 + 3 multiply-adds per entry, the rows of a lane's four columns read once, no staging tile, no scatter.  The contact-pair blocks
// through a staging tile (27 LDS reads and 54 multiply-adds per 3 x 3 block, tile by tile with 36 of 64 lanes busy, nine address selects and stores per block)
// were 11 % of the one-wave kernel's time on configs[1] (tools/fixed_iter_rate.py with the block computation stubbed out), the M table another 2 %.
// Valid for step(r) <= step(c): the upper triangle; the entries of a diagonal tile below the diagonal are never read (diag16_invert reads the upper triangle).
constexpr int kAbStride = 18;   // doubles per table row (16 + 2: rows 144 bytes apart, 16-byte aligned)
template <int N>
__device__ __forceinline__ void de_tables(const double* CP, const double* T1, const double* T2, const double* SQ, const double dt2, double* DE, const int t, const int nthreads) {
    for (int it = t; it < 9 * N; it += nthreads) {
        const int m = it / 9, k = it - 9 * m, p = k / 3, q = k - 3 * p;
        const double* Cm = CP + 9 * m;
        const double* t1 = T1 + 9 * m;
        const double d4 = dt2 * dt2;
        const double e0 = SQ[0] * SQ[0] * t1[q], e1 = SQ[1] * SQ[1] * t1[3 + q], e2 = SQ[2] * SQ[2] * t1[6 + q];
        double v = T2[9 * m + k] + Cm[p] * e0 + Cm[3 + p] * e1 + Cm[6 + p] * e2;
        v *= d4;
        v += (p == q) ? (double)(N - m) * dt2 * SQ[6 + p] * SQ[6 + p] : 0.0;
        DE[18 * m + k] = v;                                                  // D_m[p][q]
        DE[18 * m + 9 + k] = d4 * ((p == 0) ? e0 : (p == 1) ? e1 : e2);      // E_m[p][q]
    }
}
// Jb: the strip's J (36 doubles per step); act: the stance contacts in presolved order (step * 4 + contact); rows >= n_eff are zero
template <int N>
__device__ __forceinline__ void kasm_rows(const double* Jb, const double* CP, const double* DE, const double* SQ, const uint8_t* act, const int n_eff, const double s,
                                          const double dt2m, const double dtm, double* AB, const int t, const int nthreads, const int nrows) {
    const double s2 = s * s, dt4m2 = dt2m * dt2m, dt2m2 = dtm * dtm;
    for (int r = t; r < nrows; r += nthreads) {
        const bool on = r < n_eff;
        const int rr = on ? r : 0;
        const int e = rr / 3, x = rr - 3 * e, gc = act[e], j = gc >> 2;
        const double* Jp = Jb + j * 36 + 3 * (gc & 3) + x;
        const double J0 = Jp[0], J1 = Jp[12], J2 = Jp[24];
        const double* C = CP + 9 * j;
        const double* D = DE + 18 * j;
        double v[16];
        v[0] = J0; v[1] = J1; v[2] = J2;
#pragma unroll
        for (int l = 0; l < 3; ++l) v[3 + l] = -(C[3 * l] * J0 + C[3 * l + 1] * J1 + C[3 * l + 2] * J2);
        v[6] = 1.0; v[7] = -(double)j;
#pragma unroll
        for (int l = 0; l < 6; ++l) v[8 + l] = s2 * (D[3 * l] * J0 + D[3 * l + 1] * J1 + D[3 * l + 2] * J2);      // D_m J (3), E_m J (3)
        const int Ls = N - j;
        const double al = (double)(((Ls - 1) * Ls * (2 * Ls - 1)) / 6), be = (double)(((Ls - 1) * Ls) / 2);
        const double fa = SQ[3 + x] * SQ[3 + x] * dt4m2, fb = SQ[9 + x] * SQ[9 + x] * dt2m2;
        v[14] = s2 * (fa * (al + (double)j * be) + fb * (double)Ls);
        v[15] = s2 * fa * be;
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2* row = reinterpret_cast<d2*>(AB + kAbStride * r);
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = on ? (d2){v[2 * i], v[2 * i + 1]} : (d2){0.0, 0.0};
    }
}

// the four C-layout entries of upper tile (ta, tb) of lane (mcol, kq) from the table rows; ta, tb may be wave-uniform run-time values (the 4-wave kernel's tile slots)
__device__ __forceinline__ void kasm_tile(const double* AB, const int ta, const int tb, const int mcol, const int kq, const int n_eff, const double dgxy, const double dgz,
                                          double (&out)[4]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int c = 16 * tb + mcol;
    double Bv[8];
    {
        const d2* row = reinterpret_cast<const d2*>(AB + kAbStride * c + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const d2 v = row[i]; Bv[2 * i] = v[0]; Bv[2 * i + 1] = v[1]; }
    }
    const int cm3 = c - 3 * ((c * 0xAAAB) >> 17);                       // c mod 3 = the column's axis (c < 2^15)
    const double dcol = (c < n_eff) ? ((cm3 < 2) ? dgxy : dgz) : 1.0;   // (padding -> identity)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 16 * ta + kq + 4 * q;
        const d2* row = reinterpret_cast<const d2*>(AB + kAbStride * r);
        double Av[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const d2 v = row[i]; Av[2 * i] = v[0]; Av[2 * i + 1] = v[1]; }
        double v = Av[0] * Bv[0];
#pragma unroll
        for (int i = 1; i < 6; ++i) v = fma(Av[i], Bv[i], v);
        const double ft = fma(Av[7], Bv[7], Av[6] * Bv[6]);
        const int d = r + 3 * 1024 - c, dm3 = d - 3 * ((d * 0xAAAB) >> 17);
        v += (dm3 == 0) ? ft : 0.0;                                     // same axis
        v += (r == c) ? dcol : 0.0;
        out[q] = v;
    }
}

// Row kk of step i of G x (the warm start's P x^0, the rho-restart passes, the fp32-tile kernel's refinement of x_q), before the Q^1/2 s scaling, from the per-step
// torque / force sums TF: prefix sums over the steps j <= i.  Every loop runs over ALL N steps with the later ones adding exact zeros (same summation order): with
// a trip count per lane the loops were not unrolled and every trip waited for its own LDS reads.
template <int N>
__device__ __forceinline__ double gx_row(const double* CP, const double* TF, const int i, const int kk, const double dt, const double dt2, const double dtm, const double dt2m) {
    double acc = 0.0;
    if (kk < 3) {
        const double* Ci = CP + i * 9 + kk * 3;
        const double c0 = Ci[0], c1 = Ci[1], c2 = Ci[2];
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const double* Cj = CP + j * 9 + kk * 3;
            const double* tau = TF + 6 * j;
            const double term = (c0 - Cj[0]) * tau[0] + (c1 - Cj[1]) * tau[1] + (c2 - Cj[2]) * tau[2];
            acc += (j <= i) ? term : 0.0;
        }
        return acc * dt2;
    } else if (kk < 6) {
#pragma unroll 4
        for (int j = 0; j < N; ++j) { const double v = TF[6 * j + kk]; acc += (j <= i) ? (double)(i - j) * v : 0.0; }
        return acc * dt2m;
    } else {
#pragma unroll 4
        for (int j = 0; j < N; ++j) { const double v = TF[6 * j + kk - 6]; acc += (j <= i) ? v : 0.0; }
        return acc * ((kk < 9) ? dt : dtm);
    }
}

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

// ... for ONE staged QP (the *_in kernels, KArgs::done_cs): no fence and no barrier.  The completion word goes out right behind the stores and carries a checksum
// of them: every row of 16 lanes XORs the 64-bit patterns its lanes stored for the host (four DPP steps) and its first lane stores ONE 16-byte record
// {sequence number, number of records, XOR} at done_flag + 16 * row -- 4 to 16 records, at most the 256 bytes behind the staging arrays.  The host that sees the
// number in record 0 waits until every record carries it and the XOR of the records equals the XOR of what it reads back from the output arrays (srbdqp.hip
// wait_done): whatever order the writes arrive in, it never takes outputs that are not all there.  (The fence above is a round trip to the host's memory in front of
// the word, tools/pcie_probe.hip; a reduction across the workgroup would put a barrier and an LDS round trip there instead.)
// cs: this thread's share of the XOR.  BT: threads that call this (all of them alive).
__device__ __forceinline__ unsigned long long done_cs_pack(int status, int iters) { return (unsigned long long)(unsigned)status | ((unsigned long long)(unsigned)iters << 32); }
__device__ __forceinline__ unsigned dpp_xor_row(unsigned v) {
    v ^= (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1, 0, 3, 2]
    v ^= (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2, 3, 0, 1]
    v ^= (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v ^= (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror
    return v;                                                                               // every lane: the XOR over its row of 16
}
template <int BT>
__device__ __forceinline__ void signal_done_checksum(int32_t* done_flag, int32_t done_value, unsigned long long cs) {
    static_assert(BT % 16 == 0 && BT / 16 <= 16, "one 16-byte record per row of 16 lanes in 256 bytes");
    const unsigned lo = dpp_xor_row((unsigned)cs), hi = dpp_xor_row((unsigned)(cs >> 32));
    if ((threadIdx.x & 15) == 0) {
        uint4 v;
        v.x = (unsigned)done_value; v.y = (unsigned)(BT / 16); v.z = lo; v.w = hi;
        reinterpret_cast<uint4*>(done_flag)[threadIdx.x >> 4] = v;
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

// ---------------------------------------------------------------------------------------------------------
// a5: load inputs (coalesced), linearise every horizon step
// ---------------------------------------------------------------------------------------------------------
template <int N, class L>
__device__ void load_and_linearise(const KArgs& a, int b, double* sm) {
    using S = L;
    const int t = threadIdx.x;
    const char* kin_ = staged_in_base<N>(a);
    const double* gx0 = kin_ ? reinterpret_cast<const double*>(kin_ + offsetof(StagedIn<N>, x0)) : a.x0 + (size_t)b * 13;
    const double* gxr = kin_ ? reinterpret_cast<const double*>(kin_ + offsetof(StagedIn<N>, xref)) : a.xref + (size_t)b * N * 13;
    const double* gft = kin_ ? reinterpret_cast<const double*>(kin_ + offsetof(StagedIn<N>, foot)) : a.foot + (size_t)b * N * 12;
    const uint8_t* gct = kin_ ? reinterpret_cast<const uint8_t*>(kin_ + offsetof(StagedIn<N>, contact)) : a.contact + (size_t)b * N * 4;
    uint8_t* sct = reinterpret_cast<uint8_t*>(sm + S::o_ct);
    // Every global load is issued before the first LDS store, branch-free (clamped indices), so that they travel as
    // ONE batch: on the staged batch-1 path the inputs sit in GPU-mapped host memory and each dependent round is a PCIe
    // round trip of 1.2 us (tools/pcie_probe.hip) -- five conditional load -> store blocks cost five of them.
    constexpr int RX = (N * 13 + kThreads - 1) / kThreads, RF = (N * 12 + kThreads - 1) / kThreads;
    static_assert(N * 4 <= kThreads, "one thread per contact flag");
    const double* gpc = a.pcom ? (kin_ ? reinterpret_cast<const double*>(kin_ + offsetof(StagedIn<N>, pcom)) : a.pcom + (size_t)b * N * 3) : gx0;   // !pcom: a valid address, value unused
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
// CS: *cs ^= the 64-bit pattern of every value this thread stores to u_out / x_out (signal_done_checksum)
template <int N, class L, int BT = kThreads, bool CS = false>
__device__ void rollout_and_store_to(const KArgs& a, double* u_out, double* x_out, int b, double* sm, const double* uh, double* scratch /* >= 12N */, unsigned long long* cs = nullptr) {
    using S = L;
    constexpr int n = Dims<N>::n;
    const int t = threadIdx.x;
    for (int c = t; c < n; c += BT) {
        const double v = a.s * uh[c];
        u_out[(size_t)b * n + c] = v;
        if constexpr (CS) *cs ^= (unsigned long long)__double_as_longlong(v);
    }
    if (a.u_dev) for (int c = t; c < n; c += BT) a.u_dev[(size_t)b * n + c] = a.s * uh[c];
    if (!x_out) return;
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
        // (every loop of the roll-out runs over all N steps with the later ones adding exact zeros: with a trip count per lane it is not unrolled and every
        // trip waits for its own LDS reads -- 3 us of a batch-1 call went into these three loops; this way the reads of all trips are in flight together)
        // (long horizons keep the trip count per lane: k grows with the wave there, and the waves of the early steps are done after a few trips)
        double acc = 0.0;
        if constexpr (N <= 12) {
#pragma unroll
            for (int j = 0; j < N; ++j) { const double sv = sj[6 * j + comp]; acc += (j < k) ? sv : 0.0; }
        } else {
            for (int j = 0; j < k; ++j) acc += sj[6 * j + comp];
        }
        double v = x0[6 + comp] + a.dt * a.s * acc;
        if (comp == 5) v += (double)k * a.dt * x0[12];
        scratch[idx] = v;
    }
    __syncthreads();
    // phase A3 (round 5): Euler angles and CoM positions of the steps 1 .. N as one more prefix stage (6 N entries, a lane each, the reads of every trip in flight; the
    // two kinds of entries on different waves where there are several), then the rows of x as a plain gather.  Until round 5 the store loop formed them entry by
    // entry inside a five-way branch over (k, component): two or three trips of divergent arms with up to 60 dependent LDS reads each -- 4.1 k of the roll-out's 6.5 k
    // cycles in the batch-1 kernels.
    double* pa = sj;                                                     // (the per-step sums are dead behind phase A2)
    constexpr int PA_OFF = (64 * ((3 * N + 63) / 64) + 3 * N <= BT) ? 64 * ((3 * N + 63) / 64) : 3 * N;
    for (int tt = t; tt < PA_OFF + 3 * N; tt += BT) {
        if (tt < 3 * N || tt >= PA_OFF) {
            const bool posn = tt >= PA_OFF;
            const int e = posn ? tt - PA_OFF : tt;
            const int k = e / 3 + 1, comp = e % 3 + (posn ? 3 : 0);
            double acc;
            if (posn) {             // p_k = p_0 + dt sum_{l<k} v_l
                acc = x0[6 + comp];
                if constexpr (N <= 12) {
#pragma unroll
                    for (int l = 1; l < N; ++l) { const double sv = scratch[(l - 1) * 6 + comp]; acc += (l < k) ? sv : 0.0; }
                } else {
                    for (int l = 1; l < k; ++l) acc += scratch[(l - 1) * 6 + comp];
                }
            } else {                // theta_k = theta_0 + dt sum_{l<k} T_l omega_l
                acc = 0.0;
                auto term_of = [&](int l) {
                    const double* T = sm + S::o_tm + l * 9 + comp * 3;
                    const double* w = (l == 0) ? (x0 + 6) : (scratch + (l - 1) * 6);
                    return T[0] * w[0] + T[1] * w[1] + T[2] * w[2];
                };
                if constexpr (N <= 12) {
#pragma unroll
                    for (int l = 0; l < N; ++l) { const double term = term_of(l); acc += (l < k) ? term : 0.0; }
                } else {
                    for (int l = 0; l < k; ++l) acc += term_of(l);
                }
            }
            pa[6 * (k - 1) + comp] = x0[comp] + a.dt * acc;
        }
    }
    __syncthreads();
    double* xo = x_out + (size_t)b * (N + 1) * 13;
    for (int idx = t; idx < 13 * (N + 1); idx += BT) {
        const int k = idx / 13, comp = idx % 13;
        const double* src = (k == 0 || comp == 12) ? (x0 + comp) : ((comp >= 6) ? (scratch + (k - 1) * 6 + comp - 6) : (pa + (k - 1) * 6 + comp));
        const double v = *src;
        xo[idx] = v;
        if constexpr (CS) *cs ^= (unsigned long long)__double_as_longlong(v);
    }
}
template <int N, class L, int BT = kThreads>
__device__ __forceinline__ void rollout_and_store(const KArgs& a, int b, double* sm, const double* uh, double* scratch /* >= 12N */) {
    rollout_and_store_to<N, L, BT>(a, a.u_out, a.x_out, b, sm, uh, scratch);
}

}  // namespace srbdqp

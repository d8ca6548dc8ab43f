// srbdqp_gj.hpp -- kernel variant v0 ("gj"): VALU Hessian assembly + in-register symmetric sweep
// (Gauss-Jordan) inverse.  The simple, obviously-correct baseline every faster variant is A/B-ed against.
//
// Thread (r = t>>1, h = t&1) owns row r, columns CH*h .. CH*h+CH-1 of K = G'G + R s^2 + sigma I + A' rho A.
// The sweep operator keeps the matrix symmetric, so step p only needs row p broadcast through LDS
// (double-buffered, one barrier per step); after n sweeps the registers hold -K^-1 in exactly the layout the
// ADMM mat-vec wants.
#pragma once
#include "srbdqp_common.hpp"

namespace srbdqp {

template <int N>
struct GjSmem {
    using S = Smem<N>;
    static constexpr int n = Dims<N>::n;
    static constexpr int o_G = S::o_end;                        // packed G
    static constexpr int o_row = o_G + Dims<N>::gtot;           // 2 x (n + 8) sweep row buffers
    static constexpr int o_end = o_row + 2 * (n + 8);
    static constexpr size_t bytes = (size_t)o_end * sizeof(double);
};

template <int N>
__global__ __launch_bounds__(kThreads) void srbdqp_gj_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    using S = Smem<N>;
    using GS = GjSmem<N>;
    constexpr int n = Dims<N>::n, m = Dims<N>::m, CH = n / 2;
    static_assert(n <= 128, "gj variant: two threads per row of K, n <= 128");
    static_assert((S::o_rhs % 2) == 0 && (CH % 2) == 0, "16-byte alignment of the rhs halves");
    if ((int)blockIdx.x >= a.B) return;
    const int b = SRBDQP_QP_INDEX(a);
    const int t = threadIdx.x, r = t >> 1, h = t & 1;
    double* G = sm + GS::o_G;
    double* rowbuf = sm + GS::o_row;

    load_and_linearise<N, S>(a, b, sm);
    build_G<N, S>(a, sm, G);
    __syncthreads();
    build_gradient<N, S>(a, sm, G);

    // ---- a7: Hessian fragments  P[r][c] = sum_k G[k][r] G[k][c]  (+ R s^2 on the diagonal)
    double kf[CH];
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) kf[cc] = 0.0;
    if (r < n) {
        for (int k = 12 * (r / 12); k < n; ++k) {
            const int i = k / 12, kk = k - 12 * i, len = 12 * (i + 1);
            const double* row = G + g_row_off(i, kk);
            const double gr = row[r];
            // unconditional loads (a read past this row's end lands in a later row of G) + select: keeps the
            // 60 LDS reads batched instead of one exec-masked branch per element
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const int c = CH * h + cc;
                const double v = row[c];
                kf[cc] = fma(gr, (c < len) ? v : 0.0, kf[cc]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) if (CH * h + cc == r) kf[cc] += a.rs2;
    }

    if (a.mode == 1) {   // assemble only: dump P, q, l, u
        if (r < n) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) a.P_out[((size_t)b * n + r) * n + CH * h + cc] = kf[cc];
        }
        for (int c = t; c < n; c += kThreads) a.q_out[(size_t)b * n + c] = sm[S::o_q + c];
        for (int i = t; i < m; i += kThreads) {
            const RowInfo ri = row_info<N, S>(a, sm, i);
            a.l_out[(size_t)b * m + i] = ri.lo;
            a.ub_out[(size_t)b * m + i] = ri.hi;
        }
        return;
    }

    // ---- initial point: xs = x^0, xt = P x^0
    for (int c = t; c < n; c += kThreads) sm[S::o_xs + c] = a.warm_u ? a.warm_u[(size_t)b * n + c] / a.s : 0.0;
    __syncthreads();
    {
        double acc = 0.0;
        if (a.warm_u && r < n) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) acc = fma(kf[cc], sm[S::o_xs + CH * h + cc], acc);
        }
        acc += __shfl_xor(acc, 1);
        if (h == 0 && r < n) sm[S::o_xt + r] = acc;
    }

    // ---- K = P + sigma I + A' diag(rho) A, then K^-1 by n symmetric sweeps
    if (r < n) {
        const double dg = a.sigma + rho_diag<N, S>(a, sm, r);
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) if (CH * h + cc == r) kf[cc] += dg;
    }
    if (r == 0) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) rowbuf[CH * h + cc] = kf[cc];
    }
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        const double* rp = rowbuf + (p & 1) * (n + 8);
        double* rnext = rowbuf + ((p + 1) & 1) * (n + 8);
        const double inv = 1.0 / rp[p];
        if (r < n) {
            const double f = rp[r] * inv;
            if (r == p) {
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) { const int c = CH * h + cc; kf[cc] = (c == p) ? -inv : rp[c] * inv; }
            } else {
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) { const int c = CH * h + cc; kf[cc] = (c == p) ? f : fma(-f, rp[c], kf[cc]); }
            }
            if (r == p + 1) {
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) rnext[CH * h + cc] = kf[cc];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) kf[cc] = -kf[cc];

    int status;
    const int iters = admm_loop<N, S, CH>(a, b, sm, kf, &status);
    if (t == 0) {
        if (a.status) a.status[b] = status;
        if (a.iters) a.iters[b] = iters;
    }
    rollout_and_store<N, S>(a, b, sm, sm + S::o_xs, sm + S::o_rhs);
}

}  // namespace srbdqp

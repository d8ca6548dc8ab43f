// Steps either side of the QP in the reference cascade, batched over robots / feet.
//   swing-foot trajectory   g1_mujoco_sim/src/swing_trajectory.py:38-87  (call sites ros_run_simulation.py:246-256,300-312)
//   MPC -> WBID references  g1_mujoco_sim/src/wbid.py:232-297
// Both are element-wise: one thread per item, grid-stride, no LDS; they are bound by HBM traffic (104 B and 488 B
// per item).  Arithmetic is fp64 like the reference's NumPy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srbdqp {

struct SwingArgs {
    const double* p_start;   // [B][3]  foot position when the swing starts
    const double* p_final;   // [B][3]  landing position
    const double* z_middle;  // [B]     apex height (reached at progress 0.5)
    const double* progress;  // [B]     cycle progress in [0, 1]
    double* pos;             // [B][3]
    double* vel_z;           // [B] or null
    double* acc_z;           // [B] or null
    double* coeff;           // [B][7] or null: z(t) = sum_k coeff[k] t^k
    double final_velocity_z; // -0.02 in the reference (swing_trajectory.py:50)
    double first_half_share; // 0.80 in the reference (swing_trajectory.py:58)
    long long B;
};

// z(t): the sixth-order polynomial with z(0)=z_s, z'(0)=z''(0)=0, z(1/2)=z_m, z(1)=z_f, z'(1)=v_f, z''(1)=0
// (swing_trajectory.py:38-52).  The reference solves the 7x7 system numerically each time; its matrix is constant, so
// the solution is the constant inverse applied to [z_s, 0, 0, z_m, z_f, v_f, 0]: integer columns 0, 3, 4, 5 below.
__device__ __forceinline__ void swing_coefficients(double zs, double zm, double zf, double vf, double (&c)[7]) {
    c[0] = zs;
    c[1] = 0.0;
    c[2] = 0.0;
    c[3] = -42.0 * zs + 64.0 * zm - 22.0 * zf + 6.0 * vf;
    c[4] = 111.0 * zs - 192.0 * zm + 81.0 * zf - 23.0 * vf;
    c[5] = -102.0 * zs + 192.0 * zm - 90.0 * zf + 27.0 * vf;
    c[6] = 32.0 * zs - 64.0 * zm + 32.0 * zf - 10.0 * vf;
}

__global__ __launch_bounds__(256) void srbdqp_swing_kernel(SwingArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.B; i += stride) {
        const double xs = a.p_start[3 * i], ys = a.p_start[3 * i + 1], zs = a.p_start[3 * i + 2];
        const double xf = a.p_final[3 * i], yf = a.p_final[3 * i + 1], zf = a.p_final[3 * i + 2];
        const double zm = a.z_middle[i], t = a.progress[i];
        double c[7];
        swing_coefficients(zs, zm, zf, a.final_velocity_z, c);
        // x, y: a sine covers first_half_share of the distance in the first half of the cycle, the rest is linear
        // (swing_trajectory.py:54-67)
        const double share = a.first_half_share;
        const double phase = (t <= 0.5) ? share * sin(M_PI * t) : share + (t - 0.5) * (1.0 - share) * 2.0;
        a.pos[3 * i] = (1.0 - phase) * xs + phase * xf;
        a.pos[3 * i + 1] = (1.0 - phase) * ys + phase * yf;
        // z, z', z'' by Horner (swing_trajectory.py:76-89)
        a.pos[3 * i + 2] = c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * (c[4] + t * (c[5] + t * c[6])))));
        if (a.vel_z) a.vel_z[i] = c[1] + t * (2.0 * c[2] + t * (3.0 * c[3] + t * (4.0 * c[4] + t * (5.0 * c[5] + t * 6.0 * c[6]))));
        if (a.acc_z) a.acc_z[i] = 2.0 * c[2] + t * (6.0 * c[3] + t * (12.0 * c[4] + t * (20.0 * c[5] + t * 30.0 * c[6])));
        if (a.coeff) {
#pragma unroll
            for (int k = 0; k < 7; ++k) a.coeff[7 * i + k] = c[k];
        }
    }
}

struct WbidRefArgs {
    const double* x_next;    // [B][13]  the MPC's next state x_opt1[1] (run_simulation.py:111)
    const double* u0;        // [B][12]  first-step contact forces u_opt0
    const double* foot;      // [B][12]  current contact-point positions, 4 x xyz
    double* R;               // [B][9]   base orientation reference, row-major
    double* base_vel;        // [B][6]   [v, omega]
    double* base_acc;        // [B][6]   [0, 0, 0, I^-1 sum_i r_i x omega]
    double* com_acc;         // [B][3]   sum of forces / mass + gravity
    double iinv[3];          // inverse torso inertia diagonal (wbid.py:262-269)
    double mass, gravity;
    int32_t as_written;      // 1: sum the forces exactly as wbid.py:290 does (reshape (3,4), see below); 0: per axis
    long long B;
};

__global__ __launch_bounds__(256) void srbdqp_wbid_reference_kernel(WbidRefArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.B; i += stride) {
        double x[13], u[12];
#pragma unroll
        for (int k = 0; k < 13; ++k) x[k] = a.x_next[13 * i + k];
#pragma unroll
        for (int k = 0; k < 12; ++k) u[k] = a.u0[12 * i + k];
        // tf.transformations.euler_matrix(roll, pitch, yaw), default axes 'sxyz': R = Rz(yaw) Ry(pitch) Rx(roll)
        // (wbid.py:246-247)
        const double si = sin(x[0]), sj = sin(x[1]), sk = sin(x[2]), ci = cos(x[0]), cj = cos(x[1]), ck = cos(x[2]);
        const double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
        double* R = a.R + 9 * i;
        R[0] = cj * ck; R[1] = sj * sc - cs; R[2] = sj * cc + ss;
        R[3] = cj * sk; R[4] = sj * ss + cc; R[5] = sj * cs - sc;
        R[6] = -sj;     R[7] = cj * si;      R[8] = cj * ci;
        // velocity reference = [linear, angular] (wbid.py:256-258)
        double* bv = a.base_vel + 6 * i;
        bv[0] = x[9]; bv[1] = x[10]; bv[2] = x[11]; bv[3] = x[6]; bv[4] = x[7]; bv[5] = x[8];
        // angular acceleration reference = I^-1 sum_i (r_i x omega), r_i = foot_i - com (wbid.py:271-280)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double rx = a.foot[12 * i + 3 * c] - x[3], ry = a.foot[12 * i + 3 * c + 1] - x[4], rz = a.foot[12 * i + 3 * c + 2] - x[5];
            s0 += ry * x[8] - rz * x[7];
            s1 += rz * x[6] - rx * x[8];
            s2 += rx * x[7] - ry * x[6];
        }
        double* ba = a.base_acc + 6 * i;
        ba[0] = 0.0; ba[1] = 0.0; ba[2] = 0.0;
        ba[3] = a.iinv[0] * s0; ba[4] = a.iinv[1] * s1; ba[5] = a.iinv[2] * s2;
        // CoM linear acceleration = sum of forces / mass + gravity (wbid.py:287-291).  As written the reference sums
        // np.reshape(u_opt0, (3, 4)) along axis 1, i.e. the three groups of four CONSECUTIVE entries u[0:4], u[4:8],
        // u[8:12] rather than the x, y, z components of the four contacts; as_written = 1 reproduces that.
        double f0, f1, f2;
        if (a.as_written) {
            f0 = ((u[0] + u[1]) + u[2]) + u[3];
            f1 = ((u[4] + u[5]) + u[6]) + u[7];
            f2 = ((u[8] + u[9]) + u[10]) + u[11];
        } else {
            f0 = ((u[0] + u[3]) + u[6]) + u[9];
            f1 = ((u[1] + u[4]) + u[7]) + u[10];
            f2 = ((u[2] + u[5]) + u[8]) + u[11];
        }
        a.com_acc[3 * i] = f0 / a.mass;
        a.com_acc[3 * i + 1] = f1 / a.mass;
        a.com_acc[3 * i + 2] = f2 / a.mass + a.gravity;
    }
}

// ---- the step before the QP: gait schedule, landing position, input horizons (include/srbdqp_cascade.h) -----------------
struct MpcInputsArgs {
    const double* x0;        // [B][13]
    const double* feet;      // [B][12]
    const double* stamp;     // [B]
    const double* v_ref;     // [B][2]
    const uint8_t* standing; // [B] or null
    double* x_ref;           // [B][N][13]
    double* foot;            // [B][N][12]
    uint8_t* contact;        // [B][N][4]
    double* pcom;            // [B][N][3]
    double* landing;         // [B][3] or null
    double com_target[3];
    double hip_offset_y, dt;
    int32_t N, period, ds;
    long long B;
};

// A workgroup takes tiles of 32 consecutive robots: their 28 input doubles each are staged in LDS with coalesced loads (the
// algorithmic minimum of reads), then the tile's outputs -- contiguous in every output array -- are written element by
// element, 512 contiguous bytes per store instruction of a wave.  (First version: one thread per (robot, step) writing rows
// of 13 + 12 + 3 doubles: 1.2 TB/s; one thread per output element with the inputs re-read through L2: 2.4 TB/s.)  Every
// product and sum is a single rounded operation in the order of msgs.MpcNode.step (no fused multiply-add: the oracle is
// compared bit for bit).
template <int NH>   // the horizon as a compile-time constant: every index division becomes a multiply-shift
__global__ __launch_bounds__(256) void srbdqp_mpc_inputs_kernel(MpcInputsArgs a) {
#pragma clang fp contract(off)   // no fused multiply-add in this kernel (HIP's __dmul_rn / __dadd_rn do not prevent it)
    constexpr int R = 32, N = NH;
    __shared__ double sx0[R * 13], sft[R * 12], sv[R * 2];
    __shared__ int sph[R];                                               // phase of step 0 in [0, 2 period), or -1 = standing
    const int t = threadIdx.x;
    const long long tiles = (a.B + R - 1) / R;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long b0 = tile * R;
        const int nr = (int)((a.B - b0 < R) ? (a.B - b0) : R);
        for (int i = t; i < nr * 13; i += 256) sx0[i] = a.x0[b0 * 13 + i];
        for (int i = t; i < nr * 12; i += 256) sft[i] = a.feet[b0 * 12 + i];
        for (int i = t; i < nr * 2; i += 256) sv[i] = a.v_ref[b0 * 2 + i];
        if (t < nr) {   // alternating single support with a double-support overlap (msgs.AlternatingGait.contact_horizon)
            const bool stand = a.standing && a.standing[b0 + t];
            const long long k0 = (long long)floor(a.stamp[b0 + t] / a.dt + 1e-9);
            long long ph = k0 % (2LL * a.period);
            if (ph < 0) ph += 2LL * a.period;
            sph[t] = stand ? -1 : (int)ph;
        }
        __syncthreads();
        auto flags = [&](int r, int k, bool& cl, bool& cr) {
            const int p0 = sph[r];
            const int ph = (p0 + k) % (2 * a.period);
            const bool stand = p0 < 0, left_stance = ph < a.period, dsup = (ph % a.period) < a.ds;
            cl = stand || left_stance || dsup; cr = stand || !left_stance || dsup;
        };
        double* xr = a.x_ref + b0 * (N * 13);
        for (int e = t; e < nr * N * 13; e += 256) {                     // x_ref[r][k][c], reference index k + 1
            const int row = e / 13, c = e - 13 * row, r = row / N, k = row - N * r;
            const double* x0 = sx0 + 13 * r;
            const double vx = sv[2 * r], vy = sv[2 * r + 1];
            const bool moving = (vx != 0.0) || (vy != 0.0);
            const double kk = (double)(k + 1);
            double v = 0.0;
            if (c == 2) v = x0[2];
            else if (c == 3) v = moving ? x0[3] + (vx * kk) * a.dt : a.com_target[0];
            else if (c == 4) v = moving ? x0[4] + (vy * kk) * a.dt : a.com_target[1];
            else if (c == 5) v = a.com_target[2];
            else if (c == 9) v = vx;
            else if (c == 10) v = vy;
            else if (c == 12) v = x0[12];
            xr[e] = v;
        }
        double* fo = a.foot + b0 * (N * 12);
        for (int e = t; e < nr * N * 12; e += 256) {                     // foot[r][k][c]: the current contact points, repeated
            const int row = e / 12, c = e - 12 * row, r = row / N;
            fo[e] = sft[12 * r + c];
        }
        double* pc = a.pcom + b0 * (N * 3);
        for (int e = t; e < nr * N * 3; e += 256) {                      // pcom[r][k][c]: measured CoM + the commanded drift
            const int row = e / 3, c = e - 3 * row, r = row / N, k = row - N * r;
            const double vc = (c < 2) ? sv[2 * r + c] : 0.0;
            pc[e] = sx0[13 * r + 3 + c] + (vc * (double)k) * a.dt;
        }
        uint32_t* cw = reinterpret_cast<uint32_t*>(a.contact) + b0 * N;  // 4 flags per step = one aligned word
        for (int row = t; row < nr * N; row += 256) {
            const int r = row / N, k = row - N * r;
            bool cl, cr;
            flags(r, k, cl, cr);
            cw[row] = (cl ? 0x00000101u : 0u) | (cr ? 0x01010000u : 0u);
        }
        if (a.landing && t < nr * 3) {   // Raibert-style landing point of the foot that is (or goes next) in the air
            const int r = t / 3, c = t - 3 * r;
            const double* x0 = sx0 + 13 * r;
            bool cl, cr;
            flags(r, 0, cl, cr);
            const double T = (double)a.period * a.dt, side = cl ? -1.0 : 1.0;
            double v = 0.0;
            if (c == 0) v = (x0[3] + (0.5 * T) * x0[9]) + 0.03 * (x0[9] - sv[2 * r]);
            else if (c == 1) v = ((x0[4] + side * a.hip_offset_y) + (0.5 * T) * x0[10]) + 0.03 * (x0[10] - sv[2 * r + 1]);
            a.landing[b0 * 3 + t] = v;
        }
        __syncthreads();
    }
}

// u_opt0 of every QP of a solve, packed for the all-gather across GPUs (srbdqp_gather_u0_f64): out[b][c] = u[b][0][c], 12 doubles of the N x 12 a QP's plan holds
__global__ __launch_bounds__(256) void srbdqp_pack_u0_kernel(const double* __restrict__ u, double* __restrict__ out, long long items, int N) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < items) out[i] = u[(i / 12) * (12LL * N) + (i % 12)];
}

}  // namespace srbdqp

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

}  // namespace srbdqp

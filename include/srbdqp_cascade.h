/* srbdqp_cascade.h -- the steps either side of the QP in the reference's control cascade, batched on the GPU
 * (SURVEY.md 8(f) rows 2, 3 and 4).  Same library (libsrbdqp.so), same handle, same error convention as srbdqp.h.
 *
 *   srbdqp_swing_*            replaces  SwingTrajectory.calculate_coeff / calculate_position_xy / calculate_position_z /
 *                             calculate_velocity_z / calculate_acceleration_z
 *                             (g1_mujoco_sim/src/swing_trajectory.py:38-89; called from ros_run_simulation.py:246-256,
 *                             300-312), for B feet at once.
 *   srbdqp_wbid_reference_*   replaces the arithmetic of WBID.setReference(t, x_opt1, u_opt0, foot_positions_curr)
 *                             (g1_mujoco_sim/src/wbid.py:232-297): everything it computes before handing the
 *                             references to the OpenSoT tasks, for B robots at once.
 *
 * Host-buffer variants copy in and out and return when the outputs are valid; *_device_* variants take device pointers
 * and a hipStream_t (NULL = the handle's stream) and return after the launch.
 */
#ifndef SRBDQP_CASCADE_H
#define SRBDQP_CASCADE_H

#include "srbdqp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* swing_trajectory.py:50 (downward landing velocity) and :58 (share of the x-y distance covered in the first half) */
#define SRBDQP_SWING_FINAL_VELOCITY_Z (-0.02)
#define SRBDQP_SWING_FIRST_HALF_SHARE 0.80

/* p_start, p_final [B][3]; z_middle, progress [B]; pos [B][3]; vel_z, acc_z [B] (optional); coeff [B][7] (optional:
 * the polynomial's coefficients, lowest power first, = SwingTrajectory.coeff). */
int srbdqp_swing_f64(srbdqp_handle* h, int64_t B, const double* p_start, const double* p_final, const double* z_middle,
                     const double* progress, double final_velocity_z, double first_half_share, double* pos,
                     double* vel_z, double* acc_z, double* coeff);
int srbdqp_swing_device_f64(srbdqp_handle* h, int64_t B, const double* p_start, const double* p_final,
                            const double* z_middle, const double* progress, double final_velocity_z,
                            double first_half_share, double* pos, double* vel_z, double* acc_z, double* coeff,
                            void* stream);

/* x_next [B][13] (= x_opt1[1]); u0 [B][12] (= u_opt0); foot [B][12] (current contact points, srbdqp.h order);
 * R [B][9] row-major base orientation (tf euler_matrix 'sxyz'); base_vel [B][6] = [v, omega];
 * base_acc [B][6] = [0, I^-1 sum_i r_i x omega]; com_acc [B][3] = sum of forces / mass + gravity.
 * as_written != 0 sums the forces the way wbid.py:290 does (np.reshape(u_opt0, (3, 4)) summed along axis 1: groups of
 * four consecutive entries, not the per-axis sums); 0 gives the per-axis sums.  Mass, inertia: the handle's config;
 * gravity: x_next[12] is NOT used, the constant -9.80665 of wbid.py:286 is. */
int srbdqp_wbid_reference_f64(srbdqp_handle* h, int64_t B, const double* x_next, const double* u0, const double* foot,
                              int32_t as_written, double* R, double* base_vel, double* base_acc, double* com_acc);
int srbdqp_wbid_reference_device_f64(srbdqp_handle* h, int64_t B, const double* x_next, const double* u0,
                                     const double* foot, int32_t as_written, double* R, double* base_vel,
                                     double* base_acc, double* com_acc, void* stream);

/* ---- the step BEFORE the QP: gait schedule + landing position + the QP's input horizons, for B
 * robots at once.  What the reference's MPC node does between receiving /srbd_current and calling MPC.update()
 * (ros_run_simulation.py:214-218, 378-399 consume its outputs: contacts[i].active and landing_position; run_simulation.py:
 * 73-101 builds the same horizons for one robot).  The schedule itself is inside the absent module: this is the builder's own
 * design (fixed-period alternating single support with a double-support overlap, Raibert-style landing point), the same
 * one g1_locomotion_amd/msgs.py (AlternatingGait, MpcNode.step) runs on the host for one robot.
 *
 * Outputs are laid out exactly as srbdqp_solve_batch_device_f64 takes them, so a fleet's control step stays on the device:
 *   srbdqp_mpc_inputs_device_f64 -> srbdqp_solve_batch_device_f64(..., pcom) -> srbdqp_wbid_reference_device_f64. */
typedef struct srbdqp_gait {
    int32_t struct_size;          /* = sizeof(srbdqp_gait) */
    int32_t period_steps;         /* horizon steps one foot swings (0.25 s / dt, ros_run_simulation.py:148) */
    int32_t double_support_steps; /* steps of double support at the start of every half period */
    int32_t reserved0;
    double com_target[3];         /* CoM reference when v_ref = 0 (run_simulation.py:81) */
    double hip_offset_y;          /* lateral offset of the landing point from the CoM */
} srbdqp_gait;

/* x0 [B][13] measured state; feet [B][12] current contact-point positions (srbdqp.h order); stamp [B] seconds;
 * v_ref [B][2] commanded planar velocity; standing [B] non-zero = all four contacts in stance (may be NULL = walking).
 * Horizon N and dt: the handle's config.  Outputs: x_ref [B][N][13], foot [B][N][12], contact [B][N][4], pcom [B][N][3],
 * landing [B][3] (may be NULL): landing position of the foot that swings (or swings next) at the first step. */
int srbdqp_mpc_inputs_f64(srbdqp_handle* h, int64_t B, const double* x0, const double* feet, const double* stamp,
                          const double* v_ref, const uint8_t* standing, const srbdqp_gait* gait,
                          double* x_ref, double* foot, uint8_t* contact, double* pcom, double* landing);
int srbdqp_mpc_inputs_device_f64(srbdqp_handle* h, int64_t B, const double* x0, const double* feet, const double* stamp,
                                 const double* v_ref, const uint8_t* standing, const srbdqp_gait* gait,
                                 double* x_ref, double* foot, uint8_t* contact, double* pcom, double* landing, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRBDQP_CASCADE_H */

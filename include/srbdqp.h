/* srbdqp.h -- C-ABI of the MI355X-native batched SRBD convex-MPC QP engine.
 *
 * This is the drop-in boundary for the hot path of ioloizou/g1_locomotion: the per-control-step
 * body of `srbd_mpc.mpc.MPC.update(...)` (reference call site
 * g1_mujoco_sim/src/run_simulation.py:106; node `mpc_to_wbid_node`,
 * g1_mujoco_sim/launch/mpc_wbid_simulation.launch:6).  The reference has NO FFI for this path (it is a
 * duck-typed Python object, run_simulation.py:169-170,73-82,96,103,106), so these entry points are what a
 * ctypes binding of that object binds; each cites the reference interface it stands behind.
 *
 * Conventions (all evidenced on the reference's call sites):
 *   state  x[13] = roll pitch yaw | com xyz | omega xyz | v_com xyz | g      run_simulation.py:73-77
 *   input  u[12] = f_Lheel f_Ltoe f_Rheel f_Rtoe, world xyz each             ros_run_simulation.py:65,214-215
 *   arrays are C-contiguous row-major; the library keeps no caller pointer after a call returns;
 *   one handle <-> one HIP stream <-> one calling thread at a time (the reference is single threaded,
 *   run_simulation.py:135-153).
 *
 * Error convention: every function returns 0 on success, a negative SRBDQP_E_* code on failure (never
 * throws); srbdqp_last_error() gives the text.  Per-QP solver outcome comes back in status[]/iters[].
 * There is NO CPU fallback: if no HIP device is usable srbdqp_create fails with SRBDQP_E_NO_DEVICE.
 */
#ifndef SRBDQP_H
#define SRBDQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRBDQP_NX 13
#define SRBDQP_NU 12
#define SRBDQP_NC 4
#define SRBDQP_ROWS_PER_STEP 20   /* 4 contacts x (4 friction-pyramid rows + 1 normal-force row) */
#define SRBDQP_MAX_HORIZON 24     /* instantiations: N in {4, 8, 10, 12, 16, 20, 24} */

/* return codes */
#define SRBDQP_OK 0
#define SRBDQP_E_INVALID   (-1)   /* bad argument (null pointer, unsupported horizon, ...) */
#define SRBDQP_E_NO_DEVICE (-2)   /* no usable HIP device: there is no CPU fallback */
#define SRBDQP_E_HIP       (-3)   /* a HIP runtime call failed */
#define SRBDQP_E_NOMEM     (-4)

/* per-QP status[] values */
#define SRBDQP_SOLVED    1        /* primal and dual residual below tolerance */
#define SRBDQP_MAX_ITER  2        /* iteration cap reached; best iterate returned */
#define SRBDQP_PENDING   0        /* SRBDQP_FLAG_DEFER_TAIL only: the QP's continuation is deferred to the next solve on the stream / srbdqp_flush() */
#define SRBDQP_NUMERICAL (-1)     /* non-finite residual / non-positive pivot (e.g. NaN inputs); forces returned as 0 */
#define SRBDQP_CONTACT_BOUND (-2) /* more stance contacts in a step than srbdqp_config.max_contacts_per_step allows; forces 0 */

/* srbdqp_config.flags */
#define SRBDQP_FLAG_TIMING 1      /* bracket every kernel launch with HIP events (srbdqp_last_kernel_ms) */
#define SRBDQP_FLAG_SETUP4 4      /* split pipeline: set-up kernel with 4 waves per QP instead of one wave per QP (A/B) */
#define SRBDQP_FLAG_F64_TILES 8   /* _f32 calls: factor every QP's T in fp64 tiles (default: fp32 tiles for the QPs whose steps all
                                     have 0 or >= 3 stance contacts; A/B and accuracy studies) */
#define SRBDQP_FLAG_F32_TILES 16  /* _f32 calls: fp32 tiles for the eligible QPs of batches below 512 too (tests) */
#define SRBDQP_FLAG_NO_LAT 32     /* staged calls on the general kernel: the batch instantiation instead of the low-latency one (A/B) */
#define SRBDQP_FLAG_DEFER_TAIL 64 /* device-buffer solves with the rho restart on: a QP that reaches a restart mark unconverged does not hold its own
                                     launch up (a launch lasts as long as its slowest QP: up to three set-ups and 250 iterations) -- its next pass runs
                                     BESIDE the caller's next solve.  One-wave kernel (N <= 10, at most 2 stance contacts per step): the QP hands
                                     itself, with its (x, y), to the next solve enqueued on the same stream, whose first workgroups run the pass.
                                     Every other kernel: the restart passes run on a stream of the library's own behind an event.  Same passes,
                                     same results; the outputs of such a QP (about 4 % of a configs[1] batch) arrive later: its status[] reads
                                     SRBDQP_PENDING (one-wave kernel) or SRBDQP_MAX_ITER (the others: the pass it ended) until then, and
                                     srbdqp_flush() completes / waits for what is left.  A caller must not read a solve's outputs, nor hand its
                                     output arrays to another solve, before the flush.  The same holds for the solve's INPUT arrays (x0, x_ref, foot,
                                     contact, pcom, warm_u, warm_y): a continuation rebuilds its QP from the pointers of the launch it came from, one or
                                     two launches later, and the restart passes on the tail stream re-read them -- every input array of a deferred
                                     solve must stay unmodified until srbdqp_flush / srbdqp_ragged_flush has completed in stream order (a receding-horizon
                                     caller that overwrites its inputs for the next step first gets continuations of a QP that is half old, half new,
                                     with no error).  For pipelines of independent batches in their own buffers; see srbdqp_flush */
#define SRBDQP_FLAG_NO_SPIN 2     /* srbdqp_solve_staged_f64: wait with hipStreamSynchronize instead of spinning on the
                                     completion word the kernel writes to host memory */

/* srbdqp_config.kernel: which implementation of the hot path runs */
#define SRBDQP_KERNEL_AUTO  0     /* the fastest parity-green kernel */
/* 1, 2: the round-1 baselines v0 (Gauss-Jordan) and v1 (all 12N variables on the matrix cores), retired in round 2 */
#define SRBDQP_KERNEL_COMPACT 3   /* 4 waves per QP on the presolved QP (swing-contact variables eliminated), closed-form assembly */
#define SRBDQP_KERNEL_SPLIT 4     /* two kernels, one wave per QP each: set-up, then ADMM + roll-out, K^-1 handed over through
                                     HBM (<= 64 presolved variables, otherwise falls back to COMPACT); kept for A/B */
#define SRBDQP_KERNEL_WAVE 5      /* the whole solve on one wave per QP, K tiles register-resident, no barrier, no hand-over;
                                     AUTO picks it for the instantiations with <= 64 presolved variables (N <= 10 with at most 2 stance
                                     contacts per step, N = 4) at every batch size; the staged batch-1 call runs the 4-wave kernel's
                                     low-latency form */
#define SRBDQP_KERNEL_WRENCH 6    /* the general kernel: any contact pattern at every horizon (4 ... 24), wrench-space presolve
                                     (a step with >= 3 stance contacts contributes 6 coordinates instead of 3 per contact),
                                     fp64 or fp32 iterations; AUTO picks it for > 2 stance contacts per step at N > 10 and -- batches of
                                     512 QPs and more, and the staged batch-1 call -- at N <= 10, for N = 24 and for every _f32 call */

/* Everything `MPC.__init__(dt)` / `MPC.init_matrices()` hold (run_simulation.py:169-170).  Values the
 * reference keeps inside the absent module are this build's documented choices. */
typedef struct srbdqp_config {
    int32_t struct_size;          /* = sizeof(srbdqp_config), ABI check */
    int32_t horizon;              /* HORIZON_LENGTH (run_simulation.py:96) */
    int32_t device;               /* HIP device ordinal */
    int32_t flags;
    int32_t kernel;               /* SRBDQP_KERNEL_* */
    int32_t max_iter;
    int32_t check_every;
    int32_t max_contacts_per_step; /* bound on stance contact points per horizon step: 1..4; 0 = decide per call
                                    * (host-buffer API scans the contact flags, device API assumes 4).  A bound of
                                    * <= 2 selects the smaller, higher-occupancy kernel instantiation. */
    int32_t rho_restart_iter;     /* OSQP-style re-balancing of rho: a QP that has not converged after this many iterations is
                                   * re-factored with rho' = rho sqrt((r_prim/n_prim)/(r_dual/n_dual)) (clipped to [rho/10, 5 rho]) and
                                   * continues from its own (x, y) until max_iter iterations in total; iters[] counts every pass.
                                   * < 0 or >= max_iter = off; 0 (default) = automatic, by horizon: 55 (N <= 10), 70 (12), 80 (16), 125 (20), 100 (24) --
                                   * the same rule for every kernel and batch size (a QP ends with the same status alone in a staged
                                   * call and inside a batch of 65,536).  How a pass runs: in place inside the one-wave kernel (or handed
                                   * to the next solve on the stream, SRBDQP_FLAG_DEFER_TAIL); one more launch over the same grid per
                                   * pass everywhere else (only the workgroups of the QPs the pass before left at its cap do anything);
                                   * the staged batch-1 call launches a further pass only when a status[] asks for it. */
    int32_t rho_restart_count;    /* at most this many re-balancings, one every rho_restart_iter iterations, each from the rho of the
                                   * pass before it.  0 (default) = automatic: with the automatic rho_restart_iter 2 (N <= 12), 3 (N = 16),
                                   * 1 (N = 20), 2 (N = 24); with an explicit rho_restart_iter 1.  Values above 3 mean 3 (every kernel: the
                                   * one-wave kernel holds its continued passes as three straight copies of its body) */
    double dt;                    /* run_simulation.py:169 */
    double mass;                  /* wbid.py:291 model.getMass() */
    double inertia[3];            /* wbid.py:261-266 torso inertia diagonal */
    double mu;                    /* wbid.py:17 */
    double fz_min, fz_max;        /* wbid.py:123-124 */
    double q_diag[SRBDQP_NX];     /* state tracking weights */
    double r_diag;                /* force regularisation */
    double force_scale;           /* u = force_scale * u_hat */
    double rho, rho_eq_scale, sigma, alpha, eps_abs, eps_rel;   /* ADMM (OSQP Algorithm 1).  rho = 0 (the default) = automatic: 0.7
                                   * (the penalty of the friction rows; see rho_fz_scale).  The fp32 entry points raise
                                   * eps_abs / eps_rel below 2e-6 to 2e-6 (resolution of fp32 residuals). */
    double rho_fz_scale;          /* penalty of a stance contact's normal-force row (fz_min <= fz <= fz_max) relative to rho; the four
                                   * friction rows keep rho.  0 (the default) = automatic: 4 (with rho = 0.7: 14 - 30 % fewer ADMM
                                   * iterations than one common penalty and a far shorter tail, forces within 3e-3 N of the exact optimum
                                   * at every horizon) */
} srbdqp_config;

typedef struct srbdqp_handle srbdqp_handle;

/* Fill *cfg with the defaults (N=10, dt=0.04, G1 constants, ADMM constants).  cfg->struct_size must hold sizeof(srbdqp_config) of
 * the CALLER's build on entry (ABI check): a binding made against an older, shorter struct gets SRBDQP_E_INVALID back and nothing is
 * written (srbdqp_last_error(NULL) says why). */
int srbdqp_default_config(srbdqp_config* cfg);

/* ~ MPC(dt) + MPC.init_matrices() (run_simulation.py:169-170): allocate stream + workspace. */
int srbdqp_create(const srbdqp_config* cfg, srbdqp_handle** out);
int srbdqp_destroy(srbdqp_handle* h);
const char* srbdqp_last_error(const srbdqp_handle* h);   /* h may be NULL: last create() error */

/* ~ MPC.update(contact_horizon, c_horizon, p_com_horizon, x_current, one_rollout) for B independent
 * QPs (run_simulation.py:106).  HOST buffers; H2D/D2H copies happen inside on the handle's stream and
 * the call returns after the results are in the output buffers.
 *   x0      [B][13]        current state (MPC.x0, run_simulation.py:73-77)
 *   x_ref   [B][N][13]     MPC.x_ref_hor (run_simulation.py:80-82); column 2 is the linearisation yaw
 *   foot    [B][N][12]     c_horizon (run_simulation.py:94-97), world frame
 *   contact [B][N][4]      contact_horizon (run_simulation.py:100-101), nonzero = in contact
 *   pcom    [B][N][3]      p_com_horizon (run_simulation.py:103); NULL -> x_ref[:, :, 3:6]
 *   warm_u  [B][12N]       optional warm start (newtons), NULL = cold start from 0
 *   warm_y  [B][20N]       optional dual warm start, NULL = 0
 *   u_out   [B][N][12]     optimal forces in newtons; u_out[b][0] is u_opt0 (run_simulation.py:111)
 *   x_out   [B][N+1][13]   state roll-out, row 0 = x0, row 1 = x_opt1[1] (run_simulation.py:111), may be NULL
 *   y_out   [B][20N]       dual solution (for warm starts), may be NULL
 *   status  [B], iters[B]  per-QP outcome, may be NULL
 */
int srbdqp_solve_batch_f64(srbdqp_handle* h, int32_t B,
                           const double* x0, const double* x_ref, const double* foot,
                           const uint8_t* contact, const double* pcom,
                           const double* warm_u, const double* warm_y,
                           double* u_out, double* x_out, double* y_out,
                           int32_t* status, int32_t* iters);

/* Same, DEVICE buffers (HBM-resident, e.g. torch tensors' data_ptr()).  Enqueued on `stream`
 * (a hipStream_t; NULL = the handle's own stream); returns without synchronising. */
int srbdqp_solve_batch_device_f64(srbdqp_handle* h, int32_t B,
                                  const double* x0, const double* x_ref, const double* foot,
                                  const uint8_t* contact, const double* pcom,
                                  const double* warm_u, const double* warm_y,
                                  double* u_out, double* x_out, double* y_out,
                                  int32_t* status, int32_t* iters, void* stream);

/* The same two calls with fp32 buffers (BASELINE.json configs[2]: "Batch=65536, N=20, 4-contact double-support, fp32").
 * Every array above becomes float; contact stays uint8.  The closed-form assembly is computed in fp64 on chip; the ADMM
 * iterations run in fp32 on the general kernel (SRBDQP_KERNEL_WRENCH), whatever srbdqp_config.kernel says.  The wrench-space
 * matrix T is factored and inverted in fp64 tiles, or -- calls of >= 512 QPs, for the QPs whose steps all have 0 or >= 3 stance
 * contacts -- in fp32 tiles with one fp64 refinement step of x_q (SRBDQP_FLAG_F64_TILES / _F32_TILES force either): such a call
 * is two launches over the same grid.  Tolerances reachable in fp32: see DESIGN.md sections 2 and 3. */
int srbdqp_solve_batch_f32(srbdqp_handle* h, int32_t B,
                           const float* x0, const float* x_ref, const float* foot,
                           const uint8_t* contact, const float* pcom,
                           const float* warm_u, const float* warm_y,
                           float* u_out, float* x_out, float* y_out,
                           int32_t* status, int32_t* iters);
int srbdqp_solve_batch_device_f32(srbdqp_handle* h, int32_t B,
                                  const float* x0, const float* x_ref, const float* foot,
                                  const uint8_t* contact, const float* pcom,
                                  const float* warm_u, const float* warm_y,
                                  float* u_out, float* x_out, float* y_out,
                                  int32_t* status, int32_t* iters, void* stream);

/* ~ the QP-assembly half of MPC.update (what init_matrices() allocates: H, g, cone rows), produced by the SHIPPED kernel:
 * the kernel a srbdqp_solve_batch_f64 of the same batch would run (compact / one-wave, by srbdqp_config.kernel, batch size
 * and contact bound) is started in dump mode and stops right before its factorisation.  HOST buffers, for parity tests of
 * rows a5-a8 (tests/test_gpu_parity.py).  Scaled variables u_hat.
 *   P_out [B][12N][12N]   Hessian H = Bs' Q Bs + R s^2 of the PRESOLVED QP in the original variable order: rows / columns of
 *                         swing-contact variables (eliminated before anything is built) are 0
 *   q_out [B][12N]        gradient (0 for swing-contact variables)
 *   l_out, ub_out [B][20N] constraint bounds (rows 20k+5i+j; +-1e30 = unbounded): for stance contacts the values the kernel's own
 *                         ADMM loop uses (dumped by the kernel); rows of swing contacts do not exist on the device (presolve) and
 *                         are reported as the clamp 0 <= fz <= 0 they stand for
 * Configurations that solve on the general kernel (SRBDQP_KERNEL_WRENCH, or what AUTO routes to it) return
 * SRBDQP_E_INVALID: their assembly is srbdqp_assemble_wrench_f64's.
 */
int srbdqp_assemble_f64(srbdqp_handle* h, int32_t B,
                        const double* x0, const double* x_ref, const double* foot,
                        const uint8_t* contact, const double* pcom,
                        double* P_out, double* q_out, double* l_out, double* ub_out);

/* The same for the general kernel (SRBDQP_KERNEL_WRENCH), whose presolve never forms H: what it does build, dumped by the
 * shipped kernel itself right before its factorisation.  With g_j = the coordinates of step j (6 wrench coordinates when
 * the step has >= 3 stance contacts, its 3c force variables otherwise), n_g = sum_j g_j <= 6N:
 *   T_out      [B][6N][6N]   T = S + E^-1 (symmetric, leading n_g x n_g block, rest 0); K^-1 = Bd + V' T^-1 V
 *   q_out      [B][12N]      gradient in the scaled force variables (0 for swing contacts)
 *   blocks_out [B][12N][24]  per force variable u (index 12 k + 3 i + a): row u of Bd within its step (12), column u of
 *                            V (6), and -- as lane (row r = (u % 12) / 2, half h = u % 2) of the step -- V[r][6h .. 6h+5] (6)
 *   goff_out   [B][N+1]      offset of step j's coordinates in T (as doubles); goff[N] = n_g
 * HOST buffers; parity tests of rows a5-a8 on the kernel that ships (tests/test_gpu_wrench.py). */
int srbdqp_assemble_wrench_f64(srbdqp_handle* h, int32_t B,
                               const double* x0, const double* x_ref, const double* foot,
                               const uint8_t* contact, const double* pcom,
                               double* T_out, double* q_out, double* blocks_out, double* goff_out);

/* Ragged batches (BASELINE.json configs[4]: "Mixed horizon N in {8,12,16,24} with per-QP contact schedule (ragged batch,
 * bucketed kernel launch)").  One object holds an engine per horizon; a call takes the QPs in ANY order with their horizon
 * in N_per_qp[] (HOST array), sorts them into horizon buckets and launches every non-empty bucket on its own HIP stream --
 * all buckets in flight together, no host blocking (the caller's stream is made to wait for them with events).  Step-major
 * PACKED arrays: QP b owns rows [off_b, off_b + N_b) with off_b = sum of the horizons before it:
 *   x0 [B][13]   x_ref [sum N][13]   foot [sum N][12]   contact [sum N][4]   u_out [sum N][12]   x_out [sum N + B][13]
 *   (x_out of QP b starts at row off_b + b: N_b + 1 rows), status / iters [B] in the caller's QP order.
 * Every bucket runs the general kernel (any per-QP contact schedule); cfg is the template of the per-horizon engines
 * (horizon ignored, rho = 0 picks each horizon's own penalty, rho_restart_iter as for a homogeneous batch of that horizon:
 * by default the N > 10 buckets take the two-pass rho restart).  Warm starts: srbdqp_solve_ragged_warm_device_*. */
typedef struct srbdqp_ragged srbdqp_ragged;
int srbdqp_ragged_create(const srbdqp_config* cfg, const int32_t* horizons, int32_t n_horizons, srbdqp_ragged** out);
int srbdqp_ragged_destroy(srbdqp_ragged* r);
const char* srbdqp_ragged_last_error(const srbdqp_ragged* r);   /* r may be NULL: last create() error */
/* DEVICE buffers; enqueued behind `stream` (NULL = the object's own), returns without synchronising. */
int srbdqp_solve_ragged_device_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                                   const double* x0, const double* x_ref, const double* foot, const uint8_t* contact,
                                   double* u_out, double* x_out, int32_t* status, int32_t* iters, void* stream);
/* HOST buffers (copies inside; returns with the results in place). */
int srbdqp_solve_ragged_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                            const double* x0, const double* x_ref, const double* foot, const uint8_t* contact,
                            double* u_out, double* x_out, int32_t* status, int32_t* iters);
/* cfg.flags & SRBDQP_FLAG_DEFER_TAIL at srbdqp_ragged_create: the restart passes of a bucket run on the bucket's own tail stream behind its first pass -- the
 * caller's stream waits for the first passes only, the passes run beside what it enqueues next (e.g. the next call: every shared array exists three times, in
 * rotation).  The call's input AND output arrays must stay untouched until the flush has completed in stream order (the passes read the former, write the latter).  srbdqp_ragged_flush makes `stream` (NULL = the object's own) wait for the passes still running; before it the outputs of the QPs a first pass left
 * at its cap (status SRBDQP_MAX_ITER at that point) are not final.  A no-op without the flag.  The host-buffer entry points flush by themselves. */
int srbdqp_ragged_flush(srbdqp_ragged* r, void* stream);
/* The same two calls with fp32 buffers and fp32 ADMM iterations (as srbdqp_solve_batch_device_f32 / _f32; every bucket factors
 * its T in fp64 tiles: a bucket's QPs are not split by tile precision). */
int srbdqp_solve_ragged_device_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                                   const float* x0, const float* x_ref, const float* foot, const uint8_t* contact,
                                   float* u_out, float* x_out, int32_t* status, int32_t* iters, void* stream);
int srbdqp_solve_ragged_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                            const float* x0, const float* x_ref, const float* foot, const uint8_t* contact,
                            float* u_out, float* x_out, int32_t* status, int32_t* iters);
/* Ragged solves with a warm start and / or the dual solution back (what a receding-horizon fleet carries from one control
 * step to the next), DEVICE buffers, packed like u_out:  warm_u [sum N][12] (newtons) and warm_y [sum N][20] may each be NULL
 * (cold start), y_out [sum N][20] may be NULL. */
int srbdqp_solve_ragged_warm_device_f64(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                                        const double* x0, const double* x_ref, const double* foot, const uint8_t* contact,
                                        const double* warm_u, const double* warm_y,
                                        double* u_out, double* x_out, double* y_out, int32_t* status, int32_t* iters, void* stream);
int srbdqp_solve_ragged_warm_device_f32(srbdqp_ragged* r, int32_t B, const int32_t* N_per_qp,
                                        const float* x0, const float* x_ref, const float* foot, const uint8_t* contact,
                                        const float* warm_u, const float* warm_y,
                                        float* u_out, float* x_out, float* y_out, int32_t* status, int32_t* iters, void* stream);

/* Longest-first scheduling hint for the DEVICE-buffer API only (the host-buffer and the staged calls ignore it):
 * `device_iters_prev` = the iters[] array (device memory, `length` entries) of the previous control step of the same
 * batch, or NULL to switch the hint off.  A solve of more than `length` QPs is dispatched in natural order.  Subsequent device
 * solves first build a dispatch order from it (one tiny kernel on the same stream) so that QPs that needed many ADMM
 * iterations last step start first and the straggler tail overlaps the bulk of the launch.  In a receding-horizon
 * loop consecutive steps of a robot are strongly correlated; a wrong hint costs nothing but the reordering.  The
 * pointer is read at every solve; results, status[] and iters[] stay in the caller's QP order. */
int srbdqp_set_schedule_hint(srbdqp_handle* h, const int32_t* device_iters_prev, int32_t length);

/* SRBDQP_FLAG_DEFER_TAIL: complete what earlier device-buffer solves on `stream` (a hipStream_t; NULL = every stream this handle has
 * launched on) left for later -- the one-wave kernel's continuations that no later solve has picked up (one launch of the
 * continuations alone, enqueued on that stream, in which every such QP runs all the passes it has left), and the restart passes running on the library's own tail stream (the
 * stream is made to wait for them through an event).  Returns without synchronising; after it every output of every earlier solve on
 * that stream is complete in stream order -- and only then may the INPUT arrays of those solves be overwritten (SRBDQP_FLAG_DEFER_TAIL above).
 * A no-op without the flag or with nothing pending.  srbdqp_synchronize() flushes the handle's own stream first. */
int srbdqp_flush(srbdqp_handle* h, void* stream);

/* Multi-GPU without torch.distributed (SURVEY row e; one process per GPU).  The path shards by robots: QPs are independent, so a fleet of `total` QPs is cut
 * into contiguous per-rank ranges -- srbdqp_shard_range: rank r of `world` owns [*first, *first + *count), the first total % world ranks one QP more (the
 * same rule as g1_locomotion_amd/shard.py) -- and nothing crosses GPUs while they solve.  The one exchange is what the MPC's consumer reads from every robot,
 * the first-step forces u_opt0 (g1_mujoco_sim/src/ros_run_simulation.py:214-215): srbdqp_gather_u0_f64 packs u_local[b][0][0..12) of this rank's B_local
 * QPs (device memory, the u_out of a solve, [B_local][N][12]) into its slot of u0_all (device memory, [world * B_local][12], rank-major) and all-gathers the
 * slots over RCCL, in place, on `stream` (a hipStream_t; NULL = the handle's) -- 96 bytes per QP over xGMI.  rccl_comm is an ncclComm_t the caller made
 * (ncclCommInitRank with the id it distributed itself); B_local must be the same on every rank, as for any all-gather.  The library finds RCCL at the first call
 * (dlopen of librccl.so.1: not a link-time dependency); SRBDQP_E_HIP with a message if it is not there or a call fails.  Does not synchronise. */
int srbdqp_shard_range(int64_t total, int32_t world, int32_t rank, int64_t* first, int64_t* count);
int srbdqp_gather_u0_f64(srbdqp_handle* h, const double* u_local, int64_t B_local, double* u0_all, void* rccl_comm, void* stream);

/* Low-latency path for small batches (the single-robot control loop, B = 1): the library owns pinned, GPU-mapped host
 * staging arrays; the caller fills the inputs in place, calls srbdqp_solve_staged_f64 (one kernel launch that reads and
 * writes the staging memory directly over PCIe -- no hipMemcpy calls), and reads the outputs in place.  Shapes as in
 * srbdqp_solve_batch_f64 with B <= capacity.  ~ MPC.update() for one robot (run_simulation.py:106). */
typedef struct srbdqp_stage {
    int32_t capacity;             /* max B */
    int32_t reserved;
    double* x0; double* x_ref; double* foot; uint8_t* contact; double* pcom; double* warm_u; double* warm_y;
    double* u; double* x; double* y; int32_t* status; int32_t* iters;
} srbdqp_stage;
int srbdqp_stage_ptrs(srbdqp_handle* h, srbdqp_stage* out);
int srbdqp_solve_staged_f64(srbdqp_handle* h, int32_t B, int32_t use_pcom, int32_t use_warm, int32_t want_x, int32_t want_y);

/* ~ MPC.update(contact_horizon, c_horizon, p_com_horizon, x_current, one_rollout) -> (u_opt0, x_opt1) for ONE robot at control rate
 * (run_simulation.py:106,111) in ONE call: copies the inputs into the staging arrays, launches the batch-1 kernel, spins on its
 * completion word and copies the results out.  What a binding of the reference's duck-typed object binds for its update() method.
 *   x0 [13]   x_ref [N][13]   foot [N][12]   contact [N][4]   pcom [N][3] or NULL (-> x_ref[:, 3:6])       HOST pointers
 *   u0_out [12]      u_opt0: the first step's forces, newtons (run_simulation.py:111)
 *   u_out  [N][12]   the whole plan, may be NULL
 *   x_out  [N+1][13] the roll-out (row 1 = x_opt1[1], the next state), may be NULL
 *   status, iters    solver outcome of this QP (SRBDQP_SOLVED, ...), may be NULL
 * An argument that IS the matching array of srbdqp_stage_ptrs() is used in place (no copy): that is how g1_locomotion_amd/mpc.py
 * calls it, with every argument bound once.  Returns SRBDQP_OK whenever the solve ran; the QP's own outcome is *status. */
int srbdqp_update_f64(srbdqp_handle* h, const double* x0, const double* x_ref, const double* foot, const uint8_t* contact,
                      const double* pcom, double* u0_out, double* u_out, double* x_out, int32_t* status, int32_t* iters);

/* Two-phase form of the staged call, for control loops that know the contact schedule, the contact-point positions and the
 * reference horizon BEFORE the state estimate arrives.  K (and its factorisation) depends on neither x0 nor x_ref's non-yaw
 * entries, and the gradient is affine in x0:
 *   srbdqp_prepare_staged_f64  reads the staging arrays (x0 = a prediction, anything finite), runs the set-up and stores
 *                              K^-1 and dq/dx0 on the device; returns at once (asynchronous on the handle's stream);
 *   srbdqp_solve_prepared_f64  reads x0 (only) from the staging arrays again, patches the gradient, runs the ADMM iterations
 *                              and the roll-out on one wave per QP and returns with the outputs in the staging arrays.
 * Same QP, same iterates as srbdqp_solve_staged_f64 on the same inputs (the split pipeline of SRBDQP_KERNEL_SPLIT); no warm
 * start, no rho restart.  Built for the instantiations with at most 64 presolved variables (N = 10 / 8 with <= 2 stance contacts
 * per step, N = 4 any).  A QP without a stance contact is finished by prepare(): its predicted states are the free response
 * of the PREDICTED x0.  The reference's own loop measures the contact points together with the state
 * (run_simulation.py:94-97): there this is a different mode of operation, not the drop-in -- see DESIGN.md section 6. */
int srbdqp_prepare_staged_f64(srbdqp_handle* h, int32_t B, int32_t use_pcom);
int srbdqp_solve_prepared_f64(srbdqp_handle* h, int32_t B, int32_t want_x, int32_t want_y);

/* Diagnostic: device buffer [B][16] of int64 that subsequent solves fill with per-QP s_memtime stamps of the kernel's
 * phase boundaries (100 MHz constant clock); NULL switches stamping off.  Not part of the drop-in surface. */
int srbdqp_set_stamp_buffer(srbdqp_handle* h, void* device_ptr);

/* Block until everything enqueued on the handle's stream is done. */
int srbdqp_synchronize(srbdqp_handle* h);

/* With SRBDQP_FLAG_TIMING: device time (HIP events on the launch stream) of the most recent solve's
 * kernel launch, in milliseconds; synchronises that event.  Negative if unavailable. */
double srbdqp_last_kernel_ms(srbdqp_handle* h);

/* With SRBDQP_FLAG_TIMING, after a solve that ran as the split pipeline: device time of its two kernels (the set-up
 * kernel incl. the dispatch-order kernel in front of it, and the ADMM + roll-out kernel).  SRBDQP_E_INVALID otherwise. */
int srbdqp_last_kernel_parts_ms(srbdqp_handle* h, double* setup_ms, double* admm_ms);

/* Name of the kernel variant the last solve launched ("wave_f64_n10_s2", "compact_f64_n10_s4", "wrench_f32_n20", ...). */
const char* srbdqp_kernel_name(const srbdqp_handle* h);

/* How the staged one-QP call (srbdqp_update_f64, srbdqp_solve_staged_f64 with B = 1) reaches the GPU on this handle: "aql" -- dispatch packets
 * the library writes into an HSA user-mode queue of its own (csrc/srbdqp_aql.hpp; 1.4 us less per call than the runtime's launch) --, or
 * "hip: <why not>" -- hipLaunchKernelGGL on the handle's stream, the same kernels (SRBDQP_NO_AQL=1 in the environment, SRBDQP_FLAG_NO_SPIN /
 * _TIMING / _DEFER_TAIL on the handle, or the queue could not be set up), or "undecided" before the first such call.  Diagnostic. */
const char* srbdqp_batch1_launch_path(const srbdqp_handle* h);

/* Library version / build string. */
const char* srbdqp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SRBDQP_H */

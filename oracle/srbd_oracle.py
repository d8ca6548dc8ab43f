"""CPU oracle for the SRBD convex-MPC QP hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  The product path (``g1_locomotion_amd``) never does; it fails loudly when
the HIP library is missing.

PARITY UNPINNED.  The reference's implementation of this path lives in the un-vendored
submodule ``g1_mpc`` -> github.com/ioloizou/srbd_mpc (``/root/reference/.gitmodules:1-3``,
branch hint ``walking-demo`` at ``README.md:90-91``, pinned SHA unrecoverable) and is absent
from the snapshot; its QP solver (OSQP, per BASELINE.json) is not installed.  The reference
holds no test, fixture or golden vector for this path.  This file therefore restates the
*published* algorithms (single-rigid-body convex MPC, Di Carlo et al. IROS 2018; OSQP's ADMM,
Stellato et al. 2020, Algorithm 1) and anchors every convention that IS evidenced on the
reference's own call sites:

  state  x = [roll pitch yaw | com | omega | v_com | g]   g1_mujoco_sim/src/run_simulation.py:73-77
                                                          g1_mujoco_sim/src/ros_run_simulation.py:199-211
  input  u = [f_Lheel f_Ltoe f_Rheel f_Rtoe] (world xyz)  g1_mujoco_sim/src/ros_run_simulation.py:65,214-215
                                                          g1_mujoco_sim/src/wbid.py:296-297
  update(contact_horizon, c_horizon, p_com_horizon, x_current, one_rollout) -> (u_opt0, x_opt1)
                                                          g1_mujoco_sim/src/run_simulation.py:94-106,111
  gravity state = -9.80665                                g1_mujoco_sim/src/ros_run_simulation.py:58
  dt = 0.04                                               g1_mujoco_sim/src/run_simulation.py:169
  torso inertia diag(8.20564e-2, 8.05015e-2, 0.32353e-2)  g1_mujoco_sim/src/wbid.py:261-266
  mass = sum of <mass> in g1_description/g1_23dof.urdf = 34.13385728 (model.getMass(), wbid.py:291)
  mu = 0.8, fz in [10, 1000] N stance / 0 swing           g1_mujoco_sim/src/wbid.py:17,123-124
                                                          g1_mujoco_sim/src/ros_run_simulation.py:235-244

Everything else (weights, discretisation, row ordering, ADMM constants) is this build's own
documented choice -- see DESIGN.md section "Problem specification".

The ADMM below (``admm_solve``) is the algorithm the HIP kernels implement, operation for
operation.  ``solve_reference`` is an *independent* high-accuracy solver (active-set KKT
solve + verification) used to pin the QP optimum; ``kkt_residuals`` is solver independent.
"""
from __future__ import annotations

from dataclasses import replace, dataclass, field, asdict
import numpy as np

NX = 13          # state size (12 dynamic + gravity)
NU = 12          # input size (4 contact points x 3)
NC = 4           # contact points
ROWS_PER_CONTACT = 5
INF = 1.0e30     # "infinite" bound, as OSQP's OSQP_INFTY

# status codes shared with include/srbdqp.h
STATUS_SOLVED = 1
STATUS_MAX_ITER = 2
STATUS_NUMERICAL = -1


@dataclass
class SrbdParams:
    """Every constant of the path.  Mirrors ``srbdqp_config`` in include/srbdqp.h."""
    dt: float = 0.04
    mass: float = 34.13385728
    inertia: tuple = (8.20564e-2, 8.05015e-2, 0.32353e-2)
    mu: float = 0.8
    fz_min: float = 10.0
    fz_max: float = 1000.0
    gravity: float = -9.80665
    # cost: 0.5 * sum_k (x_{k+1}-xref_k)' Q (x_{k+1}-xref_k) + 0.5 * u_k' R u_k   (NOT IN TREE: own choice)
    q_diag: tuple = (300.0, 300.0, 150.0,  400.0, 400.0, 600.0,
                     1.0, 1.0, 1.0,  20.0, 20.0, 20.0,  0.0)
    r_diag: float = 1.0e-4
    # fixed variable scaling u = force_scale * u_hat (stands in for OSQP's Ruiz equilibration)
    force_scale: float = 100.0
    # ADMM (OSQP Algorithm 1) constants
    rho: float = 0.7
    # penalty of a stance contact's normal-force row (fz_min <= fz <= fz_max) relative to rho (round 3): the friction rows
    # and the bound row see different curvature, and one common penalty leaves the bound rows under-weighted
    # (auto_rho(), auto_rho_fz_scale())
    rho_fz_scale: float = 4.0
    rho_eq_scale: float = 1.0e3
    sigma: float = 1.0e-6
    alpha: float = 1.6
    eps_abs: float = 1.0e-6
    eps_rel: float = 1.0e-6
    max_iter: int = 250
    check_every: int = 5
    # one OSQP-style re-balancing of rho (presolved path only): a QP that has not converged after rho_restart_iter
    # iterations is re-factored with rho' = rho sqrt((r_prim/n_prim)/(r_dual/n_dual)), clipped to [rho/10, 5 rho] (wider clips stop further from the optimum on the same residual test), and
    # continues from its own (x, y) until max_iter iterations in total.  0 (or >= max_iter) = off = the default (100 solves
    # 99.9 % instead of 99.4 % of the config-2 QPs; on the GPU the second pass costs ~20 % of the batch throughput).
    rho_restart_iter: int = 0
    # ... and at most this many of them, one every rho_restart_iter iterations, each from the rho of the pass before it (the clip is per step); the last pass runs
    # to max_iter.  Round 3: three re-balancings 50 iterations apart solve 99.99 % of the N = 10 single-support QPs inside the 250-iteration cap, one solves
    # 99.78 %, none 99.22 % (the one-wave kernel does them in place; every other kernel re-balances once, as a second launch).
    rho_restart_count: int = 1
    # presolve: variables of swing contacts (force clamped to 0) are eliminated before the ADMM (kernel v2);
    # False = keep all 12N variables and clamp through the bounds (kernel v0/v1)
    eliminate_swing: bool = True

    def as_dict(self):
        return asdict(self)


# --------------------------------------------------------------------------------------
# a5: SRBD linearisation
# --------------------------------------------------------------------------------------
def rot_z(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def skew(r):
    return np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])


def linearise(p: SrbdParams, yaw: float, r: np.ndarray):
    """Discrete (forward-Euler) SRBD matrices for one horizon step.

    yaw : linearisation yaw psi_k.   r : (4,3) lever arms c_i - p_com.
    Returns A (13,13), B (13,12) with x_{k+1} = A x_k + B u_k.
    """
    Rz = rot_z(yaw)
    Ib_inv = np.diag(1.0 / np.asarray(p.inertia, dtype=np.float64))
    Iw_inv = Rz @ Ib_inv @ Rz.T
    A = np.eye(NX)
    A[0:3, 6:9] = p.dt * Rz.T          # euler-rate ~= Rz(psi)' omega
    A[3:6, 9:12] = p.dt * np.eye(3)    # p' = v
    A[11, 12] = p.dt                   # v_z' += g
    B = np.zeros((NX, NU))
    for i in range(NC):
        B[6:9, 3 * i:3 * i + 3] = p.dt * (Iw_inv @ skew(r[i]))
        B[9:12, 3 * i:3 * i + 3] = (p.dt / p.mass) * np.eye(3)
    return A, B


# --------------------------------------------------------------------------------------
# a6: horizon condensation
# --------------------------------------------------------------------------------------
def condense(p: SrbdParams, yaw_hor, foot_hor, pcom_hor):
    """A_qp (13N,13), B_qp (13N,12N):  X = A_qp x0 + B_qp U,  X = [x_1..x_N]."""
    N = len(yaw_hor)
    Ak, Bk = [], []
    for k in range(N):
        r = np.asarray(foot_hor[k], dtype=np.float64).reshape(NC, 3) - np.asarray(pcom_hor[k], dtype=np.float64)
        A, B = linearise(p, float(yaw_hor[k]), r)
        Ak.append(A)
        Bk.append(B)
    A_qp = np.zeros((NX * N, NX))
    B_qp = np.zeros((NX * N, NU * N))
    acc = np.eye(NX)
    for i in range(N):
        acc = Ak[i] @ acc
        A_qp[NX * i:NX * (i + 1)] = acc
        for j in range(i + 1):
            blk = Bk[j]
            for l in range(j + 1, i + 1):
                blk = Ak[l] @ blk
            B_qp[NX * i:NX * (i + 1), NU * j:NU * (j + 1)] = blk
    return A_qp, B_qp


# --------------------------------------------------------------------------------------
# a7/a8: Hessian, gradient, friction-cone rows
# --------------------------------------------------------------------------------------
def cone_block(mu):
    """5x3 constraint block of one contact point: rows (fx-mu fz, -fx-mu fz, fy-mu fz, -fy-mu fz, fz)."""
    return np.array([[1.0, 0.0, -mu], [-1.0, 0.0, -mu], [0.0, 1.0, -mu], [0.0, -1.0, -mu], [0.0, 0.0, 1.0]])


def build_qp(p: SrbdParams, x0, x_ref, foot_hor, contact_hor, pcom_hor=None):
    """Dense QP in SCALED variables u_hat = u / force_scale.

    min 0.5 uh' P uh + q' uh   s.t.  l <= A uh <= ub
    Returns dict(P,q,A,l,u,A_qp,B_qp).  Row 20k+5i+j = step k, contact i, cone row j.
    """
    x0 = np.asarray(x0, dtype=np.float64).reshape(NX)
    x_ref = np.asarray(x_ref, dtype=np.float64)
    N = x_ref.shape[0]
    foot_hor = np.asarray(foot_hor, dtype=np.float64).reshape(N, NU)
    contact_hor = np.asarray(contact_hor).reshape(N, NC)
    if pcom_hor is None:
        pcom_hor = x_ref[:, 3:6]
    pcom_hor = np.asarray(pcom_hor, dtype=np.float64).reshape(N, 3)
    A_qp, B_qp = condense(p, x_ref[:, 2], foot_hor, pcom_hor)
    s = p.force_scale
    Qd = np.tile(np.asarray(p.q_diag, dtype=np.float64), N)
    Bs = B_qp * s
    P = Bs.T @ (Qd[:, None] * Bs) + (p.r_diag * s * s) * np.eye(NU * N)
    P = 0.5 * (P + P.T)
    q = Bs.T @ (Qd * (A_qp @ x0 - x_ref.reshape(-1)))
    C = cone_block(p.mu)
    m = ROWS_PER_CONTACT * NC * N
    A = np.zeros((m, NU * N))
    l = np.full(m, -INF)
    u = np.zeros(m)
    for k in range(N):
        for i in range(NC):
            r0 = 20 * k + 5 * i
            c0 = NU * k + 3 * i
            A[r0:r0 + 5, c0:c0 + 3] = C
            c = 1.0 if contact_hor[k, i] else 0.0
            l[r0 + 4] = c * p.fz_min / s
            u[r0 + 4] = c * p.fz_max / s
    return dict(P=P, q=q, A=A, l=l, u=u, A_qp=A_qp, B_qp=B_qp)


def closed_form_hessian_gradient(p: SrbdParams, x0, x_ref, foot_hor, contact_hor, pcom_hor=None):
    """The assembly the compact HIP kernel uses, restated in NumPy for the tests: P and q of the PRESOLVED QP (stance
    contacts only, ordered by step then contact index) without ever forming B_qp.

    With T_k = R_z(psi_k)', C_k = sum_{l<=k} T_l and J_j = I_w^-1 [r]x per contact, block (i, j) of B_qp is
    theta: dt^2 (C_i - C_j) J_j, p: (i-j) dt^2/m I, omega: dt J_j, v: dt/m I (gravity row zero), so for contacts e
    (step j) and e' (step m >= j)
        (Bs'Q Bs)[e, e'] = s^2 ( J_e' M(j,m) J_e' + diag(w_p dt^4/m^2 Sp(j,m) + w_v dt^2/m^2 (N-m)) ),
        M(j,m) = dt^4 (T2(m) + (C_m - C_j)' W_th T1(m)) + (N-m) dt^2 W_om,
        T1(m) = sum_{i>=m} (C_i - C_m),  T2(m) = sum_{i>=m} (C_i - C_m)' W_th (C_i - C_m),  Sp = sum_{i>=m} (i-j)(i-m),
    and Bs'Q v is a suffix sum per step followed by one 3-vector product per contact.
    Returns (P (n_eff, n_eff), q (n_eff,), vi) with vi = indices of the kept variables in the full 12N vector."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(NX)
    x_ref = np.asarray(x_ref, dtype=np.float64)
    N = x_ref.shape[0]
    foot_hor = np.asarray(foot_hor, dtype=np.float64).reshape(N, NC, 3)
    contact_hor = np.asarray(contact_hor).reshape(N, NC) != 0
    pcom = x_ref[:, 3:6] if pcom_hor is None else np.asarray(pcom_hor, dtype=np.float64).reshape(N, 3)
    dt, s, inv_m = p.dt, p.force_scale, 1.0 / p.mass
    w = np.asarray(p.q_diag, dtype=np.float64)
    W_th, W_p, W_om, W_v = np.diag(w[0:3]), w[3:6], np.diag(w[6:9]), w[9:12]
    T = [rot_z(float(x_ref[k, 2])).T for k in range(N)]
    C = np.cumsum(np.array(T), axis=0)
    Ib_inv = np.diag(1.0 / np.asarray(p.inertia, dtype=np.float64))
    contacts = [(k, i) for k in range(N) for i in range(NC) if contact_hor[k, i]]
    J = []
    for k, i in contacts:
        Rz = T[k].T
        J.append(Rz @ Ib_inv @ Rz.T @ skew(foot_hor[k, i] - pcom[k]))
    T1 = [sum((C[i] - C[m] for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]
    T2 = [sum(((C[i] - C[m]).T @ W_th @ (C[i] - C[m]) for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]
    na = len(contacts)
    P = np.zeros((3 * na, 3 * na))
    for e in range(na):
        for e2 in range(e, na):
            j, m = contacts[e][0], contacts[e2][0]
            M = dt ** 4 * (T2[m] + (C[m] - C[j]).T @ W_th @ T1[m]) + (N - m) * dt ** 2 * W_om
            sp = sum((i - j) * (i - m) for i in range(m, N))
            blk = J[e].T @ M @ J[e2] + np.diag(W_p * dt ** 4 * inv_m ** 2 * sp + W_v * dt ** 2 * inv_m ** 2 * (N - m))
            P[3 * e:3 * e + 3, 3 * e2:3 * e2 + 3] = s * s * blk
            P[3 * e2:3 * e2 + 3, 3 * e:3 * e + 3] = s * s * blk.T
    P += p.r_diag * s * s * np.eye(3 * na)
    # gradient: e_i = Q (free response - x_ref) per step, then the suffix sums
    xf = x0.copy()
    err = np.zeros((N, 12))
    Ak = [linearise(p, float(x_ref[k, 2]), np.zeros((NC, 3)))[0] for k in range(N)]
    for i in range(N):
        xf = Ak[i] @ xf
        err[i] = w[:12] * (xf[:12] - x_ref[i, :12])
    q = np.zeros(3 * na)
    for e, (j, _) in enumerate(contacts):
        g = sum((dt ** 2 * (C[i] - C[j]).T @ err[i, 0:3] + dt * err[i, 6:9] for i in range(j, N)), np.zeros(3))
        hp = sum(((i - j) * err[i, 3:6] for i in range(j, N)), np.zeros(3))
        hv = sum((err[i, 9:12] for i in range(j, N)), np.zeros(3))
        q[3 * e:3 * e + 3] = s * (J[e].T @ g + dt ** 2 * inv_m * hp + dt * inv_m * hv)
    vi = np.array([NU * k + 3 * i + a for k, i in contacts for a in range(3)], dtype=int)
    return P, q, vi


def closed_form_hessian_rank6(p: SrbdParams, x_ref, foot_hor, contact_hor, pcom_hor=None):
    """The form of closed_form_hessian_gradient()'s P that the HIP kernels assemble since round 4 (srbdqp_common.hpp: de_tables, kasm_rows,
    kasm_tile), restated in NumPy for the tests.  M(j, m) = D_m - C_j' E_m with ONE pair of 3 x 3 matrices per step,
        D_m = dt^4 (T2(m) + C_m' W_th T1(m)) + (N - m) dt^2 W_om,     E_m = dt^4 W_th T1(m),
    and Sp(j, m) = alpha_m + (m - j) beta_m (alpha_m = sum_{i>=m} (i - m)^2, beta_m = sum_{i>=m} (i - m)), so for variables r = (contact e of
    step j, axis x) and c = (contact e' of step m >= j, axis y)
        P[r, c] = a_r . b_c + [x == y] (f0_c - j f1_c),
        a_r = [J_e[:, x]; -C_j J_e[:, x]],   b_c = s^2 [D_m J_e'[:, y]; E_m J_e'[:, y]],
        f0_c = s^2 (fa_y (alpha_m + m beta_m) + fb_y (N - m)),   f1_c = s^2 fa_y beta_m,   fa = W_p dt^4 / mass^2,  fb = W_v dt^2 / mass^2:
    one table row (a, b, f0, f1, j) per presolved variable, an entry = 6 + 2 multiply-adds.  Only the upper triangle is formed that way (the
    kernels never read the rest); it is mirrored here.  Returns P (n_eff, n_eff) without the R s^2 term's partner terms changed: the same matrix
    as closed_form_hessian_gradient()[0]."""
    x_ref = np.asarray(x_ref, dtype=np.float64)
    N = x_ref.shape[0]
    foot_hor = np.asarray(foot_hor, dtype=np.float64).reshape(N, NC, 3)
    contact_hor = np.asarray(contact_hor).reshape(N, NC) != 0
    pcom = x_ref[:, 3:6] if pcom_hor is None else np.asarray(pcom_hor, dtype=np.float64).reshape(N, 3)
    dt, s, inv_m = p.dt, p.force_scale, 1.0 / p.mass
    w = np.asarray(p.q_diag, dtype=np.float64)
    W_th, W_p, W_om, W_v = np.diag(w[0:3]), w[3:6], np.diag(w[6:9]), w[9:12]
    T = [rot_z(float(x_ref[k, 2])).T for k in range(N)]
    C = np.cumsum(np.array(T), axis=0)
    Ib_inv = np.diag(1.0 / np.asarray(p.inertia, dtype=np.float64))
    contacts = [(k, i) for k in range(N) for i in range(NC) if contact_hor[k, i]]
    T1 = [sum((C[i] - C[m] for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]
    T2 = [sum(((C[i] - C[m]).T @ W_th @ (C[i] - C[m]) for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]
    D = [dt ** 4 * (T2[m] + C[m].T @ W_th @ T1[m]) + (N - m) * dt ** 2 * W_om for m in range(N)]
    E = [dt ** 4 * W_th @ T1[m] for m in range(N)]
    fa, fb = W_p * dt ** 4 * inv_m ** 2, W_v * dt ** 2 * inv_m ** 2
    rows = []                                           # (a, b, f0, f1, j, axis) per presolved variable
    for k, i in contacts:
        Rz = T[k].T
        Jk = Rz @ Ib_inv @ Rz.T @ skew(foot_hor[k, i] - pcom[k])
        L = N - k
        alpha, beta = (L - 1) * L * (2 * L - 1) // 6, (L - 1) * L // 2
        for ax in range(3):
            a = np.concatenate([Jk[:, ax], -(C[k] @ Jk[:, ax])])
            b = s * s * np.concatenate([D[k] @ Jk[:, ax], E[k] @ Jk[:, ax]])
            rows.append((a, b, s * s * (fa[ax] * (alpha + k * beta) + fb[ax] * L), s * s * fa[ax] * beta, k, ax))
    n = len(rows)
    P = np.zeros((n, n))
    for r in range(n):
        for c in range(r, n):
            v = rows[r][0] @ rows[c][1]
            if rows[r][5] == rows[c][5]:
                v += rows[c][2] - rows[r][4] * rows[c][3]
            P[r, c] = P[c, r] = v
    return P + p.r_diag * s * s * np.eye(n)


def wrench_reduce(p: SrbdParams, x_ref, foot_hor, contact_hor, pcom_hor=None, rho=None):
    """The presolve of the general (any contact pattern) HIP kernel, restated in NumPy for the tests: the reduced-KKT
    matrix K = P + sigma I + A' rho A of the stance-contact QP, inverted through the rank structure of the SRBD.

    A step's 3c stance-force variables act on the body only through the 6-vector wrench g = W u, W = [J_e ...; I I ...]
    (angular acceleration I_w^-1 sum r x f, total force).  So the condensed Hessian is P = Om' S_w Om + R s^2 I with
    Om = blockdiag(W_j) and the 6N x 6N wrench-space matrix S_w(j, m) = s^2 blockdiag(M(j, m), diag(d(j, m)))
    (closed_form_hessian_gradient()'s tables), and A' rho A is diagonal per contact (rho diag(2, 2, 4 mu^2 + rho_fz_scale)), hence
    K = D + Y' S Y with D diagonal.  Per step the kernel keeps g_j = 6 wrench coordinates if the step has >= 3 stance
    contacts (Y_j = W_j) and the 3c force variables themselves otherwise (Y_j = I), and applies
        K^-1 = Bd + V' T^-1 V,   T = S + E^-1,  E = Y D^-1 Y' (block diagonal),  V = E^-1 Y D^-1,
        Bd = D^-1 - D^-1 Y' V   (block diagonal; zero on the identity steps)
    (Woodbury twice).  Only T (n_g x n_g, n_g <= 6N) is factored: double support at N = 20 is a 120 x 120 problem
    instead of 240 x 240.  Returns dict(T, V, Bd, D, gsz, goff, n_g, vi, contacts)."""
    x_ref = np.asarray(x_ref, dtype=np.float64)
    N = x_ref.shape[0]
    foot_hor = np.asarray(foot_hor, dtype=np.float64).reshape(N, NC, 3)
    contact_hor = np.asarray(contact_hor).reshape(N, NC) != 0
    pcom = x_ref[:, 3:6] if pcom_hor is None else np.asarray(pcom_hor, dtype=np.float64).reshape(N, 3)
    rho = p.rho if rho is None else float(rho)
    dt, s, inv_m = p.dt, p.force_scale, 1.0 / p.mass
    w = np.asarray(p.q_diag, dtype=np.float64)
    W_th, W_p, W_om, W_v = np.diag(w[0:3]), w[3:6], np.diag(w[6:9]), w[9:12]
    Tm = [rot_z(float(x_ref[k, 2])).T for k in range(N)]
    C = np.cumsum(np.array(Tm), axis=0)
    Ib_inv = np.diag(1.0 / np.asarray(p.inertia, dtype=np.float64))
    T1 = [sum((C[i] - C[m] for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]
    T2 = [sum(((C[i] - C[m]).T @ W_th @ (C[i] - C[m]) for i in range(m, N)), np.zeros((3, 3))) for m in range(N)]

    def S_w(j, m):   # 6 x 6 wrench-space block, j <= m
        M = dt ** 4 * (T2[m] + (C[m] - C[j]).T @ W_th @ T1[m]) + (N - m) * dt ** 2 * W_om
        sp = sum((i - j) * (i - m) for i in range(m, N))
        out = np.zeros((6, 6))
        out[0:3, 0:3] = M
        out[3:6, 3:6] = np.diag(W_p * dt ** 4 * inv_m ** 2 * sp + W_v * dt ** 2 * inv_m ** 2 * (N - m))
        return s * s * out

    contacts = [(k, i) for k in range(N) for i in range(NC) if contact_hor[k, i]]
    nu_ = 3 * len(contacts)
    vi = np.array([NU * k + 3 * i + a for k, i in contacts for a in range(3)], dtype=int)
    dxy = p.r_diag * s * s + p.sigma + 2.0 * rho
    dz = p.r_diag * s * s + p.sigma + (4.0 * p.mu ** 2 + p.rho_fz_scale) * rho
    D = np.tile(np.array([dxy, dxy, dz]), len(contacts))
    Wj, uoff, csz = [], [], []
    off = 0
    for k in range(N):
        cs = [i for i in range(NC) if contact_hor[k, i]]
        Rz = Tm[k].T
        blk = np.zeros((6, 3 * len(cs)))
        for q, i in enumerate(cs):
            blk[0:3, 3 * q:3 * q + 3] = Rz @ Ib_inv @ Rz.T @ skew(foot_hor[k, i] - pcom[k])
            blk[3:6, 3 * q:3 * q + 3] = np.eye(3)
        Wj.append(blk); uoff.append(off); csz.append(len(cs))
        off += 3 * len(cs)
    gsz = np.array([6 if c >= 3 else 3 * c for c in csz], dtype=int)
    goff = np.concatenate([[0], np.cumsum(gsz)]).astype(int)
    n_g = int(goff[-1])
    Z = [np.eye(6) if csz[k] >= 3 else Wj[k] for k in range(N)]           # wrench coordinates -> g coordinates
    Y = [Wj[k] if csz[k] >= 3 else np.eye(3 * csz[k]) for k in range(N)]   # force variables -> g coordinates
    S = np.zeros((n_g, n_g))
    for j in range(N):
        for m in range(j, N):
            if gsz[j] == 0 or gsz[m] == 0:
                continue
            blk = Z[j].T @ S_w(j, m) @ Z[m]
            S[goff[j]:goff[j + 1], goff[m]:goff[m + 1]] = blk
            if m > j:
                S[goff[m]:goff[m + 1], goff[j]:goff[j + 1]] = blk.T
    S = 0.5 * (S + S.T)
    T = S.copy()
    V = np.zeros((n_g, nu_))
    Bd = np.zeros((nu_, nu_))
    for k in range(N):
        if gsz[k] == 0:
            continue
        us = slice(uoff[k], uoff[k] + 3 * csz[k]); gs = slice(goff[k], goff[k + 1])
        Dk = D[us]
        E = (Y[k] / Dk) @ Y[k].T
        Einv = np.linalg.inv(E)
        Einv = 0.5 * (Einv + Einv.T)
        T[gs, gs] += Einv
        V[gs, us] = Einv @ (Y[k] / Dk)
        Bd[us, us] = np.diag(1.0 / Dk) - (Y[k] / Dk).T @ V[gs, us]
    return dict(T=T, V=V, Bd=Bd, D=D, S=S, gsz=gsz, goff=goff, n_g=n_g, vi=vi, contacts=contacts, csz=np.array(csz))


def wrench_kinv_op(wr):
    """x~ = K^-1 rhs as the general kernel applies it: Bd rhs + V' (T^-1 (V rhs)), T^-1 explicit (W'W of its Cholesky factor)."""
    Lc = np.linalg.cholesky(wr["T"])
    Linv = _tri_inv(Lc)
    Tinv = Linv.T @ Linv
    V, Bd = wr["V"], wr["Bd"]
    return lambda rhs: Bd @ rhs + V.T @ (Tinv @ (V @ rhs))


def auto_rho(N: int) -> float:
    """The engine's default ADMM penalty of the friction rows (srbdqp_config.rho = 0): 0.7 at every horizon, together with
    auto_rho_fz_scale() = 4 on the normal-force rows.  Round-3 scan (numpy ADMM against the exact active-set optimum, horizons
    8 ... 24 x {single, double, mixed} support, rho in {0.5 ... 2} x rho_fz_scale in {1 ... 6}): with ONE common penalty the
    fastest rho grows with the horizon (rounds 1-2 used 1 / 1.5 / 2) and above 2 the residual test is passed further and
    further from the optimum; what the long horizons were asking for is weight on the two-sided normal-force rows, not on the
    friction rows.  (0.7, 4) needs 14-26 % fewer iterations at N = 10 and 20-30 % fewer at N = 20 / 24 than the old rule, with
    a far shorter tail (N = 10: p99 48-63 iterations against 88-100) and forces within 3e-3 N (p99) of the exact optimum at
    every horizon (old rule: 6e-4 N at N = 10, 2.4e-2 N at N = 20)."""
    return 0.7


def auto_rho_fz_scale(N: int) -> float:
    """The engine's default penalty ratio of the normal-force rows (srbdqp_config.rho_fz_scale = 0), see auto_rho(): 4 at every
    horizon.  (6 at N >= 16 needs another 10 - 20 % fewer iterations, but stops 1e-2 - 3e-2 N (p99) from the exact optimum
    instead of 2e-3 N: measured, not taken.)"""
    return 4.0


def default_restart(N: int, one_wave: bool = False):
    """(rho_restart_iter, rho_restart_count) the engine picks by default (srbdqp.hip restart_iter_of), by horizon: N <= 10: 55 x 2, N = 12: 70 x 2, N = 16: 80 x 3,
    N = 20: 125 x 1, N = 24: 100 x 2 (each re-balancing from the rho of the pass before it; the cap max_iter is on the total).  The same for every kernel and batch
    size since round 4 (one_wave is ignored: round 3 applied the N <= 10 rule only where the one-wave kernel ran a call of 4096 QPs or more, and one re-balancing
    after 100 / 125 iterations at N > 10)."""
    if N <= 10:
        return 55, 2
    return {12: (70, 2), 16: (80, 3), 24: (100, 2)}.get(N, (125, 1))


def default_params(N: int, **kw) -> "SrbdParams":
    """SrbdParams as the engine runs horizon N when NOTHING is configured: params_for(N) plus the automatic rho restart."""
    it, cnt = default_restart(N)
    kw.setdefault("rho_restart_iter", it)
    kw.setdefault("rho_restart_count", cnt)
    return params_for(N, **kw)


def params_for(N: int, **kw) -> SrbdParams:
    """SrbdParams as the engine runs horizon N by default (rho = auto_rho(N), rho_fz_scale = auto_rho_fz_scale(N) unless given)."""
    kw.setdefault("rho", auto_rho(N))
    kw.setdefault("rho_fz_scale", auto_rho_fz_scale(N))
    return SrbdParams(**kw)


def rho_vector(p: SrbdParams, l, u):
    """Per-row ADMM penalty: rho for the friction rows, rho * rho_fz_scale for the two-sided normal-force rows of stance
    contacts, rho * rho_eq_scale for equalities (OSQP's rule; the clamped rows of swing contacts when they are kept)."""
    rho = np.full(l.shape, p.rho)
    rho[l > -INF / 2] = p.rho * p.rho_fz_scale
    rho[(u - l) < 1e-12] = p.rho * p.rho_eq_scale
    return rho


# --------------------------------------------------------------------------------------
# a9: ADMM (OSQP Algorithm 1, reduced-KKT form)
# --------------------------------------------------------------------------------------
def admm_solve(p: SrbdParams, P, q, A, l, u, x_init=None, y_init=None, dtype=np.float64, trace=None, info=None, kinv_op=None):
    """The algorithm the HIP kernel runs, in the same order of operations.

    K = P + sigma I + A' diag(rho) A is factored once (Cholesky); every iteration applies the
    inverse.  P x is tracked by recursion (no matvec with P inside the loop).
    kinv_op: optional callable applying K^-1 (the general kernel's wrench-space form, wrench_kinv_op()); default = the
    explicit dense inverse.
    Returns (x, z, y, iters, status).
    """
    P = P.astype(dtype); q = q.astype(dtype); A = A.astype(dtype)
    l = l.astype(dtype); u = u.astype(dtype)
    n, m = P.shape[0], A.shape[0]
    rho = rho_vector(p, l, u).astype(dtype)
    sigma, alpha = dtype(p.sigma), dtype(p.alpha)
    if kinv_op is None:
        K = P + sigma * np.eye(n, dtype=dtype) + (A.T * rho) @ A
        Lc = np.linalg.cholesky(K.astype(np.float64)).astype(dtype) if dtype == np.float64 else _chol(K)
        Linv = _tri_inv(Lc)
        Kinv = (Linv.T @ Linv).astype(dtype)
        kinv_op = lambda r: Kinv @ r
    x = np.zeros(n, dtype) if x_init is None else np.asarray(x_init, dtype).copy()
    y = np.zeros(m, dtype) if y_init is None else np.asarray(y_init, dtype).copy()
    z = np.clip(A @ x, l, u)
    Px = P @ x
    qn = np.max(np.abs(q))
    status, iters = STATUS_MAX_ITER, p.max_iter
    e_prim_last, vote_ok = np.inf, True
    for k in range(1, p.max_iter + 1):
        rhs = sigma * x - q + A.T @ (rho * z - y)
        xt = kinv_op(rhs)
        zt = A @ xt
        # P xt from the KKT identity: (P + sigma I) xt + A'(rho (zt - z) + y) = sigma x - q
        Pxt = sigma * (x - xt) - q - A.T @ (rho * (zt - z) + y)
        x = alpha * xt + (1 - alpha) * x
        Px = alpha * Pxt + (1 - alpha) * Px
        zh = alpha * zt + (1 - alpha) * z
        zn = np.clip(zh + y / rho, l, u)
        y = y + rho * (zh - zn)
        z = zn
        # Residual pre-test (kernels: one ballot, no reduction): a full convergence check at iteration k is only made
        # if at iteration k-1 every row satisfied |Ax - z| <= e_prim of the LAST full check (inf before the first one).
        if (k + 1) % p.check_every == 0:
            vote_ok = bool(np.max(np.abs(A @ x - z)) <= e_prim_last)
        if k % p.check_every == 0 or k == p.max_iter:
            if not (vote_ok or k == p.max_iter):
                continue
            Ax = A @ x
            Aty = A.T @ y
            # the four maxima (and |q|) are rounded to float32 before the comparison: the kernels reduce them across
            # the workgroup in fp32 (one DPP v_max_f32 per step); rounding is monotone, so it commutes with max
            f32 = lambda v: float(np.float32(v))
            r_prim = f32(np.max(np.abs(Ax - z)))
            r_dual = f32(np.max(np.abs(Px + q + Aty)))
            e_prim = p.eps_abs + p.eps_rel * f32(max(np.max(np.abs(Ax)), np.max(np.abs(z))))
            e_dual = p.eps_abs + p.eps_rel * max(f32(max(np.max(np.abs(Px)), np.max(np.abs(Aty)))), f32(qn))
            e_prim_last = e_prim
            if info is not None:   # the fp32 maxima of the last full check (what the kernels hand to the restart rule)
                info.update(r_prim=r_prim, n_prim=f32(max(np.max(np.abs(Ax)), np.max(np.abs(z)))), r_dual=r_dual,
                            n_dual=max(f32(max(np.max(np.abs(Px)), np.max(np.abs(Aty)))), f32(qn)))
            if trace is not None:
                trace.append((k, float(r_prim), float(r_dual)))
            if not np.isfinite(r_prim + r_dual):
                status, iters = STATUS_NUMERICAL, k
                break
            if r_prim <= e_prim and r_dual <= e_dual:
                status, iters = STATUS_SOLVED, k
                break
    return x, z, y, iters, status


def admm_solve_split(p: SrbdParams, P, q, A, l, u, wr, x_init=None, y_init=None, dtype=np.float32, info=None, tile_dtype=np.float64):
    """The recursions of the general HIP kernel (srbdqp_wrench.hpp), in `dtype` (float32 = its fp32 instantiation).

    Same algorithm as admm_solve(), rearranged so that the large gradient q never enters the iteration:
      x~ = x_q + K^-1 (sigma x + A'(rho z - y)),  x_q = -K^-1 q computed ONCE in float64  (K^-1 = Bd + V' T^-1 V, `wr`
      from wrench_reduce(); T^-1, V, Bd are rounded to `dtype` for the loop), and the dual residual is carried as
      c = P x + q with c~ = sigma (x - x~) - A' nu.  In exact arithmetic the iterates equal admm_solve()'s.
    The four maxima of a check are float32 in the kernels whatever `dtype` is.  Returns (x, z, y, iters, status)."""
    n, m = P.shape[0], A.shape[0]
    rho64 = rho_vector(p, l, u)
    T = wr["T"]
    if tile_dtype == np.float32:   # the fp32-tile instantiation: T rounded to float32, factored and inverted in float32
        Linv = _tri_inv(_chol(T.astype(np.float32)))
        Tinv64 = (Linv.T @ Linv).astype(np.float64)
    else:
        Linv = _tri_inv(np.linalg.cholesky(T))
        Tinv64 = Linv.T @ Linv
    V64, Bd64 = wr["V"], wr["Bd"]
    kinv64 = lambda r: Bd64 @ r + V64.T @ (Tinv64 @ (V64 @ r))
    xq = -kinv64(q)
    if tile_dtype == np.float32:   # one step of iterative refinement with the float64 residual K x_q + q
        K = P + p.sigma * np.eye(n) + (A.T * rho64) @ A
        xq = xq - kinv64(K @ xq + q)
    xq = xq.astype(dtype)
    Tinv, V, Bd = Tinv64.astype(dtype), V64.astype(dtype), Bd64.astype(dtype)
    A = A.astype(dtype); rho = rho64.astype(dtype)
    lo = np.maximum(l, -INF).astype(dtype); hi = np.minimum(u, INF).astype(dtype)
    sigma, alpha = dtype(p.sigma), dtype(p.alpha)
    oma = dtype(1.0 - p.alpha)
    x = np.zeros(n, dtype) if x_init is None else np.asarray(x_init, dtype).copy()
    y = np.zeros(m, dtype) if y_init is None else np.asarray(y_init, dtype).copy()
    z = np.clip(A @ x, lo, hi)
    c = (P @ x.astype(np.float64) + q).astype(dtype)
    qd = q.astype(dtype)
    f32 = lambda v: float(np.float32(v))
    qn = f32(np.max(np.abs(q)))
    status, iters = STATUS_MAX_ITER, p.max_iter
    e_prim_last, vote_ok = np.inf, True
    for k in range(1, p.max_iter + 1):
        w = sigma * x + A.T @ (rho * z - y)
        xt = xq + (Bd @ w + V.T @ (Tinv @ (V @ w)))
        zt = A @ xt
        ct = sigma * (x - xt) - A.T @ (rho * (zt - z) + y)
        x = alpha * xt + oma * x
        c = alpha * ct + oma * c
        zh = alpha * zt + oma * z
        zn = np.clip(zh + y / rho, lo, hi)
        y = y + rho * (zh - zn)
        z = zn
        if (k + 1) % p.check_every == 0:
            vote_ok = bool(np.max(np.abs(A @ x - z)) <= e_prim_last)
        if k % p.check_every == 0 or k == p.max_iter:
            if not (vote_ok or k == p.max_iter):
                continue
            Ax = A @ x
            Aty = A.T @ y
            r_prim = f32(np.max(np.abs(Ax - z)))
            r_dual = f32(np.max(np.abs(c + Aty)))
            n_prim = f32(max(np.max(np.abs(Ax)), np.max(np.abs(z))))
            n_dual = max(f32(max(np.max(np.abs(c - qd)), np.max(np.abs(Aty)))), qn)
            e_prim = p.eps_abs + p.eps_rel * n_prim
            e_dual = p.eps_abs + p.eps_rel * n_dual
            e_prim_last = e_prim
            if info is not None:
                info.update(r_prim=r_prim, n_prim=n_prim, r_dual=r_dual, n_dual=n_dual)
            if not np.isfinite(r_prim + r_dual):
                status, iters = STATUS_NUMERICAL, k
                break
            if r_prim <= e_prim and r_dual <= e_dual:
                status, iters = STATUS_SOLVED, k
                break
    return x.astype(np.float64), z.astype(np.float64), y.astype(np.float64), iters, status


def fp32_tiles_ok(contact_hor) -> bool:
    """The general kernel's rule for factoring T in float32 tiles (fp32 calls, batches >= 512): every step has 0 or >= 3
    stance contacts, i.e. every g coordinate is a wrench coordinate (cond(T) ~ 5e4; a step kept in force variables
    carries the conditioning of K, 1e8, into T)."""
    c = (np.asarray(contact_hor).reshape(-1, NC) != 0).sum(axis=1)
    return bool(np.all((c == 0) | (c >= 3)))


def update_split(p: SrbdParams, x0, x_ref, foot_hor, contact_hor, pcom_hor=None, warm=None, dtype=np.float32, tile_dtype=np.float64):
    """Oracle twin of the general kernel's fp32 (or fp64) path: like update(), through wrench_reduce() +
    admm_solve_split().  With dtype=float32 the inputs are first rounded to float32, as the _f32 entry points see them.
    tile_dtype: np.float64, np.float32 or "auto" (float32 where fp32_tiles_ok(), as the engine does for batches >= 512);
    the pass after a rho restart always factors in float64 tiles."""
    if isinstance(tile_dtype, str):
        tile_dtype = np.float32 if (dtype == np.float32 and fp32_tiles_ok(contact_hor)) else np.float64
    if dtype == np.float32:
        x0, x_ref, foot_hor = (np.asarray(v, np.float32).astype(np.float64) for v in (x0, x_ref, foot_hor))
        pcom_hor = None if pcom_hor is None else np.asarray(pcom_hor, np.float32).astype(np.float64)
    qp = build_qp(p, x0, x_ref, foot_hor, contact_hor, pcom_hor)
    n, m = qp["P"].shape[0], qp["A"].shape[0]
    red, vi, ri = presolve(qp, contact_hor)
    uh = np.zeros(n); y = np.zeros(m)
    if len(vi) == 0:
        iters, status = 0, STATUS_SOLVED
    else:
        wr = wrench_reduce(p, x_ref, foot_hor, contact_hor, pcom_hor)
        xi, yi = (None, None) if warm is None else (np.asarray(warm[0])[vi], np.asarray(warm[1])[ri])
        args = (red["P"], red["q"], red["A"], red["l"], red["u"])
        if 0 < p.rho_restart_iter < p.max_iter:   # the general kernel's rho restart passes (same rule as solve_with_restart: up to rho_restart_count
            # re-balancings, each from the rho of the pass before it, the cap on the total; every pass after the first factors in float64 tiles)
            nre = max(int(p.rho_restart_count), 1)
            pc, w, td, xs_, ys_, iters = p, wr, tile_dtype, xi, yi, 0
            for k in range(nre + 1):
                left = p.max_iter - iters
                cap = p.rho_restart_iter if (k < nre and p.rho_restart_iter < left) else left
                info = {}
                xr_, _, yr_, it, status = admm_solve_split(replace(pc, max_iter=cap), *args, w, xs_, ys_, dtype=dtype, info=info, tile_dtype=td)
                iters += it
                if status != STATUS_MAX_ITER or cap == left:
                    break
                pc = replace(pc, rho=restart_rho(pc, info))
                w = wrench_reduce(pc, x_ref, foot_hor, contact_hor, pcom_hor)
                xs_, ys_, td = xr_, yr_, np.float64
        else:
            xr_, _, yr_, iters, status = admm_solve_split(p, *args, wr, xi, yi, dtype=dtype, tile_dtype=tile_dtype)
        uh[vi] = xr_
        y[ri] = yr_
    N = np.asarray(x_ref).shape[0]
    return dict(u=(uh * p.force_scale).reshape(N, NU), x=rollout(qp, x0, uh, p.force_scale), iters=iters, status=status,
                u_hat=uh, y=y, qp=qp)


def _chol(K):
    """Unblocked lower Cholesky in the matrix' own dtype (used for the fp32 oracle)."""
    K = K.copy()
    n = K.shape[0]
    for j in range(n):
        K[j, j] = np.sqrt(K[j, j] - np.dot(K[j, :j], K[j, :j]))
        K[j + 1:, j] = (K[j + 1:, j] - K[j + 1:, :j] @ K[j, :j]) / K[j, j]
    return np.tril(K)


def _tri_inv(L):
    n = L.shape[0]
    W = np.zeros_like(L)
    for j in range(n):
        W[j, j] = 1.0 / L[j, j]
        for i in range(j + 1, n):
            W[i, j] = -np.dot(L[i, j:i], W[j:i, j]) / L[i, i]
    return W


# --------------------------------------------------------------------------------------
# a10: update() = assemble + solve + rollout
# --------------------------------------------------------------------------------------
def rollout(qp, x0, u_hat, force_scale):
    """x horizon (N+1,13): row 0 = x0, row k = predicted x_k."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(NX)
    X = qp["A_qp"] @ x0 + (qp["B_qp"] * force_scale) @ u_hat
    return np.vstack([x0[None, :], X.reshape(-1, NX)])


def presolve(qp, contact_hor):
    """Fixed-variable elimination: drop the 3 variables and 5 rows of every swing contact (their force is 0).
    Returns (reduced qp dict, var_index, row_index).  The reduced problem has the same optimum on the kept variables."""
    on = np.asarray(contact_hor).reshape(-1).astype(bool)
    vi = np.where(np.repeat(on, 3))[0]
    ri = np.where(np.repeat(on, ROWS_PER_CONTACT))[0]
    red = dict(P=qp["P"][np.ix_(vi, vi)], q=qp["q"][vi], A=qp["A"][np.ix_(ri, vi)], l=qp["l"][ri], u=qp["u"][ri])
    return red, vi, ri


def restart_rho(p: SrbdParams, info):
    """OSQP's rho update from the fp32 maxima of the last check, in double: rho sqrt((r_p/n_p)/(r_d/n_d)), clipped (p.rho = the rho of the pass that ended)."""
    num = float(info["r_prim"]) / max(float(info["n_prim"]), 1e-30)
    den = float(info["r_dual"]) / max(float(info["n_dual"]), 1e-30)
    if not (num > 0.0 and den > 0.0 and np.isfinite(num) and np.isfinite(den)):
        return float(p.rho)
    return float(min(max(p.rho * np.sqrt(num / den), p.rho * 0.1), p.rho * 5.0))


def solve_with_restart(p: SrbdParams, P, q, A, l, u, x_init=None, y_init=None, dtype=np.float64):
    """admm_solve() plus the single rho re-balancing of the presolved path (what the compact / split kernels run).
    Returns (x, z, y, iters, status); iters counts both passes."""
    if p.rho_restart_iter <= 0 or p.rho_restart_iter >= p.max_iter:   # off, or the cap comes first
        return admm_solve(p, P, q, A, l, u, x_init, y_init, dtype=dtype)
    s = dtype(p.force_scale)
    x, y, done, pc = x_init, y_init, 0, p
    for k in range(max(int(p.rho_restart_count), 1) + 1):
        left = p.max_iter - done                                   # the cap is on the total
        cap = p.rho_restart_iter if (k < max(int(p.rho_restart_count), 1) and p.rho_restart_iter < left) else left
        info = {}
        x, z, y, it, st = admm_solve(replace(pc, max_iter=cap), P, q, A, l, u, x, y, dtype=dtype, info=info)
        done += it
        if st != STATUS_MAX_ITER or cap == left:
            break
        x = (x * s) / s                                            # the next pass re-reads the forces in newtons
        pc = replace(pc, rho=restart_rho(pc, info))
    return x, z, y, done, st


def update(p: SrbdParams, x0, x_ref, foot_hor, contact_hor, pcom_hor=None, warm=None, dtype=np.float64):
    """Oracle twin of MPC.update: returns dict(u (N,12) in newtons, x (N+1,13), iters, status, ...)."""
    qp = build_qp(p, x0, x_ref, foot_hor, contact_hor, pcom_hor)
    n, m = qp["P"].shape[0], qp["A"].shape[0]
    xi, yi = (None, None) if warm is None else warm
    if p.eliminate_swing:
        red, vi, ri = presolve(qp, contact_hor)
        uh = np.zeros(n); y = np.zeros(m)
        if len(vi) == 0:
            iters, status = 0, STATUS_SOLVED
        else:
            xr_, _, yr_, iters, status = solve_with_restart(p, red["P"], red["q"], red["A"], red["l"], red["u"],
                                                            None if xi is None else np.asarray(xi)[vi],
                                                            None if yi is None else np.asarray(yi)[ri], dtype=dtype)
            uh[vi] = xr_
            y[ri] = yr_
    else:
        uh, _, y, iters, status = admm_solve(p, qp["P"], qp["q"], qp["A"], qp["l"], qp["u"], xi, yi, dtype=dtype)
    N = np.asarray(x_ref).shape[0]
    uh64 = np.asarray(uh, dtype=np.float64)
    return dict(u=(uh64 * p.force_scale).reshape(N, NU), x=rollout(qp, x0, uh64, p.force_scale),
                iters=iters, status=status, u_hat=uh64, y=np.asarray(y, dtype=np.float64), qp=qp)


# --------------------------------------------------------------------------------------
# independent reference solution + solver-independent acceptance
# --------------------------------------------------------------------------------------
def kkt_residuals(P, q, A, l, u, x, y):
    """Solver-independent optimality measures (all should be ~0 at the optimum)."""
    Ax = A @ x
    stat = np.max(np.abs(P @ x + q + A.T @ y))
    prim = max(0.0, float(np.max(l - Ax)), float(np.max(Ax - u)))
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    fin_u, fin_l = u < INF / 2, l > -INF / 2
    comp = max(float(np.max(np.abs(yp[fin_u] * (u - Ax)[fin_u]), initial=0.0)),
               float(np.max(np.abs(ym[fin_l] * (Ax - l)[fin_l]), initial=0.0)))
    dual_inf = max(float(np.max(yp[~fin_u], initial=0.0)), float(np.max(-ym[~fin_l], initial=0.0)))
    return dict(stationarity=float(stat), primal=prim, complementarity=comp, dual_sign=dual_inf)


def solve_reference(p: SrbdParams, qp, max_outer=200):
    """Independent high-accuracy solve: primal active-set iteration on the reduced problem
    (inactive contacts eliminated), each step an exact KKT solve, verified by kkt_residuals.
    Returns (x, y) in the scaled variables with KKT residuals <= ~1e-9.
    """
    P, q, A, l, u = qp["P"], qp["q"], qp["A"], qp["l"], qp["u"]
    n, m = P.shape[0], A.shape[0]
    eq_rows = np.where((u - l) < 1e-12)[0]           # fz == 0 rows of swing contacts
    fixed = np.zeros(n, bool)
    for r in eq_rows:
        c = np.where(A[r] != 0)[0][0]                 # the fz column
        fixed[c - 2:c + 1] = True                     # fx, fy, fz of that contact are all forced to 0
    free = ~fixed
    rows = np.array([r for r in range(m) if np.all(free[np.where(A[r] != 0)[0]])], dtype=int)
    Pf, qf, Af, lf, uf = P[np.ix_(free, free)], q[free], A[np.ix_(rows, free)], l[rows], u[rows]
    # start from a tight ADMM estimate of the active set
    pt = SrbdParams(**{**p.as_dict(), "eps_abs": 1e-9, "eps_rel": 1e-9, "max_iter": 20000, "check_every": 25})
    xf, _, yf, _, _ = admm_solve(pt, Pf, qf, Af, lf, uf)
    Axf = Af @ xf
    act_lo = (yf < -1e-7) & (lf > -INF / 2)
    act_up = (yf > 1e-7)
    for _ in range(max_outer):
        act = np.where(act_lo | act_up)[0]
        b = np.where(act_up[act], uf[act], lf[act])
        Aa = Af[act]
        na = len(act)
        KKT = np.block([[Pf, Aa.T], [Aa, np.zeros((na, na))]])
        sol = np.linalg.solve(KKT, np.concatenate([-qf, b]))
        xs, lam = sol[:Pf.shape[0]], sol[Pf.shape[0]:]
        ys = np.zeros(len(rows)); ys[act] = lam
        Axs = Af @ xs
        viol_lo = (Axs < lf - 1e-10) & ~act_lo
        viol_up = (Axs > uf + 1e-10) & ~act_up
        bad_lo = act_lo & (ys > 1e-12)
        bad_up = act_up & (ys < -1e-12)
        if not (viol_lo.any() or viol_up.any() or bad_lo.any() or bad_up.any()):
            break
        act_lo = (act_lo | viol_lo) & ~bad_lo
        act_up = (act_up | viol_up) & ~bad_up
    else:
        raise RuntimeError("active-set reference did not converge")
    x = np.zeros(n); x[free] = xs
    y = np.zeros(m); y[rows] = ys
    # multipliers of the eliminated contacts: recover a valid dual from stationarity
    g = P @ x + q + A.T @ y                           # nonzero only on fixed columns
    for r in eq_rows:
        c = np.where(A[r] != 0)[0][0]
        gx, gy, gz = g[c - 2], g[c - 1], g[c]
        # rows r-4..r-1 are the cone rows (upper bound 0, multipliers >= 0), row r the equality
        y[r - 4], y[r - 3] = max(-gx, 0.0), max(gx, 0.0)
        y[r - 2], y[r - 1] = max(-gy, 0.0), max(gy, 0.0)
        y[r] = -gz + p.mu * (y[r - 4] + y[r - 3] + y[r - 2] + y[r - 1])
    return x, y


# --------------------------------------------------------------------------------------
# synthetic inputs: product-side data (bench.py times the engine on them), re-exported here
# --------------------------------------------------------------------------------------
import os as _os
import sys as _sys
_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from g1_locomotion_amd.synth import COM_TARGET, HIP_Y, HEEL_X, TOE_X, synthetic_batch  # noqa: E402,F401

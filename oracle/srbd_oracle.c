/* srbd_oracle.c -- plain-C CPU restatement of the SRBD convex-MPC QP hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Used by tests/ (as a second, compiled checker next to oracle/srbd_oracle.py) and by bench.py's
 * `cpu_baseline` leg ("kind": "port").  Never linked, imported or executed by the product path.
 *
 * PARITY UNPINNED: the reference's implementation (submodule g1_mpc -> github.com/ioloizou/srbd_mpc,
 * /root/reference/.gitmodules:1-3) is absent from the snapshot and holds no fixtures; see the header of
 * oracle/srbd_oracle.py for the call-site evidence each convention follows
 * (g1_mujoco_sim/src/run_simulation.py:73-111, ros_run_simulation.py:58,65,199-215, wbid.py:17,123-124,261-266).
 *
 * Algorithm (same as srbd_oracle.py, dense and generic on purpose):
 *   linearise  : A_k (13x13), B_k (13x12) forward-Euler SRBD                 [srbd_oracle.py linearise()]
 *   condense   : A_qp, B_qp by dense block products                          [condense()]
 *   assemble   : P = Bs' Q Bs + R s^2, q = Bs' Q (A_qp x0 - x_ref), cone rows [build_qp()]
 *   solve      : K = P + sigma I + A' rho A, Cholesky, K^-1, OSQP-style ADMM [admm_solve()]
 *   rollout    : X = A_qp x0 + Bs u_hat                                      [rollout()]
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define NX 13
#define NU 12
#define NC 4
#define INF 1.0e30

typedef struct {
    double dt, mass, inertia[3], mu, fz_min, fz_max;
    double q_diag[NX], r_diag, force_scale;
    double rho, rho_eq_scale, sigma, alpha, eps_abs, eps_rel;
    double rho_fz_scale;   /* penalty of a stance contact's normal-force row relative to rho (srbd_oracle.py SrbdParams) */
    int max_iter, check_every;
    int rho_restart_iter;  /* presolved path: OSQP-style re-balancing of rho after this many iterations (0 = off) */
    int eliminate_swing;   /* presolve: drop the variables/rows of swing contacts (kernel v2); 0 = clamp via bounds */
    int rho_restart_count; /* at most this many re-balancings, one every rho_restart_iter iterations (<= 1: one) */
} srbd_oracle_params;

static void matmul(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) C[i * n + j] = 0.0;
        for (int l = 0; l < k; ++l) {
            const double a = A[i * k + l];
            if (a == 0.0) continue;
            for (int j = 0; j < n; ++j) C[i * n + j] += a * B[l * n + j];
        }
    }
}

static void linearise(const srbd_oracle_params* p, double yaw, const double* r /*4x3*/, double* A, double* B) {
    const double c = cos(yaw), s = sin(yaw);
    const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    double Iw[9], tmp[9], RzT[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RzT[i * 3 + j] = Rz[j * 3 + i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tmp[i * 3 + j] = Rz[i * 3 + j] / p->inertia[j];
    matmul(tmp, RzT, Iw, 3, 3, 3);
    memset(A, 0, sizeof(double) * NX * NX);
    memset(B, 0, sizeof(double) * NX * NU);
    for (int i = 0; i < NX; ++i) A[i * NX + i] = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i * NX + 6 + j] = p->dt * RzT[i * 3 + j];
    for (int i = 0; i < 3; ++i) A[(3 + i) * NX + 9 + i] = p->dt;
    A[11 * NX + 12] = p->dt;
    for (int ci = 0; ci < NC; ++ci) {
        const double rx = r[ci * 3], ry = r[ci * 3 + 1], rz = r[ci * 3 + 2];
        const double S[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        double IS[9];
        matmul(Iw, S, IS, 3, 3, 3);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[(6 + i) * NU + 3 * ci + j] = p->dt * IS[i * 3 + j];
        for (int i = 0; i < 3; ++i) B[(9 + i) * NU + 3 * ci + i] = p->dt / p->mass;
    }
}

/* workspace size in doubles for horizon N */
size_t srbd_oracle_work_doubles(int N) {
    const size_t n = 12 * (size_t)N, s = 13 * (size_t)N, m = 20 * (size_t)N;
    return (size_t)N * (NX * NX + NX * NU) + s * NX + s * n + 3 * n * n + 8 * n + 8 * m + 4 * s + 4 * N + 1024;
}

int srbd_oracle_solve(const srbd_oracle_params* p, int N, const double* x0, const double* xref, const double* foot,
                      const unsigned char* contact, const double* pcom, double* u_out, double* x_out,
                      double* P_out, double* q_out, int* iters_out, int* status_out, double* work) {
    const int n = NU * N, s = NX * N, m = 20 * N;
    double* w = work;
    double* Ak = w; w += (size_t)N * NX * NX;
    double* Bk = w; w += (size_t)N * NX * NU;
    double* Aqp = w; w += (size_t)s * NX;
    double* Bqp = w; w += (size_t)s * n;
    double* P = w; w += (size_t)n * n;
    double* K = w; w += (size_t)n * n;
    double* Kinv = w; w += (size_t)n * n;
    double* q = w; w += n;
    double* x = w; w += n;
    double* xt = w; w += n;
    double* rhs = w; w += n;
    double* Px = w; w += n;
    double* Atw = w; w += n;
    double* tmpn = w; w += n;
    double* z = w; w += m;
    double* y = w; w += m;
    double* lo = w; w += m;
    double* hi = w; w += m;
    double* rho = w; w += m;
    double* zt = w; w += m;
    double* wv = w; w += m;
    double* e = w; w += s;
    double* blk = w; w += NX * NU;
    double* blk2 = w; w += NX * NU;
    double* acc = w; w += NX * NX;
    double* acc2 = w; w += NX * NX;
    const double sc = p->force_scale, mu = p->mu;

    /* a5 linearise */
    for (int k = 0; k < N; ++k) {
        double r[12];
        for (int ci = 0; ci < NC; ++ci) for (int j = 0; j < 3; ++j) {
            const double pc = pcom ? pcom[k * 3 + j] : xref[k * NX + 3 + j];
            r[ci * 3 + j] = foot[k * NU + 3 * ci + j] - pc;
        }
        linearise(p, xref[k * NX + 2], r, Ak + (size_t)k * NX * NX, Bk + (size_t)k * NX * NU);
    }
    /* a6 condense */
    memset(Bqp, 0, sizeof(double) * (size_t)s * n);
    memset(acc, 0, sizeof(double) * NX * NX);
    for (int i = 0; i < NX; ++i) acc[i * NX + i] = 1.0;
    for (int i = 0; i < N; ++i) {
        matmul(Ak + (size_t)i * NX * NX, acc, acc2, NX, NX, NX);
        memcpy(acc, acc2, sizeof(double) * NX * NX);
        memcpy(Aqp + (size_t)i * NX * NX, acc, sizeof(double) * NX * NX);
        for (int j = 0; j <= i; ++j) {
            memcpy(blk, Bk + (size_t)j * NX * NU, sizeof(double) * NX * NU);
            for (int l = j + 1; l <= i; ++l) {
                matmul(Ak + (size_t)l * NX * NX, blk, blk2, NX, NX, NU);
                memcpy(blk, blk2, sizeof(double) * NX * NU);
            }
            for (int a = 0; a < NX; ++a) for (int c = 0; c < NU; ++c)
                Bqp[(size_t)(NX * i + a) * n + NU * j + c] = sc * blk[a * NU + c];
        }
    }
    /* a7 Hessian + gradient (scaled variables) */
    for (int a = 0; a < s; ++a) {
        double v = -xref[a];
        for (int c = 0; c < NX; ++c) v += Aqp[(size_t)a * NX + c] * x0[c];
        e[a] = p->q_diag[a % NX] * v;
    }
    memset(P, 0, sizeof(double) * (size_t)n * n);
    for (int a = 0; a < s; ++a) {
        const double qa = p->q_diag[a % NX];
        if (qa == 0.0) continue;
        const double* row = Bqp + (size_t)a * n;
        const int cmax = NU * (a / NX + 1);
        for (int i = 0; i < cmax; ++i) {
            const double v = qa * row[i];
            if (v == 0.0) continue;
            double* Pi = P + (size_t)i * n;
            for (int j = 0; j < cmax; ++j) Pi[j] += v * row[j];
        }
    }
    for (int i = 0; i < n; ++i) P[(size_t)i * n + i] += p->r_diag * sc * sc;
    for (int i = 0; i < n; ++i) {
        double v = 0.0;
        for (int a = 0; a < s; ++a) v += Bqp[(size_t)a * n + i] * e[a];
        q[i] = v;
    }
    if (P_out) memcpy(P_out, P, sizeof(double) * (size_t)n * n);
    if (q_out) memcpy(q_out, q, sizeof(double) * n);
    /* presolve: the contacts that take part in the solve */
    int nc = 0;
    int* cmap = (int*)w; w += 4 * N;                 /* compact contact -> original contact (4k+ci) */
    for (int gc = 0; gc < 4 * N; ++gc) if (!p->eliminate_swing || contact[gc]) cmap[nc++] = gc;
    const int nr = 3 * nc, mr = 5 * nc;
    /* a8 bounds (compact rows) */
    for (int e = 0; e < nc; ++e) {
        const int r0 = 5 * e;
        const int on = contact[cmap[e]] != 0;
        for (int j = 0; j < 4; ++j) { lo[r0 + j] = -INF; hi[r0 + j] = 0.0; rho[r0 + j] = p->rho; }
        lo[r0 + 4] = on ? p->fz_min / sc : 0.0;
        hi[r0 + 4] = on ? p->fz_max / sc : 0.0;
        rho[r0 + 4] = on ? p->rho * p->rho_fz_scale : p->rho * p->rho_eq_scale;
    }
    /* a9: K (compact), Cholesky, inverse */
#define VIDX(v) (3 * cmap[(v) / 3] + (v) % 3)       /* compact variable -> original variable */
    for (int i = 0; i < nr; ++i) tmpn[i] = q[VIDX(i)];   /* compact gradient */
    double* qc = tmpn; double* tmp2 = Atw;               /* note: Atw is reused below only after qc is copied */
    (void)tmp2;
    double* qcv = e;                                     /* e[] (length s >= nr) is free now: keep the compact q there */
    for (int i = 0; i < nr; ++i) qcv[i] = qc[i];
    double qn = 0.0, e_prim_last = INFINITY, last_rp = 0, last_np = 0, last_rd = 0, last_nd = 0;
    int status = 2, iters = 0, vote_ok = 1, iters_base = 0;
    const int restart = (p->eliminate_swing && p->rho_restart_iter > 0 && p->rho_restart_iter < p->max_iter) ? p->rho_restart_iter : 0;
    /* up to rho_restart_count re-balancings, one every `restart` iterations (each from the rho of the pass before it); the last pass runs to the cap */
    const int npass = restart ? 1 + (p->rho_restart_count > 1 ? p->rho_restart_count : 1) : 1;
    double rho_cur = p->rho;
    for (int pass = 0; pass < npass; ++pass) {
    const int left = p->max_iter - iters_base;                                      /* the cap is on the total */
    const int cap = (pass + 1 < npass && restart < left) ? restart : left;
    for (int i = 0; i < nr; ++i) for (int j = 0; j < nr; ++j) K[(size_t)i * nr + j] = P[(size_t)VIDX(i) * n + VIDX(j)];
    for (int e_ = 0; e_ < nc; ++e_) {
        const int r0 = 5 * e_, c0 = 3 * e_;
        K[(size_t)(c0) * nr + c0] += p->sigma + rho[r0] + rho[r0 + 1];
        K[(size_t)(c0 + 1) * nr + c0 + 1] += p->sigma + rho[r0 + 2] + rho[r0 + 3];
        K[(size_t)(c0 + 2) * nr + c0 + 2] += p->sigma + mu * mu * (rho[r0] + rho[r0 + 1] + rho[r0 + 2] + rho[r0 + 3]) + rho[r0 + 4];
    }
    if (nr > 0) {
    for (int j = 0; j < nr; ++j) {            /* lower Cholesky in place */
        double d = K[(size_t)j * nr + j];
        for (int k = 0; k < j; ++k) d -= K[(size_t)j * nr + k] * K[(size_t)j * nr + k];
        if (!(d > 0.0)) { *status_out = -1; *iters_out = 0; return -1; }
        d = sqrt(d);
        K[(size_t)j * nr + j] = d;
        for (int i = j + 1; i < nr; ++i) {
            double v = K[(size_t)i * nr + j];
            for (int k = 0; k < j; ++k) v -= K[(size_t)i * nr + k] * K[(size_t)j * nr + k];
            K[(size_t)i * nr + j] = v / d;
        }
    }
    }
    /* W = L^-1 (lower) into Kinv's lower part, then Kinv = W' W */
    double* W = Kinv;
    memset(W, 0, sizeof(double) * (size_t)nr * nr);
    for (int j = 0; j < nr; ++j) {
        W[(size_t)j * nr + j] = 1.0 / K[(size_t)j * nr + j];
        for (int i = j + 1; i < nr; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v += K[(size_t)i * nr + k] * W[(size_t)k * nr + j];
            W[(size_t)i * nr + j] = -v / K[(size_t)i * nr + i];
        }
    }
    memcpy(K, W, sizeof(double) * (size_t)nr * nr);   /* K now holds W */
    for (int i = 0; i < nr; ++i) for (int j = 0; j <= i; ++j) {
        double v = 0.0;
        for (int k = i; k < nr; ++k) v += K[(size_t)k * nr + i] * K[(size_t)k * nr + j];
        Kinv[(size_t)i * nr + j] = v;
    }
    for (int i = 0; i < nr; ++i) for (int j = i + 1; j < nr; ++j) Kinv[(size_t)i * nr + j] = Kinv[(size_t)j * nr + i];

    /* ADMM on the compact problem */
#define A_ROW(vec, j) ((j) == 0 ? (vec)[0] - mu * (vec)[2] : (j) == 1 ? -(vec)[0] - mu * (vec)[2] : \
                       (j) == 2 ? (vec)[1] - mu * (vec)[2] : (j) == 3 ? -(vec)[1] - mu * (vec)[2] : (vec)[2])
#define AT_APPLY(out, wv_)                                                                         \
    for (int e_ = 0; e_ < nc; ++e_) {                                                              \
        const double* pw = (wv_) + 5 * e_; double* po = (out) + 3 * e_;                            \
        po[0] = pw[0] - pw[1]; po[1] = pw[2] - pw[3];                                              \
        po[2] = -mu * (pw[0] + pw[1] + pw[2] + pw[3]) + pw[4];                                     \
    }
    if (pass == 0) {
        memset(x, 0, sizeof(double) * n);
        memset(y, 0, sizeof(double) * m);
        memset(Px, 0, sizeof(double) * n);
    } else {   /* continue from the first pass: x re-read in newtons, P x and z recomputed, y kept */
        for (int i = 0; i < nr; ++i) x[i] = (sc * x[i]) / sc;
        for (int i = 0; i < nr; ++i) {
            double v = 0.0;
            for (int j = 0; j < nr; ++j) v += P[(size_t)VIDX(i) * n + VIDX(j)] * x[j];
            Px[i] = v;
        }
    }
    for (int i = 0; i < mr; ++i) { const double* v = x + 3 * (i / 5); z[i] = fmin(fmax(A_ROW(v, i % 5), lo[i]), hi[i]); }
    qn = 0.0;
    for (int i = 0; i < nr; ++i) qn = fmax(qn, fabs(qcv[i]));
    status = 2; iters = cap; vote_ok = 1;
    e_prim_last = INFINITY;
    if (nr == 0) { status = 1; iters = 0; }
    for (int k = 1; k <= cap && nr > 0; ++k) {
        for (int i = 0; i < mr; ++i) wv[i] = rho[i] * z[i] - y[i];
        AT_APPLY(Atw, wv);
        for (int i = 0; i < nr; ++i) rhs[i] = p->sigma * x[i] - qcv[i] + Atw[i];
        for (int i = 0; i < nr; ++i) {
            const double* Ki = Kinv + (size_t)i * nr;
            double v = 0.0;
            for (int j = 0; j < nr; ++j) v += Ki[j] * rhs[j];
            xt[i] = v;
        }
        for (int i = 0; i < mr; ++i) {
            const double* v = xt + 3 * (i / 5);
            zt[i] = A_ROW(v, i % 5);
            wv[i] = rho[i] * (zt[i] - z[i]) + y[i];       /* nu */
        }
        AT_APPLY(tmpn, wv);
        for (int i = 0; i < nr; ++i) {
            const double pxt = p->sigma * (x[i] - xt[i]) - qcv[i] - tmpn[i];
            x[i] = p->alpha * xt[i] + (1.0 - p->alpha) * x[i];
            Px[i] = p->alpha * pxt + (1.0 - p->alpha) * Px[i];
        }
        for (int i = 0; i < mr; ++i) {
            const double zh = p->alpha * zt[i] + (1.0 - p->alpha) * z[i];
            const double zn = fmin(fmax(zh + y[i] / rho[i], lo[i]), hi[i]);
            y[i] = y[i] + rho[i] * (zh - zn);
            z[i] = zn;
        }
        /* residual pre-test (see srbd_oracle.py): full check at k only if |Ax - z| <= e_prim_last held at k-1 */
        if ((k + 1) % p->check_every == 0) {
            double rpm = 0;
            for (int i = 0; i < mr; ++i) {
                const double* v = x + 3 * (i / 5);
                rpm = fmax(rpm, fabs(A_ROW(v, i % 5) - z[i]));
            }
            vote_ok = (rpm <= e_prim_last);
        }
        if ((k % p->check_every == 0 && vote_ok) || k == cap) {
            double rp = 0, rd = 0, nax = 0, nz = 0, npx = 0, naty = 0;
            for (int i = 0; i < mr; ++i) {
                const double* v = x + 3 * (i / 5);
                const double ax = A_ROW(v, i % 5);
                rp = fmax(rp, fabs(ax - z[i])); nax = fmax(nax, fabs(ax)); nz = fmax(nz, fabs(z[i]));
            }
            AT_APPLY(Atw, y);
            for (int i = 0; i < nr; ++i) {
                const double r_ = fabs(Px[i] + qcv[i] + Atw[i]);
                if (r_ != r_) rd = r_; else if (rd == rd) rd = fmax(rd, r_);
                npx = fmax(npx, fabs(Px[i])); naty = fmax(naty, fabs(Atw[i]));
            }
            /* the maxima are compared after rounding to float (the kernels reduce them in fp32; see srbd_oracle.py) */
            rp = (double)(float)rp; rd = (double)(float)rd;
            if (!(rp <= INF) || !(rd <= INF)) { status = -1; iters = k; break; }
            const double ep = p->eps_abs + p->eps_rel * (double)(float)fmax(nax, nz);
            e_prim_last = ep;
            const double ed = p->eps_abs + p->eps_rel * fmax((double)(float)fmax(npx, naty), (double)(float)qn);
            last_rp = rp; last_np = (double)(float)fmax(nax, nz); last_rd = rd; last_nd = fmax((double)(float)fmax(npx, naty), (double)(float)qn);
            if (rp <= ep && rd <= ed) { status = 1; iters = k; break; }
        }
    }
    if (!(pass + 1 < npass && status == 2 && iters_base + cap < p->max_iter)) break;
    {   /* OSQP's re-balancing from the fp32 maxima of the last check (srbd_oracle.py restart_rho) */
        const double num = last_rp / fmax(last_np, 1e-30), den = last_rd / fmax(last_nd, 1e-30);
        double r1 = rho_cur;
        if (num > 0.0 && den > 0.0 && num <= INF && den <= INF) r1 = fmin(fmax(rho_cur * sqrt(num / den), rho_cur * 0.1), rho_cur * 5.0);
        rho_cur = r1;
        for (int i = 0; i < mr; ++i) rho[i] = (i % 5 == 4) ? r1 * p->rho_fz_scale : r1;   /* (restart: presolved path only, every row a stance row) */
        iters_base += cap;
    }
    }   /* pass */
    iters += iters_base;
    /* expand the compact solution (x is compact here) into the full variable vector */
    for (int i = 0; i < n; ++i) rhs[i] = 0.0;
    for (int i = 0; i < nr; ++i) rhs[VIDX(i)] = x[i];
    memcpy(x, rhs, sizeof(double) * n);
    for (int i = 0; i < n; ++i) u_out[i] = sc * x[i];
    if (x_out) {
        for (int c = 0; c < NX; ++c) x_out[c] = x0[c];
        for (int a = 0; a < s; ++a) {
            double v = 0.0;
            for (int c = 0; c < NX; ++c) v += Aqp[(size_t)a * NX + c] * x0[c];
            const double* row = Bqp + (size_t)a * n;
            for (int c = 0; c < n; ++c) v += row[c] * x[c];
            x_out[NX + a] = v;
        }
    }
    *iters_out = iters;
    *status_out = status;
    return 0;
}

typedef struct {
    const srbd_oracle_params* p; int N, B, tid, nthreads;
    const double *x0, *xref, *foot, *pcom; const unsigned char* contact;
    double *u_out, *x_out; int *iters, *status;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    const int N = j->N, n = NU * N;
    double* work = (double*)malloc(sizeof(double) * srbd_oracle_work_doubles(N));
    if (!work) return NULL;
    for (int b = j->tid; b < j->B; b += j->nthreads) {
        int it = 0, st = 0;
        srbd_oracle_solve(j->p, N, j->x0 + (size_t)b * NX, j->xref + (size_t)b * N * NX, j->foot + (size_t)b * N * NU,
                          j->contact + (size_t)b * N * NC, j->pcom ? j->pcom + (size_t)b * N * 3 : NULL,
                          j->u_out + (size_t)b * n, j->x_out ? j->x_out + (size_t)b * (N + 1) * NX : NULL, NULL, NULL,
                          &it, &st, work);
        if (j->iters) j->iters[b] = it;
        if (j->status) j->status[b] = st;
    }
    free(work);
    return NULL;
}

int srbd_oracle_solve_batch(const srbd_oracle_params* p, int N, int B, const double* x0, const double* xref,
                            const double* foot, const unsigned char* contact, const double* pcom, double* u_out,
                            double* x_out, int* iters, int* status, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){p, N, B, t, nthreads, x0, xref, foot, pcom, contact, u_out, x_out, iters, status};
        if (nthreads == 1) worker(&jobs[t]);
        else if (pthread_create(&th[t], NULL, worker, &jobs[t]) != 0) return -1;
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return 0;
}

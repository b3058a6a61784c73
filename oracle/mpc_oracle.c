/*
 * mpc_oracle.c -- CPU restatement of the reference MPC step (see mpc_oracle.h for scope and
 * parity status).  TEST INFRASTRUCTURE ONLY: never linked into or called from the product path.
 *
 * Compile WITHOUT -ffast-math and WITH -ffp-contract=off: stages S1-S3 must reproduce numpy's
 * operation order bit for bit where the outputs are integers (path indices, reaches_end).
 */
#include "mpc_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------
 * main/lib/mpc.py:46-58  smooth_yaw  (in place, sequential: each element sees the already-smoothed
 * predecessor)
 * ---------------------------------------------------------------------------------------------- */
void orc_smooth_yaw(double *yaw, int64_t n)
{
    for (int64_t i = 0; i + 1 < n; ++i) {
        double dyaw = yaw[i + 1] - yaw[i];
        while (dyaw >= M_PI / 2.0) {
            yaw[i + 1] -= M_PI * 2.0;
            dyaw = yaw[i + 1] - yaw[i];
        }
        while (dyaw <= -M_PI / 2.0) {
            yaw[i + 1] += M_PI * 2.0;
            dyaw = yaw[i + 1] - yaw[i];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * main/lib/trajectories.py:100-126  calc_nearest_index_in_direction
 *   dist = sqrt(dx*dx + dy*dy) over the tail cx[start:], the three smallest (numpy argpartition +
 *   argsort; ties are numpy-implementation-defined there -- restated as "lower index first", ties
 *   have measure zero for real-valued inputs), then the adjacency rules of :111-120.
 * ---------------------------------------------------------------------------------------------- */
int orc_nearest_index_in_direction(double x, double y, const double *cx, const double *cy,
                                   int64_t ncourse, int64_t start_index, int forward,
                                   int64_t *out_index)
{
    int64_t len = ncourse - start_index;
    if (len < 0) len = 0; /* numpy slice past the end is empty */
    if (len >= 3) {
        double bd[3] = {INFINITY, INFINITY, INFINITY};
        int64_t bi[3] = {-1, -1, -1};
        for (int64_t k = 0; k < len; ++k) {
            double dx = cx[start_index + k] - x;
            double dy = cy[start_index + k] - y;
            double d = sqrt(dx * dx + dy * dy);
            /* strict '<' keeps the lower index first among equal distances */
            if (d < bd[0]) {
                bd[2] = bd[1]; bi[2] = bi[1];
                bd[1] = bd[0]; bi[1] = bi[0];
                bd[0] = d; bi[0] = k;
            } else if (d < bd[1]) {
                bd[2] = bd[1]; bi[2] = bi[1];
                bd[1] = d; bi[1] = k;
            } else if (d < bd[2]) {
                bd[2] = d; bi[2] = k;
            }
        }
        if (llabs((long long)(bi[1] - bi[2])) == 2) {
            *out_index = bi[0] + start_index;
            return ORC_OK;
        }
        if (llabs((long long)(bi[0] - bi[1])) == 1) {
            int64_t a = bi[0], b = bi[1];
            *out_index = (forward ? (a > b ? a : b) : (a < b ? a : b)) + start_index;
            return ORC_OK;
        }
        *out_index = start_index;
        return ORC_NEAREST_ANOMALY; /* raise Exception("something wrong") */
    }
    if (len == 2) {
        *out_index = forward ? 1 + start_index : start_index;
        return ORC_OK;
    }
    *out_index = start_index; /* len <= 1 */
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * main/lib/mpc.py:89-112  _calc_ref_trajectory.  ov_in == NULL is the reference's ov=None branch (the
 * only one taken with the stock MAX_ITER=1); with MAX_ITER>1 the reference feeds the previous pass's
 * solved speeds, which orc_mpc_step passes as ov_in.
 * ---------------------------------------------------------------------------------------------- */
static int calc_ref_trajectory_impl(const orc_params *p, double sx, double sy, double sv,
                                    const double *ov_in, const double *cx, const double *cy,
                                    const double *cyaw, const double *cv, int64_t cv_cut,
                                    int64_t ncourse, int64_t start_idx,
                                    double *xref, int64_t *idx, uint8_t *reaches_end,
                                    int64_t *target_ind)
{
    const int T = p->T;
    int64_t s = start_idx;
    int st = orc_nearest_index_in_direction(sx, sy, cx, cy, ncourse, start_idx, 1, &s);
    *target_ind = s;
    if (st != ORC_OK) return st;
    if (ncourse < 1) return ORC_BAD_INPUT;

    /* ov = ones(T+1) * max(state.v, 10/3.6); travel = cumsum(|ov|*dt)  (sequential accumulate) */
    double vref = sv > p->min_ref_speed ? sv : p->min_ref_speed;
    double travel = 0.0;
    for (int k = 0; k <= T; ++k) {
        double ovk = ov_in ? ov_in[k] : vref;
        double c = fabs(ovk) * p->dt;
        travel = (k == 0) ? c : travel + c;
        int64_t ik = (int64_t)rint(travel / p->dl); /* np.rint: round-half-even */
        ik += s;
        if (ik > ncourse - 1) ik = ncourse - 1;
        idx[k] = ik;
        xref[0 * (T + 1) + k] = cx[ik];
        xref[1 * (T + 1) + k] = cy[ik];
        /* mpc.py:107 leaves the speed reference at 0; the mpc_with_speed variant sets xref[2] = cv[idx]
         * (mpc_with_speed.py:103-104) with cv zeroed from a cut-off index on (:276-282) */
        xref[2 * (T + 1) + k] = (cv && (cv_cut < 0 || ik < cv_cut)) ? cv[ik] : 0.0;
        xref[3 * (T + 1) + k] = cyaw[ik];
        reaches_end[k] = (ik == ncourse - 1);
    }
    return ORC_OK;
}

int orc_calc_ref_trajectory(const orc_params *p, double sx, double sy, double sv, const double *cx,
                            const double *cy, const double *cyaw, int64_t ncourse, int64_t start_idx,
                            double *xref, int64_t *idx, uint8_t *reaches_end, int64_t *target_ind)
{
    return calc_ref_trajectory_impl(p, sx, sy, sv, NULL, cx, cy, cyaw, NULL, -1, ncourse, start_idx, xref, idx,
                                    reaches_end, target_ind);
}

/* ------------------------------------------------------------------------------------------------
 * main/lib/simulation.py:35-47 Simulation.step  +  main/bicycle/main.py:28-41 Bicycle.step
 * state = [x, y, v, yaw]
 * ---------------------------------------------------------------------------------------------- */
void orc_plant_step(const orc_params *p, double state[4], double a, double delta)
{
    double d = delta < p->max_steer ? delta : p->max_steer; /* min(delta, MAX_STEER) */
    d = d > -p->max_steer ? d : -p->max_steer;              /* max(.., -MAX_STEER)   */
    double v = state[2], th = state[3];
    double xc_dot = v * cos(th);
    double yc_dot = v * sin(th);
    double theta_dot = (v / p->L) * tan(d);
    state[0] += xc_dot * p->dt;
    state[1] += yc_dot * p->dt;
    state[3] += theta_dot * p->dt;
    v += a * p->dt;
    v = v < p->max_speed ? v : p->max_speed;
    v = v > p->min_speed ? v : p->min_speed;
    state[2] = v;
}

/* main/lib/mpc.py:115-129 _predict_motion */
void orc_predict_motion(const orc_params *p, const double x0[4], const double *oa, const double *od,
                        double *xbar)
{
    const int T = p->T;
    double st[4] = {x0[0], x0[1], x0[2], x0[3]};
    for (int r = 0; r < 4; ++r) xbar[r * (T + 1)] = x0[r];
    for (int i = 1; i <= T; ++i) {
        orc_plant_step(p, st, oa[i - 1], od[i - 1]);
        for (int r = 0; r < 4; ++r) xbar[r * (T + 1) + i] = st[r];
    }
}

/* main/lib/mpc.py:61-82 _get_linear_model_matrix ; A 4x4, B 4x2 row-major */
void orc_linear_model_matrix(double v, double phi, double delta, double dt, double L, double A[16],
                             double B[8], double C[4])
{
    memset(A, 0, 16 * sizeof(double));
    memset(B, 0, 8 * sizeof(double));
    memset(C, 0, 4 * sizeof(double));
    A[0 * 4 + 0] = 1.0;
    A[1 * 4 + 1] = 1.0;
    A[2 * 4 + 2] = 1.0;
    A[3 * 4 + 3] = 1.0;
    A[0 * 4 + 2] = dt * cos(phi);
    A[0 * 4 + 3] = -dt * v * sin(phi);
    A[1 * 4 + 2] = dt * sin(phi);
    A[1 * 4 + 3] = dt * v * cos(phi);
    A[3 * 4 + 2] = dt * tan(delta) / L;
    double cd = cos(delta);
    B[2 * 2 + 0] = dt;
    B[3 * 2 + 1] = dt * v / (L * (cd * cd));
    C[0] = dt * v * sin(phi) * phi;
    C[1] = -dt * v * cos(phi) * phi;
    C[3] = -dt * v * delta / (L * (cd * cd));
}

int orc_nx(const orc_params *p) { return p->nx == 5 ? 5 : 4; }
int orc_nvar(const orc_params *p) { return 2 * p->T + (p->nx == 5 ? 1 : 0); }

/* main/lib/mpc.py:132-138 */
static void xy_cost_mtx(double angle, double P[4])
{
    double c = cos(angle), s = sin(angle);
    P[0] = c * c;
    P[1] = c * s;
    P[2] = c * s;
    P[3] = s * s;
}

/* ------------------------------------------------------------------------------------------------
 * main/lib/mpc.py:141-194  QP build, condensed.
 *   z_t = fresp_t + Sens_t u,  Sens_0 = 0, fresp_0 = x0,
 *   Sens_{t+1} = A_t Sens_t + B_t E_t,  fresp_{t+1} = A_t fresp_t + C_t          (:176-178, :189)
 *   cost (cvxpy carries no 1/2):  sum_t e_t' Q_t e_t + sum_t u_t' R_t u_t + sum_t du_t' Rd du_t
 *   => H = 2 (S'QS + Rbar + D'Rd D),  g = 2 S'Q (fresp - xref); multipliers of G u <= h are then the
 *   reference problem's own.
 * ---------------------------------------------------------------------------------------------- */
/* The acceleration-state variant (main/lib/mpc_jerk.py, p->nx == 5): state [x, y, v, yaw, acc] with
 *   A[4][4] = 1, A[2][4] = dt, B[4][0] = dt                                     (mpc_jerk.py:67,73,78)
 * on top of the stock model, `x[:4, 0] == x0` only (:193) -- acc_0 is a FREE variable, carried here as the last
 * decision variable (index 2T, n = 2T + 1) -- and the cost term (x[4,t+1] - x[4,t])^2 for t < T-1 (:190).
 * xref / xbar / Qf keep their four stock rows: the fifth ones are zero in the reference (xref = zeros, xbar = 0 * xref,
 * Qf[4][4] = 0, and :174 weighs x[2:4] only). */
int orc_build_qp(const orc_params *p, const double *xref, const double *xbar, const double x0[4],
                 const uint8_t *reaches_end, double speed, double *H, double *g, double *G, double *h,
                 uint8_t *skip, double *fresp, double *Sens)
{
    const int NX = orc_nx(p), T = p->T, n = orc_nvar(p), m = 8 * T, W = T + 1;
    memset(H, 0, sizeof(double) * n * n);
    memset(g, 0, sizeof(double) * n);
    memset(G, 0, sizeof(double) * m * n);
    memset(h, 0, sizeof(double) * m);
    memset(skip, 0, m);
    memset(Sens, 0, sizeof(double) * NX * W * n);
    memset(fresp, 0, sizeof(double) * NX * W);

    for (int r = 0; r < 4; ++r) fresp[r * W] = x0[r];
    if (NX == 5) Sens[(0 * NX + 4) * n + 2 * T] = 1.0; /* x[4,0] = acc_0 */
    /* Sens stored as [t][r][c] -> index ((t*NX + r) * n + c) */
    for (int t = 0; t < T; ++t) {
        double A4[16], B4[8], C4[4], A[25], Bm[10], C[5];
        /* dref[0,t] == 0 always (mpc.py:96) */
        orc_linear_model_matrix(xbar[2 * W + t], xbar[3 * W + t], 0.0, p->dt, p->L, A4, B4, C4);
        memset(A, 0, sizeof(A)); memset(Bm, 0, sizeof(Bm)); memset(C, 0, sizeof(C));
        for (int r = 0; r < 4; ++r) {
            for (int k = 0; k < 4; ++k) A[r * NX + k] = A4[r * 4 + k];
            Bm[r * 2 + 0] = B4[r * 2 + 0]; Bm[r * 2 + 1] = B4[r * 2 + 1];
            C[r] = C4[r];
        }
        if (NX == 5) { A[4 * NX + 4] = 1.0; A[2 * NX + 4] = p->dt; Bm[4 * 2 + 0] = p->dt; }
        for (int r = 0; r < NX; ++r) {
            double acc = C[r];
            for (int k = 0; k < NX; ++k) acc += A[r * NX + k] * fresp[k * W + t];
            fresp[r * W + t + 1] = acc;
            for (int c = 0; c < n; ++c) {
                double s = 0.0;
                for (int k = 0; k < NX; ++k) s += A[r * NX + k] * Sens[((t * NX) + k) * n + c];
                Sens[(((t + 1) * NX) + r) * n + c] = s;
            }
            Sens[(((t + 1) * NX) + r) * n + 2 * t + 0] += Bm[r * 2 + 0];
            Sens[(((t + 1) * NX) + r) * n + 2 * t + 1] += Bm[r * 2 + 1];
        }
    }

    /* state cost, t = 1..T (:160-173) */
    for (int t = 1; t <= T; ++t) {
        double Q[16];
        memset(Q, 0, sizeof(Q));
        if (!reaches_end[t]) {
            double P1[4], P2[4];
            double ref_yaw_perp = xref[3 * W + t] + 0.5 * M_PI;
            double ref_yaw = xref[3 * W + t];
            xy_cost_mtx(ref_yaw_perp, P1);
            xy_cost_mtx(ref_yaw, P2);
            Q[0 * 4 + 0] = P1[0] * p->w_perp + P2[0] * p->w_para;
            Q[0 * 4 + 1] = P1[1] * p->w_perp + P2[1] * p->w_para;
            Q[1 * 4 + 0] = P1[2] * p->w_perp + P2[2] * p->w_para;
            Q[1 * 4 + 1] = P1[3] * p->w_perp + P2[3] * p->w_para;
            Q[2 * 4 + 2] = p->Q_v_yaw[0];
            Q[3 * 4 + 3] = p->Q_v_yaw[1];
        } else {
            for (int r = 0; r < 4; ++r) Q[r * 4 + r] = p->Qf[r] * (double)T; /* Qf * T, mpc.py:28 */
        }
        const double *St = &Sens[(t * NX) * n];
        double e[4], Qe[4];
        for (int r = 0; r < 4; ++r) e[r] = fresp[r * W + t] - xref[r * W + t];
        for (int r = 0; r < 4; ++r) {
            Qe[r] = 0.0;
            for (int k = 0; k < 4; ++k) Qe[r] += Q[r * 4 + k] * e[k];
        }
        for (int i = 0; i < n; ++i) {
            double QSi[4]; /* (Q S_t)[:, i] */
            for (int r = 0; r < 4; ++r) {
                QSi[r] = 0.0;
                for (int k = 0; k < 4; ++k) QSi[r] += Q[r * 4 + k] * St[k * n + i];
            }
            for (int r = 0; r < 4; ++r) g[i] += 2.0 * St[r * n + i] * Qe[r];
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int r = 0; r < 4; ++r) s += St[r * n + j] * QSi[r];
                H[j * n + i] += 2.0 * s;
            }
        }
    }
    /* mpc_jerk.py:190  jerk_penalty_weight * (x[4,t+1] - x[4,t])^2,  t < T-1 */
    if (NX == 5) {
        for (int t = 0; t + 1 < T; ++t) {
            const double *Sa = &Sens[(t * NX + 4) * n], *Sb = &Sens[((t + 1) * NX + 4) * n];
            const double df = fresp[4 * W + t + 1] - fresp[4 * W + t];
            for (int i = 0; i < n; ++i) {
                const double di = Sb[i] - Sa[i];
                g[i] += 2.0 * p->jerk_weight * df * di;
                for (int j = 0; j < n; ++j) H[i * n + j] += 2.0 * p->jerk_weight * di * (Sb[j] - Sa[j]);
            }
        }
    }
    /* input cost (:180-183) and input-difference cost (:186) */
    for (int t = 0; t < T; ++t) {
        const double *Rt = reaches_end[t] ? p->R_end : p->R;
        H[(2 * t) * n + 2 * t] += 2.0 * Rt[0];
        H[(2 * t + 1) * n + 2 * t + 1] += 2.0 * Rt[1];
    }
    for (int t = 0; t + 1 < T; ++t) {
        for (int c = 0; c < 2; ++c) {
            int i = 2 * t + c, j = 2 * (t + 1) + c;
            H[i * n + i] += 2.0 * p->Rd[c];
            H[j * n + j] += 2.0 * p->Rd[c];
            H[i * n + j] -= 2.0 * p->Rd[c];
            H[j * n + i] -= 2.0 * p->Rd[c];
        }
    }
    /* symmetrise against rounding asymmetry */
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double s = 0.5 * (H[i * n + j] + H[j * n + i]);
            H[i * n + j] = s;
            H[j * n + i] = s;
        }

    /* inequality rows, canonical order (mpc_oracle.h) */
    const double dmax = p->max_dsteer * p->dt;
    int status = ORC_OK;
    for (int t = 0; t + 1 < T; ++t) { /* D (:187) */
        int r0 = 2 * t, r1 = 2 * t + 1;
        G[r0 * n + 2 * (t + 1) + 1] = 1.0;
        G[r0 * n + 2 * t + 1] = -1.0;
        h[r0] = dmax;
        G[r1 * n + 2 * (t + 1) + 1] = -1.0;
        G[r1 * n + 2 * t + 1] = 1.0;
        h[r1] = dmax;
    }
    for (int t = 0; t <= T; ++t) { /* VU (:190), VL (:191) */
        int ru = 2 * T - 2 + t, rl = 3 * T - 1 + t;
        const double *Sv = &Sens[(t * NX + 2) * n];
        for (int c = 0; c < n; ++c) {
            G[ru * n + c] = Sv[c];
            G[rl * n + c] = -Sv[c];
        }
        h[ru] = speed - fresp[2 * W + t];
        h[rl] = fresp[2 * W + t] - p->min_speed;
        if (t == 0) { /* x[2,0] is pinned to x0.v by :189 -> constant rows */
            skip[ru] = 1;
            skip[rl] = 1;
            /* ECOS accepts violations below its feasibility tolerance (1e-8) */
            if (h[ru] < -1e-8 || h[rl] < -1e-8) status = ORC_INFEASIBLE;
        }
    }
    for (int t = 0; t < T; ++t) { /* AU (:192), AL (:193) */
        int ru = 4 * T + t, rl = 5 * T + t;
        G[ru * n + 2 * t] = 1.0;
        h[ru] = p->max_accel;
        G[rl * n + 2 * t] = -1.0;
        h[rl] = -p->max_decel;
    }
    for (int t = 0; t < T; ++t) { /* S (:194) */
        int r0 = 6 * T + 2 * t, r1 = r0 + 1;
        G[r0 * n + 2 * t + 1] = 1.0;
        h[r0] = p->max_steer;
        G[r1 * n + 2 * t + 1] = -1.0;
        h[r1] = p->max_steer;
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * Exact solve of the strictly convex QP (stands in for cvxpy->ECOS, mpc.py:196-199).
 * Goldfarb & Idnani (1983) dual active-set method:
 *   J = L^{-T} Q with H = L L',  R upper triangular, L^{-1} N = Q [R; 0] for the working set's
 *   normals N (in ">=" form n_i = -G_i').  A constraint is added with ONE Householder reflection on
 *   the trailing block of J (instead of a Givens sweep; same Q up to sign) and dropped with a Givens
 *   sweep on R / the leading block of J.  The HIP kernel follows the same sequence of decisions.
 *   Selection rule: among the violated rows the one with the largest  viol_i^2 / (n_i' H^-1 n_i)  (a static
 *   steepest-edge weight: the dual ascent a full step on row i would give from the unconstrained optimum;
 *   n_i' H^-1 n_i = ||J' n_i||^2 is computed once per solve from J = L^-T), ties -> lowest row index.  Against
 *   "largest raw violation" this needs ~40 % fewer iterations on the closed-loop workload (T = 20: 22.9 -> 13.6).
 *   Ratio test ties -> lowest working-set position.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_VIOL_TOL 1e-10
#define ORC_DEP_TOL 1e-18 /* ||d2||^2 <= tol * ||d||^2  => normal is in the span of the working set */

/* Entering-row key with its low 20 mantissa bits cleared: keys closer than 2^-32 relative are ties -> lowest row id.
 * (Same rule as csrc/jsim_mpc.hip: jsim_key_trunc.) */
static double orc_key_trunc(double k)
{
    uint64_t b;
    memcpy(&b, &k, 8);
    b &= ~(uint64_t)0xFFFFF;
    memcpy(&k, &b, 8);
    return k;
}

/* Dual step length with its low 7 mantissa bits cleared (ties -> lowest working-set position); jsim_ratio_trunc. */
static double orc_ratio_trunc(double t)
{
    uint64_t b;
    memcpy(&b, &t, 8);
    b &= ~(uint64_t)127;
    memcpy(&t, &b, 8);
    return t;
}

int orc_solve_qp(int n, int m, const double *H, const double *g, const double *G, const double *h,
                 const uint8_t *skip, double *u, double *lam, int32_t *n_iter_out)
{
    int status = ORC_OK;
    double *Lm = (double *)malloc(sizeof(double) * n * n);
    double *J = (double *)malloc(sizeof(double) * n * n);
    double *Rm = (double *)calloc((size_t)n * n, sizeof(double));
    double *d = (double *)malloc(sizeof(double) * n);
    double *z = (double *)malloc(sizeof(double) * n);
    double *r = (double *)malloc(sizeof(double) * n);
    double *w = (double *)malloc(sizeof(double) * n);
    double *lact = (double *)malloc(sizeof(double) * n);
    int *act = (int *)malloc(sizeof(int) * n);
    uint8_t *inact = (uint8_t *)calloc(m, 1);
    double *wgt = (double *)malloc(sizeof(double) * m);
    int q = 0, iters = 0;
    const int max_iters = 50 * n + 100;

    /* Cholesky H = L L' */
    memcpy(Lm, H, sizeof(double) * n * n);
    for (int k = 0; k < n; ++k) {
        double dk = Lm[k * n + k];
        for (int j = 0; j < k; ++j) dk -= Lm[k * n + j] * Lm[k * n + j];
        if (!(dk > 0.0)) { status = ORC_BAD_INPUT; goto done; }
        dk = sqrt(dk);
        Lm[k * n + k] = dk;
        for (int i = k + 1; i < n; ++i) {
            double s = Lm[i * n + k];
            for (int j = 0; j < k; ++j) s -= Lm[i * n + j] * Lm[k * n + j];
            Lm[i * n + k] = s / dk;
        }
        for (int j = k + 1; j < n; ++j) Lm[k * n + j] = 0.0;
    }
    /* X = L^{-1} (lower), J = X' */
    for (int c = 0; c < n; ++c) {
        for (int i = 0; i < n; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int j = c; j < i; ++j) s -= Lm[i * n + j] * J[c * n + j]; /* J[c][j] = X[j][c] */
            J[c * n + i] = (i < c) ? 0.0 : s / Lm[i * n + i];
        }
    }
    /* J currently holds X' stored as J[c][i] = X[i][c]  => J[row c][col i] = (L^{-T})[c][i]. */
    /* unconstrained optimum u = -J J' g */
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += J[i * n + j] * g[i];
        d[j] = s;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += J[i * n + j] * d[j];
        u[i] = -s;
    }
    memset(lam, 0, sizeof(double) * m);
    /* static steepest-edge weights  w_i = n_i' H^-1 n_i = ||J' n_i||^2  (J = L^-T here) */
    for (int i = 0; i < m; ++i) {
        double acc = 0.0;
        if (!skip[i]) {
            /* rows of G are sparse (1, 2 or t non-zeros): accumulate G_i J over the non-zero columns only */
            for (int j = 0; j < n; ++j) d[j] = 0.0;
            for (int c = 0; c < n; ++c) {
                const double gc = G[i * n + c];
                if (gc != 0.0)
                    for (int j = c; j < n; ++j) d[j] += gc * J[c * n + j]; /* J = L^-T is upper triangular here */
            }
            for (int j = 0; j < n; ++j) acc += d[j] * d[j];
        }
        wgt[i] = acc > 0.0 ? acc : 1.0;
    }

    for (;;) {
        /* step 1: violated row with the largest viol^2 / w */
        int p = -1;
        double vmax = 0.0;
        for (int i = 0; i < m; ++i) {
            if (skip[i] || inact[i]) continue;
            double s = -h[i];
            for (int c = 0; c < n; ++c) s += G[i * n + c] * u[c];
            if (s > ORC_VIOL_TOL * (1.0 + fabs(h[i]))) {
                const double key = orc_key_trunc(s * s / wgt[i]);
                if (key > vmax) {
                    vmax = key;
                    p = i;
                }
            }
        }
        if (p < 0) break; /* optimal */
        double lplus = 0.0;

        for (;;) { /* step 2 */
            if (++iters > max_iters) { status = ORC_INFEASIBLE; goto finish; }
            /* d = J' n+,  n+ = -G_p' */
            double dd = 0.0, zn = 0.0;
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s -= G[p * n + i] * J[i * n + j];
                d[j] = s;
                dd += s * s;
                if (j >= q) zn += s * s;
            }
            /* z = J2 d2 */
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = q; j < n; ++j) s += J[i * n + j] * d[j];
                z[i] = s;
            }
            /* r = R^{-1} d1 */
            for (int j = q - 1; j >= 0; --j) {
                double s = d[j];
                for (int k = j + 1; k < q; ++k) s -= Rm[j * n + k] * r[k];
                r[j] = s / Rm[j * n + j];
            }
            /* ratio test on the dual */
            int l = -1;
            double t1 = INFINITY;
            for (int k = 0; k < q; ++k) {
                if (r[k] > 0.0) {
                    double tk = orc_ratio_trunc(lact[k] / r[k]);
                    if (tk < t1) { t1 = tk; l = k; }
                }
            }
            int dependent = !(zn > ORC_DEP_TOL * dd);
            double viol = -h[p];
            for (int c = 0; c < n; ++c) viol += G[p * n + c] * u[c];
            double t2 = dependent ? INFINITY : viol / zn;
            if (t2 < 0.0) t2 = 0.0;

            if (isinf(t1) && isinf(t2)) { status = ORC_INFEASIBLE; goto finish; }

            int full = (t2 <= t1);
            double t = full ? t2 : t1;
            if (!dependent)
                for (int i = 0; i < n; ++i) u[i] += t * z[i];
            for (int k = 0; k < q; ++k) {
                lact[k] -= t * r[k];
                if (lact[k] < 0.0) lact[k] = 0.0;
            }
            lplus += t;

            if (full) {
                /* add p: Householder P on d2 -> rho e1, J2 <- J2 P, new R column [d1; rho] */
                double nrm = sqrt(zn);
                double sg = d[q] >= 0.0 ? 1.0 : -1.0;
                double rho = -sg * nrm;
                double v0 = d[q] + sg * nrm;           /* v = d2 + sg*nrm*e1 */
                double beta = 1.0 / (nrm * (nrm + fabs(d[q]))); /* 2 / v'v */
                for (int i = 0; i < n; ++i) {
                    /* w_i = beta * (J2 v)_i = beta * (z_i + sg*nrm*J[i][q]) */
                    w[i] = beta * (z[i] + sg * nrm * J[i * n + q]);
                }
                for (int i = 0; i < n; ++i) {
                    J[i * n + q] -= w[i] * v0;
                    for (int j = q + 1; j < n; ++j) J[i * n + j] -= w[i] * d[j];
                }
                for (int k = 0; k < q; ++k) Rm[k * n + q] = d[k];
                Rm[q * n + q] = rho;
                act[q] = p;
                lact[q] = lplus;
                inact[p] = 1;
                ++q;
                break; /* back to step 1 */
            }
            /* partial step: drop working-set position l */
            {
                inact[act[l]] = 0;
                for (int j = l; j + 1 < q; ++j) {
                    /* column j+1 moves to column j; rotate rows (j, j+1) to zero the subdiagonal */
                    double a = Rm[j * n + j + 1], b = Rm[(j + 1) * n + j + 1];
                    double hh = hypot(a, b);
                    double c = 1.0, s = 0.0;
                    if (hh > 0.0) { c = a / hh; s = b / hh; }
                    for (int k = j + 1; k < q; ++k) {
                        double x1 = Rm[j * n + k], x2 = Rm[(j + 1) * n + k];
                        Rm[j * n + k] = c * x1 + s * x2;
                        Rm[(j + 1) * n + k] = -s * x1 + c * x2;
                    }
                    for (int i = 0; i < n; ++i) {
                        double x1 = J[i * n + j], x2 = J[i * n + j + 1];
                        J[i * n + j] = c * x1 + s * x2;
                        J[i * n + j + 1] = -s * x1 + c * x2;
                    }
                }
                /* shift columns left */
                for (int j = l; j + 1 < q; ++j) {
                    for (int k = 0; k <= j; ++k) Rm[k * n + j] = Rm[k * n + j + 1];
                    act[j] = act[j + 1];
                    lact[j] = lact[j + 1];
                }
                --q;
            }
        }
    }
finish:
    for (int k = 0; k < q; ++k) lam[act[k]] = lact[k];
    *n_iter_out = iters;
done:
    free(Lm); free(J); free(Rm); free(d); free(z); free(r); free(w); free(lact); free(act); free(inact); free(wgt);
    return status;
}

void orc_active_mask(int m, int n, const double *lam, const double *g, uint32_t *mask)
{
    double gmax = 1.0;
    for (int i = 0; i < n; ++i)
        if (fabs(g[i]) > gmax) gmax = fabs(g[i]);
    const double thr = 1e-9 * gmax;
    for (int i = 0; i < (m + 31) / 32; ++i) mask[i] = 0u;
    for (int i = 0; i < m; ++i)
        if (lam[i] > thr) mask[i >> 5] |= (1u << (i & 31));
}

/* ------------------------------------------------------------------------------------------------
 * main/lib/mpc.py:214-242 + :284-303  one controller step for one ego
 * ---------------------------------------------------------------------------------------------- */
int orc_mpc_step(const orc_params *p, double sx, double sy, double syaw, double sv, const double *cx,
                 const double *cy, const double *cyaw, int64_t ncourse, int64_t target_ind,
                 double speed, const double *oa_in, const double *od_in, orc_step_out *out)
{
    return orc_mpc_step_cv(p, sx, sy, syaw, sv, cx, cy, cyaw, NULL, -1, ncourse, target_ind, speed, oa_in, od_in, out);
}

/* the same step with a per-point speed reference cv (NULL: none) zeroed from index cv_cut on (< 0: nowhere):
 * main/lib/mpc_with_speed.py:85-110,276-282 */
int orc_mpc_step_cv(const orc_params *p, double sx, double sy, double syaw, double sv, const double *cx,
                    const double *cy, const double *cyaw, const double *cv, int64_t cv_cut, int64_t ncourse,
                    int64_t target_ind, double speed, const double *oa_in, const double *od_in, orc_step_out *out)
{
    const int NX = orc_nx(p), T = p->T, n = orc_nvar(p), m = 8 * T, W = T + 1;
    double x0[4] = {sx, sy, sv, syaw}; /* mpc.py:291 */
    double *oa = (double *)calloc(T, sizeof(double));
    double *od = (double *)calloc(T, sizeof(double));
    double *xref = (double *)malloc(sizeof(double) * 4 * W);
    double *xbar = (double *)malloc(sizeof(double) * 4 * W);
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * W);
    uint8_t *rend = (uint8_t *)malloc(W);
    double *H = (double *)malloc(sizeof(double) * n * n);
    double *g = (double *)malloc(sizeof(double) * n);
    double *G = (double *)malloc(sizeof(double) * m * n);
    double *h = (double *)malloc(sizeof(double) * m);
    uint8_t *skip = (uint8_t *)malloc(m);
    double *fresp = (double *)malloc(sizeof(double) * NX * W);
    double *Sens = (double *)malloc(sizeof(double) * NX * W * n);
    double *u = (double *)malloc(sizeof(double) * n);
    double *lam = (double *)malloc(sizeof(double) * m);
    double *ovprev = (double *)malloc(sizeof(double) * W);
    int have_ov = 0;
    int status = ORC_OK;
    int32_t iters_total = 0;
    int64_t tind = target_ind;

    if (oa_in && od_in) { /* mpc.py:225-227 */
        memcpy(oa, oa_in, sizeof(double) * T);
        memcpy(od, od_in, sizeof(double) * T);
    }
    int passes = p->max_iter > 0 ? p->max_iter : 1;
    for (int it = 0; it < passes; ++it) { /* mpc.py:231 */
        status = calc_ref_trajectory_impl(p, sx, sy, sv, have_ov ? ovprev : NULL, cx, cy, cyaw, cv, cv_cut, ncourse,
                                          tind, xref, idx, rend, &tind);
        if (status != ORC_OK) break;
        orc_predict_motion(p, x0, oa, od, xbar);
        status = orc_build_qp(p, xref, xbar, x0, rend, speed, H, g, G, h, skip, fresp, Sens);
        if (status != ORC_OK) break;
        int32_t iters = 0;
        status = orc_solve_qp(n, m, H, g, G, h, skip, u, lam, &iters);
        iters_total += iters;
        if (status != ORC_OK) break;
        for (int t = 0; t < T; ++t) {
            oa[t] = u[2 * t];
            od[t] = u[2 * t + 1];
        }
        /* predicted states of the linearised model = the cvxpy x variable at the optimum */
        for (int t = 0; t <= T; ++t) {
            double zt[4];
            for (int r = 0; r < 4; ++r) {
                double s = fresp[r * W + t];
                const double *Sr = &Sens[(t * NX + r) * n];
                for (int c = 0; c < n; ++c) s += Sr[c] * u[c];
                zt[r] = s;
            }
            if (out->ox) out->ox[t] = zt[0];
            if (out->oy) out->oy[t] = zt[1];
            if (out->ov) out->ov[t] = zt[2];
            if (out->oyaw) out->oyaw[t] = zt[3];
            ovprev[t] = zt[2];
        }
        have_ov = 1;
    }
    out->status = status;
    out->n_iter = iters_total;
    out->target_ind = tind;
    if (out->xref && status != ORC_NEAREST_ANOMALY) memcpy(out->xref, xref, sizeof(double) * 4 * W);
    if (status != ORC_NEAREST_ANOMALY) {
        if (out->xbar) memcpy(out->xbar, xbar, sizeof(double) * 4 * W);
        if (out->idx) memcpy(out->idx, idx, sizeof(int64_t) * W);
        if (out->reaches_end) memcpy(out->reaches_end, rend, W);
    }
    if (status == ORC_OK) {
        if (out->oa) memcpy(out->oa, oa, sizeof(double) * T);
        if (out->od) memcpy(out->od, od, sizeof(double) * T);
        if (out->lam) memcpy(out->lam, lam, sizeof(double) * m);
        if (out->active_mask) orc_active_mask(m, n, lam, g, out->active_mask);
        if (out->H) memcpy(out->H, H, sizeof(double) * n * n);
        if (out->g) memcpy(out->g, g, sizeof(double) * n);
    } else if (out->active_mask) {
        for (int i = 0; i < (m + 31) / 32; ++i) out->active_mask[i] = 0u;
    }
    free(oa); free(od); free(xref); free(xbar); free(idx); free(rend); free(H); free(g); free(G);
    free(h); free(skip); free(fresp); free(Sens); free(u); free(lam); free(ovprev);
    return status;
}

int orc_mpc_step_batch(const orc_params *p, int32_t B, const double *x0, const int32_t *path_id,
                       const int32_t *path_len, const double *speed, const double *cx,
                       const double *cy, const double *cyaw, const int64_t *path_off,
                       int64_t *target_ind, double *oa, double *od, double *ox, double *oy,
                       double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                       int32_t *status, int32_t *n_iter, int32_t n_threads)
{
    return orc_mpc_step_batch_cv(p, B, x0, path_id, path_len, speed, cx, cy, cyaw, NULL, NULL, path_off, target_ind, oa,
                                 od, ox, oy, ov, oyaw, xref, active_mask, status, n_iter, n_threads);
}

int orc_mpc_step_batch_cv(const orc_params *p, int32_t B, const double *x0, const int32_t *path_id,
                          const int32_t *path_len, const double *speed, const double *cx,
                          const double *cy, const double *cyaw, const double *cv, const int32_t *cv_cut,
                          const int64_t *path_off, int64_t *target_ind, double *oa, double *od, double *ox,
                          double *oy, double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                          int32_t *status, int32_t *n_iter, int32_t n_threads)
{
    const int T = p->T, W = T + 1, MW = (8 * T + 31) / 32;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int32_t b = 0; b < B; ++b) {
        const int64_t off = path_off[path_id[b]];
        orc_step_out o;
        memset(&o, 0, sizeof(o));
        double *tmp = (double *)malloc(sizeof(double) * (2 * T + 4 * W + 4 * W));
        o.oa = tmp;
        o.od = tmp + T;
        o.ox = tmp + 2 * T;
        o.oy = o.ox + W;
        o.ov = o.oy + W;
        o.oyaw = o.ov + W;
        o.xref = o.oyaw + W;
        o.active_mask = active_mask ? &active_mask[(size_t)b * MW] : NULL;
        int st = orc_mpc_step_cv(p, x0[4 * b + 0], x0[4 * b + 1], x0[4 * b + 3], x0[4 * b + 2], cx + off,
                                 cy + off, cyaw + off, cv ? cv + off : NULL, cv_cut ? (int64_t)cv_cut[b] : -1,
                                 (int64_t)path_len[b], target_ind[b], speed[b],
                                 oa ? &oa[(size_t)b * T] : NULL, od ? &od[(size_t)b * T] : NULL, &o);
        if (st != ORC_NEAREST_ANOMALY) target_ind[b] = o.target_ind;
        if (st == ORC_OK) {
            if (oa) memcpy(&oa[(size_t)b * T], o.oa, sizeof(double) * T);
            if (od) memcpy(&od[(size_t)b * T], o.od, sizeof(double) * T);
            if (ox) memcpy(&ox[(size_t)b * W], o.ox, sizeof(double) * W);
            if (oy) memcpy(&oy[(size_t)b * W], o.oy, sizeof(double) * W);
            if (ov) memcpy(&ov[(size_t)b * W], o.ov, sizeof(double) * W);
            if (oyaw) memcpy(&oyaw[(size_t)b * W], o.oyaw, sizeof(double) * W);
        }
        if (st == ORC_INFEASIBLE) { /* batched form of "outputs None -> next call cold-starts from zeros" */
            if (oa) memset(&oa[(size_t)b * T], 0, sizeof(double) * T);
            if (od) memset(&od[(size_t)b * T], 0, sizeof(double) * T);
        }
        if (xref && st != ORC_NEAREST_ANOMALY)
            memcpy(&xref[(size_t)b * 4 * W], o.xref, sizeof(double) * 4 * W);
        if (status) status[b] = st;
        if (n_iter) n_iter[b] = o.n_iter;
        free(tmp);
    }
    return ORC_OK;
}

/* main/lib/mpc.py:305-312 (element-wise product as written in the reference) */
double orc_xref_deviation(const double *cx, const double *cy, const double *cyaw, int64_t target_ind,
                          double ox0, double oy0)
{
    double ref_yaw_perp = cyaw[target_ind] + M_PI / 2;
    double dx = cx[target_ind] - ox0;
    double dy = cy[target_ind] - oy0;
    double a = cos(ref_yaw_perp) * dx;
    double b = sin(ref_yaw_perp) * dy;
    return sqrt(a * a + b * b);
}

/* main/lib/mpc.py:314-330 */
int orc_is_goal(const orc_params *p, double sx, double sy, double sv, double goal_x, double goal_y,
                int64_t target_ind, int64_t ncourse)
{
    double d = hypot(sx - goal_x, sy - goal_y);
    int isgoal = d <= p->goal_dis;
    if (llabs((long long)(target_ind - ncourse)) >= 5) isgoal = 0;
    int isstop = fabs(sv) <= p->stop_speed;
    return isgoal && isstop;
}

/* ------------------------------------------------------------------------------------------------
 * main/scenarios/mpc_intersection.py:99-163 -- the per-vehicle loop for B independent egos, n_ticks ticks each:
 * MPC.step (warm start carried, failure path ai = MAX_DECEL / di kept, mpc.py:298-303) -> Simulation.step
 * (lib/simulation.py:35-47) -> `if mpc.is_goal(state): break` (:101), a finished ego (or one older than max_age
 * ticks) restarting from its spawn state with a cold controller -- the bookkeeping of the product's
 * jsim_loop_advance.  hist [n_ticks][B][2] = applied (di, ai) (NULL: not recorded).  OpenMP over egos: each thread
 * runs whole egos (every tick of an ego depends on its previous one).  Used as the closed-loop checker of
 * jsim_mpc_run_ticks and as bench.py's cpu_baseline.
 * ---------------------------------------------------------------------------------------------- */
int orc_closed_loop(const orc_params *p, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id,
                    const int32_t *path_len, const double *speed, const double *cx, const double *cy,
                    const double *cyaw, const int64_t *path_off, int64_t *target_ind, double *oa, double *od,
                    double *di_ai, const double *x0_spawn, const int64_t *target_spawn, int32_t *age,
                    int32_t max_age, double *hist, int64_t *n_respawn, int64_t *n_iter_sum, int64_t *n_fail,
                    int32_t n_threads)
{
    const int T = p->T, W = T + 1;
    int64_t resp = 0, iters = 0, fails = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : resp, iters, fails)
#endif
    for (int32_t b = 0; b < B; ++b) {
        const int64_t off = path_off[path_id[b]];
        const int64_t full = path_off[path_id[b] + 1] - off;
        double *tmp = (double *)malloc(sizeof(double) * (2 * T + 4 * W));
        double *woa = &oa[(size_t)b * T], *wod = &od[(size_t)b * T];
        double st4[4] = {x0[4 * b], x0[4 * b + 1], x0[4 * b + 2], x0[4 * b + 3]};
        for (int32_t k = 0; k < n_ticks; ++k) {
            orc_step_out o;
            memset(&o, 0, sizeof(o));
            o.oa = tmp; o.od = tmp + T; o.ox = tmp + 2 * T; o.oy = o.ox + W; o.ov = o.oy + W; o.oyaw = o.ov + W;
            int st = orc_mpc_step_cv(p, st4[0], st4[1], st4[3], st4[2], cx + off, cy + off, cyaw + off, NULL, -1,
                                     (int64_t)path_len[b], target_ind[b], speed[b], woa, wod, &o);
            iters += o.n_iter;
            if (st != ORC_NEAREST_ANOMALY) target_ind[b] = o.target_ind;
            double di = di_ai[2 * b], ai;
            if (st == ORC_OK) {
                memcpy(woa, o.oa, sizeof(double) * T);
                memcpy(wod, o.od, sizeof(double) * T);
                di = wod[0];
                ai = woa[0];
            } else {
                ++fails;
                if (st == ORC_INFEASIBLE) { memset(woa, 0, sizeof(double) * T); memset(wod, 0, sizeof(double) * T); }
                ai = p->max_decel;
            }
            di_ai[2 * b] = di;
            di_ai[2 * b + 1] = ai;
            if (hist) { hist[((size_t)k * B + b) * 2] = di; hist[((size_t)k * B + b) * 2 + 1] = ai; }
            orc_plant_step(p, st4, ai, di);
            const int goal = orc_is_goal(p, st4[0], st4[1], st4[2], cx[off + full - 1], cy[off + full - 1], target_ind[b],
                                         (int64_t)path_len[b]);
            if (goal || (max_age > 0 && age[b] + 1 >= max_age)) {
                for (int r = 0; r < 4; ++r) st4[r] = x0_spawn[4 * b + r];
                target_ind[b] = target_spawn[b];
                memset(woa, 0, sizeof(double) * T);
                memset(wod, 0, sizeof(double) * T);
                di_ai[2 * b] = 0.0; di_ai[2 * b + 1] = 0.0;
                age[b] = 0;
                ++resp;
            } else {
                age[b] += 1;
            }
        }
        for (int r = 0; r < 4; ++r) x0[4 * b + r] = st4[r];
        free(tmp);
    }
    if (n_respawn) *n_respawn += resp;
    if (n_iter_sum) *n_iter_sum += iters;
    if (n_fail) *n_fail += fails;
    return ORC_OK;
}

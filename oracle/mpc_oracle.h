/*
 * mpc_oracle.h -- CPU restatement (plain C, fp64) of the reference's per-timestep MPC solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the reported CPU baseline.  The product (libjsim_mpc.so, HIP) never
 * links, imports or falls back to it.
 *
 * What it restates (all citations relative to /root/reference):
 *   main/lib/mpc.py:46-58     smooth_yaw
 *   main/lib/mpc.py:61-82     _get_linear_model_matrix
 *   main/lib/mpc.py:89-112    _calc_ref_trajectory
 *   main/lib/mpc.py:115-129   _predict_motion
 *   main/lib/mpc.py:132-138   _get_xy_cost_mtx_for_orientation
 *   main/lib/mpc.py:141-211   _linear_mpc_control   (QP build; the cvxpy->ECOS solve is replaced, see below)
 *   main/lib/mpc.py:214-242   _iterative_linear_mpc_control (MAX_ITER passes)
 *   main/lib/mpc.py:284-330   MPC.step / get_current_xref_deviation / is_goal
 *   main/lib/mpc_with_speed.py:85-110,276-282  the variant's speed reference cv (orc_mpc_step_cv); its other
 *                             differences are parameter values (weights, MAX_DECEL, speed limit)
 *   main/lib/mpc_jerk.py:59-83,144-199  the acceleration-state variant (orc_params.nx == 5): 5x5 model, free x[4,0]
 *                             carried as decision variable 2T, jerk cost; built by the same generic condensation
 *   main/lib/trajectories.py:100-126  calc_nearest_index_in_direction
 *   main/lib/simulation.py:22-47      Simulation.step (plant, clamps)
 *   main/bicycle/main.py:28-41        Bicycle.step (explicit Euler kinematic bicycle)
 *
 * Parity status:
 *   S1-S3, S5 (reference window, rollout, linearisation, deviation/goal, plant): PINNED by golden
 *     vectors generated in the build container by importing the reference's own Python modules
 *     (tests/golden/make_golden.py -> the .npz files under tests/golden/).
 *   S4, assembly (which cost terms over which t, the reaches_end switches, the constraint list and its order,
 *     main/lib/mpc.py:141-194): PINNED since round 3 by the reference's OWN `_linear_mpc_control`, executed unmodified
 *     under a recording stand-in for the eight cvxpy names it touches (tests/golden/cvxpy_recorder.py).  The sparse
 *     problem it emitted -- 4 x 150 whole MPC.step calls committed as tests/golden/ref_qp_T*.npz, 4 x 1000 more compared
 *     at generation time (ref_qp_sweep.json), fresh ones in tests/test_ref_qp_live.py -- equals this oracle's condensed
 *     (H, g, G, h) after generic elimination of the states to <= 1e-12 (g: 1e-11), row for row in the emitted order,
 *     and the reference's S5 lines (:199-211, :298-303) run on the optimum give this oracle's outputs (u* <= 2e-9,
 *     integers and xref bit for bit, active sets identical wherever the multipliers are unique).
 *   S4, the numbers ECOS returns: the reference calls cvxpy -> ECOS (main/lib/mpc.py:196-197); neither is
 *     installed/pinned anywhere and the reference holds no golden vectors for it, so against ECOS's own
 *     stopping tolerance (1e-8, OPTIMAL_INACCURATE accepted) this stage stays "PARITY UNPINNED".  The QP is
 *     strictly convex (lambda_min(H) >= 2*min(R) > 0), so its optimum is unique; the oracle solves it exactly
 *     (Goldfarb-Idnani dual active set); the stand-in's optimum of the emitted problem comes from an unrelated
 *     solver (null-space elimination + Mehrotra interior point + KKT polish) and agrees to <= 2e-9.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 4
#define ORC_NU 2

/* Mirrors main/config/mpc_config.json + the Simulation class constants + the MPC ctor arguments. */
typedef struct orc_params {
    int32_t T;            /* horizon (mpc_config.json "T") */
    int32_t max_iter;     /* MAX_ITER; 1 in the stock config */
    double dt;            /* MPC ctor dt */
    double dl;            /* MPC ctor dl (course tick) */
    double L;             /* car_dimensions.distance_back_to_front_wheel */
    double w_perp, w_para;
    double R[2];          /* input cost diag */
    double Rd[2];         /* input difference cost diag */
    double Q_v_yaw[2];    /* state cost diag on [v, yaw] */
    double Qf[4];         /* terminal cost diag as in the JSON; multiplied by T inside (mpc.py:28) */
    double R_end[2];      /* diag(10,10) used when reaches_end[t] (mpc.py:181) */
    double max_dsteer;    /* rad/s (deg2rad already applied, mpc.py:37) */
    double max_accel;     /* MAX_ACCEL */
    double max_decel;     /* MAX_DECEL (negative) */
    double max_steer;     /* Simulation.MAX_STEER  (rad) */
    double max_speed;     /* Simulation.MAX_SPEED  (plant clamp) */
    double min_speed;     /* Simulation.MIN_SPEED */
    double min_ref_speed; /* 10/3.6, mpc.py:99 */
    double goal_dis;      /* GOAL_DIS */
    double stop_speed;    /* STOP_SPEED */
    int32_t nx;           /* 4: main/lib/mpc.py; 5: main/lib/mpc_jerk.py (acceleration state, free acc_0, n = 2T + 1 variables) */
    int32_t reserved_;
    double jerk_weight;   /* jerk_penalty_weight (mpc_jerk.py:31), nx == 5 only */
} orc_params;

int orc_nx(const orc_params *p);   /* 4 or 5 */
int orc_nvar(const orc_params *p); /* 2T, or 2T + 1 with the acceleration state */

/* status codes shared with the product C-ABI (include/jsim_mpc.h) */
enum {
    ORC_OK = 0,
    ORC_INFEASIBLE = 1,       /* QP infeasible / active-set did not converge -> reference failure path */
    ORC_NEAREST_ANOMALY = 2,  /* trajectories.py:120 raise Exception("something wrong") */
    ORC_BAD_INPUT = 3
};

/* Canonical inequality-row order (defines "active-constraint indices"); follows the reference's
 * constraint-list order mpc.py:187-194, abs(e)<=b => row +e<=b then row -e<=b:
 *   D : steer rate, t=0..T-2, rows 2t (+), 2t+1 (-)      -> [0, 2T-2)
 *   VU: v_t <= speed, t=0..T                              -> [2T-2, 3T-1)
 *   VL: v_t >= MIN_SPEED, t=0..T                          -> [3T-1, 4T)
 *   AU: a_t <= MAX_ACCEL, t=0..T-1                        -> [4T, 5T)
 *   AL: a_t >= MAX_DECEL, t=0..T-1                        -> [5T, 6T)
 *   S : +-delta_t <= MAX_STEER, rows 6T+2t (+), 6T+2t+1 (-) -> [6T, 8T)
 * Decision variables of the condensed QP: u[2t] = a_t, u[2t+1] = delta_t.
 */
static inline int orc_num_ineq(int T) { return 8 * T; }

void orc_smooth_yaw(double *yaw, int64_t n);

int orc_nearest_index_in_direction(double x, double y, const double *cx, const double *cy,
                                   int64_t ncourse, int64_t start_index, int forward,
                                   int64_t *out_index);

/* xref: [4][T+1] row-major; idx: [T+1]; reaches_end: [T+1] */
int orc_calc_ref_trajectory(const orc_params *p, double sx, double sy, double sv,
                            const double *cx, const double *cy, const double *cyaw, int64_t ncourse,
                            int64_t start_idx, double *xref, int64_t *idx, uint8_t *reaches_end,
                            int64_t *target_ind);

/* Simulation.step: state = [x, y, v, yaw] (MPC order) updated in place */
void orc_plant_step(const orc_params *p, double state[4], double a, double delta);

/* xbar: [4][T+1] row-major, rows x,y,v,yaw */
void orc_predict_motion(const orc_params *p, const double x0[4], const double *oa, const double *od,
                        double *xbar);

void orc_linear_model_matrix(double v, double phi, double delta, double dt, double L,
                             double A[16], double B[8], double C[4]);

/* Condensed QP  min 1/2 u'Hu + g'u  s.t. G u <= h  (n = 2T, m = 8T), equal to the reference's
 * cvxpy problem after eliminating the state variables through the dynamics equalities.
 * H: n*n row-major, g: n, G: m*n row-major, h: m, skip: m (1 = zero row, never a candidate),
 * fresp: [4][T+1] free response (u = 0), Sens: [4*(T+1)][n] sensitivities d z_t / d u.
 * Returns ORC_INFEASIBLE when a zero row is violated (x0.v outside [MIN_SPEED, speed]).
 */
int orc_build_qp(const orc_params *p, const double *xref, const double *xbar, const double x0[4],
                 const uint8_t *reaches_end, double speed, double *H, double *g, double *G,
                 double *h, uint8_t *skip, double *fresp, double *Sens);

/* Goldfarb-Idnani dual active-set on a dense strictly convex QP.  lam: m multipliers (0 for rows not in
 * the final working set).  Returns ORC_OK / ORC_INFEASIBLE. */
int orc_solve_qp(int n, int m, const double *H, const double *g, const double *G, const double *h,
                 const uint8_t *skip, double *u, double *lam, int32_t *n_iter);

/* active-set bitmask from multipliers: bit i set <=> lam[i] > 1e-9 * max(1, ||g||_inf) */
void orc_active_mask(int m, int n, const double *lam, const double *g, uint32_t *mask /*ceil(m/32)*/);

typedef struct orc_step_out {
    /* all caller-allocated */
    double *oa;       /* [T]   */
    double *od;       /* [T]   */
    double *ox;       /* [T+1] */
    double *oy;       /* [T+1] */
    double *ov;       /* [T+1] */
    double *oyaw;     /* [T+1] */
    double *xref;     /* [4][T+1] */
    double *xbar;     /* [4][T+1]  (may be NULL) */
    int64_t *idx;     /* [T+1]     (may be NULL) */
    uint8_t *reaches_end; /* [T+1] (may be NULL) */
    double *lam;      /* [8T]      (may be NULL) */
    uint32_t *active_mask; /* [ceil(8T/32)] (may be NULL) */
    double *H;        /* [n*n] (may be NULL) */
    double *g;        /* [n]   (may be NULL) */
    int64_t target_ind;
    int32_t n_iter;
    int32_t status;
} orc_step_out;

/* One MPC.step for one ego.  state = (x, y, yaw, v) as in lib/simulation.py State;  oa_in/od_in: warm
 * start [T] or NULL (zeros, mpc.py:225-227).  On ORC_INFEASIBLE the outputs oa..oyaw are left
 * untouched (the reference returns None) and the caller applies ai = MAX_DECEL (mpc.py:298-301). */
int orc_mpc_step(const orc_params *p, double sx, double sy, double syaw, double sv,
                 const double *cx, const double *cy, const double *cyaw, int64_t ncourse,
                 int64_t target_ind, double speed, const double *oa_in, const double *od_in,
                 orc_step_out *out);

int orc_mpc_step_cv(const orc_params *p, double sx, double sy, double syaw, double sv, const double *cx,
                    const double *cy, const double *cyaw, const double *cv, int64_t cv_cut, int64_t ncourse,
                    int64_t target_ind, double speed, const double *oa_in, const double *od_in, orc_step_out *out);

/* Batched form used by bench.py's cpu_baseline and the parity tests.  Layouts are the product's
 * (include/jsim_mpc.h): x0 [B][4] = (x,y,v,yaw); oa/od [B][T] in/out; ox.. [B][T+1]; xref [B][4][T+1];
 * active_mask [B][ceil(8T/32)]; paths concatenated with path_off[n_paths+1]. Any output may be NULL.
 * n_threads > 1 uses OpenMP when compiled with it. */
int orc_mpc_step_batch(const orc_params *p, int32_t B, const double *x0, const int32_t *path_id,
                       const int32_t *path_len, const double *speed, const double *cx,
                       const double *cy, const double *cyaw, const int64_t *path_off,
                       int64_t *target_ind, double *oa, double *od, double *ox, double *oy,
                       double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                       int32_t *status, int32_t *n_iter, int32_t n_threads);

int orc_mpc_step_batch_cv(const orc_params *p, int32_t B, const double *x0, const int32_t *path_id,
                          const int32_t *path_len, const double *speed, const double *cx,
                          const double *cy, const double *cyaw, const double *cv, const int32_t *cv_cut,
                          const int64_t *path_off, int64_t *target_ind, double *oa, double *od, double *ox,
                          double *oy, double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                          int32_t *status, int32_t *n_iter, int32_t n_threads);

double orc_xref_deviation(const double *cx, const double *cy, const double *cyaw, int64_t target_ind,
                          double ox0, double oy0);
int orc_is_goal(const orc_params *p, double sx, double sy, double sv, double goal_x, double goal_y,
                int64_t target_ind, int64_t ncourse);

/* n_ticks ticks of the per-vehicle loop (main/scenarios/mpc_intersection.py:99-163) for B independent egos: MPC.step with the
 * carried warm start -> Simulation.step -> goal test / respawn, OpenMP over egos.  All state arrays in/out (layouts of the
 * batched step; di_ai [B][2], age [B]); hist [n_ticks][B][2] may be NULL; the three counters are incremented. */
int orc_closed_loop(const orc_params *p, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id,
                    const int32_t *path_len, const double *speed, const double *cx, const double *cy,
                    const double *cyaw, const int64_t *path_off, int64_t *target_ind, double *oa, double *od,
                    double *di_ai, const double *x0_spawn, const int64_t *target_spawn, int32_t *age,
                    int32_t max_age, double *hist, int64_t *n_respawn, int64_t *n_iter_sum, int64_t *n_fail,
                    int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif

/*
 * jsim_mpc.h -- C-ABI of libjsim_mpc.so: the MI355X (gfx950) batched replacement for the reference's
 * per-timestep MPC solve.  extern "C", plain pointers and sizes, no torch / C++ types.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo root):
 *   jsim_mpc_create      <- module-level config load main/lib/mpc.py:15-39 (main/config/mpc_config.json)
 *                           + MPC.__init__ scalars (dl, dt, car_dimensions.L)   main/lib/mpc.py:246-275
 *   jsim_mpc_set_paths   <- MPC.__init__ cx/cy/cyaw + MPC.set_trajectory_fromarray  main/lib/mpc.py:257-261,279-282
 *                           (truncation = per-ego path_len passed to jsim_mpc_step)
 *   jsim_mpc_step        <- MPC.step -> _iterative_linear_mpc_control -> _calc_ref_trajectory,
 *                           _predict_motion, _linear_mpc_control (cvxpy->ECOS)    main/lib/mpc.py:284-303,214-242,89-211
 *                           with calc_nearest_index_in_direction                  main/lib/trajectories.py:100-126
 *                           and Simulation.step / Bicycle.step for the rollout    main/lib/simulation.py:35-47, main/bicycle/main.py:28-41
 *   jsim_plant_step      <- HistorySimulation.step / Simulation.step (the per-vehicle loop's plant update)
 *                           main/lib/simulation.py:35-47,58-61 ; main/scenarios/mpc_intersection.py:163
 *   jsim_loop_advance    <- the rest of the loop body: plant update, history, `if mpc.is_goal(state): break`
 *                           main/scenarios/mpc_intersection.py:99-101,163 ; main/lib/simulation.py:64-88 (History)
 *   jsim_mpc_run_ticks   <- `for i in itertools.count():` itself, main/scenarios/mpc_intersection.py:99 (K iterations per call)
 *   jsim_mpc_set_path_speed / _set_speed_cutoff / _update_cfg <- the data-only MPC variants main/lib/mpc_with_speed.py,
 *                           main/lib/mpc_sensitivity.py
 *   jsim_mpc_xref_deviation_goal <- MPC.get_current_xref_deviation / MPC.is_goal  main/lib/mpc.py:305-330
 *
 * Conventions
 *   - All array arguments of jsim_mpc_step / jsim_plant_step / ..._goal are DEVICE pointers (HBM), caller-owned,
 *     no allocation inside those calls; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     jsim_mpc_set_paths takes HOST pointers (one-time upload into the context's resident path table).
 *   - Ego-major rows: a wavefront owns one ego, so every per-ego array is one contiguous row
 *     (x0 [B][4], oa/od [B][T], ox/oy/ov/oyaw [B][T+1], xref [B][4][T+1], active_mask [B][ceil(8T/32)]).
 *   - State order is the MPC's [x, y, v, yaw] (main/lib/mpc.py:291), NOT State's (x, y, yaw, v).
 *   - Return value: 0 on success, negative on error (-EINVAL style); never throws.  jsim_last_error()
 *     returns a message for the last failing call on that context (or globally when ctx == NULL).
 *   - Per-ego status (int32): 0 ok; 1 QP infeasible / not converged (reference: prints
 *     "Error: Cannot solve mpc..." and returns None, main/lib/mpc.py:207-209 -- outputs ox..oyaw are left
 *     untouched, oa/od are zeroed = the reference's cold start after None, caller applies ai = MAX_DECEL);
 *     2 nearest-index anomaly (reference raises Exception("something wrong"), main/lib/trajectories.py:120 --
 *     nothing but status is written for that ego).
 *   - Active-constraint indices (bit i of active_mask) use the canonical row order of the reference's
 *     constraint list main/lib/mpc.py:187-194:
 *       D  steer-rate rows  2t (+), 2t+1 (-), t = 0..T-2        [0, 2T-2)
 *       VU v_t <= speed,     t = 0..T                            [2T-2, 3T-1)
 *       VL v_t >= MIN_SPEED, t = 0..T                            [3T-1, 4T)
 *       AU a_t <= MAX_ACCEL                                      [4T, 5T)
 *       AL a_t >= MAX_DECEL                                      [5T, 6T)
 *       S  +-delta_t <= MAX_STEER rows 6T+2t (+), 6T+2t+1 (-)    [6T, 8T)
 *     bit set <=> the row is in the final working set with multiplier > 1e-9 * max(1, ||g||_inf).
 */
#ifndef JSIM_MPC_H
#define JSIM_MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSIM_ABI_VERSION 2 /* 2: jsim_cfg ends with nx, reserved_, jerk_weight */
#define JSIM_MAX_T 48 /* two (2T)x(2T+1) fp64 tiles must fit the 160 KiB LDS of one CU */

typedef struct jsim_cfg {
    int32_t T;        /* horizon; 1 <= T <= JSIM_MAX_T (stock 13) */
    int32_t max_iter; /* MAX_ITER (1..16; stock value 1): linearisation passes per step, main/lib/mpc.py:231 */
    double dt, dl, L;
    double w_perp, w_para;
    double R[2], Rd[2], Q_v_yaw[2];
    double Qf[4];     /* as in the JSON; multiplied by T inside (main/lib/mpc.py:28) */
    double R_end[2];  /* diag(10,10), main/lib/mpc.py:181 */
    double max_dsteer; /* rad/s */
    double max_accel, max_decel;
    double max_steer, max_speed, min_speed; /* Simulation.* main/lib/simulation.py:23-25 */
    double min_ref_speed;                   /* 10/3.6, main/lib/mpc.py:99 */
    double goal_dis, stop_speed;
    int32_t nx;          /* 4: main/lib/mpc.py.  5: main/lib/mpc_jerk.py -- acceleration state x[4] with A[4][4] = 1, A[2][4] = dt,
                          * B[4][0] = dt (:67-78), x[4,0] free (:193): the condensed QP has 2T + 1 variables (acc_0 last), oa =
                          * u[0,:] is the input of that state; runs on the LDS-resident kernel */
    int32_t reserved_;
    double jerk_weight;  /* jerk_penalty_weight of (x[4,t+1] - x[4,t])^2, t < T-1 (main/lib/mpc_jerk.py:31,190); nx == 5 only */
} jsim_cfg;

typedef struct jsim_ctx jsim_ctx;

enum { JSIM_OK = 0, JSIM_INFEASIBLE = 1, JSIM_NEAREST_ANOMALY = 2 };

int jsim_abi_version(void);
const char *jsim_last_error(const jsim_ctx *ctx);

int jsim_mpc_create(const jsim_cfg *cfg, int device_id, jsim_ctx **out);
void jsim_mpc_destroy(jsim_ctx *ctx);

/* HOST pointers.  cyaw must already be smoothed (MPC.__init__ does it on the host, in place). */
int jsim_mpc_set_paths(jsim_ctx *ctx, const double *cx, const double *cy, const double *cyaw,
                       const int64_t *path_off /*[n_paths+1]*/, int32_t n_paths);

/* One MPC.step for B egos.  In/out: target_ind [B], oa/od [B][T] (warm start in, solution out).
 * Optional outputs may be NULL: ox, oy, ov, oyaw, xref, active_mask, n_iter. */
int jsim_mpc_step(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                  const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa,
                  double *od, double *ox, double *oy, double *ov, double *oyaw, double *xref,
                  uint32_t *active_mask, int32_t *status, int32_t *n_iter, void *stream);

/* Same, plus stage-level outputs for parity tests (any may be NULL):
 * xbar [B][4][T+1], ref_idx [B][T+1], H [B][n][n] (lower triangle valid), g [B][n], lam [B][8T]; n = 2T (2T + 1 with nx == 5). */
int jsim_mpc_step_debug(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                        const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa,
                        double *od, double *ox, double *oy, double *ov, double *oyaw, double *xref,
                        uint32_t *active_mask, int32_t *status, int32_t *n_iter, double *xbar,
                        int64_t *ref_idx, double *H, double *g, double *lam, void *stream);

/* Controller output + plant update for B egos: (di, ai) = (od[b][0], oa[b][0]) when status[b]==0, else
 * ai = MAX_DECEL and di = di_prev[b] (main/lib/mpc.py:298-303); then Simulation.step on x0 in place.
 * di_ai [B][2] in/out (previous steer in, applied (steer, accel) out). */
int jsim_plant_step(jsim_ctx *ctx, int32_t B, double *x0, const double *oa, const double *od,
                    const int32_t *status, double *di_ai, void *stream);

/* One tick of closed-loop bookkeeping for a batch, the rest of the per-vehicle loop body
 * (main/scenarios/mpc_intersection.py:99-163): (di, ai) selection + plant step as jsim_plant_step, optional
 * history record hist[tick][B][2] (tick = a device counter this call increments, so the call can be replayed
 * from a hipGraph), and replacement of finished egos: an ego for which MPC.is_goal holds on its new state
 * (main/lib/mpc.py:314-330; the loop's `break`, :101) or whose age reaches max_age ticks (<=0: never) restarts
 * from x0_spawn/target_spawn with a cold controller (oa = od = 0, di = ai = 0). n_respawn (optional) counts them. */
int jsim_loop_advance(jsim_ctx *ctx, int32_t B, double *x0, double *oa, double *od, const int32_t *status,
                      double *di_ai, int64_t *target_ind, const int32_t *path_id, const int32_t *path_len,
                      const double *x0_spawn, const int64_t *target_spawn, int32_t *age, int32_t max_age,
                      double *hist, int32_t *tick, int32_t hist_cap, uint64_t *n_respawn, void *stream);

/* ---- MPC variants that differ from main/lib/mpc.py only in data ----
 * jsim_mpc_set_path_speed: per-point speed reference cv (HOST array aligned with cx/cy/cyaw of jsim_mpc_set_paths, or NULL
 *   to switch it off): xref[2] = cv[idx] instead of 0, main/lib/mpc_with_speed.py:85-110.
 * jsim_mpc_set_speed_cutoff: caller-owned DEVICE array [B] (or NULL): ego b's reference is zeroed from that path index on
 *   (set_trajectory_fromarray(trajectory, cutoff_idx), main/lib/mpc_with_speed.py:276-282); entries < 0 mean no cut-off.
 * jsim_mpc_update_cfg: replace weights / limits between solves (same T), which is what main/lib/mpc_sensitivity.py does by
 *   re-reading its JSON inside every solve (:153-166).  The other differences of the variants are parameter values:
 *   mpc_with_speed uses w_perp = 10, Q_v_yaw = (20, 0.5), MAX_DECEL = -5 and speed limit Simulation.MAX_SPEED. */
int jsim_mpc_set_path_speed(jsim_ctx *ctx, const double *cv);
int jsim_mpc_set_speed_cutoff(jsim_ctx *ctx, const int32_t *cv_cut);
int jsim_mpc_update_cfg(jsim_ctx *ctx, const jsim_cfg *cfg);

/* Per-ego weights and limits: a whole sensitivity sweep (main/scenarios/mpc_sensitivity_analysis*.py run the loop once per
 * parameter set, rewriting config/mpc_config_sensitivity.json in between) as ONE batch -- ego b solves with its own row of
 * cfg, a caller-owned DEVICE array [B][JSIM_EGO_CFG_DOUBLES] in the JSON's own units:
 *   { w_perp, w_para, R[0], R[1], Rd[0], Rd[1], Q_v_yaw[0], Q_v_yaw[1], Qf[0..3] (unscaled; x T inside, mpc.py:28),
 *     MAX_DSTEER [rad/s], MAX_ACCEL, MAX_DECEL, reserved }.
 * Everything else (T, dt, dl, L, R_end, speed limits, goal test) stays with the context's jsim_cfg.  NULL switches back. */
#define JSIM_EGO_CFG_DOUBLES 16
int jsim_mpc_set_ego_config(jsim_ctx *ctx, const double *cfg);

/* ---- the loop glue that produces the truncated path, main/scenarios/mpc_intersection.py:104-140 ----
 * jsim_loop_set_geometry: the car's two collision circles (offsets of their centres from the rear axle along the body
 *   axis, radius) = car_dimensions.circle_centers / .radius, main/lib/car_dimensions.py:62-79; precomputes the circle
 *   centres of every path point (main/lib/trajectories.py:11-55).  Call after jsim_mpc_set_paths or before; either order.
 * jsim_loop_predict_obstacles: MovingObstaclesPrediction.state_prediction for n_obs obstacles (<= 8), n_steps samples
 *   (<= 64; the loop uses len(arange(0, 7.0, DT)) = 35), main/lib/moving_obstacles_prediction.py:21-47.
 *   obst [n_obs][6] = (x, y, v, yaw, a, steer) as MovingObstacle*.get() returns them; pred [n_obs][n_steps][3] = (x, y, yaw).
 * jsim_loop_pre_tick: per ego -- progress index traj_idx on the FULL path (in/out; skipped when it sits on the last point
 *   of the previous truncated path, :106-109), resample_curve of the remaining path with the accelerate-to-MAX_SPEED
 *   spacing (:111-120, main/lib/trajectories.py:58-86), check_collision_moving_cars against the predictions with
 *   +-frame_window frame offsets (main/lib/collision_avoidance.py:68-124), get_cutoff_curve_by_position_idx minus `margin`
 *   (:133-140, main/lib/collision_avoidance.py:168-180).  path_len [B] out = cut-off length (full length if no collision) --
 *   exactly what jsim_mpc_step takes.  prev_path_len [B]: previous tick's path_len, -1 before the first tick.
 *   status: 0 ok, 2 nearest-index anomaly, 4 resampled path longer than the kernel's 320-point table. */
int jsim_loop_set_geometry(jsim_ctx *ctx, double cc_front, double cc_rear, double radius);
/* n_ticks ticks of the WHOLE scenario loop (main/scenarios/mpc_intersection.py:99-163) for B egos and n_obs scripted obstacle
 * vehicles: obstacle get() -> prediction -> progress index / resample / collision / cut-off (jsim_loop_pre_tick) -> MPC.step ->
 * plant, history, goal (jsim_loop_advance) -> obstacle step().  Arguments as in jsim_mpc_run_ticks, jsim_loop_pre_tick
 * (traj_idx, prev_path_len in/out; path_len, col_flag, pre_status out) and jsim_loop_obstacles (obs_state in/out, obs_param,
 * obs_get [n_obs][6] out).  speed_cutoff = 1: the glue of main/scenarios/mpc_intersection_new_ref.py:122-139 -- the path
 * stays whole (path_len must hold the full lengths) and the cut-off index goes to the buffer registered with
 * jsim_mpc_set_speed_cutoff.  With a register kernel (T = 13, 15, 16, 20, 25, 30, 32, 40) and MAX_ITER = 1 it is three launches: obstacles rolled forward n_ticks ticks, their predictions for
 * every tick, and ONE fused launch in which every ego's wave runs its own glue + solve + plant n_ticks times; otherwise the
 * same ticks as separate launches.  Identical results either way. */
int jsim_loop_run_scenario(jsim_ctx *ctx, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id, int32_t *path_len,
                           const double *speed, int64_t *target_ind, double *oa, double *od, double *ox, double *oy, double *ov,
                           double *oyaw, double *xref, uint32_t *active_mask, int32_t *status, int32_t *n_iter, double *di_ai,
                           const double *x0_spawn, const int64_t *target_spawn, int32_t *age, int32_t max_age, double *hist,
                           int32_t *tick, int32_t hist_cap, uint64_t *n_respawn, int64_t *traj_idx, int32_t *prev_path_len,
                           int32_t *col_flag, int32_t *pre_status, int32_t frame_window, int32_t margin, int32_t n_obs,
                           double *obs_state, const double *obs_param, double *obs_get, int32_t n_steps, int32_t speed_cutoff,
                           void *stream);
/* Obstacles of another shape than the ego (main/scenarios/overtaking_cyclist_bidirectional_road.py:122-133: the cyclist's
 * BicycleRealDimensions): their two circles and wheelbase -- used by the prediction (MovingObstaclesPrediction(...,
 * car_dimensions=bicycle_dimensions)) and by the collision rows of check_collision_moving_bicycle,
 * main/lib/collision_avoidance.py:126-166 (min_distance = car radius + bicycle radius, same row order).  Without this call
 * the obstacles have the ego's geometry (check_collision_moving_cars). */
int jsim_loop_set_obstacle_geometry(jsim_ctx *ctx, double cc_front, double cc_rear, double radius, double wheelbase);
/* Scripted obstacle vehicles of main/lib/moving_obstacles.py -- MovingObstacleTIntersection (:166-232, kind 0),
 * MovingObstacleRoundabout (:28-124, kind 1), MovingObstacleArterial (:126-164, kind 2): state [n_obs][4] = (xc, yc, theta,
 * counter) in/out, param [n_obs][8] = (direction +-1, turning 0/1, speed, offset seconds (<= 0: none), x_turn, dt, kind,
 * initial_speed).  Writes the `get()` tuples (x, y, v, yaw, 0, steer) of the CURRENT state to `get` (may be NULL) -- the input
 * of jsim_loop_predict_obstacles -- and, when do_step != 0, advances the state by one `step()`. */
int jsim_loop_obstacles(jsim_ctx *ctx, int32_t n_obs, double *state, const double *param, double *get, int32_t do_step,
                        void *stream);
int jsim_loop_predict_obstacles(jsim_ctx *ctx, int32_t n_obs, const double *obst, int32_t n_steps, double *pred, void *stream);
int jsim_loop_pre_tick(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id, int64_t *traj_idx,
                       const int32_t *prev_path_len, int32_t *path_len, int32_t *col_flag, double *col_xy,
                       int32_t *first_idx, int32_t *status, int32_t frame_window, int32_t margin,
                       int32_t *dbg_res_idx /*[B][320]*/, int32_t *dbg_n_res, void *stream);

/* n_ticks consecutive closed-loop ticks, each = jsim_mpc_step followed by jsim_loop_advance, with identical results.
 * For the horizons that have the fused register-resident kernel (T = 13, 20) this is ONE launch in which every
 * wavefront runs all n_ticks for its own ego (egos are independent, so none waits for the slowest solve of a tick);
 * other horizons fall back to 2 * n_ticks launches.  The per-step outputs (ox .. n_iter) hold the LAST tick's values. */
int jsim_mpc_run_ticks(jsim_ctx *ctx, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id,
                       const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa, double *od,
                       double *ox, double *oy, double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                       int32_t *status, int32_t *n_iter, double *di_ai, const double *x0_spawn,
                       const int64_t *target_spawn, int32_t *age, int32_t max_age, double *hist, int32_t *tick,
                       int32_t hist_cap, uint64_t *n_respawn, void *stream);

/* Launch order of the fused closed-loop launches (jsim_mpc_run_ticks, jsim_loop_run_scenario).  No reference counterpart:
 * the reference runs one vehicle per process (main/scenarios/mpc_intersection.py:99); this is scheduling of the batch only.
 * With 512 <= B <= 65536 workgroup b of a launch is given the ego that ranked b-th by the active-set iterations it needed in
 * the previous launch of this context (most first), so that the egos far from their paths do not start last; results are
 * those of the identity order bit for bit.  On by default (JSIM_LAUNCH_ORDER=0 in the environment turns it off);
 * jsim_mpc_set_launch_order overrides the environment for one context.  jsim_mpc_get_launch_order copies the order used by
 * the last launch and the iteration counts that launch recorded to HOST arrays [B] (either may be NULL); it synchronises. */
int jsim_mpc_set_launch_order(jsim_ctx *ctx, int32_t enabled);
int jsim_mpc_get_launch_order(jsim_ctx *ctx, int32_t B, int32_t *order, uint32_t *work);

/* Per-ego running totals of active-set iterations of the fused launches (jsim_mpc_run_ticks, jsim_loop_run_scenario) since the
 * last reset.  No reference counterpart (the reference never sees its solver's iterations, main/lib/mpc.py:196-199); this is
 * the measurement hook bench.py takes mean iterations, algorithmic flops and the straggler statistic of the TIMED launches from.
 * Copies totals [B] to a HOST array (may be NULL) and, with reset != 0, clears them; it synchronises. */
int jsim_mpc_iter_totals(jsim_ctx *ctx, int32_t B, uint64_t *totals, int32_t reset);

/* ---- the route planner: A* over motion primitives, a batch of route queries at once ----
 * Replaces MotionPrimitiveSearch(scenario, car_dimensions, mps, margin).run() -- main/lib/mp_search_ww_generic.py:136-257 with
 * main/lib/a_star.py:31-78 and main/lib/obstacles.py:157-176 -- which every scenario script calls once before its loop
 * (main/scenarios/mpc_intersection.py:63-64) and whose (M, 3) [x, y, yaw] trajectory becomes the MPC's path.  One wavefront
 * searches one route; n_routes routes in one launch.  HOST pointers in and out (a one-time precompute: the output is what
 * jsim_mpc_set_paths takes).
 *   start, goal [R][3] (x, y, theta); goal_box [R][4] = (x1, y1, x2, y2) of the scenario's goal_area box; tol [R] =
 *   allowed_goal_theta_difference.  Obstacles as half-plane sets a x + b y + c <= 0 (Obstacle.to_convex(margin)): hp [.][3],
 *   hp_off [n_obs_total + 1] per obstacle, route_obs_off [R + 1] = each route's obstacles.  Primitives: mp_pts
 *   [n_prim][n_pts][3], mp_len [n_prim] (total_length), cc_pts [.][2] / cc_off [n_prim + 1] = the collision-check points of each
 *   primitive in its own frame (_create_collision_points, :118-136).  wh [5] = (dist, theta, steering, obstacle, center) of the
 *   heuristic, wc [4] = (dist, steering, obstacle, center) of the edge cost (:29-33; the scenarios use the defaults
 *   (1, 2.7, 15, 0, 0) / (1, 5, 0.1, 0)).
 *   node_cap: search workspace per route in nodes, ~100 B each (16384 covers the reference's 18 standard routes a hundred
 *   times over; the longest lane change of the two-lane scenario makes 85k; a route that reports status 4 wants more).
 *   Out: status [R] (0 found; 1 no solution -- the reference raises Exception("No solution found."); 4 search workspace
 *   exhausted; 5 obstacle / primitive set too large for the kernel; 6 path longer than max_path), cost [R], n_prims [R],
 *   prims [R][max_path] (primitive index per segment), nodes [R][max_path + 1][3], traj [R][max_path * (n_pts - 1)][3] (the first
 *   n_prims * (n_pts - 1) rows are the trajectory, path_to_full_trajectory :245-257), n_expanded [R]. */
int jsim_plan_routes(int device_id, int32_t n_routes, const double *start, const double *goal, const double *goal_box,
                     const double *tol, const double *hp, const int32_t *hp_off, int32_t n_obs_total,
                     const int32_t *route_obs_off, const double *mp_pts, const double *mp_len, int32_t n_prim, int32_t n_pts,
                     const double *cc_pts, const int32_t *cc_off, const double *wh, const double *wc, int32_t max_path,
                     int32_t node_cap, int32_t *status, double *cost, int32_t *n_prims, int32_t *prims, double *nodes, double *traj,
                     int32_t *n_expanded);

/* ---- the job's one exchange: the final trajectory gather over RCCL / xGMI ----
 * The reference has no multi-process code at all (its only multi-ego code is the serial Python loop of
 * main/scenarios/interactive_mpc.py:119-172); egos are independent (main/lib/mpc.py:141-211), so ranks own contiguous shards
 * of the ego batch and exchange nothing while solving.  These four calls are the gather of the per-rank result blocks:
 *   jsim_comm_unique_id: fills a 128-byte ncclUniqueId (rank 0 calls it and hands the bytes to the other ranks by any means).
 *   jsim_comm_init:      ncclCommInitRank on the context's device; the communicator then belongs to the context.
 *   jsim_mpc_gather:     ncclAllGather of bytes_per_rank bytes from `local` into `out` [n_ranks * bytes_per_rank] (DEVICE pointers)
 *                        on `stream`; comm == NULL uses the context's communicator, otherwise a caller-owned ncclComm_t.
 *   jsim_comm_destroy:   ncclCommDestroy (also done by jsim_mpc_destroy).
 * librccl is loaded on first use (dlopen; override with JSIM_RCCL_LIB): libjsim_mpc.so itself links only libamdhip64. */
int jsim_comm_unique_id(void *id128);
int jsim_comm_init(jsim_ctx *ctx, const void *id128, int32_t n_ranks, int32_t rank);
int jsim_mpc_gather(jsim_ctx *ctx, void *comm, const void *local, void *out, size_t bytes_per_rank, void *stream);
int jsim_comm_destroy(jsim_ctx *ctx);

/* deviation [B] (needs ox[b][0], oy[b][0]) and is_goal [B] (int32 0/1); goal = last point of the FULL path. */
int jsim_mpc_xref_deviation_goal(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                                 const int32_t *path_len, const int64_t *target_ind, const double *ox,
                                 const double *oy, double *deviation, int32_t *is_goal, void *stream);

#ifdef __cplusplus
}
#endif
#endif

// jsim_mpc.hip -- MI355X (gfx950 / CDNA4) batched receding-horizon MPC step + the C-ABI of
// include/jsim_mpc.h.  Hand-written HIP, no CUDA-compat layers.  Compile with -ffp-contract=off:
// stages S1-S3 follow numpy's operation order (integer outputs are bit-exact); where a fused
// multiply-add is wanted it is written as fma().
//
// One wavefront (64 lanes) owns one ego for the whole step:
//   S1  reference window      nearest path point (3 smallest distances, wave arg-min), travel -> idx, gather xref
//                             reference: main/lib/mpc.py:89-112, main/lib/trajectories.py:100-126
//   S2  nonlinear rollout     lane t holds time step t; transcendental work lane-parallel, the three
//                             running sums sequential (same order as the reference's python loop)
//                             reference: main/lib/mpc.py:115-129, main/lib/simulation.py:35-47, main/bicycle/main.py:28-41
//   S3  linearisation         A_t,B_t,C_t coefficients per lane (delta_bar == 0 => v, yaw rows are integrators)
//                             reference: main/lib/mpc.py:61-82
//   S4a condense + Hessian    sensitivities generated on the fly from prefix sums; H = 2 S'QS on the fp64
//                             matrix cores (v_mfma_f64_16x16x4_f64, K = the 4 state components of one time
//                             step), block-triangular zero tiles skipped; g, R, Rd terms on the VALU
//                             reference: main/lib/mpc.py:141-186 (cost), :176-178,189 (dynamics eliminated)
//   S4b exact QP solve        Goldfarb-Idnani dual active set (entering row: largest viol^2 / n'H^-1 n); H -> L (Cholesky) -> J = L^-T and the working
//                             set's R factor live in LDS (lane i owns row i); constraints are evaluated from
//                             their structure (boxes, steer-rate differences, speed = prefix sums of accel)
//                             reference: main/lib/mpc.py:187-199 (constraints; cvxpy->ECOS solve replaced)
//   S5  outputs               predicted states of the linearised model, warm start, target_ind, active mask
//                             reference: main/lib/mpc.py:200-205, 293-303
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "jsim_mpc.h"

#define JSIM_VIOL_TOL 1e-10
// Entering-row key: viol^2 / (n'H^-1 n) with its low 20 mantissa bits cleared (keys closer than 2^-32 relative count as
// tied and go to the lowest canonical row id, < 512).  The register kernels store 511 - id in the lowest 9 of them.
// 20 bits: rows that COINCIDE (v_1 <= speed and a_0 <= MAX_ACCEL when v_0 = speed - MAX_ACCEL * dt) have mathematically equal keys
// whose roundings differ by a few ulps; with 9 cleared bits one such pair in ~100 straddled a truncation boundary and the kernel and
// the oracle entered different rows of the pair (same u*, different but equally valid multipliers).
#define JSIM_KEY_MASK 0xFFFFF
__device__ __forceinline__ double jsim_key_trunc(double k)
{
    return __hiloint2double(__double2hiint(k), __double2loint(k) & ~JSIM_KEY_MASK);
}
// Dual ratio test: a step length keeps its upper 57 bits (the low 7 are cleared; the register kernel stores the
// working-set position there), steps closer than 2^-45 relative are ties -> lowest position.
__device__ __forceinline__ double jsim_ratio_trunc(double t)
{
    return __hiloint2double(__double2hiint(t), __double2loint(t) & ~127);
}
__device__ __forceinline__ double jsim_ratio_pack(double t, int k)
{
    return __hiloint2double(__double2hiint(t), (__double2loint(t) & ~127) | k);
}
__device__ __forceinline__ double jsim_key_pack(double k, int id)
{
    return __hiloint2double(__double2hiint(k), (__double2loint(k) & ~JSIM_KEY_MASK) | (511 - id));
}
#define JSIM_DEP_TOL 1e-18
#define JSIM_ACT_TOL 1e-9
#define JSIM_FEAS_TOL 1e-8

typedef double v4d __attribute__((ext_vector_type(4)));

struct KP {
    int T, n, ld, B;
    int pass; // 0: first linearisation; k >= 1: re-linearisation pass k of MAX_ITER (mpc.py:231): skips egos that failed,
              // takes the travel distances from the previous pass's predicted speeds (P.ov) and accumulates n_iter
    double dt, dl, L, w_perp, w_para;
    double R0, R1, Rd0, Rd1, Qv, Qyaw, Qf0, Qf1, Qf2, Qf3, Re0, Re1; // Qf* already multiplied by T
    double dmax, amax, amin, smax, vmax_plant, vmin, vref_min;
    double jerkw; // acceleration-state variant (main/lib/mpc_jerk.py, n = 2T + 1): jerk_penalty_weight
    const double2 *pxy;
    const double *pyaw;
    const double *pcv;   // per-point speed reference of the mpc_with_speed variant, or NULL (xref[2] = 0, mpc.py:107)
    const int *cv_cut;   // [B] index from which that reference is zeroed, or NULL
    const double *pe;    // [B][JSIM_EGO_CFG_DOUBLES] per-ego weights / limits (jsim_mpc_set_ego_config), or NULL
    const long long *poff;
    const double *x0;
    const int *path_id;
    const int *path_len;
    const double *speed;
    long long *target_ind;
    double *oa, *od, *ox, *oy, *ov, *oyaw, *xref;
    unsigned *amask;
    int *status;
    int *n_iter;
    double *dbg_xbar;
    long long *dbg_idx;
    double *dbg_H, *dbg_g, *dbg_lam;
    long long *dbg_clk; // diagnostic build only (-DJSIM_STAMPS): [B][16] s_memtime stamps at phase boundaries
    int dbg_max_gi;     // diagnostic builds only (-DJSIM_STAMPS / -DJSIM_SPAN, env JSIM_DEBUG_MAX_GI): stop the active-set loop after this many
                        // outer iterations (results are then NOT the optimum); -1 = off
};

// per-ego weights / limits over the launch constants (uniform loads: the row depends on the workgroup only)
template <class KPT>
__device__ __forceinline__ void jsim_apply_ego_cfg(KPT &P, const double *pe, int ego, int T, double dt)
{
    const double *w = pe + (size_t)ego * JSIM_EGO_CFG_DOUBLES;
    P.w_perp = w[0]; P.w_para = w[1]; P.R0 = w[2]; P.R1 = w[3]; P.Rd0 = w[4]; P.Rd1 = w[5]; P.Qv = w[6]; P.Qyaw = w[7];
    P.Qf0 = w[8] * T; P.Qf1 = w[9] * T; P.Qf2 = w[10] * T; P.Qf3 = w[11] * T;
    P.dmax = w[12] * dt; P.amax = w[13]; P.amin = w[14];
}

// ---------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ---------------------------------------------------------------------------------------------------
// Between two phases of a ONE-wave workgroup that hand data from lane to lane through LDS.  The hardware needs nothing -- a wave's LDS
// operations are carried out in program order -- and the compiler only has to keep that order: a wavefront-scope fence.
// __syncthreads() (rounds 1-2) drops its s_barrier in a 64-thread workgroup but keeps `s_waitcnt lgkmcnt(0)`: every hand-over waited
// for its writes to COMPLETE before the reads were even issued.  (-DJSIM_LDS_SYNC_BARRIER: the old form, for A/B runs.)
#ifdef JSIM_LDS_SYNC_BARRIER
#define LDS_SYNC() __syncthreads()
#else
#define LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

#ifdef JSIM_STAMPS
#define STAMP(i) do { if (P.dbg_clk && lane == 0) P.dbg_clk[(size_t)ego * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#elif defined(JSIM_SPAN) /* only the first and the last stamp of a step: two s_memtime per solve, nothing else changes */
#define STAMP(i) do { if (((i) == 0 || (i) == 11) && P.dbg_clk && lane == 0) P.dbg_clk[(size_t)ego * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double rdlane(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uni(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v)
{
    return ((long long)uni((int)(v >> 32)) << 32) | (long long)(unsigned)uni((int)v);
}

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// lexicographic arg-min on (v, id): smaller v, ties -> smaller id
__device__ __forceinline__ void wargmin(double &v, int &id)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double ov = __shfl_xor(v, o);
        int oi = __shfl_xor(id, o);
        bool take = (ov < v) || (ov == v && oi < id);
        v = take ? ov : v;
        id = take ? oi : id;
    }
}
// arg-max on (v, id): larger v, ties -> smaller id
__device__ __forceinline__ void wargmax(double &v, int &id)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double ov = __shfl_xor(v, o);
        int oi = __shfl_xor(id, o);
        bool take = (ov > v) || (ov == v && oi < id);
        v = take ? ov : v;
        id = take ? oi : id;
    }
}
// exclusive prefix sum over lanes
__device__ __forceinline__ double wexscan(double v, int lane)
{
    double s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(s, o);
        if (lane >= o) s += t;
    }
    double e = __shfl_up(s, 1);
    return lane == 0 ? 0.0 : e;
}

// ---------------------------------------------------------------------------------------------------
// the step kernel.  RPL = rows of the n x n factors owned by one lane (n <= 64*RPL).
// ---------------------------------------------------------------------------------------------------
enum { TQ_PA = 0, TQ_PB, TQ_PAP, TQ_PBP, TQ_KT, TQ_QXX, TQ_QXY, TQ_QYY, TQ_QV, TQ_QYAW, TQ_QEX, TQ_QEY,
       TQ_QEV, TQ_QEYAW, TQ_REND, TQ_ONE, TQ_ZERO /* constant rows for the register kernels' operand tables */, TQ_COUNT };

// n = 2T decision variables (2T + 1 with the free acc_0 of the acceleration-state variant); ld = row stride of J / R (odd);
// vectors are sized nv = n rounded up to even so that each stays 16-byte aligned
__host__ __device__ static inline int jsim_nvar(int T, int jerk) { return 2 * T + (jerk ? 1 : 0); }
__host__ __device__ static inline int jsim_ld(int n) { return (n + 1) | 1; }
__host__ __device__ static inline size_t jsim_lds_doubles(int T, int jerk)
{
    const size_t n = (size_t)jsim_nvar(T, jerk), ld = (size_t)jsim_ld((int)n), nv = (n + 1) & ~(size_t)1, tp = (size_t)T + 2;
    //      Jm + Rm (even)           dvec uvec      lamv gsv     rdg  ldg  actv(int)   mask                                   tq
    return ((2 * n * ld + 1) & ~(size_t)1) + nv + (nv + 4) + nv + 2 * nv + nv + nv + nv / 2 + 2 + ((8 * (size_t)T + 31) / 32 + 1) / 2 + 1 +
           TQ_COUNT * tp;
}

// JERK: the acceleration-state variant of main/lib/mpc_jerk.py.  Decision vector z = [u0_0, delta_0, .., u0_{T-1}, delta_{T-1}, acc_0]:
// the state [x, y, v, yaw] responds to the EFFECTIVE accelerations  a~_t = acc_t + u0_t = acc_0 + u0_t + dt * sum_{r<t} u0_r
// (v_{t+1} = v_t + dt acc_t + dt u0_t, acc_{t+1} = acc_t + dt u0_t; :67-78) exactly as the stock model responds to a_t, so the
// state-cost Hessian / gradient are the stock ones in (a~, delta) coordinates, carried to z by  a~ = E z  (suffix sums down
// the a-rows and a-columns); the speed rows become  dt * sum_{s<t} A_s  with  A_s = J[acc_0] + J[2s] + dt * sum_{r<s} J[2r].
template <int RPL, bool JERK>
__global__ __launch_bounds__(64) void mpc_step_kernel(const KP Pin)
{
    KP P = Pin;
    if (Pin.pe && (int)blockIdx.x < Pin.B) jsim_apply_ego_cfg(P, Pin.pe, blockIdx.x, Pin.T, Pin.dt);
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int ego = blockIdx.x;
    if (ego >= P.B) return;
    if (P.pass > 0 && P.status[ego] != JSIM_OK) return; // failed in an earlier pass of this step: the failure stands
    const int it_base = (P.pass > 0 && P.n_iter) ? P.n_iter[ego] : 0;
    const int T = P.T, n = P.n, ld = P.ld, tp = T + 2;
    const int nh = 2 * T;             // inputs (u0 / a, delta) x T; n = nh + 1 with the free acc_0 (JERK)
    const int nv = (n + 1) & ~1;
    const int MW = (8 * T + 31) >> 5;

    double *Jm = lds;
    double *Rm = Jm + (size_t)n * ld;
    double *dvec = lds + (((size_t)2 * n * ld + 1) & ~(size_t)1);
    double *uvec = dvec + nv;         // nv + 4
    double *lamv = uvec + nv + 4;     // nv
    double *gsv = lamv + nv;          // 2nv  Givens (c, s)
    double *rdg = gsv + 2 * nv;       // nv   1 / R[k][k]
    double *ldg = rdg + nv;           // nv   1 / L[k][k]
    int *actv = (int *)(ldg + nv);    // n ints
    unsigned *maskw = (unsigned *)(ldg + nv + nv / 2 + 1);
    double *tq = ldg + nv + nv / 2 + 2 + (MW + 1) / 2 + 1;

    // ------------------------------------------------------------------ inputs (wave-uniform)
    const int pid = P.path_id[ego];
    const long long off = P.poff[pid];
    const long long M = P.path_len[ego];
    const long long s0 = P.target_ind[ego];
    const double sx = P.x0[4 * ego + 0], sy = P.x0[4 * ego + 1], sv = P.x0[4 * ego + 2], syaw = P.x0[4 * ego + 3];
    const double speed = P.speed[ego];
    const double dt = P.dt;

    STAMP(0);
    // ------------------------------------------------------------------ S1: nearest index in direction
    long long tind = s0;
    int status = JSIM_OK;
    {
        long long len = M - s0;
        if (len < 0) len = 0;
        if (M < 1) {
            status = 3;
        } else if (len >= 3) {
            double b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
            int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
            const double2 *pp = P.pxy + off + s0;
            for (int k = lane; k < (int)len; k += 64) {
                double2 p = pp[k];
                double dx = p.x - sx, dy = p.y - sy;
                double d = sqrt(dx * dx + dy * dy); // numpy: sqrt(add.reduce(x*x))
                if (d < b0) { b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = d; i0 = k; }
                else if (d < b1) { b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b2 = d; i2 = k; }
            }
            int gi[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double dm = b0;
                int im = i0;
                wargmin(dm, im);
                gi[r] = im;
                if (i0 == im) { b0 = b1; i0 = i1; b1 = b2; i1 = i2; b2 = INFINITY; i2 = 0x7fffffff; }
            }
            int g0 = uni(gi[0]), g1 = uni(gi[1]), g2 = uni(gi[2]);
            int d12 = g1 - g2; d12 = d12 < 0 ? -d12 : d12;
            int d01 = g0 - g1; d01 = d01 < 0 ? -d01 : d01;
            if (d12 == 2) tind = g0 + s0;
            else if (d01 == 1) tind = (g0 > g1 ? g0 : g1) + s0;
            else status = JSIM_NEAREST_ANOMALY;
        } else if (len == 2) {
            tind = s0 + 1;
        }
    }
    if (status != JSIM_OK) { // reference raises: nothing is updated
        if (lane == 0) {
            P.status[ego] = status;
            if (P.n_iter) P.n_iter[ego] = it_base;
        }
        return;
    }

    STAMP(1);
    // ------------------------------------------------------------------ S1: travel -> idx -> xref
    const int tl_idx = lane < T ? lane : T; // lanes > T mirror lane T (keeps loads in range)
    double xr, yr, yawr, vr = 0.0;
    bool rend;
    long long ik;
    {
        double vref = (P.vref_min > sv) ? P.vref_min : sv; // python max(state.v, 10/3.6)
        double cstep = fabs(vref) * dt;
        double trav = cstep;
        if (P.pass == 0) {
            for (int j = 1; j <= T; ++j)
                if (j <= tl_idx) trav = trav + cstep; // np.cumsum: sequential
        } else { // ov = the previous pass's predicted speeds (mpc.py:232,101)
            const double *ovp = P.ov + (size_t)ego * (T + 1);
            trav = fabs(ovp[0]) * dt;
            for (int j = 1; j <= T; ++j)
                if (j <= tl_idx) trav = trav + fabs(ovp[j]) * dt;
        }
        ik = (long long)rint(trav / P.dl) + tind;
        if (ik > M - 1) ik = M - 1;
        double2 pr = P.pxy[off + ik];
        xr = pr.x; yr = pr.y; yawr = P.pyaw[off + ik];
        rend = (ik == M - 1);
        if (P.pcv) { // mpc_with_speed variant: xref[2] = cv[idx], cv zeroed from the ego's cut-off index on
            const int cut = P.cv_cut ? P.cv_cut[ego] : -1;
            vr = (cut < 0 || ik < cut) ? P.pcv[off + ik] : 0.0;
        }
    }

    // infeasible constant rows x[2,0] <= speed, x[2,0] >= MIN_SPEED (ECOS feasibility tolerance)
    if (speed - sv < -JSIM_FEAS_TOL || sv - P.vmin < -JSIM_FEAS_TOL) status = JSIM_INFEASIBLE;

    STAMP(2);
    // ------------------------------------------------------------------ S2: rollout of the warm start
    const bool tl = lane < T;
    double wa_t = tl ? P.oa[(size_t)ego * T + lane] : 0.0;
    double wd_t = tl ? P.od[(size_t)ego * T + lane] : 0.0;
    double bx = sx, by = sy, bv = sv, bth = syaw, sn, cs;
    {
        double dc = (P.smax < wd_t) ? P.smax : wd_t;   // min(delta, MAX_STEER)
        dc = (-P.smax > dc) ? -P.smax : dc;            // max(.., -MAX_STEER)
        double tan_t = tan(dc);
        double vcur = sv;
        for (int j = 0; j < T; ++j) {
            double aj = rdlane(wa_t, j);
            double vn = vcur + aj * dt;
            vn = (P.vmax_plant < vn) ? P.vmax_plant : vn;
            vn = (P.vmin > vn) ? P.vmin : vn;
            vcur = vn;
            if (lane == j + 1) bv = vn;
        }
        double w = ((bv / P.L) * tan_t) * dt;
        double thcur = syaw;
        for (int j = 0; j < T; ++j) {
            thcur = thcur + rdlane(w, j);
            if (lane == j + 1) bth = thcur;
        }
        sincos(bth, &sn, &cs);
        double ix = (bv * cs) * dt, iy = (bv * sn) * dt;
        double xcur = sx, ycur = sy;
        for (int j = 0; j < T; ++j) {
            xcur = xcur + rdlane(ix, j);
            ycur = ycur + rdlane(iy, j);
            if (lane == j + 1) { bx = xcur; by = ycur; }
        }
    }
    if (P.dbg_xbar && lane <= T) {
        double *xb = P.dbg_xbar + (size_t)ego * 4 * (T + 1);
        xb[0 * (T + 1) + lane] = bx; xb[1 * (T + 1) + lane] = by;
        xb[2 * (T + 1) + lane] = bv; xb[3 * (T + 1) + lane] = bth;
    }
    if (P.dbg_idx && lane <= T) P.dbg_idx[(size_t)ego * (T + 1) + lane] = ik;
    if (P.xref && lane <= T) {
        double *xf = P.xref + (size_t)ego * 4 * (T + 1);
        xf[0 * (T + 1) + lane] = xr; xf[1 * (T + 1) + lane] = yr;
        xf[2 * (T + 1) + lane] = vr; xf[3 * (T + 1) + lane] = yawr;
    }
    if (status == JSIM_INFEASIBLE) {
        // reference: solver reports infeasible -> None outputs; target_ind/xref are still stored (mpc.py:293)
        if (tl) { P.oa[(size_t)ego * T + lane] = 0.0; P.od[(size_t)ego * T + lane] = 0.0; }
        if (P.amask && lane < MW) P.amask[(size_t)ego * MW + lane] = 0u;
        if (lane == 0) {
            P.status[ego] = status;
            P.target_ind[ego] = tind;
            if (P.n_iter) P.n_iter[ego] = it_base;
        }
        return;
    }

    STAMP(3);
    // ------------------------------------------------------------------ S3: linearisation coefficients (lane t < T)
    double al = 0, be = 0, alp = 0, bep = 0, ccx = 0, ccy = 0, kt = 0;
    if (tl) {
        al = dt * cs;                 // A[0,2]
        be = (-dt * bv) * sn;         // A[0,3]
        alp = dt * sn;                // A[1,2]
        bep = (dt * bv) * cs;         // A[1,3]
        ccx = ((dt * bv) * sn) * bth; // C[0]
        ccy = ((-dt * bv) * cs) * bth; // C[1]
        kt = (dt * bv) / P.L;         // B[3,1] with delta_bar = 0
    }
    {
        // exclusive prefix sums over time: x_t = x0 + sum_{r<t}(al_r v_r + be_r yaw_r + ccx_r)
        double PA = wexscan(al, lane), PB = wexscan(be, lane), PAP = wexscan(alp, lane), PBP = wexscan(bep, lane);
        double FX = sx + wexscan(fma(al, sv, fma(be, syaw, ccx)), lane);
        double FY = sy + wexscan(fma(alp, sv, fma(bep, syaw, ccy)), lane);
        double Qxx = 0, Qxy = 0, Qyy = 0, qv = 0, qyaw = 0;
        if (lane >= 1 && lane <= T) {
            if (!rend) {
                double a1 = yawr + 0.5 * M_PI;
                double c1 = cos(a1), s1 = sin(a1), c2 = cos(yawr), s2 = sin(yawr);
                Qxx = (c1 * c1) * P.w_perp + (c2 * c2) * P.w_para;
                Qxy = (c1 * s1) * P.w_perp + (c2 * s2) * P.w_para;
                Qyy = (s1 * s1) * P.w_perp + (s2 * s2) * P.w_para;
                qv = P.Qv; qyaw = P.Qyaw;
            } else {
                Qxx = P.Qf0; Qyy = P.Qf1; qv = P.Qf2; qyaw = P.Qf3;
            }
        }
        double ex = FX - xr, ey = FY - yr, ev = sv - vr, eyaw = syaw - yawr;
        if (lane <= T + 1) {
            const int t = lane;
            bool in = lane <= T;
            tq[TQ_PA * tp + t] = PA; tq[TQ_PB * tp + t] = PB; tq[TQ_PAP * tp + t] = PAP; tq[TQ_PBP * tp + t] = PBP;
            tq[TQ_KT * tp + t] = kt;
            tq[TQ_QXX * tp + t] = in ? Qxx : 0; tq[TQ_QXY * tp + t] = in ? Qxy : 0; tq[TQ_QYY * tp + t] = in ? Qyy : 0;
            tq[TQ_QV * tp + t] = in ? qv : 0; tq[TQ_QYAW * tp + t] = in ? qyaw : 0;
            tq[TQ_QEX * tp + t] = in ? fma(Qxx, ex, Qxy * ey) : 0;
            tq[TQ_QEY * tp + t] = in ? fma(Qxy, ex, Qyy * ey) : 0;
            tq[TQ_QEV * tp + t] = in ? qv * ev : 0;
            tq[TQ_QEYAW * tp + t] = in ? qyaw * eyaw : 0;
            tq[TQ_REND * tp + t] = (in && rend) ? 1.0 : 0.0;
        }
    }
    LDS_SYNC();

    STAMP(4);
    // ------------------------------------------------------------------ S4a: H = 2 S'QS on the fp64 matrix cores
    // v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
    // D[row = (lane>>4) + 4*reg][col = lane&15].  k = state component of time step t; tile columns = 8 time steps.
    {
        const int ntile = (nh + 15) >> 4;
        const int kk = lane >> 4, cc = lane & 15;
        for (int ti = 0; ti < ntile; ++ti) {
            const int ca = 16 * ti + cc;
            const bool va = ca < nh;
            const int sa = va ? (ca >> 1) : (T - 1);
            const bool isda = ca & 1;
            const double kta = tq[TQ_KT * tp + sa];
            const double coefa = va ? (isda ? kta : dt) : 0.0;
            const double pax = isda ? tq[TQ_PB * tp + sa + 1] : tq[TQ_PA * tp + sa + 1];
            const double pay = isda ? tq[TQ_PBP * tp + sa + 1] : tq[TQ_PAP * tp + sa + 1];
            const double c23a = !va ? 0.0 : (kk == 2 ? (isda ? 0.0 : dt) : (kk == 3 ? (isda ? kta : 0.0) : 0.0));
            const double *ptxa = isda ? &tq[TQ_PB * tp] : &tq[TQ_PA * tp];
            const double *ptya = isda ? &tq[TQ_PBP * tp] : &tq[TQ_PAP * tp];
            for (int tj = 0; tj <= ti; ++tj) {
                const int cb = 16 * tj + cc;
                const bool vb = cb < nh;
                const int sb = vb ? (cb >> 1) : (T - 1);
                const bool isdb = cb & 1;
                const double ktb = tq[TQ_KT * tp + sb];
                const double coefb = vb ? (isdb ? ktb : dt) : 0.0;
                const double pbx = isdb ? tq[TQ_PB * tp + sb + 1] : tq[TQ_PA * tp + sb + 1];
                const double pby = isdb ? tq[TQ_PBP * tp + sb + 1] : tq[TQ_PAP * tp + sb + 1];
                const double c23b = !vb ? 0.0 : (kk == 2 ? (isdb ? 0.0 : dt) : (kk == 3 ? (isdb ? ktb : 0.0) : 0.0));
                const double *ptxb = isdb ? &tq[TQ_PB * tp] : &tq[TQ_PA * tp];
                const double *ptyb = isdb ? &tq[TQ_PBP * tp] : &tq[TQ_PAP * tp];
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                for (int t = 8 * ti + 1; t <= T; ++t) { // S_t has no entry in tile ti before t = 8*ti + 1
                    double aval;
                    {
                        const bool v1 = t > sa, v2 = t >= sa + 2;
                        double S0 = v2 ? coefa * (ptxa[t] - pax) : 0.0;
                        double S1 = v2 ? coefa * (ptya[t] - pay) : 0.0;
                        double S23 = v1 ? c23a : 0.0;
                        aval = kk == 0 ? S0 : (kk == 1 ? S1 : S23);
                    }
                    double bval;
                    {
                        const bool v1 = t > sb, v2 = t >= sb + 2;
                        double S0 = v2 ? coefb * (ptxb[t] - pbx) : 0.0;
                        double S1 = v2 ? coefb * (ptyb[t] - pby) : 0.0;
                        double S23 = v1 ? c23b : 0.0;
                        double qxx = tq[TQ_QXX * tp + t], qxy = tq[TQ_QXY * tp + t], qyy = tq[TQ_QYY * tp + t];
                        double q23 = kk == 2 ? tq[TQ_QV * tp + t] : tq[TQ_QYAW * tp + t];
                        bval = kk == 0 ? fma(qxx, S0, qxy * S1) : (kk == 1 ? fma(qxy, S0, qyy * S1) : q23 * S23);
                    }
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval, bval, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + kk + 4 * r, col = 16 * tj + cc;
                    if (row < nh && col < nh) {
                        double hv = 2.0 * acc[r];
                        Rm[row * ld + col] = hv;
                        if (ti != tj) Rm[col * ld + row] = hv;
                    }
                }
            }
        }
    }
    LDS_SYNC();

    STAMP(5);
    // g = 2 S'Q(f - xref); [JERK: state-cost terms from (a~, delta) to z coordinates;] then the input cost R / R_end
    // (mpc.py:180-183) and the input-difference cost Rd (mpc.py:186) [and the jerk cost, mpc_jerk.py:190]
    double gmax;
    {
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            if (i < nh) {
                const int t = i >> 1, c = i & 1;
                // g_i
                const bool isd = c;
                const double ks = tq[TQ_KT * tp + t];
                const double coef = isd ? ks : dt;
                const double px = isd ? tq[TQ_PB * tp + t + 1] : tq[TQ_PA * tp + t + 1];
                const double py = isd ? tq[TQ_PBP * tp + t + 1] : tq[TQ_PAP * tp + t + 1];
                const double *ptx = isd ? &tq[TQ_PB * tp] : &tq[TQ_PA * tp];
                const double *pty = isd ? &tq[TQ_PBP * tp] : &tq[TQ_PAP * tp];
                const double *q23 = isd ? &tq[TQ_QEYAW * tp] : &tq[TQ_QEV * tp];
                double acc = 0.0;
                for (int tt = t + 1; tt <= T; ++tt) {
                    acc = fma(coef, q23[tt], acc);
                    if (tt >= t + 2) {
                        acc = fma(coef * (ptx[tt] - px), tq[TQ_QEX * tp + tt], acc);
                        acc = fma(coef * (pty[tt] - py), tq[TQ_QEY * tp + tt], acc);
                    }
                }
                dvec[i] = 2.0 * acc;
            }
        }
        if (JERK) {
            LDS_SYNC();
            // rows: H <- E'H.  Lane j walks column j upwards: row u0_t = row a~_t + dt * (sum of the rows a~_{t'}, t' > t); the
            // total is the acc_0 row.  (E: a~_t = acc_0 + u0_t + dt * sum_{r<t} u0_r.)
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int j = lane + 64 * rr;
                if (j < nh) {
                    double S = 0.0;
                    for (int t = T - 1; t >= 0; --t) {
                        const double old = Rm[(2 * t) * ld + j];
                        Rm[(2 * t) * ld + j] = fma(dt, S, old);
                        S += old;
                    }
                    Rm[nh * ld + j] = S;
                }
            }
            // g the same way (lane t: suffix sum over the later time steps)
            double gs = 0.0, gt = 0.0;
            if (lane < T) {
                for (int t = T - 1; t > lane; --t) gs += dvec[2 * t];
                gt = gs + dvec[2 * lane]; // lane 0: the total
            }
            LDS_SYNC();
            if (lane < T) dvec[2 * lane] = fma(dt, gs, dvec[2 * lane]);
            if (lane == 0) dvec[nh] = gt;
            // columns: H <- (E'H) E, lane i walks row i (the new acc_0 row included)
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int i = lane + 64 * rr;
                if (i < n) {
                    double S = 0.0;
                    for (int t = T - 1; t >= 0; --t) {
                        const double old = Rm[i * ld + 2 * t];
                        Rm[i * ld + 2 * t] = fma(dt, S, old);
                        S += old;
                    }
                    Rm[i * ld + nh] = S;
                }
            }
            LDS_SYNC();
        }
        double gl = 0.0;
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            if (i < nh) {
                const int t = i >> 1, c = i & 1;
                const double Rt = (tq[TQ_REND * tp + t] != 0.0) ? (c ? P.Re1 : P.Re0) : (c ? P.R1 : P.R0);
                const double rdc = 2.0 * (c ? P.Rd1 : P.Rd0);
                const int nd = (T >= 2) ? ((t == 0 || t == T - 1) ? 1 : 2) : 0;
                double dg = 2.0 * Rt + nd * rdc;
                if (JERK && c == 0 && t + 1 < T) dg += 2.0 * P.jerkw * (dt * dt); // (x[4,t+1] - x[4,t])^2 = (dt u0_t)^2
                Rm[i * ld + i] += dg;
                if (t + 1 < T) Rm[i * ld + i + 2] -= rdc;
                if (t >= 1) Rm[i * ld + i - 2] -= rdc;
            }
            if (i < n) {
                const double gi = dvec[i];
                gl = fmax(gl, fabs(gi));
                if (P.dbg_g) P.dbg_g[(size_t)ego * n + i] = gi;
            }
        }
        gmax = uni(wmax(gl));
    }
    LDS_SYNC();
    if (P.dbg_H) {
        double *Hd = P.dbg_H + (size_t)ego * n * n;
        for (int e = lane; e < n * n; e += 64) {
            int r = e / n, c = e - r * n;
            Hd[e] = (c <= r) ? Rm[r * ld + c] : Rm[c * ld + r];
        }
    }

    STAMP(6);
    // ------------------------------------------------------------------ S4b: Cholesky H = L L' (in place, lower), lane = row
    for (int k = 0; k < n; ++k) {
        double s[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            s[rr] = (i >= k && i < n) ? Rm[i * ld + k] : 0.0;
        }
        for (int j = 0; j < k; ++j) {
            const double lkj = Rm[k * ld + j];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int i = lane + 64 * rr;
                if (i >= k && i < n) s[rr] = fma(-Rm[i * ld + j], lkj, s[rr]);
            }
        }
        double dkk = (RPL == 1 || k < 64) ? rdlane(s[0], k & 63) : rdlane(s[RPL - 1], k & 63);
        if (!(dkk > 0.0)) dkk = 1e-300; // H is SPD by construction (>= 2 min(R) I)
        const double lkk = sqrt(dkk);
        const double inv = 1.0 / lkk;
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            if (i == k) { Rm[k * ld + k] = lkk; ldg[k] = inv; }
            else if (i > k && i < n) Rm[i * ld + k] = s[rr] * inv;
        }
        LDS_SYNC();
    }

    STAMP(7);
    // ------------------------------------------------------------------ J = L^-T : lane r owns row r, J[r][i] = (L^-1)[i][r]
    for (int i = 0; i < n; ++i) {
        double s[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) s[rr] = (lane + 64 * rr == i) ? 1.0 : 0.0;
        for (int j = 0; j < i; ++j) {
            const double lij = Rm[i * ld + j];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int r = lane + 64 * rr;
                if (r <= j && r < n) s[rr] = fma(-lij, Jm[r * ld + j], s[rr]);
            }
        }
        const double inv = ldg[i];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int r = lane + 64 * rr;
            if (r < n) Jm[r * ld + i] = (r <= i) ? s[rr] * inv : 0.0;
        }
    }
    LDS_SYNC();

    STAMP(8);
    // ------------------------------------------------------------------ unconstrained optimum u = -J J' g
    double u[RPL];
    {
        double tc[RPL];
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) tc[rr] = 0.0;
        for (int i = 0; i < n; ++i) {
            const double gi = dvec[i];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int j = lane + 64 * rr;
                if (j < n) tc[rr] = fma(Jm[i * ld + j], gi, tc[rr]);
            }
        }
        LDS_SYNC();
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int j = lane + 64 * rr;
            if (j < n) dvec[j] = tc[rr];
        }
        LDS_SYNC();
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) u[rr] = 0.0;
        for (int j = 0; j < n; ++j) {
            const double tj = dvec[j];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int i = lane + 64 * rr;
                if (i < n) u[rr] = fma(Jm[i * ld + j], tj, u[rr]);
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int i = lane + 64 * rr;
            u[rr] = -u[rr];
            if (i < n) uvec[i] = u[rr];
        }
        if (lane < 4) uvec[n + lane] = 0.0;
        if (lane < MW) maskw[lane] = 0u;
    }
    LDS_SYNC();

    STAMP(9);
    // ------------------------------------------------------------------ Goldfarb-Idnani dual active-set iterations
    // static steepest-edge weights of the entering-row rule, per time lane t: 1 / ||J'n||^2 for the rate rows
    // (J[2t+3] - J[2t+1]), the speed rows (dt * sum_{s<t} J[2s]), the accel rows (J[2t]) and the steer rows (J[2t+1]).
    // Rm (L is dead, R not yet started) is the scratch for the cumulative accel rows.
    double iwD = 1.0, iwV = 1.0, iwA = 1.0, iwS = 1.0;
    {
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int j = lane + 64 * rr;
            if (j < n) {
                double c = 0.0, pj = 0.0;
                const double jt = JERK ? Jm[nh * ld + j] : 0.0;
                for (int ss = 0; ss < T; ++ss) {
                    const double a = Jm[(2 * ss) * ld + j];
                    c += JERK ? jt + a + dt * pj : a; // JERK: A_s = J[acc_0] + J[2s] + dt * sum_{r<s} J[2r]
                    pj += a;
                    Rm[(ss + 1) * ld + j] = c;
                }
            }
        }
        LDS_SYNC();
        const int t = lane;
        if (t <= T) {
            double wD = 0.0, wV = 0.0, wA = 0.0, wS = 0.0;
            for (int j = 0; j < n; ++j) {
                if (t >= 1) { const double c = Rm[t * ld + j]; wV = fma(c, c, wV); }
                if (t < T) {
                    const double ja = Jm[(2 * t) * ld + j], js = Jm[(2 * t + 1) * ld + j];
                    wA = fma(ja, ja, wA); wS = fma(js, js, wS);
                    if (t + 1 < T) { const double e = Jm[(2 * t + 3) * ld + j] - js; wD = fma(e, e, wD); }
                }
            }
            wV *= dt * dt;
            iwD = 1.0 / (wD > 0.0 ? wD : 1.0); iwV = 1.0 / (wV > 0.0 ? wV : 1.0);
            iwA = 1.0 / (wA > 0.0 ? wA : 1.0); iwS = 1.0 / (wS > 0.0 ? wS : 1.0);
        }
        LDS_SYNC();
    }
    int q = 0, iters = 0;
    const int max_iters = 50 * n + 100;
    unsigned abits = 0; // lane t: which of the 8 rows of time step t are in the working set
    for (;;) {
        // ---- step 1: among the violated rows the one with the largest viol^2 / (n'H^-1 n) (ties -> lowest canonical index)
        int p;
        double violp;
        {
            const int t = lane;
            double a = 0.0, dl_ = 0.0, dnext = 0.0;
            if (t < T) { double2 ad = *(const double2 *)&uvec[2 * t]; a = ad.x; dl_ = ad.y; dnext = uvec[2 * t + 3]; }
            double aeff = a; // JERK: the effective acceleration a~_t = acc_0 + u0_t + dt * sum_{r<t} u0_r
            if (JERK) aeff = t < T ? uvec[nh] + a + dt * wexscan(a, lane) : 0.0;
            const double vt = sv + dt * wexscan(aeff, lane);
            double best = 0.0, bviol = 0.0;
            int bid = 0x7fffffff;
#define CONSIDER(valid, bit, viol_expr, habs, id_expr, iw)                               \
    if ((valid) && !(abits & (1u << (bit)))) {                                           \
        const double vv = (viol_expr);                                                   \
        const int id_ = (id_expr);                                                       \
        const double key_ = jsim_key_trunc(vv * vv * (iw));                              \
        if (vv > JSIM_VIOL_TOL * (1.0 + (habs)) && (key_ > best || (key_ == best && id_ < bid))) { best = key_; bid = id_; bviol = vv; } \
    }
            CONSIDER(t + 1 < T, 0, (dnext - dl_) - P.dmax, P.dmax, 2 * t, iwD)
            CONSIDER(t + 1 < T, 1, (dl_ - dnext) - P.dmax, P.dmax, 2 * t + 1, iwD)
            CONSIDER(t >= 1 && t <= T, 2, vt - speed, fabs(speed - sv), 2 * T - 2 + t, iwV)
            CONSIDER(t >= 1 && t <= T, 3, P.vmin - vt, fabs(sv - P.vmin), 3 * T - 1 + t, iwV)
            CONSIDER(t < T, 4, a - P.amax, fabs(P.amax), 4 * T + t, iwA)
            CONSIDER(t < T, 5, P.amin - a, fabs(P.amin), 5 * T + t, iwA)
            CONSIDER(t < T, 6, dl_ - P.smax, P.smax, 6 * T + 2 * t, iwS)
            CONSIDER(t < T, 7, -dl_ - P.smax, P.smax, 6 * T + 2 * t + 1, iwS)
#undef CONSIDER
            const int mybid = bid;
            wargmax(best, bid);
            p = uni(bid);
            // the violation of the winner: held by the one lane whose candidate won
            violp = uni(wsum(mybid == p && p != 0x7fffffff ? bviol : 0.0));
        }
        if (p == 0x7fffffff) break; // optimal

        // decode row p
        int kind, tp_, neg = 0;
        if (p < 2 * T - 2) { kind = 0; tp_ = p >> 1; neg = p & 1; }
        else if (p < 3 * T - 1) { kind = 1; tp_ = p - (2 * T - 2); }
        else if (p < 4 * T) { kind = 2; tp_ = p - (3 * T - 1); }
        else if (p < 5 * T) { kind = 3; tp_ = p - 4 * T; }
        else if (p < 6 * T) { kind = 4; tp_ = p - 5 * T; }
        else { kind = 5; tp_ = (p - 6 * T) >> 1; neg = (p - 6 * T) & 1; }
        const unsigned pbit = kind == 0 ? (neg ? 2u : 1u) : kind == 1 ? 4u : kind == 2 ? 8u : kind == 3 ? 16u
                              : kind == 4 ? 32u : (neg ? 128u : 64u);
        double lplus = 0.0;
        bool failed = false;

        for (;;) { // ---- step 2
            if (++iters > max_iters) { failed = true; break; }
            // d = J' n+, n+ = -G_p'   (lane j = column j of J)
            double d[RPL];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int j = lane + 64 * rr;
                double dj = 0.0;
                if (j < n) {
                    if (kind == 0) {
                        double e = Jm[(2 * tp_ + 1) * ld + j] - Jm[(2 * tp_ + 3) * ld + j]; // -(J[d_{t+1}] - J[d_t])
                        dj = neg ? -e : e;
                    } else if (kind == 1 || kind == 2) {
                        double s = 0.0, pj = 0.0;
                        const double jt = JERK ? Jm[nh * ld + j] : 0.0;
                        for (int ss = 0; ss < tp_; ++ss) {
                            const double a = Jm[(2 * ss) * ld + j];
                            s += JERK ? jt + a + dt * pj : a;
                            pj += a;
                        }
                        dj = (kind == 1) ? -dt * s : dt * s;
                    } else if (kind == 3) dj = -Jm[(2 * tp_) * ld + j];
                    else if (kind == 4) dj = Jm[(2 * tp_) * ld + j];
                    else dj = neg ? Jm[(2 * tp_ + 1) * ld + j] : -Jm[(2 * tp_ + 1) * ld + j];
                    dvec[j] = dj;
                }
                d[rr] = dj;
            }
            double dd_l = 0.0, zn_l = 0.0;
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int j = lane + 64 * rr;
                const double sq = d[rr] * d[rr];
                dd_l += sq;
                if (j >= q) zn_l += sq;
            }
            const double dd = uni(wsum(dd_l)), zn = uni(wsum(zn_l));
            LDS_SYNC();
            // z = J2 d2 (lane i = row i)
            double z[RPL];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) z[rr] = 0.0;
            for (int j = q; j < n; ++j) {
                const double dj = dvec[j];
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int i = lane + 64 * rr;
                    if (i < n) z[rr] = fma(Jm[i * ld + j], dj, z[rr]);
                }
            }
            // r = R^-1 d1 (back substitution; lane k = row k of R)
            double r[RPL];
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) r[rr] = d[rr];
            for (int j = q - 1; j >= 0; --j) {
                const double rj = ((RPL == 1 || j < 64) ? rdlane(r[0], j & 63) : rdlane(r[RPL - 1], j & 63)) * rdg[j];
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int i = lane + 64 * rr;
                    if (i == j) r[rr] = rj;
                    else if (i < j) r[rr] = fma(-Rm[i * ld + j], rj, r[rr]);
                }
            }
            // dual ratio test
            int l;
            double t1;
            {
                double bt = INFINITY;
                int bk = 0x7fffffff;
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int k = lane + 64 * rr;
                    if (k < q && r[rr] > 0.0) {
                        double tk = jsim_ratio_trunc(lamv[k] / r[rr]);
                        if (tk < bt) { bt = tk; bk = k; }
                    }
                }
                wargmin(bt, bk);
                t1 = uni(bt);
                l = uni(bk);
            }
            const bool dependent = !(zn > JSIM_DEP_TOL * dd);
            double t2 = dependent ? INFINITY : violp / zn;
            if (t2 < 0.0) t2 = 0.0;
            if (isinf(t1) && isinf(t2)) { failed = true; break; }
            const bool full = (t2 <= t1);
            const double tstep = full ? t2 : t1;
#pragma unroll
            for (int rr = 0; rr < RPL; ++rr) {
                const int i = lane + 64 * rr;
                if (!dependent && i < n) { u[rr] = fma(tstep, z[rr], u[rr]); uvec[i] = u[rr]; }
                if (i < q) {
                    double lk = fma(-tstep, r[rr], lamv[i]);
                    lamv[i] = lk < 0.0 ? 0.0 : lk;
                }
            }
            lplus += tstep;

            if (full) {
                // add p: one Householder reflection on d2; J2 <- J2 P ; new R column [d1; rho]
                const double nrm = sqrt(zn);
                const double dq = (RPL == 1 || q < 64) ? rdlane(d[0], q & 63) : rdlane(d[RPL - 1], q & 63);
                const double sg = dq >= 0.0 ? 1.0 : -1.0;
                const double rho = -sg * nrm;
                const double v0 = dq + sg * nrm;
                const double beta = 1.0 / (nrm * (nrm + fabs(dq)));
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int i = lane + 64 * rr;
                    if (i < n) {
                        const double w = beta * fma(sg * nrm, Jm[i * ld + q], z[rr]);
                        Jm[i * ld + q] = fma(-w, v0, Jm[i * ld + q]);
                        for (int j = q + 1; j < n; ++j) Jm[i * ld + j] = fma(-w, dvec[j], Jm[i * ld + j]);
                    }
                    if (i < q) Rm[i * ld + q] = d[rr];
                    if (i == q) { Rm[q * ld + q] = rho; rdg[q] = 1.0 / rho; actv[q] = p; lamv[q] = lplus; }
                }
                if (lane == tp_) abits |= pbit;
                ++q;
                LDS_SYNC();
                break;
            }
            // partial step: drop working-set position l
            {
                const int pdrop = actv[l];
                // Givens sweep on R (lane k = column k), coefficients kept for the J pass
                for (int j = l; j + 1 < q; ++j) {
                    const double a = Rm[j * ld + j + 1], b = Rm[(j + 1) * ld + j + 1];
                    const double hh = sqrt(a * a + b * b);
                    double c = 1.0, s = 0.0;
                    if (hh > 0.0) { c = a / hh; s = b / hh; }
#pragma unroll
                    for (int rr = 0; rr < RPL; ++rr) {
                        const int k = lane + 64 * rr;
                        if (k >= j + 1 && k < q) {
                            const double x1 = Rm[j * ld + k], x2 = Rm[(j + 1) * ld + k];
                            Rm[j * ld + k] = fma(c, x1, s * x2);
                            Rm[(j + 1) * ld + k] = fma(-s, x1, c * x2);
                        }
                    }
                    if (lane == 0) { gsv[2 * j] = c; gsv[2 * j + 1] = s; }
                    LDS_SYNC();
                }
                // same rotations on the columns of J (lane i = row i, carried column)
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int i = lane + 64 * rr;
                    if (i < n) {
                        double carry = Jm[i * ld + l];
                        for (int j = l; j + 1 < q; ++j) {
                            const double nx = Jm[i * ld + j + 1];
                            const double c = gsv[2 * j], s = gsv[2 * j + 1];
                            Jm[i * ld + j] = fma(c, carry, s * nx);
                            carry = fma(-s, carry, c * nx);
                        }
                        Jm[i * ld + q - 1] = carry;
                    }
                }
                // shift R's columns, the working-set list and the multipliers left
                int an[RPL];
                double ln[RPL];
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int k = lane + 64 * rr;
                    if (k < q) {
                        for (int j = (k > l ? k : l); j + 1 < q; ++j) Rm[k * ld + j] = Rm[k * ld + j + 1];
                    }
                    an[rr] = (k >= l && k + 1 < q) ? actv[k + 1] : 0;
                    ln[rr] = (k >= l && k + 1 < q) ? lamv[k + 1] : 0.0;
                }
                LDS_SYNC();
#pragma unroll
                for (int rr = 0; rr < RPL; ++rr) {
                    const int k = lane + 64 * rr;
                    if (k >= l && k + 1 < q) { actv[k] = an[rr]; lamv[k] = ln[rr]; rdg[k] = 1.0 / Rm[k * ld + k]; }
                }
                --q;
                // clear the dropped row's working-set bit on its time lane
                {
                    int dk, dt_, dneg = 0;
                    if (pdrop < 2 * T - 2) { dk = 0; dt_ = pdrop >> 1; dneg = pdrop & 1; }
                    else if (pdrop < 3 * T - 1) { dk = 1; dt_ = pdrop - (2 * T - 2); }
                    else if (pdrop < 4 * T) { dk = 2; dt_ = pdrop - (3 * T - 1); }
                    else if (pdrop < 5 * T) { dk = 3; dt_ = pdrop - 4 * T; }
                    else if (pdrop < 6 * T) { dk = 4; dt_ = pdrop - 5 * T; }
                    else { dk = 5; dt_ = (pdrop - 6 * T) >> 1; dneg = (pdrop - 6 * T) & 1; }
                    const unsigned dbit = dk == 0 ? (dneg ? 2u : 1u) : dk == 1 ? 4u : dk == 2 ? 8u : dk == 3 ? 16u
                                          : dk == 4 ? 32u : (dneg ? 128u : 64u);
                    if (lane == dt_) abits &= ~dbit;
                }
                LDS_SYNC();
                // violation of p at the new point
                if (kind == 0) {
                    double e = uvec[2 * tp_ + 3] - uvec[2 * tp_ + 1];
                    violp = (neg ? -e : e) - P.dmax;
                } else if (kind == 1 || kind == 2) {
                    double a = (lane < tp_) ? uvec[2 * lane] : 0.0;
                    if (JERK) {
                        const double u0 = lane < T ? uvec[2 * lane] : 0.0;
                        a = (lane < tp_) ? uvec[nh] + u0 + dt * wexscan(u0, lane) : 0.0;
                    }
                    double vt = sv + dt * uni(wsum(a));
                    violp = (kind == 1) ? vt - speed : P.vmin - vt;
                } else if (kind == 3) violp = uvec[2 * tp_] - P.amax;
                else if (kind == 4) violp = P.amin - uvec[2 * tp_];
                else violp = (neg ? -uvec[2 * tp_ + 1] : uvec[2 * tp_ + 1]) - P.smax;
            }
        }
        if (failed) { status = JSIM_INFEASIBLE; break; }
    }

    STAMP(10);
    // ------------------------------------------------------------------ S5: outputs
    if (status != JSIM_OK) {
        if (tl) { P.oa[(size_t)ego * T + lane] = 0.0; P.od[(size_t)ego * T + lane] = 0.0; }
        if (P.amask && lane < MW) P.amask[(size_t)ego * MW + lane] = 0u;
        if (lane == 0) {
            P.status[ego] = status;
            P.target_ind[ego] = tind;
            if (P.n_iter) P.n_iter[ego] = it_base + iters;
        }
        return;
    }
    {
        const double thr = JSIM_ACT_TOL * fmax(1.0, gmax);
        if (P.dbg_lam) {
            for (int e = lane; e < 8 * T; e += 64) P.dbg_lam[(size_t)ego * 8 * T + e] = 0.0;
            LDS_SYNC(); // drains the zero fill (vmcnt) before other lanes store multipliers to the same words
        }
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
            const int k = lane + 64 * rr;
            if (k < q) {
                const double lk = lamv[k];
                const int id = actv[k];
                if (lk > thr) atomicOr(&maskw[id >> 5], 1u << (id & 31));
                if (P.dbg_lam) P.dbg_lam[(size_t)ego * 8 * T + id] = lk;
            }
        }
        LDS_SYNC();
        if (P.amask && lane < MW) P.amask[(size_t)ego * MW + lane] = maskw[lane];
    }
    {
        double a = 0.0, dl_ = 0.0;
        if (tl) { double2 ad = *(const double2 *)&uvec[2 * lane]; a = ad.x; dl_ = ad.y; }
        // predicted states of the linearised model at u* (= the cvxpy x variable)
        double aeff = a;
        if (JERK) aeff = tl ? uvec[nh] + a + dt * wexscan(a, lane) : 0.0;
        const double vt = sv + dt * wexscan(aeff, lane);
        const double yt = syaw + wexscan(kt * dl_, lane);
        const double xt = sx + wexscan(fma(al, vt, fma(be, yt, ccx)), lane);
        const double yy = sy + wexscan(fma(alp, vt, fma(bep, yt, ccy)), lane);
        if (tl) { P.oa[(size_t)ego * T + lane] = a; P.od[(size_t)ego * T + lane] = dl_; }
        if (lane <= T) {
            const size_t o = (size_t)ego * (T + 1) + lane;
            if (P.ox) P.ox[o] = xt;
            if (P.oy) P.oy[o] = yy;
            if (P.ov) P.ov[o] = vt;
            if (P.oyaw) P.oyaw[o] = yt;
        }
    }
    if (lane == 0) {
        P.status[ego] = JSIM_OK;
        P.target_ind[ego] = tind;
        if (P.n_iter) P.n_iter[ego] = it_base + iters;
    }
    STAMP(11);
}

#include "mpc_step_reg.inc"
#include "reg_common.inc"
#include "mpc_step_reg4.inc"

// Horizons with a register-resident kernel.  One wavefront per ego: every row stored for 13 <= T <= 20 (3T+1 <= 63 lanes), virtual
// speed rows for 21 <= T <= 31; four wavefronts per ego (two lanes per row) for T = 32 and 40 (half rows in panels of eight).  The
// kernels are templates on T, fully unrolled -- a horizon is fast if it is in one of these lists (7 s of compile time per
// instantiation) and runs on the LDS kernel otherwise (any T <= 48; 4-6 x slower: tools/dev/horizon_ab.py).  BASELINE.json's
// configurations use 13 (the reference's stock horizon), 20, 30 and 40; the others are there so that a horizon near them does not
// fall off that cliff.  Mirrored by config.ONE_WAVE_HORIZONS / FOUR_WAVE_HORIZONS / HELP_HORIZONS (tests/test_host_cpu.py compares them).
#define JSIM_ONE_WAVE_HORIZONS(X) X(13) X(15) X(16) X(20) X(25) X(30)
#define JSIM_FOUR_WAVE_HORIZONS(X) X(32) X(40)
// one-wave horizons that also have the form with three helper wavefronts per ego, taken at B <= 256 (one ego per CU at most).  Measured
// at 256 egos, closed loop (tools/dev/help_ab13.py): T = 13 +8 %, 15 +6 %, 16 +11 %, 20 +11-14 %, 25 +6 %; T = 30 LOSES 2 % (448
// registers, 72 KB of LDS: handing 61 rows of 60 doubles over costs what the helpers save) and is left out.
#define JSIM_HELP_HORIZONS(X) X(13) X(15) X(16) X(20) X(25)
// ... and those whose form with the loop glue inside the launch (PRE) has helpers too: the reference's stock horizon and the headline's.
// (T = 16 with PRE and helpers is a build the ISA guard refuses -- vector code in front of a join block's exec restore, section 5 fact 6 of
// DESIGN.md -- and is not instantiated.)
#define JSIM_HELP_PRE_HORIZONS(X) X(13) X(20)

// ---- Split build (build.py's default: one translation unit per horizon, compiled in parallel -- 3 minutes of one core otherwise).
// -DJSIM_KERNEL_TU=<T>: this file up to here plus the explicit instantiations of horizon T's register kernels, nothing else.
// -DJSIM_SPLIT_BUILD  : the rest of the library, with those instantiations declared `extern template` (their host stubs and code
//                       objects come from the kernel translation units).  Neither: everything in one unit, as before.
// The `#if` lists below repeat the three horizon lists above (tests/test_host_cpu.py compares them).
#define JSIM_REG_ARGS const KP, const TickP, const PreK
#if defined(JSIM_KERNEL_TU)
#if JSIM_KERNEL_TU == 32 || JSIM_KERNEL_TU == 40                                                            /* four-wave horizons */
template __global__ void mpc_step_reg4_kernel<JSIM_KERNEL_TU, true>(JSIM_REG_ARGS);
template __global__ void mpc_step_reg4_kernel<JSIM_KERNEL_TU, false>(JSIM_REG_ARGS);
#elif JSIM_KERNEL_TU == 13 || JSIM_KERNEL_TU == 15 || JSIM_KERNEL_TU == 16 || JSIM_KERNEL_TU == 20 || JSIM_KERNEL_TU == 25 || JSIM_KERNEL_TU == 30   /* one-wave horizons */
template __global__ void mpc_step_reg_kernel<JSIM_KERNEL_TU, true, 1, false>(JSIM_REG_ARGS);
template __global__ void mpc_step_reg_kernel<JSIM_KERNEL_TU, false, (JSIM_KERNEL_TU == 13 ? 2 : 1), false>(JSIM_REG_ARGS);
#if JSIM_KERNEL_TU == 20
template __global__ void mpc_step_reg_kernel<20, false, 2, false>(JSIM_REG_ARGS);
#endif
#if JSIM_KERNEL_TU == 13 || JSIM_KERNEL_TU == 15 || JSIM_KERNEL_TU == 16 || JSIM_KERNEL_TU == 20 || JSIM_KERNEL_TU == 25                            /* helper-wavefront horizons */
template __global__ void mpc_step_reg_kernel<JSIM_KERNEL_TU, false, 1, true>(JSIM_REG_ARGS);
#endif
#if JSIM_KERNEL_TU == 13 || JSIM_KERNEL_TU == 20                                                                                                    /* helper-wavefront horizons with the glue */
template __global__ void mpc_step_reg_kernel<JSIM_KERNEL_TU, true, 1, true>(JSIM_REG_ARGS);
#endif
#else
#error "JSIM_KERNEL_TU: not a horizon with a register kernel"
#endif
#else /* !JSIM_KERNEL_TU: the library proper */
#if defined(JSIM_SPLIT_BUILD)
#define JSIM_X(t) extern template __global__ void mpc_step_reg_kernel<t, true, 1, false>(JSIM_REG_ARGS); \
                  extern template __global__ void mpc_step_reg_kernel<t, false, (t == 13 ? 2 : 1), false>(JSIM_REG_ARGS);
JSIM_ONE_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
extern template __global__ void mpc_step_reg_kernel<20, false, 2, false>(JSIM_REG_ARGS);
#define JSIM_X(t) extern template __global__ void mpc_step_reg_kernel<t, false, 1, true>(JSIM_REG_ARGS);
JSIM_HELP_HORIZONS(JSIM_X)
#undef JSIM_X
#define JSIM_X(t) extern template __global__ void mpc_step_reg_kernel<t, true, 1, true>(JSIM_REG_ARGS);
JSIM_HELP_PRE_HORIZONS(JSIM_X)
#undef JSIM_X
#define JSIM_X(t) extern template __global__ void mpc_step_reg4_kernel<t, true>(JSIM_REG_ARGS); \
                  extern template __global__ void mpc_step_reg4_kernel<t, false>(JSIM_REG_ARGS);
JSIM_FOUR_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
#endif
#if defined(JSIM_DEV_NO_REG)
static bool has_reg_kernel(int) { return false; }
#elif defined(JSIM_DEV_ONLY_T40)
static bool has_reg_kernel(int T) { return T == 40; }
#elif defined(JSIM_DEV_ONLY_T30)
static bool has_reg_kernel(int T) { return T == 30; }
#elif defined(JSIM_DEV_ONLY_T20)
static bool has_reg_kernel(int T) { return T == 20; }
#else
static bool has_reg_kernel(int T)
{
#define JSIM_X(t) if (T == t) return true;
    JSIM_ONE_WAVE_HORIZONS(JSIM_X) JSIM_FOUR_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
    return false;
}
#endif
static bool has_fused_glue(int T) { return has_reg_kernel(T); }

static void launch_reg(int T, int B, hipStream_t s, const KP &P, const TickP &K, const PreK *Q = nullptr)
{
    static const PreK none = {};
    // more egos than SIMDs (256 CUs x 4): the T = 20 form built for two waves per SIMD (mpc_step_reg.inc, WPE)
    static const int w2_min_b = [] { const char *e = getenv("JSIM_W2_MIN_B"); return e ? atoi(e) : 1025; }();
    // at most one ego per CU: the form with three helper wavefronts per ego (mpc_step_reg.inc, HELP) -- it needs a CU to itself (four
    // wavefronts of 270-350 registers), so the default threshold is the device's CU count (256 on an MI355X in SPX mode)
    static const int help_env_b = [] { const char *e = getenv("JSIM_HELP_MAX_B"); return e ? atoi(e) : -1; }();
    int help_max_b = help_env_b;
    if (help_max_b < 0) {
        static int cus[64];   // per device id, 0 = not asked yet
        int dev = 0;
        help_max_b = 256;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (cus[dev] == 0) {
                int n = 0;
                cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
            }
            help_max_b = cus[dev];
        }
    }
#if defined(JSIM_DEV_NO_REG) /* development builds of the planner / glue: no register kernel is instantiated */
    (void)T; (void)B; (void)s; (void)P; (void)K; (void)Q; (void)none; (void)w2_min_b; (void)help_max_b;
    return;
#elif defined(JSIM_DEV_ONLY_T40) /* development builds: only the T = 40 kernel is instantiated (seconds instead of minutes to compile) */
    if (T == 40) {
        if (Q) hipLaunchKernelGGL((mpc_step_reg4_kernel<40, true>), dim3(B), dim3(256), 0, s, P, K, *Q);
        else hipLaunchKernelGGL((mpc_step_reg4_kernel<40, false>), dim3(B), dim3(256), 0, s, P, K, none);
    }
    return;
#elif defined(JSIM_DEV_ONLY_T30) /* development builds: only the T = 30 one-wave kernel */
    if (T == 30) {
        if (Q) hipLaunchKernelGGL((mpc_step_reg_kernel<30, true, 1>), dim3(B), dim3(64), 0, s, P, K, *Q);
        else hipLaunchKernelGGL((mpc_step_reg_kernel<30, false, 1>), dim3(B), dim3(64), 0, s, P, K, none);
    }
    return;
#elif defined(JSIM_DEV_ONLY_T20) /* development builds: only the T = 20 one-wave kernels */
    if (T == 20) {
        if (Q) hipLaunchKernelGGL((mpc_step_reg_kernel<20, true, 1>), dim3(B), dim3(64), 0, s, P, K, *Q);
        else if (B <= help_max_b) hipLaunchKernelGGL((mpc_step_reg_kernel<20, false, 1, true>), dim3(B), dim3(256), 0, s, P, K, none);
        else if (B >= w2_min_b) hipLaunchKernelGGL((mpc_step_reg_kernel<20, false, 2>), dim3(B), dim3(64), 0, s, P, K, none);
        else hipLaunchKernelGGL((mpc_step_reg_kernel<20, false, 1>), dim3(B), dim3(64), 0, s, P, K, none);
    }
    return;
#else
    if (Q) { // the loop glue inside the launch
        if (B <= help_max_b) { // (with helper wavefronts, like the plain closed loop below)
#define JSIM_X(t) if (T == t) { hipLaunchKernelGGL((mpc_step_reg_kernel<t, true, 1, true>), dim3(B), dim3(256), 0, s, P, K, *Q); return; }
            JSIM_HELP_PRE_HORIZONS(JSIM_X)
#undef JSIM_X
        }
#define JSIM_X(t) if (T == t) { hipLaunchKernelGGL((mpc_step_reg_kernel<t, true, 1>), dim3(B), dim3(64), 0, s, P, K, *Q); return; }
        JSIM_ONE_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
#define JSIM_X(t) if (T == t) { hipLaunchKernelGGL((mpc_step_reg4_kernel<t, true>), dim3(B), dim3(256), 0, s, P, K, *Q); return; }
        JSIM_FOUR_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
        return;
    }
    // register budgets (WPE): T = 13 fits 256 registers without scratch -- two waves per SIMD at every batch size; T = 20 has a
    // 256-register form for batches above one ego per SIMD; every other horizon one wave per SIMD
    // at most one ego per CU: three helper wavefronts per ego (mpc_step_reg.inc, HELP) -- every one-wave horizon
    if (B <= help_max_b) {
#define JSIM_X(t) if (T == t) { hipLaunchKernelGGL((mpc_step_reg_kernel<t, false, 1, true>), dim3(B), dim3(256), 0, s, P, K, none); return; }
        JSIM_HELP_HORIZONS(JSIM_X)
#undef JSIM_X
    }
    if (T == 13) { hipLaunchKernelGGL((mpc_step_reg_kernel<13, false, 2>), dim3(B), dim3(64), 0, s, P, K, none); return; }
    if (T == 20 && B >= w2_min_b) { hipLaunchKernelGGL((mpc_step_reg_kernel<20, false, 2>), dim3(B), dim3(64), 0, s, P, K, none); return; }
#define JSIM_X(t) if (T == t && t != 13) { hipLaunchKernelGGL((mpc_step_reg_kernel<t, false, (t == 13 ? 2 : 1)>), dim3(B), dim3(64), 0, s, P, K, none); return; }
    JSIM_ONE_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
#define JSIM_X(t) if (T == t) { hipLaunchKernelGGL((mpc_step_reg4_kernel<t, false>), dim3(B), dim3(256), 0, s, P, K, none); return; }
    JSIM_FOUR_WAVE_HORIZONS(JSIM_X)
#undef JSIM_X
#endif
}

// ---------------------------------------------------------------------------------------------------
// plant update for the per-vehicle loop (Simulation.step, main/lib/simulation.py:35-47) and the
// controller's (di, ai) selection with the failure path (main/lib/mpc.py:298-303)
// ---------------------------------------------------------------------------------------------------
struct PlantP {
    int B, T;
    double dt, L, smax, vmax, vmin, max_decel;
    const double *pe; // per-ego configuration (MAX_DECEL at [14]) or NULL
};

__global__ __launch_bounds__(256) void plant_step_kernel(PlantP P, double *x0, const double *oa, const double *od,
                                                         const int *status, double *di_ai)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    double di = di_ai[2 * b], ai;
    if (status[b] == JSIM_OK) { di = od[(size_t)b * P.T]; ai = oa[(size_t)b * P.T]; }
    else ai = P.pe ? P.pe[(size_t)b * JSIM_EGO_CFG_DOUBLES + 14] : P.max_decel;
    di_ai[2 * b] = di;
    di_ai[2 * b + 1] = ai;
    double x = x0[4 * b], y = x0[4 * b + 1], v = x0[4 * b + 2], th = x0[4 * b + 3];
    double dc = (P.smax < di) ? P.smax : di;
    dc = (-P.smax > dc) ? -P.smax : dc;
    const double xd = v * cos(th), yd = v * sin(th), thd = (v / P.L) * tan(dc);
    x += xd * P.dt; y += yd * P.dt; th += thd * P.dt;
    v += ai * P.dt;
    v = (P.vmax < v) ? P.vmax : v;
    v = (P.vmin > v) ? P.vmin : v;
    x0[4 * b] = x; x0[4 * b + 1] = y; x0[4 * b + 2] = v; x0[4 * b + 3] = th;
}

// Closed-loop bookkeeping of the per-vehicle loop for a batch (main/scenarios/mpc_intersection.py:99-163):
// (di, ai) selection + plant step as above, history record, and replacement of finished egos -- the loop's
// `if mpc.is_goal(state): break` (:101) becomes "respawn at the ego's spawn state with a cold controller".
struct LoopP {
    int B, T, max_age;
    double dt, L, smax, vmax, vmin, max_decel, goal_dis, stop_speed;
    const double2 *pxy;
    const long long *poff;
    const double *pe; // per-ego configuration (MAX_DECEL at [14]) or NULL
};

__global__ __launch_bounds__(256) void loop_advance_kernel(LoopP P, double *x0, double *oa, double *od,
                                                           const int *status, double *di_ai, long long *target_ind,
                                                           const int *path_id, const int *path_len,
                                                           const double *x0_spawn, const long long *target_spawn,
                                                           int *age, double *hist, int *tick, int hist_cap,
                                                           unsigned long long *n_respawn)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    double di = di_ai[2 * b], ai;
    if (status[b] == JSIM_OK) { di = od[(size_t)b * P.T]; ai = oa[(size_t)b * P.T]; }
    else ai = P.pe ? P.pe[(size_t)b * JSIM_EGO_CFG_DOUBLES + 14] : P.max_decel;
    di_ai[2 * b] = di;
    di_ai[2 * b + 1] = ai;
    if (hist) {
        const int k = *tick; // device tick counter: bumped by a 1-thread kernel after this one (graph-replay safe)
        if (k < hist_cap) { hist[((size_t)k * P.B + b) * 2] = di; hist[((size_t)k * P.B + b) * 2 + 1] = ai; }
    }
    double x = x0[4 * b], y = x0[4 * b + 1], v = x0[4 * b + 2], th = x0[4 * b + 3];
    double dc = (P.smax < di) ? P.smax : di;
    dc = (-P.smax > dc) ? -P.smax : dc;
    const double xd = v * cos(th), yd = v * sin(th), thd = (v / P.L) * tan(dc);
    x += xd * P.dt; y += yd * P.dt; th += thd * P.dt;
    v += ai * P.dt;
    v = (P.vmax < v) ? P.vmax : v;
    v = (P.vmin > v) ? P.vmin : v;
    // MPC.is_goal on the new state (main/lib/mpc.py:314-330)
    const long long off = P.poff[path_id[b]];
    const long long full = P.poff[path_id[b] + 1] - off;
    const double2 g = P.pxy[off + full - 1];
    const long long ti = target_ind[b];
    bool isgoal = hypot(x - g.x, y - g.y) <= P.goal_dis;
    long long df = ti - (long long)path_len[b];
    if ((df < 0 ? -df : df) >= 5) isgoal = false;
    const bool done = (isgoal && fabs(v) <= P.stop_speed) || (age[b] + 1 >= P.max_age);
    if (done) {
        x = x0_spawn[4 * b]; y = x0_spawn[4 * b + 1]; v = x0_spawn[4 * b + 2]; th = x0_spawn[4 * b + 3];
        target_ind[b] = target_spawn[b];
        for (int t = 0; t < P.T; ++t) { oa[(size_t)b * P.T + t] = 0.0; od[(size_t)b * P.T + t] = 0.0; }
        di_ai[2 * b] = 0.0; di_ai[2 * b + 1] = 0.0;
        age[b] = 0;
        if (n_respawn) atomicAdd(n_respawn, 1ull);
    } else {
        age[b] += 1;
    }
    x0[4 * b] = x; x0[4 * b + 1] = y; x0[4 * b + 2] = v; x0[4 * b + 3] = th;
}

__global__ void tick_increment_kernel(int *tick) { *tick += 1; }
__global__ void tick_add_kernel(int *tick, int n) { *tick += n; }

struct GoalP {
    int B, T;
    double goal_dis, stop_speed;
    const double2 *pxy;
    const double *pyaw;
    const long long *poff;
};

// MPC.get_current_xref_deviation (main/lib/mpc.py:305-312) and MPC.is_goal (:314-330)
__global__ __launch_bounds__(256) void deviation_goal_kernel(GoalP P, const double *x0, const int *path_id,
                                                             const int *path_len, const long long *target_ind,
                                                             const double *ox, const double *oy, double *deviation,
                                                             int *is_goal)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const long long off = P.poff[path_id[b]];
    const long long full = P.poff[path_id[b] + 1] - off;
    const long long ti = target_ind[b];
    if (deviation) {
        const double2 rp = P.pxy[off + ti];
        const double yp = P.pyaw[off + ti] + M_PI / 2;
        const double dx = rp.x - ox[(size_t)b * (P.T + 1)], dy = rp.y - oy[(size_t)b * (P.T + 1)];
        const double a = cos(yp) * dx, c = sin(yp) * dy;
        deviation[b] = sqrt(a * a + c * c);
    }
    if (is_goal) {
        const double2 g = P.pxy[off + full - 1]; // goal = last point of the path given to MPC.__init__
        const double d = hypot(x0[4 * b] - g.x, x0[4 * b + 1] - g.y);
        bool isgoal = d <= P.goal_dis;
        long long df = ti - (long long)path_len[b];
        if ((df < 0 ? -df : df) >= 5) isgoal = false;
        const bool isstop = fabs(x0[4 * b + 2]) <= P.stop_speed;
        is_goal[b] = (isgoal && isstop) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------
struct jsim_ctx {
    jsim_cfg cfg;
    int device;
    double2 *d_pxy;
    double *d_pyaw;
    long long *d_poff;
    int n_paths;
    long long n_points;
    // per-vehicle loop glue (row f1): host copy of the paths, car circle geometry, per-point circle centres, predictions
    double *h_cx, *h_cy, *h_cyaw;
    int have_geom;
    double cc0, cc1, col_radius;
    double *d_get_all;     // [ticks][n_obs][6]  obstacle get() tuples of a fused scenario run
    double2 *d_pred_all;   // [ticks][n_obs][n_steps][2]
    double4 *d_bc_all;     // [ticks][n_obs] bounding circles of the predictions
    double4 *d_pred_bc;    // [JSIM_MAX_OBS] of the current single-tick prediction
    size_t get_all_cap, pred_all_cap; // in elements
    double occ0, occ1, ocol_radius, oL; // the obstacles' circles / wheelbase (jsim_loop_set_obstacle_geometry); default: the ego's
    int have_ogeom;
    double2 *d_pcc;
    double2 *d_pred_cc;
    int pred_n_obs, pred_n_steps;
    double *d_pcv;          // speed reference per path point (mpc_with_speed variant) or NULL
    const int *cv_cut;      // caller-owned device array [B] or NULL
    size_t lds_bytes;
    const double *d_pe; // per-ego weights / limits (caller-owned device array) or NULL
    void *comm;         // ncclComm_t of jsim_comm_init (RCCL), or NULL
    int *d_order;       // launch order of the fused closed-loop launches (prepare_launch_order) ..
    unsigned *d_work;   // .. and the per-ego iteration count of the previous launch it is derived from
    unsigned long long *d_iters; // per-ego running totals of active-set iterations over the fused launches (jsim_mpc_iter_totals)
    int iters_cap;
    int order_cap;
    int order_mode;     // 0: from the environment (default on), 1: on, -1: off (jsim_mpc_set_launch_order)
    int use_reg_kernel; // 1: register-resident fast path available for this T (and not disabled)
    int dbg_max_gi;
    long long *dbg_clk; // diagnostic builds only
    char err[512];
};

static thread_local char g_err[512] = "";

static int fail(jsim_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof(g_err), "%s", buf);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", buf);
    return code;
}

#define HIP_TRY(ctx, call)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return fail(ctx, -5, "%s failed: %s", #call, hipGetErrorString(e_));             \
    } while (0)

// Every entry point that launches, allocates or copies runs on the context's device and leaves the caller's current device
// as it found it (two contexts on different GPUs in one process; a caller whose current device is another one).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess; // a failed hipGetDevice / hipSetDevice: the entry point must not carry on on the caller's device
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    bool ok() const { return err == hipSuccess; }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define JSIM_GUARD_OK(ctx_)                                                                                   \
    do {                                                                                                       \
        if (!dev_guard.ok()) return fail(ctx_, -5, "switching to the context's device failed: %s", hipGetErrorString(dev_guard.err)); \
    } while (0)

extern "C" int jsim_abi_version(void) { return JSIM_ABI_VERSION; }

extern "C" const char *jsim_last_error(const jsim_ctx *ctx) { return ctx ? ctx->err : g_err; }

extern "C" int jsim_mpc_create(const jsim_cfg *cfg, int device_id, jsim_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, -22, "jsim_mpc_create: null argument");
    if (cfg->T < 1 || cfg->T > JSIM_MAX_T) return fail(nullptr, -22, "jsim_mpc_create: T=%d outside [1, %d]", cfg->T, JSIM_MAX_T);
    if (cfg->max_iter < 1 || cfg->max_iter > 16) return fail(nullptr, -22, "jsim_mpc_create: MAX_ITER=%d (1..16)", cfg->max_iter);
    if (!(cfg->dt > 0) || !(cfg->dl > 0) || !(cfg->L > 0)) return fail(nullptr, -22, "jsim_mpc_create: dt, dl, L must be positive");
    if (!(cfg->R[0] > 0) || !(cfg->R[1] > 0) || !(cfg->R_end[0] > 0) || !(cfg->R_end[1] > 0))
        return fail(nullptr, -22, "jsim_mpc_create: R / R_end must be positive (strict convexity)");
    if (cfg->nx != 4 && cfg->nx != 5) return fail(nullptr, -22, "jsim_mpc_create: NX=%d (4: lib/mpc.py, 5: lib/mpc_jerk.py)", cfg->nx);
    if (cfg->nx == 5 && !(cfg->jerk_weight >= 0)) return fail(nullptr, -22, "jsim_mpc_create: jerk_weight must be >= 0");
    int ndev = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(nullptr, -19, "jsim_mpc_create: no HIP device");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, -22, "jsim_mpc_create: device %d of %d", device_id, ndev);
    const size_t lds_bytes = jsim_lds_doubles(cfg->T, cfg->nx == 5) * sizeof(double);
    if (lds_bytes > 160 * 1024) return fail(nullptr, -22, "jsim_mpc_create: T=%d needs %zu B of LDS (> 160 KiB)", cfg->T, lds_bytes);
    DeviceGuard dev_guard(device_id);   // the caller's current device is left as it was
    JSIM_GUARD_OK(nullptr);
    // dynamic LDS above the 64 KiB default has to be granted per kernel; done once here so that the step call
    // itself is pure launches (it may be captured into a hipGraph)
    HIP_TRY(nullptr, hipFuncSetAttribute((const void *)mpc_step_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(nullptr, hipFuncSetAttribute((const void *)mpc_step_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(nullptr, hipFuncSetAttribute((const void *)mpc_step_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(nullptr, hipFuncSetAttribute((const void *)mpc_step_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    jsim_ctx *c = new (std::nothrow) jsim_ctx();
    if (!c) return fail(nullptr, -12, "jsim_mpc_create: out of memory");
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->device = device_id;
    c->lds_bytes = lds_bytes;
    // register-resident fast path: instantiated for the stock horizon (13) and the benchmark horizon (20);
    // JSIM_FORCE_LDS_KERNEL=1 routes those through the generic LDS-resident kernel too (used by the tests to
    // cover both kernels on the same inputs)
    const char *force = getenv("JSIM_FORCE_LDS_KERNEL");
    c->use_reg_kernel = cfg->nx == 4 && has_reg_kernel(cfg->T) && !(force && force[0] == '1');
    const char *mg = getenv("JSIM_DEBUG_MAX_GI");
    c->dbg_max_gi = mg ? atoi(mg) : -1;
    *out = c;
    return 0;
}

static void free_paths(jsim_ctx *c)
{
    if (c->d_pxy) (void)hipFree(c->d_pxy);
    if (c->d_pyaw) (void)hipFree(c->d_pyaw);
    if (c->d_poff) (void)hipFree(c->d_poff);
    if (c->d_pcc) (void)hipFree(c->d_pcc);
    if (c->d_pcv) (void)hipFree(c->d_pcv);
    c->d_pcv = nullptr;
    delete[] c->h_cx; delete[] c->h_cy; delete[] c->h_cyaw;
    c->h_cx = c->h_cy = c->h_cyaw = nullptr; c->d_pcc = nullptr;
    c->d_pxy = nullptr; c->d_pyaw = nullptr; c->d_poff = nullptr;
    c->n_paths = 0; c->n_points = 0;
}

extern "C" void jsim_mpc_destroy(jsim_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->comm) (void)jsim_comm_destroy(ctx);
    DeviceGuard dev_guard(ctx->device);   // (a failed switch still frees: hipFree takes the pointers' own device)
    free_paths(ctx);
    if (ctx->d_pred_cc) (void)hipFree(ctx->d_pred_cc);
    if (ctx->d_get_all) (void)hipFree(ctx->d_get_all);
    if (ctx->d_pred_all) (void)hipFree(ctx->d_pred_all);
    if (ctx->d_bc_all) (void)hipFree(ctx->d_bc_all);
    if (ctx->d_pred_bc) (void)hipFree(ctx->d_pred_bc);
    if (ctx->d_order) (void)hipFree(ctx->d_order);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_iters) (void)hipFree(ctx->d_iters);
    delete ctx;
}

// circle centres of every path point (lib/trajectories.py:11-55 with the two body-axis circles of
// lib/car_dimensions.py:66-79): (cos(yaw) * x_off - sin(yaw) * 0.0) + x, (sin(yaw) * x_off + cos(yaw) * 0.0) + y
static int upload_circle_centres(jsim_ctx *ctx)
{
    const long long N = ctx->n_points;
    if (N <= 0 || !ctx->h_cx) return 0;
    std::vector<double2> h((size_t)2 * N);
    for (long long i = 0; i < N; ++i) {
        const double c = std::cos(ctx->h_cyaw[i]), s = std::sin(ctx->h_cyaw[i]);
        h[2 * i].x = c * ctx->cc0 - s * 0.0 + ctx->h_cx[i];     h[2 * i].y = s * ctx->cc0 + c * 0.0 + ctx->h_cy[i];
        h[2 * i + 1].x = c * ctx->cc1 - s * 0.0 + ctx->h_cx[i]; h[2 * i + 1].y = s * ctx->cc1 + c * 0.0 + ctx->h_cy[i];
    }
    if (ctx->d_pcc) { (void)hipFree(ctx->d_pcc); ctx->d_pcc = nullptr; }
    HIP_TRY(ctx, hipMalloc(&ctx->d_pcc, sizeof(double2) * 2 * N));
    HIP_TRY(ctx, hipMemcpy(ctx->d_pcc, h.data(), sizeof(double2) * 2 * N, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int jsim_mpc_set_paths(jsim_ctx *ctx, const double *cx, const double *cy, const double *cyaw,
                                  const int64_t *path_off, int32_t n_paths)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_set_paths: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (!cx || !cy || !cyaw || !path_off || n_paths < 1) return fail(ctx, -22, "jsim_mpc_set_paths: bad argument");
    if (path_off[0] != 0) return fail(ctx, -22, "jsim_mpc_set_paths: path_off[0] must be 0");
    for (int i = 0; i < n_paths; ++i)
        if (path_off[i + 1] <= path_off[i]) return fail(ctx, -22, "jsim_mpc_set_paths: path %d is empty", i);
    const long long N = path_off[n_paths];
    free_paths(ctx);
    // resident path table: xy interleaved (one 16-byte load per point in the nearest-index scan) + yaw
    double2 *h = new (std::nothrow) double2[N];
    if (!h) return fail(ctx, -12, "jsim_mpc_set_paths: out of host memory");
    for (long long i = 0; i < N; ++i) { h[i].x = cx[i]; h[i].y = cy[i]; }
    hipError_t e = hipMalloc(&ctx->d_pxy, sizeof(double2) * N);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_pyaw, sizeof(double) * N);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_poff, sizeof(long long) * (n_paths + 1));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_pxy, h, sizeof(double2) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_pyaw, cyaw, sizeof(double) * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_poff, path_off, sizeof(long long) * (n_paths + 1), hipMemcpyHostToDevice);
    delete[] h;
    if (e != hipSuccess) { free_paths(ctx); return fail(ctx, -5, "jsim_mpc_set_paths: %s", hipGetErrorString(e)); }
    ctx->n_paths = n_paths;
    ctx->n_points = N;
    ctx->h_cx = new (std::nothrow) double[N]; ctx->h_cy = new (std::nothrow) double[N]; ctx->h_cyaw = new (std::nothrow) double[N];
    if (!ctx->h_cx || !ctx->h_cy || !ctx->h_cyaw) return fail(ctx, -12, "jsim_mpc_set_paths: out of host memory");
    memcpy(ctx->h_cx, cx, sizeof(double) * N); memcpy(ctx->h_cy, cy, sizeof(double) * N); memcpy(ctx->h_cyaw, cyaw, sizeof(double) * N);
    if (ctx->have_geom) return upload_circle_centres(ctx);
    return 0;
}

static void fill_kp(const jsim_ctx *ctx, int32_t B, KP &P)
{
    const jsim_cfg &c = ctx->cfg;
    memset(&P, 0, sizeof(P));
    P.T = c.T; P.n = jsim_nvar(c.T, c.nx == 5); P.ld = jsim_ld(P.n); P.B = B;
    P.jerkw = c.jerk_weight;
    P.dt = c.dt; P.dl = c.dl; P.L = c.L; P.w_perp = c.w_perp; P.w_para = c.w_para;
    P.R0 = c.R[0]; P.R1 = c.R[1]; P.Rd0 = c.Rd[0]; P.Rd1 = c.Rd[1]; P.Qv = c.Q_v_yaw[0]; P.Qyaw = c.Q_v_yaw[1];
    P.Qf0 = c.Qf[0] * c.T; P.Qf1 = c.Qf[1] * c.T; P.Qf2 = c.Qf[2] * c.T; P.Qf3 = c.Qf[3] * c.T; // mpc.py:28
    P.Re0 = c.R_end[0]; P.Re1 = c.R_end[1];
    P.dmax = c.max_dsteer * c.dt; P.amax = c.max_accel; P.amin = c.max_decel; P.smax = c.max_steer;
    P.vmax_plant = c.max_speed; P.vmin = c.min_speed; P.vref_min = c.min_ref_speed;
    P.pxy = ctx->d_pxy; P.pyaw = ctx->d_pyaw; P.poff = ctx->d_poff;
    P.pcv = ctx->d_pcv; P.cv_cut = ctx->cv_cut; P.pe = ctx->d_pe;
}

// ---------------------------------------------------------------------------------------------------
// Launch order of the fused closed-loop launches.  A launch ends when its slowest ego does, and with more egos than
// the chip holds at once (1024 one-wave egos at T = 13 / 20 / 30, 512 four-wave egos at T = 40) the workgroups of the
// second round start when a slot frees up: an ego far from its path -- four to five times the mean number of active-set
// iterations per tick, tick after tick -- that starts late ends the launch late.  Workgroups are dispatched in blockIdx
// order, so workgroup b is given ego order[b], the egos ranked by the iterations they needed in the PREVIOUS launch, most
// first (longest-processing-time-first list scheduling).  Egos are independent: the order changes when an ego is solved,
// never what is computed for it.  JSIM_LAUNCH_ORDER=0 in the environment keeps the identity order (A/B runs).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void launch_order_kernel(const unsigned *work, int B, int *order)
{
    __shared__ unsigned tile[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned wi = i < B ? work[i] : 0u;
    int rank = 0; // egos ahead of ego i: more work, or the same work and a lower index (a permutation whatever the values)
    for (int base = 0; base < B; base += 1024) {
        __syncthreads();
        for (int j = threadIdx.x; j < 1024; j += 256) tile[j] = base + j < B ? work[base + j] : 0u;
        __syncthreads();
        const int n = B - base < 1024 ? B - base : 1024;
        for (int j = 0; j < n; ++j) {
            const unsigned wj = tile[j];
            rank += (wj > wi || (wj == wi && base + j < i)) ? 1 : 0;
        }
    }
    if (i < B) order[rank] = i;
}

static int prepare_launch_order(jsim_ctx *ctx, int B, hipStream_t s, TickP &K)
{
    static const int enabled = [] { const char *e = getenv("JSIM_LAUNCH_ORDER"); return !(e && e[0] == '0'); }();
    K.order = nullptr; K.work = nullptr;
    const bool on = ctx->order_mode ? ctx->order_mode > 0 : (bool)enabled;
    if (!on || B < 512 || B > 65536) return 0; // fewer egos than slots: all start at once; beyond: O(B^2) ranking not worth it
    if (ctx->order_cap < B) {
        if (ctx->d_order) (void)hipFree(ctx->d_order);
        if (ctx->d_work) (void)hipFree(ctx->d_work);
        ctx->d_order = nullptr; ctx->d_work = nullptr; ctx->order_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_order, sizeof(int) * (size_t)B));
        HIP_TRY(ctx, hipMalloc(&ctx->d_work, sizeof(unsigned) * (size_t)B));
        ctx->order_cap = B;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned) * (size_t)B, s));
    }
    hipLaunchKernelGGL(launch_order_kernel, dim3((B + 255) / 256), dim3(256), 0, s, ctx->d_work, B, ctx->d_order);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned) * (size_t)B, s));
    K.order = ctx->d_order; K.work = ctx->d_work;
    return 0;
}

// Per-ego running totals of active-set iterations, added to by every fused launch (one 8-byte update per ego and tick).
static int prepare_iter_totals(jsim_ctx *ctx, int B, hipStream_t s, TickP &K)
{
    K.iters = nullptr;
    if (ctx->iters_cap < B) {
        if (ctx->d_iters) (void)hipFree(ctx->d_iters);
        ctx->d_iters = nullptr; ctx->iters_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_iters, sizeof(unsigned long long) * (size_t)B));
        ctx->iters_cap = B;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_iters, 0, sizeof(unsigned long long) * (size_t)B, s));
    }
    K.iters = ctx->d_iters;
    return 0;
}

extern "C" int jsim_mpc_iter_totals(jsim_ctx *ctx, int32_t B, uint64_t *totals, int32_t reset)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_iter_totals: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B <= 0 || B > ctx->iters_cap) return fail(ctx, -22, "jsim_mpc_iter_totals: B=%d, but the fused launches of this context had at most %d egos", B, ctx->iters_cap);
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (totals) HIP_TRY(ctx, hipMemcpy(totals, ctx->d_iters, sizeof(uint64_t) * (size_t)B, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(ctx, hipMemset(ctx->d_iters, 0, sizeof(unsigned long long) * (size_t)ctx->iters_cap));
    return 0;
}

extern "C" int jsim_mpc_set_launch_order(jsim_ctx *ctx, int32_t enabled)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_set_launch_order: null ctx");
    ctx->order_mode = enabled ? 1 : -1;
    return 0;
}

extern "C" int jsim_mpc_get_launch_order(jsim_ctx *ctx, int32_t B, int32_t *order, uint32_t *work)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_get_launch_order: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B <= 0 || B > ctx->order_cap) return fail(ctx, -22, "jsim_mpc_get_launch_order: B=%d, but the last ordered launch had %d egos", B, ctx->order_cap);
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (order) HIP_TRY(ctx, hipMemcpy(order, ctx->d_order, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    if (work) HIP_TRY(ctx, hipMemcpy(work, ctx->d_work, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return 0;
}

static int launch_step(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id, const int32_t *path_len,
                       const double *speed, int64_t *target_ind, double *oa, double *od, double *ox, double *oy,
                       double *ov, double *oyaw, double *xref, uint32_t *active_mask, int32_t *status, int32_t *n_iter,
                       double *xbar, int64_t *ref_idx, double *H, double *g, double *lam, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_step: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0) return fail(ctx, -22, "jsim_mpc_step: B=%d", B);
    if (B == 0) return 0;
    if (!x0 || !path_id || !path_len || !speed || !target_ind || !oa || !od || !status)
        return fail(ctx, -22, "jsim_mpc_step: a required device pointer is null");
    if (!ctx->d_pxy) return fail(ctx, -22, "jsim_mpc_step: jsim_mpc_set_paths has not been called");
    const jsim_cfg &c = ctx->cfg;
    KP P;
    fill_kp(ctx, B, P);
    P.x0 = x0; P.path_id = path_id; P.path_len = path_len; P.speed = speed;
    P.target_ind = (long long *)target_ind; P.oa = oa; P.od = od; P.ox = ox; P.oy = oy; P.ov = ov; P.oyaw = oyaw;
    P.xref = xref; P.amask = active_mask; P.status = status; P.n_iter = n_iter;
    P.dbg_xbar = xbar; P.dbg_idx = (long long *)ref_idx; P.dbg_H = H; P.dbg_g = g; P.dbg_lam = lam;
    P.dbg_clk = ctx->dbg_clk;
    P.dbg_max_gi = ctx->dbg_max_gi;

    const size_t lds_bytes = ctx->lds_bytes;
    hipStream_t s = (hipStream_t)stream;
    TickP K;
    memset(&K, 0, sizeof(K));
    K.n_ticks = 1; // plain MPC.step: one tick, no plant/bookkeeping
    if (c.max_iter > 1 && !ov) return fail(ctx, -22, "jsim_mpc_step: MAX_ITER=%d needs the ov buffer (the next pass's travel distances)", c.max_iter);
    for (int pass = 0; pass < c.max_iter; ++pass) { // _iterative_linear_mpc_control, main/lib/mpc.py:231-236: one launch per pass
        P.pass = pass;
        if (ctx->use_reg_kernel) launch_reg(c.T, B, s, P, K);
        else if (c.nx == 5) {
            if (P.n <= 64) hipLaunchKernelGGL((mpc_step_kernel<1, true>), dim3(B), dim3(64), lds_bytes, s, P);
            else hipLaunchKernelGGL((mpc_step_kernel<2, true>), dim3(B), dim3(64), lds_bytes, s, P);
        } else if (P.n <= 64) hipLaunchKernelGGL((mpc_step_kernel<1, false>), dim3(B), dim3(64), lds_bytes, s, P);
        else hipLaunchKernelGGL((mpc_step_kernel<2, false>), dim3(B), dim3(64), lds_bytes, s, P);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int jsim_mpc_step(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                             const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa,
                             double *od, double *ox, double *oy, double *ov, double *oyaw, double *xref,
                             uint32_t *active_mask, int32_t *status, int32_t *n_iter, void *stream)
{
    return launch_step(ctx, B, x0, path_id, path_len, speed, target_ind, oa, od, ox, oy, ov, oyaw, xref, active_mask,
                       status, n_iter, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int jsim_mpc_step_debug(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                                   const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa,
                                   double *od, double *ox, double *oy, double *ov, double *oyaw, double *xref,
                                   uint32_t *active_mask, int32_t *status, int32_t *n_iter, double *xbar,
                                   int64_t *ref_idx, double *H, double *g, double *lam, void *stream)
{
    return launch_step(ctx, B, x0, path_id, path_len, speed, target_ind, oa, od, ox, oy, ov, oyaw, xref, active_mask,
                       status, n_iter, xbar, ref_idx, H, g, lam, stream);
}

extern "C" int jsim_plant_step(jsim_ctx *ctx, int32_t B, double *x0, const double *oa, const double *od,
                               const int32_t *status, double *di_ai, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_plant_step: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0) return fail(ctx, -22, "jsim_plant_step: B=%d", B);
    if (B == 0) return 0;
    if (!x0 || !oa || !od || !status || !di_ai) return fail(ctx, -22, "jsim_plant_step: null device pointer");
    const jsim_cfg &c = ctx->cfg;
    PlantP P = {B, c.T, c.dt, c.L, c.max_steer, c.max_speed, c.min_speed, c.max_decel, ctx->d_pe};
    hipLaunchKernelGGL(plant_step_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, x0, oa, od, status, di_ai);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int jsim_mpc_xref_deviation_goal(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id,
                                            const int32_t *path_len, const int64_t *target_ind, const double *ox,
                                            const double *oy, double *deviation, int32_t *is_goal, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_xref_deviation_goal: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0) return fail(ctx, -22, "jsim_mpc_xref_deviation_goal: B=%d", B);
    if (B == 0) return 0;
    if (!x0 || !path_id || !path_len || !target_ind || (deviation && (!ox || !oy)))
        return fail(ctx, -22, "jsim_mpc_xref_deviation_goal: null device pointer");
    if (!ctx->d_pxy) return fail(ctx, -22, "jsim_mpc_xref_deviation_goal: no paths set");
    const jsim_cfg &c = ctx->cfg;
    GoalP P = {B, c.T, c.goal_dis, c.stop_speed, ctx->d_pxy, ctx->d_pyaw, ctx->d_poff};
    hipLaunchKernelGGL(deviation_goal_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, x0, path_id,
                       path_len, (const long long *)target_ind, ox, oy, deviation, is_goal);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int jsim_loop_advance(jsim_ctx *ctx, int32_t B, double *x0, double *oa, double *od, const int32_t *status,
                                 double *di_ai, int64_t *target_ind, const int32_t *path_id, const int32_t *path_len,
                                 const double *x0_spawn, const int64_t *target_spawn, int32_t *age, int32_t max_age,
                                 double *hist, int32_t *tick, int32_t hist_cap, uint64_t *n_respawn, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_advance: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0) return fail(ctx, -22, "jsim_loop_advance: B=%d", B);
    if (B == 0) return 0;
    if (!x0 || !oa || !od || !status || !di_ai || !target_ind || !path_id || !path_len || !x0_spawn || !target_spawn || !age)
        return fail(ctx, -22, "jsim_loop_advance: null device pointer");
    if (hist && !tick) return fail(ctx, -22, "jsim_loop_advance: hist needs a device tick counter");
    if (!ctx->d_pxy) return fail(ctx, -22, "jsim_loop_advance: no paths set");
    const jsim_cfg &c = ctx->cfg;
    LoopP P = {B, c.T, max_age > 0 ? max_age : 0x7fffffff, c.dt, c.L, c.max_steer, c.max_speed, c.min_speed,
               c.max_decel, c.goal_dis, c.stop_speed, ctx->d_pxy, ctx->d_poff, ctx->d_pe};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loop_advance_kernel, dim3((B + 255) / 256), dim3(256), 0, s, P, x0, oa, od, status, di_ai,
                       (long long *)target_ind, path_id, path_len, x0_spawn, (const long long *)target_spawn, age, hist,
                       tick, hist_cap, (unsigned long long *)n_respawn);
    if (tick) hipLaunchKernelGGL(tick_increment_kernel, dim3(1), dim3(1), 0, s, tick);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#if defined(JSIM_STAMPS) || defined(JSIM_SPAN)
// diagnostic build only (not declared in include/jsim_mpc.h, not built into libjsim_mpc.so)
extern "C" int jsim_debug_set_clock_buffer(jsim_ctx *ctx, long long *dev_buf)
{
    if (!ctx) return -22;
    ctx->dbg_clk = dev_buf;
    return 0;
}
#endif

extern "C" int jsim_mpc_run_ticks(jsim_ctx *ctx, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id,
                                  const int32_t *path_len, const double *speed, int64_t *target_ind, double *oa,
                                  double *od, double *ox, double *oy, double *ov, double *oyaw, double *xref,
                                  uint32_t *active_mask, int32_t *status, int32_t *n_iter, double *di_ai,
                                  const double *x0_spawn, const int64_t *target_spawn, int32_t *age, int32_t max_age,
                                  double *hist, int32_t *tick, int32_t hist_cap, uint64_t *n_respawn, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_run_ticks: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0 || n_ticks < 0) return fail(ctx, -22, "jsim_mpc_run_ticks: B=%d n_ticks=%d", B, n_ticks);
    if (B == 0 || n_ticks == 0) return 0;
    if (!x0 || !path_id || !path_len || !speed || !target_ind || !oa || !od || !status || !di_ai || !x0_spawn ||
        !target_spawn || !age)
        return fail(ctx, -22, "jsim_mpc_run_ticks: a required device pointer is null");
    if (hist && !tick) return fail(ctx, -22, "jsim_mpc_run_ticks: hist needs a device tick counter");
    if (!ctx->d_pxy) return fail(ctx, -22, "jsim_mpc_run_ticks: jsim_mpc_set_paths has not been called");
    const jsim_cfg &c = ctx->cfg;
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->use_reg_kernel || c.max_iter > 1) {
        // horizons without the fused register kernel, or several linearisation passes per step: the same ticks as separate launches
        for (int k = 0; k < n_ticks; ++k) {
            int rc = launch_step(ctx, B, x0, path_id, path_len, speed, target_ind, oa, od, ox, oy, ov, oyaw, xref,
                                 active_mask, status, n_iter, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
            if (rc) return rc;
            rc = jsim_loop_advance(ctx, B, x0, oa, od, status, di_ai, target_ind, path_id, path_len, x0_spawn,
                                   target_spawn, age, max_age, hist, tick, hist_cap, n_respawn, stream);
            if (rc) return rc;
        }
        return 0;
    }
    KP P;
    fill_kp(ctx, B, P);
    P.x0 = x0; P.path_id = path_id; P.path_len = path_len; P.speed = speed;
    P.target_ind = (long long *)target_ind; P.oa = oa; P.od = od; P.ox = ox; P.oy = oy; P.ov = ov; P.oyaw = oyaw;
    P.xref = xref; P.amask = active_mask; P.status = status; P.n_iter = n_iter;
    P.dbg_clk = nullptr; P.dbg_max_gi = ctx->dbg_max_gi;
    TickP K;
    memset(&K, 0, sizeof(K));
    K.n_ticks = n_ticks; K.advance = 1; K.max_age = max_age > 0 ? max_age : 0x7fffffff; K.hist_cap = hist_cap;
    K.max_decel = c.max_decel; K.goal_dis = c.goal_dis; K.stop_speed = c.stop_speed;
    K.x0w = x0; K.di_ai = di_ai; K.x0_spawn = x0_spawn; K.target_spawn = (const long long *)target_spawn; K.age = age;
    K.hist = hist; K.tick = tick; K.n_respawn = (unsigned long long *)n_respawn;
    if (int rc_ = prepare_launch_order(ctx, B, s, K)) return rc_;
    if (int rc_ = prepare_iter_totals(ctx, B, s, K)) return rc_;
    launch_reg(c.T, B, s, P, K);
    if (tick) hipLaunchKernelGGL(tick_add_kernel, dim3(1), dim3(1), 0, s, tick, n_ticks);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// The job's one exchange: all-gather of the per-rank result blocks over RCCL (xGMI).  Egos are independent
// (main/lib/mpc.py:141-211 couples nothing), so nothing is exchanged on the solve path; this is the final trajectory gather.
// librccl is loaded on first use -- the library itself links only libamdhip64 and single-GPU users never touch RCCL.
// ---------------------------------------------------------------------------------------------------
struct JsimNcclId { char internal[128]; }; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value like the original
namespace {
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, JsimNcclId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}
static RcclApi g_rccl;

static int rccl_load(jsim_ctx *ctx)
{
    if (g_rccl.h) return 0;
    const char *names[] = {getenv("JSIM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if (n && n[0] && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(ctx, -2, "RCCL not found (librccl.so.1; set JSIM_RCCL_LIB): %s", dlerror());
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, JsimNcclId, int))dlsym(h, "ncclCommInitRank");
    a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    a.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString)
        return fail(ctx, -2, "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    g_rccl = a;
    return 0;
}

extern "C" int jsim_comm_unique_id(void *id128)
{
    if (!id128) return fail(nullptr, -22, "jsim_comm_unique_id: null argument");
    if (int rc = rccl_load(nullptr)) return rc;
    const int r = g_rccl.GetUniqueId(id128);
    if (r) return fail(nullptr, -5, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    return 0;
}

extern "C" int jsim_comm_init(jsim_ctx *ctx, const void *id128, int32_t n_ranks, int32_t rank)
{
    if (!ctx) return fail(nullptr, -22, "jsim_comm_init: null ctx");
    if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, -22, "jsim_comm_init: bad argument (rank %d of %d)", rank, n_ranks);
    if (ctx->comm) return fail(ctx, -17, "jsim_comm_init: this context already has a communicator");
    if (int rc = rccl_load(ctx)) return rc;
    DeviceGuard dev_guard(ctx->device); // ncclCommInitRank binds the communicator to the current device
    JSIM_GUARD_OK(ctx);
    JsimNcclId id;
    memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, n_ranks, id, rank);
    if (r) return fail(ctx, -5, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, g_rccl.GetErrorString(r));
    ctx->comm = comm;
    return 0;
}

extern "C" int jsim_mpc_gather(jsim_ctx *ctx, void *comm, const void *local, void *out, size_t bytes_per_rank, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_gather: null ctx");
    void *c = comm ? comm : ctx->comm;
    if (!c) return fail(ctx, -22, "jsim_mpc_gather: no communicator (jsim_comm_init, or pass an ncclComm_t)");
    if (bytes_per_rank == 0) return 0;
    if (!local || !out) return fail(ctx, -22, "jsim_mpc_gather: null device pointer");
    if (int rc = rccl_load(ctx)) return rc;
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    const int r = g_rccl.AllGather(local, out, bytes_per_rank, /* ncclChar */ 0, c, (hipStream_t)stream);
    if (r) return fail(ctx, -5, "ncclAllGather: %s", g_rccl.GetErrorString(r));
    return 0;
}

extern "C" int jsim_comm_destroy(jsim_ctx *ctx)
{
    if (!ctx) return fail(nullptr, -22, "jsim_comm_destroy: null ctx");
    if (!ctx->comm) return 0;
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    const int r = g_rccl.h ? g_rccl.CommDestroy(ctx->comm) : 0;
    ctx->comm = nullptr;
    if (r) return fail(ctx, -5, "ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return 0;
}

#include "planner.inc"

// Route planner: HOST pointers in and out -- a one-time precompute whose (M, 3) output is what
// jsim_mpc_set_paths takes; device buffers are allocated, filled, searched (one wavefront per route) and read back inside the call.
extern "C" int jsim_plan_routes(int device_id, int32_t n_routes, const double *start, const double *goal, const double *goal_box,
                                const double *tol, const double *hp, const int32_t *hp_off, int32_t n_obs_total,
                                const int32_t *route_obs_off, const double *mp_pts, const double *mp_len, int32_t n_prim,
                                int32_t n_pts, const double *cc_pts, const int32_t *cc_off, const double *wh, const double *wc,
                                int32_t max_path, int32_t node_cap, int32_t *status, double *cost, int32_t *n_prims, int32_t *prims,
                                double *nodes, double *traj, int32_t *n_expanded)
{
    if (n_routes < 0 || n_prim < 1 || n_prim > JPL_MAX_PRIM || n_pts < 2 || max_path < 1 || n_obs_total < 0 || node_cap < 64 || node_cap > (1 << 24))
        return fail(nullptr, -22, "jsim_plan_routes: bad sizes (routes %d, primitives %d (max %d), points %d, max_path %d)", n_routes, n_prim,
                    JPL_MAX_PRIM, n_pts, max_path);
    if (n_routes == 0) return 0;
    if (!start || !goal || !goal_box || !tol || !hp_off || !route_obs_off || !mp_pts || !mp_len || !cc_pts || !cc_off || !wh || !wc ||
        !status || !cost || !n_prims || !prims || !nodes || !traj || !n_expanded || (n_obs_total > 0 && !hp))
        return fail(nullptr, -22, "jsim_plan_routes: null argument");
    // the offset tables index device arrays from inside the kernel: they must be what they claim to be
    for (int k = 0; k < n_obs_total; ++k)
        if (hp_off[k] < 0 || hp_off[k + 1] < hp_off[k]) return fail(nullptr, -22, "jsim_plan_routes: hp_off is not non-decreasing at %d", k);
    if (hp_off[0] != 0) return fail(nullptr, -22, "jsim_plan_routes: hp_off[0] = %d", hp_off[0]);
    for (int k = 0; k < n_routes; ++k)
        if (route_obs_off[k] < 0 || route_obs_off[k + 1] < route_obs_off[k] || route_obs_off[k + 1] > n_obs_total)
            return fail(nullptr, -22, "jsim_plan_routes: route_obs_off[%d..%d] = %d, %d with %d obstacles", k, k + 1, route_obs_off[k], route_obs_off[k + 1], n_obs_total);
    for (int k = 0; k < n_prim; ++k)
        if (cc_off[k] < 0 || cc_off[k + 1] < cc_off[k]) return fail(nullptr, -22, "jsim_plan_routes: cc_off is not non-decreasing at %d", k);
    int ndev = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, -19, "jsim_plan_routes: device %d of %d", device_id, ndev);
    DeviceGuard dev_guard(device_id);
    JSIM_GUARD_OK(nullptr);
    const int R = n_routes, cap = node_cap, seg = n_pts - 1;
    // sizes in 64 bits BEFORE anything is allocated: per route ~84 B per node slot (+ the hash table) and the output arrays
    if (max_path > 4096 || n_pts > 4096) return fail(nullptr, -22, "jsim_plan_routes: max_path %d / points per primitive %d above 4096", max_path, n_pts);
    {
        const unsigned long long per_route = 84ull * (unsigned long long)cap * 3ull + 8ull * 3ull * (unsigned long long)max_path * (unsigned long long)seg;
        if ((unsigned long long)R * per_route > (64ull << 30))
            return fail(nullptr, -12, "jsim_plan_routes: %d routes x node_cap %d x max_path %d would need more than 64 GiB of device memory", R, cap, max_path);
    }
    int hash_cap = 128;
    while (hash_cap < 2 * cap) hash_cap <<= 1;
    // a circle around every obstacle (the kernel's exact pruning of the collision test): vertices = the pairwise intersections of its
    // half-plane boundaries that satisfy all the others; bounded iff the largest angular gap between consecutive normals is below pi
    std::vector<double> obc((size_t)(n_obs_total > 0 ? n_obs_total : 1) * 3, 0.0);
    for (int o = 0; o < n_obs_total; ++o) {
        const double *q = hp + (size_t)hp_off[o] * 3;
        const int n = hp_off[o + 1] - hp_off[o];
        double *c = &obc[(size_t)o * 3];
        c[0] = 0.0; c[1] = 0.0; c[2] = INFINITY;
        if (n < 3 || n > 64) continue;
        double ang[64];
        for (int i = 0; i < n; ++i) ang[i] = atan2(q[3 * i + 1], q[3 * i]);
        std::sort(ang, ang + n);
        double gap = ang[0] + 2.0 * M_PI - ang[n - 1];
        for (int i = 1; i < n; ++i) gap = std::max(gap, ang[i] - ang[i - 1]);
        if (!(gap < M_PI - 1e-9)) continue;                       // unbounded (or degenerate): no circle, never pruned
        double sx = 0.0, sy = 0.0, vx[64 * 2], vy[64 * 2];
        int nv = 0;
        for (int i = 0; i < n && nv < 128; ++i)
            for (int j = i + 1; j < n && nv < 128; ++j) {
                const double a1 = q[3 * i], b1 = q[3 * i + 1], c1 = q[3 * i + 2], a2 = q[3 * j], b2 = q[3 * j + 1], c2 = q[3 * j + 2];
                const double det = a1 * b2 - a2 * b1;
                if (fabs(det) <= 1e-12 * (fabs(a1) + fabs(b1)) * (fabs(a2) + fabs(b2))) continue;
                const double px = (b1 * c2 - b2 * c1) / det, py = (a2 * c1 - a1 * c2) / det;
                bool in = std::isfinite(px) && std::isfinite(py);
                for (int k = 0; k < n && in; ++k) {
                    const double v = q[3 * k] * px + q[3 * k + 1] * py + q[3 * k + 2];
                    in = v <= 1e-9 * (1.0 + fabs(q[3 * k + 2]) + (fabs(q[3 * k]) + fabs(q[3 * k + 1])) * (fabs(px) + fabs(py)));
                }
                if (in) { vx[nv] = px; vy[nv] = py; sx += px; sy += py; ++nv; }
            }
        if (nv < 3) continue;                                     // (an empty or degenerate set: left to the exact test)
        const double mx = sx / nv, my = sy / nv;
        double rr = 0.0;
        for (int i = 0; i < nv; ++i) rr = std::max(rr, hypot(vx[i] - mx, vy[i] - my));
        c[0] = mx; c[1] = my; c[2] = rr * (1.0 + 1e-9) + 1e-9;
    }
    const size_t n_hp = (size_t)hp_off[n_obs_total], n_cc = (size_t)cc_off[n_prim];
    std::vector<void *> owned;
    auto dalloc = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes ? bytes : 8) != hipSuccess) return nullptr; owned.push_back(q); return q; };
    auto put = [&](const void *src, size_t bytes) -> void * {
        void *q = dalloc(bytes);
        if (q && bytes && hipMemcpy(q, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return q;
    };
    auto cleanup = [&]() { for (void *q : owned) (void)hipFree(q); };
    PlanP P;
    memset(&P, 0, sizeof(P));
    P.n_routes = R; P.n_prim = n_prim; P.n_pts = n_pts; P.max_path = max_path; P.node_cap = cap;
    P.wh_dist = wh[0]; P.wh_theta = wh[1]; P.wh_steer = wh[2]; P.wh_obst = wh[3]; P.wh_center = wh[4];
    P.wc_dist = wc[0]; P.wc_steer = wc[1]; P.wc_obst = wc[2]; P.wc_center = wc[3];
    P.start = (const double *)put(start, sizeof(double) * 3 * R); P.goal = (const double *)put(goal, sizeof(double) * 3 * R);
    P.goal_box = (const double *)put(goal_box, sizeof(double) * 4 * R); P.tol = (const double *)put(tol, sizeof(double) * R);
    P.hp = (const double *)put(hp, sizeof(double) * 3 * n_hp); P.hp_off = (const int *)put(hp_off, sizeof(int) * (n_obs_total + 1));
    P.route_obs_off = (const int *)put(route_obs_off, sizeof(int) * (R + 1));
    P.obc = (const double *)put(obc.data(), sizeof(double) * obc.size());
    P.mp_pts = (const double *)put(mp_pts, sizeof(double) * 3 * (size_t)n_prim * n_pts); P.mp_len = (const double *)put(mp_len, sizeof(double) * n_prim);
    P.cc_pts = (const double *)put(cc_pts, sizeof(double) * 2 * n_cc); P.cc_off = (const int *)put(cc_off, sizeof(int) * (n_prim + 1));
    P.nx = (double *)dalloc(sizeof(double) * (size_t)R * cap); P.ny = (double *)dalloc(sizeof(double) * (size_t)R * cap);
    P.nth = (double *)dalloc(sizeof(double) * (size_t)R * cap); P.ng = (double *)dalloc(sizeof(double) * (size_t)R * cap);
    P.nparent = (int *)dalloc(sizeof(int) * (size_t)R * cap); P.nprim = (int *)dalloc(sizeof(int) * (size_t)R * cap);
    P.htab = (int *)dalloc(sizeof(int) * (size_t)R * hash_cap); P.hash_cap = hash_cap;
    P.ov_gh = (double *)dalloc(sizeof(double) * (size_t)R * cap); P.ov_g = (double *)dalloc(sizeof(double) * (size_t)R * cap);
    P.ov_id = (int *)dalloc(sizeof(int) * (size_t)R * cap);
    P.status = (int *)dalloc(sizeof(int) * R); P.n_prims = (int *)dalloc(sizeof(int) * R); P.n_expanded = (int *)dalloc(sizeof(int) * R);
    P.prims = (int *)dalloc(sizeof(int) * (size_t)R * max_path); P.cost = (double *)dalloc(sizeof(double) * R);
    P.nodes = (double *)dalloc(sizeof(double) * (size_t)R * (max_path + 1) * 3);
    P.traj = (double *)dalloc(sizeof(double) * (size_t)R * max_path * seg * 3);
    if (!P.start || !P.goal || !P.goal_box || !P.tol || !P.hp || !P.hp_off || !P.route_obs_off || !P.obc || !P.mp_pts || !P.mp_len || !P.cc_pts ||
        !P.cc_off || !P.nx || !P.ny || !P.nth || !P.ng || !P.nparent || !P.nprim || !P.htab || !P.ov_gh || !P.ov_g || !P.ov_id || !P.status || !P.n_prims || !P.n_expanded || !P.prims ||
        !P.cost || !P.nodes || !P.traj) {
        cleanup();
        return fail(nullptr, -12, "jsim_plan_routes: device allocation / upload failed");
    }
    hipError_t e = hipMemset(P.cost, 0, sizeof(double) * R);
    if (e == hipSuccess) e = hipMemset(P.htab, 0xff, sizeof(int) * (size_t)R * hash_cap);
    if (e == hipSuccess) e = hipMemset(P.traj, 0, sizeof(double) * (size_t)R * max_path * seg * 3);
    if (e == hipSuccess) e = hipMemset(P.nodes, 0, sizeof(double) * (size_t)R * (max_path + 1) * 3);
    if (e == hipSuccess) e = hipMemset(P.prims, 0xff, sizeof(int) * (size_t)R * max_path);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(plan_astar_kernel, dim3(R), dim3(64), 0, 0, P);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
#ifdef JPL_STAMPS
    {
        static long long clk[64][32];
        (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(jpl_clk), sizeof(clk));
        for (int r_ = 0; r_ < R && r_ < 64; ++r_)
            if (clk[r_][7] > 1000) {
                fprintf(stderr, "jpl route %d: %lld expansions, %lld nodes, %lld open at the end; per expansion:", r_, clk[r_][7], clk[r_][13], clk[r_][14]);
                for (int k = 0; k < 32; ++k) if (k != 7 && k != 13 && k != 14) fprintf(stderr, " [%d] %.2f", k, (double)clk[r_][k] / clk[r_][7]);
                fprintf(stderr, "\n");
            }
    }
#endif
    if (e == hipSuccess) e = hipMemcpy(status, P.status, sizeof(int) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cost, P.cost, sizeof(double) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(n_prims, P.n_prims, sizeof(int) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(n_expanded, P.n_expanded, sizeof(int) * R, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(prims, P.prims, sizeof(int) * (size_t)R * max_path, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(nodes, P.nodes, sizeof(double) * (size_t)R * (max_path + 1) * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(traj, P.traj, sizeof(double) * (size_t)R * max_path * seg * 3, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(nullptr, -5, "jsim_plan_routes: %s", hipGetErrorString(e));
    return 0;
}

// the largest double x with sqrt(x) <= thr (IEEE sqrt is correctly rounded and monotone): comparing a squared distance with it
// decides exactly what comparing its square root with thr decides
static double jsim_sqrt_threshold(double thr)
{
    double x = thr * thr;
    while (std::sqrt(x) > thr) x = std::nextafter(x, 0.0);
    while (std::sqrt(std::nextafter(x, INFINITY)) <= thr) x = std::nextafter(x, INFINITY);
    return x;
}

extern "C" int jsim_loop_set_geometry(jsim_ctx *ctx, double cc_front, double cc_rear, double radius)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_set_geometry: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (!(radius > 0)) return fail(ctx, -22, "jsim_loop_set_geometry: radius must be positive");
    ctx->cc0 = cc_front; ctx->cc1 = cc_rear; ctx->col_radius = radius; ctx->have_geom = 1;
    if (!ctx->have_ogeom) { ctx->occ0 = cc_front; ctx->occ1 = cc_rear; ctx->ocol_radius = radius; ctx->oL = ctx->cfg.L; }
    if (!ctx->d_pred_cc) HIP_TRY(ctx, hipMalloc(&ctx->d_pred_cc, sizeof(double2) * JSIM_MAX_OBS * JSIM_MAX_PRED * 2));
    if (!ctx->d_pred_bc) HIP_TRY(ctx, hipMalloc(&ctx->d_pred_bc, sizeof(double4) * JSIM_MAX_OBS));
    return upload_circle_centres(ctx);
}

extern "C" int jsim_loop_set_obstacle_geometry(jsim_ctx *ctx, double cc_front, double cc_rear, double radius, double wheelbase)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_set_obstacle_geometry: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (!(radius > 0) || !(wheelbase > 0)) return fail(ctx, -22, "jsim_loop_set_obstacle_geometry: radius and wheelbase must be positive");
    ctx->occ0 = cc_front; ctx->occ1 = cc_rear; ctx->ocol_radius = radius; ctx->oL = wheelbase; ctx->have_ogeom = 1;
    return 0;
}

extern "C" int jsim_loop_predict_obstacles(jsim_ctx *ctx, int32_t n_obs, const double *obst, int32_t n_steps, double *pred,
                                           void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_predict_obstacles: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (!ctx->have_geom) return fail(ctx, -22, "jsim_loop_predict_obstacles: jsim_loop_set_geometry has not been called");
    if (n_obs < 0 || n_obs > JSIM_MAX_OBS || n_steps < 1 || n_steps > JSIM_MAX_PRED)
        return fail(ctx, -22, "jsim_loop_predict_obstacles: n_obs=%d (max %d), n_steps=%d (max %d)", n_obs, JSIM_MAX_OBS, n_steps, JSIM_MAX_PRED);
    ctx->pred_n_obs = n_obs; ctx->pred_n_steps = n_steps;
    if (n_obs == 0) return 0;
    if (!obst || !pred) return fail(ctx, -22, "jsim_loop_predict_obstacles: null device pointer");
    ObsP P = {n_obs, n_steps, ctx->cfg.dt, ctx->oL, ctx->occ0, ctx->occ1, obst, pred, ctx->d_pred_cc, ctx->d_pred_bc};
    hipLaunchKernelGGL(obstacle_predict_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int jsim_loop_pre_tick(jsim_ctx *ctx, int32_t B, const double *x0, const int32_t *path_id, int64_t *traj_idx,
                                  const int32_t *prev_path_len, int32_t *path_len, int32_t *col_flag, double *col_xy,
                                  int32_t *first_idx, int32_t *status, int32_t frame_window, int32_t margin,
                                  int32_t *dbg_res_idx, int32_t *dbg_n_res, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_pre_tick: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0 || frame_window < 0 || frame_window > 32 || margin < 0) return fail(ctx, -22, "jsim_loop_pre_tick: bad argument");
    if (B == 0) return 0;
    if (!x0 || !path_id || !traj_idx || !prev_path_len || !path_len || !col_flag || !status)
        return fail(ctx, -22, "jsim_loop_pre_tick: null device pointer");
    if (!ctx->d_pxy || !ctx->d_pcc) return fail(ctx, -22, "jsim_loop_pre_tick: paths / geometry not set");
    if (dbg_res_idx && !dbg_n_res) return fail(ctx, -22, "jsim_loop_pre_tick: dbg_res_idx needs dbg_n_res");
    const jsim_cfg &c = ctx->cfg;
    PreP P;
    memset(&P, 0, sizeof(P));
    P.B = B; P.n_obs = ctx->pred_n_obs; P.n_steps = ctx->pred_n_steps; P.frame_window = frame_window; P.margin = margin;
    P.dt = c.dt; P.max_accel = c.max_accel; P.max_speed = c.max_speed; P.thr = ctx->col_radius + ctx->ocol_radius; // min_distance: 2 * radius, or car radius + bicycle radius
    P.thr_sq = jsim_sqrt_threshold(P.thr);
    P.pxy = ctx->d_pxy; P.pcc = ctx->d_pcc; P.poff = ctx->d_poff; P.pred_cc = ctx->d_pred_cc; P.pred_bc = ctx->d_pred_bc;
    P.x0 = x0; P.path_id = path_id; P.traj_idx = (long long *)traj_idx; P.prev_path_len = prev_path_len; P.path_len = path_len;
    P.col_flag = col_flag; P.col_xy = col_xy; P.first_idx = first_idx; P.status = status;
    P.dbg_res_idx = dbg_res_idx; P.dbg_n_res = dbg_n_res;
    hipLaunchKernelGGL(loop_pre_tick_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- MPC variants of the reference that differ from main/lib/mpc.py only in data ----
extern "C" int jsim_mpc_set_path_speed(jsim_ctx *ctx, const double *cv)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_set_path_speed: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (ctx->d_pcv) { (void)hipFree(ctx->d_pcv); ctx->d_pcv = nullptr; }
    if (!cv) return 0; // back to the plain controller (no speed reference)
    if (ctx->n_points <= 0) return fail(ctx, -22, "jsim_mpc_set_path_speed: call jsim_mpc_set_paths first");
    HIP_TRY(ctx, hipMalloc(&ctx->d_pcv, sizeof(double) * ctx->n_points));
    HIP_TRY(ctx, hipMemcpy(ctx->d_pcv, cv, sizeof(double) * ctx->n_points, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int jsim_mpc_set_speed_cutoff(jsim_ctx *ctx, const int32_t *cv_cut)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_set_speed_cutoff: null ctx");
    ctx->cv_cut = cv_cut;
    return 0;
}

extern "C" int jsim_mpc_update_cfg(jsim_ctx *ctx, const jsim_cfg *cfg)
{
    if (!ctx || !cfg) return fail(ctx, -22, "jsim_mpc_update_cfg: null argument");
    if (cfg->T != ctx->cfg.T) return fail(ctx, -22, "jsim_mpc_update_cfg: the horizon cannot change (T=%d -> %d)", ctx->cfg.T, cfg->T);
    if (cfg->nx != ctx->cfg.nx) return fail(ctx, -22, "jsim_mpc_update_cfg: NX cannot change (%d -> %d)", ctx->cfg.nx, cfg->nx);
    if (cfg->max_iter < 1 || cfg->max_iter > 16) return fail(ctx, -22, "jsim_mpc_update_cfg: MAX_ITER=%d (1..16)", cfg->max_iter);
    if (!(cfg->dt > 0) || !(cfg->dl > 0) || !(cfg->L > 0) || !(cfg->R[0] > 0) || !(cfg->R[1] > 0) || !(cfg->R_end[0] > 0) ||
        !(cfg->R_end[1] > 0))
        return fail(ctx, -22, "jsim_mpc_update_cfg: dt, dl, L, R, R_end must be positive");
    ctx->cfg = *cfg;
    return 0;
}

extern "C" int jsim_mpc_set_ego_config(jsim_ctx *ctx, const double *cfg)
{
    if (!ctx) return fail(nullptr, -22, "jsim_mpc_set_ego_config: null ctx");
    ctx->d_pe = cfg;
    return 0;
}

extern "C" int jsim_loop_obstacles(jsim_ctx *ctx, int32_t n_obs, double *state, const double *param, double *get, int32_t do_step,
                                   void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_obstacles: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (n_obs < 0 || n_obs > JSIM_MAX_OBS) return fail(ctx, -22, "jsim_loop_obstacles: n_obs=%d (max %d)", n_obs, JSIM_MAX_OBS);
    if (n_obs == 0) return 0;
    if (!state || !param) return fail(ctx, -22, "jsim_loop_obstacles: null device pointer");
    ObsStepP P = {n_obs, do_step ? 1 : 0, ctx->have_ogeom ? ctx->oL : ctx->cfg.L, state, param, get};
    hipLaunchKernelGGL(obstacle_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// an ego that was just respawned (age == 0 after the advance) starts its run like a new one: progress index 0, no previous path
__global__ __launch_bounds__(256) void glue_reset_kernel(int B, const int *age, long long *traj_idx, int *prev_len)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && age[b] == 0) { traj_idx[b] = 0; prev_len[b] = -1; }
}

// The whole scenario loop (main/scenarios/mpc_intersection.py:99-163) for n_ticks ticks.  With a one-wave register kernel and
// one linearisation pass it is THREE launches: the scripted obstacles rolled forward n_ticks ticks (they do not depend on the
// egos), their predictions for every tick, and the fused K-tick kernel with the loop glue inside each ego's tick loop.
// Otherwise the same ticks as the separate calls a host loop would make.
extern "C" int jsim_loop_run_scenario(jsim_ctx *ctx, int32_t B, int32_t n_ticks, double *x0, const int32_t *path_id,
                                      int32_t *path_len, const double *speed, int64_t *target_ind, double *oa, double *od,
                                      double *ox, double *oy, double *ov, double *oyaw, double *xref, uint32_t *active_mask,
                                      int32_t *status, int32_t *n_iter, double *di_ai, const double *x0_spawn,
                                      const int64_t *target_spawn, int32_t *age, int32_t max_age, double *hist, int32_t *tick,
                                      int32_t hist_cap, uint64_t *n_respawn, int64_t *traj_idx, int32_t *prev_path_len,
                                      int32_t *col_flag, int32_t *pre_status, int32_t frame_window, int32_t margin,
                                      int32_t n_obs, double *obs_state, const double *obs_param, double *obs_get,
                                      int32_t n_steps, int32_t speed_cutoff, void *stream)
{
    if (!ctx) return fail(nullptr, -22, "jsim_loop_run_scenario: null ctx");
    DeviceGuard dev_guard(ctx->device);
    JSIM_GUARD_OK(ctx);
    if (B < 0 || n_ticks < 0 || frame_window < 0 || frame_window > 32 || margin < 0)
        return fail(ctx, -22, "jsim_loop_run_scenario: bad argument");
    if (B == 0 || n_ticks == 0) return 0;
    if (!x0 || !path_id || !path_len || !speed || !target_ind || !oa || !od || !status || !di_ai || !x0_spawn || !target_spawn ||
        !age || !traj_idx || !prev_path_len || !col_flag || !pre_status)
        return fail(ctx, -22, "jsim_loop_run_scenario: a required device pointer is null");
    if (n_obs < 0 || n_obs > JSIM_MAX_OBS || n_steps < 1 || n_steps > JSIM_MAX_PRED)
        return fail(ctx, -22, "jsim_loop_run_scenario: n_obs=%d (max %d), n_steps=%d (max %d)", n_obs, JSIM_MAX_OBS, n_steps, JSIM_MAX_PRED);
    if (n_obs > 0 && (!obs_state || !obs_param || !obs_get)) return fail(ctx, -22, "jsim_loop_run_scenario: null obstacle pointer");
    if (hist && !tick) return fail(ctx, -22, "jsim_loop_run_scenario: hist needs a device tick counter");
    if (!ctx->d_pxy || !ctx->d_pcc || !ctx->have_geom) return fail(ctx, -22, "jsim_loop_run_scenario: paths / geometry not set");
    const jsim_cfg &c = ctx->cfg;
    hipStream_t s = (hipStream_t)stream;
    if (speed_cutoff && !ctx->cv_cut)
        return fail(ctx, -22, "jsim_loop_run_scenario: the speed-cut-off glue needs jsim_mpc_set_speed_cutoff first");
    int32_t *const glue_out = speed_cutoff ? const_cast<int32_t *>(ctx->cv_cut) : path_len; // where the cut-off index goes
    if (!ctx->use_reg_kernel || !has_fused_glue(c.T) || c.max_iter > 1 || (ctx->cv_cut && !speed_cutoff)) {
        // tick by tick, as ScenarioLoop.tick does
        for (int k = 0; k < n_ticks; ++k) {
            int rc = jsim_loop_obstacles(ctx, n_obs, obs_state, obs_param, obs_get, 0, stream);
            if (rc) return rc;
            ctx->pred_n_obs = n_obs; ctx->pred_n_steps = n_steps;
            if (n_obs > 0) {
                ObsP OP = {n_obs, n_steps, c.dt, ctx->oL, ctx->occ0, ctx->occ1, obs_get, nullptr, ctx->d_pred_cc, ctx->d_pred_bc};
                hipLaunchKernelGGL(obstacle_predict_kernel, dim3(1), dim3(64), 0, s, OP);
            }
            rc = jsim_loop_pre_tick(ctx, B, x0, path_id, traj_idx, prev_path_len, glue_out, col_flag, nullptr, nullptr, pre_status,
                                    frame_window, margin, nullptr, nullptr, stream);
            if (rc) return rc;
            // the previous tmp_trajectory: the truncated path, or always the full one (mpc_intersection_new_ref.py:131)
            HIP_TRY(ctx, hipMemcpyAsync(prev_path_len, path_len, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, s));
            rc = launch_step(ctx, B, x0, path_id, path_len, speed, target_ind, oa, od, ox, oy, ov, oyaw, xref, active_mask, status,
                             n_iter, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
            if (rc) return rc;
            rc = jsim_loop_advance(ctx, B, x0, oa, od, status, di_ai, target_ind, path_id, path_len, x0_spawn, target_spawn, age,
                                   max_age, hist, tick, hist_cap, n_respawn, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(glue_reset_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, age, (long long *)traj_idx, prev_path_len);
            rc = jsim_loop_obstacles(ctx, n_obs, obs_state, obs_param, obs_get, 1, stream);
            if (rc) return rc;
        }
        return 0;
    }
    // obstacles: n_ticks ticks of get() tuples, then every tick's prediction
    const size_t need_get = (size_t)n_ticks * (n_obs > 0 ? n_obs : 1) * 6;
    const size_t need_pred = (size_t)n_ticks * (n_obs > 0 ? n_obs : 1) * n_steps * 2;
    if (need_get > ctx->get_all_cap) {
        if (ctx->d_get_all) (void)hipFree(ctx->d_get_all);
        ctx->d_get_all = nullptr; ctx->get_all_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_get_all, sizeof(double) * need_get));
        ctx->get_all_cap = need_get;
    }
    if (need_pred > ctx->pred_all_cap) {
        if (ctx->d_pred_all) (void)hipFree(ctx->d_pred_all);
        ctx->d_pred_all = nullptr; ctx->pred_all_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_pred_all, sizeof(double2) * need_pred));
        ctx->pred_all_cap = need_pred;
        if (ctx->d_bc_all) (void)hipFree(ctx->d_bc_all);
        ctx->d_bc_all = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->d_bc_all, sizeof(double4) * (need_pred / ((size_t)n_steps * 2))));
    }
    if (n_obs > 0) {
        ObsStepP SP = {n_obs, 1, ctx->have_ogeom ? ctx->oL : c.L, obs_state, obs_param, nullptr};
        hipLaunchKernelGGL(obstacle_rollout_kernel, dim3(1), dim3(64), 0, s, SP, n_ticks, ctx->d_get_all);
        ObsP OP = {n_obs, n_steps, c.dt, ctx->oL, ctx->occ0, ctx->occ1, ctx->d_get_all, nullptr, ctx->d_pred_all, ctx->d_bc_all};
        hipLaunchKernelGGL(obstacle_predict_kernel, dim3(n_ticks), dim3(64), 0, s, OP);
        // the last get() tuples, as after n_ticks host ticks
        HIP_TRY(ctx, hipMemcpyAsync(obs_get, ctx->d_get_all + (size_t)(n_ticks - 1) * n_obs * 6, sizeof(double) * n_obs * 6,
                                    hipMemcpyDeviceToDevice, s));
    }
    ctx->pred_n_obs = n_obs; ctx->pred_n_steps = n_steps;
    KP P;
    fill_kp(ctx, B, P);
    P.x0 = x0; P.path_id = path_id; P.path_len = path_len; P.speed = speed;
    P.target_ind = (long long *)target_ind; P.oa = oa; P.od = od; P.ox = ox; P.oy = oy; P.ov = ov; P.oyaw = oyaw;
    P.xref = xref; P.amask = active_mask; P.status = status; P.n_iter = n_iter;
    P.dbg_clk = nullptr; P.dbg_max_gi = ctx->dbg_max_gi;
    TickP K;
    memset(&K, 0, sizeof(K));
    K.n_ticks = n_ticks; K.advance = 1; K.max_age = max_age > 0 ? max_age : 0x7fffffff; K.hist_cap = hist_cap;
    K.max_decel = c.max_decel; K.goal_dis = c.goal_dis; K.stop_speed = c.stop_speed;
    K.x0w = x0; K.di_ai = di_ai; K.x0_spawn = x0_spawn; K.target_spawn = (const long long *)target_spawn; K.age = age;
    K.hist = hist; K.tick = tick; K.n_respawn = (unsigned long long *)n_respawn;
    PreK Q;
    memset(&Q, 0, sizeof(Q));
    Q.pre.B = B; Q.pre.n_obs = n_obs; Q.pre.n_steps = n_steps; Q.pre.frame_window = frame_window; Q.pre.margin = margin;
    Q.pre.dt = c.dt; Q.pre.max_accel = c.max_accel; Q.pre.max_speed = c.max_speed; Q.pre.thr = ctx->col_radius + ctx->ocol_radius;
    Q.pre.thr_sq = jsim_sqrt_threshold(Q.pre.thr);
    Q.pre.pxy = ctx->d_pxy; Q.pre.pcc = ctx->d_pcc; Q.pre.poff = ctx->d_poff;
    Q.pred_cc_all = ctx->d_pred_all; Q.pred_bc_all = ctx->d_bc_all; Q.traj_idx = (long long *)traj_idx; Q.prev_len = prev_path_len; Q.path_len_out = path_len;
    Q.col_flag = col_flag; Q.pre_status = pre_status;
    Q.speed_cutoff = speed_cutoff ? 1 : 0; Q.cut_io = glue_out;
    if (int rc_ = prepare_launch_order(ctx, B, s, K)) return rc_;
    if (int rc_ = prepare_iter_totals(ctx, B, s, K)) return rc_;
    launch_reg(c.T, B, s, P, K, &Q);
    if (tick) hipLaunchKernelGGL(tick_add_kernel, dim3(1), dim3(1), 0, s, tick, n_ticks);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
#endif /* !JSIM_KERNEL_TU */

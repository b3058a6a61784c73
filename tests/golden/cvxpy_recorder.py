"""A RECORDING stand-in for the eight cvxpy names main/lib/mpc.py touches.  TEST INFRASTRUCTURE, build container only.

Why it exists.  The reference assembles its QP in `_linear_mpc_control` (main/lib/mpc.py:141-211) with cvxpy and solves it with
ECOS.  Neither package is installed, pinned (poetry.lock has no such entries) or fetchable here, so until round 3 that function
had never executed in any form and stage S4 of the oracle was compared only with restatements written by the builder.  This
module is registered as `cvxpy` in `sys.modules` by tests/golden/make_golden_refqp.py (and by tests/test_ref_qp_live.py) so
that the reference's OWN code runs, unmodified, and writes down the problem it builds: which terms, over which `t`, with which
`reaches_end` switches, in which constraint order.  It is NOT cvxpy and NOT ECOS:

  * PINNED by it: the assembly -- the sparse problem  minimise z'Pz + q'z + c0  s.t.  A z = b,  G z <= h  exactly as emitted by
    the reference (z = the entries of its two `Variable`s, column-major: x[:,0], ..., x[:,T], u[:,0], ..., u[:,T-1]), rows in
    the order of the reference's `constraints` list, and the reference's own S5 lines (:199-211, :298-303) run on the optimum.
  * STILL UNPINNED: the numbers ECOS itself would return (its stopping tolerance: feastol = abstol = reltol = 1e-8 by cvxpy's
    defaults, `OPTIMAL_INACCURATE` accepted).  `Problem.solve` here returns the exact optimum of the emitted problem, which is
    unique (the cost is strictly convex in u and x is an affine function of u), computed by a solver that shares nothing with
    oracle/ or csrc/: generic null-space elimination of the equalities, a Mehrotra primal-dual interior-point method (the
    algorithm class of ECOS) on the reduced problem, then a KKT re-solve on the identified active rows with iterative refinement.

Conventions this stand-in fixes (cvxpy leaves them to its canonicaliser, so they are choices, stated here once):
  * `abs(e) <= c` with e of length k becomes the 2k rows (+e_0 <= c, -e_0 <= c, +e_1 <= c, -e_1 <= c, ...);
  * `e >= c` becomes `-e <= -c`; a vector constraint contributes its rows in element order;
  * infeasibility is declared when some point cannot satisfy A z = b, G z <= h + 1e-8 (ECOS's feastol), status "infeasible".

Supported surface (anything else raises, so a change in how the reference uses cvxpy cannot go unnoticed):
  Variable((rows, cols)) with 2-D basic indexing; affine arithmetic  ndarray -/+ expr, expr -/+ expr, ndarray @ expr, -expr;
  expr == rhs, expr <= scalar, expr >= scalar; abs(expr) <= scalar; quad_form(expr, constant matrix); sums of quad_forms
  starting from the float 0.0; Minimize; Problem(objective, constraints).solve(solver=ECOS, verbose=False), .status; Variable.value.
"""
from __future__ import annotations

import numpy as np

ECOS = "ECOS"
OPTIMAL = "optimal"
OPTIMAL_INACCURATE = "optimal_inaccurate"
INFEASIBLE = "infeasible"

RECORDS: list = []        # one dict per Problem.solve() call, appended in call order (the generator drains it)
FEASTOL = 1e-8


class Variable:
    _next_id = 0

    def __init__(self, shape):
        if not (isinstance(shape, tuple) and len(shape) == 2):
            raise NotImplementedError("stand-in: only 2-D Variables")
        self.shape = tuple(int(s) for s in shape)
        self.size = self.shape[0] * self.shape[1]
        self.id = Variable._next_id
        Variable._next_id += 1
        self.value = None

    def __getitem__(self, key):
        if not (isinstance(key, tuple) and len(key) == 2):
            raise NotImplementedError("stand-in: Variable indexing needs two subscripts")
        r, c = self.shape
        flat = np.arange(self.size).reshape((c, r)).T          # flat[i, j] = column-major position of entry (i, j)
        sel = np.atleast_1d(flat[key])
        if sel.ndim != 1:
            raise NotImplementedError("stand-in: only scalar / vector slices of a Variable")
        coef = np.zeros((sel.shape[0], self.size))
        coef[np.arange(sel.shape[0]), sel] = 1.0
        return Affine({self: coef}, np.zeros(sel.shape[0]))


class Affine:
    """k affine functions of the Variables: sum_v coef[v] @ vec(v) + const."""
    __array_ufunc__ = None          # numpy defers  ndarray - expr,  ndarray @ expr  to the reflected methods below

    def __init__(self, coefs, const):
        self.coefs = coefs
        self.const = np.asarray(const, dtype=np.float64)

    @property
    def k(self):
        return self.const.shape[0]

    @staticmethod
    def _lift(other, k):
        if isinstance(other, Affine):
            return other
        a = np.asarray(other, dtype=np.float64)
        if a.ndim == 0:
            a = np.full(k, float(a))
        if a.shape != (k,):
            raise NotImplementedError(f"stand-in: constant of shape {a.shape} against an expression of length {k}")
        return Affine({}, a)

    def __neg__(self):
        return Affine({v: -c for v, c in self.coefs.items()}, -self.const)

    def __add__(self, other):
        o = self._lift(other, self.k)
        if o.k != self.k:
            raise NotImplementedError("stand-in: length mismatch in +")
        coefs = {v: c.copy() for v, c in self.coefs.items()}
        for v, c in o.coefs.items():
            coefs[v] = coefs[v] + c if v in coefs else c.copy()
        return Affine(coefs, self.const + o.const)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-self._lift(other, self.k))

    def __rsub__(self, other):
        return (-self) + self._lift(other, self.k)

    def __rmatmul__(self, M):
        M = np.asarray(M, dtype=np.float64)
        if M.ndim != 2 or M.shape[1] != self.k:
            raise NotImplementedError("stand-in: matrix @ expression shape")
        return Affine({v: M @ c for v, c in self.coefs.items()}, M @ self.const)

    def __eq__(self, other):                       # expression == rhs   ->   (self - rhs) == 0
        return Constraint("eq", self - other)

    def __le__(self, other):                       # expression <= rhs   ->   (self - rhs) <= 0
        return Constraint("le", self - other)

    def __ge__(self, other):                       # expression >= rhs   ->   (rhs - self) <= 0
        return Constraint("le", -(self - other))

    __hash__ = None


class Abs:
    def __init__(self, expr):
        self.expr = expr

    def __le__(self, bound):
        e = self.expr
        bound = np.asarray(bound, dtype=np.float64)
        if bound.ndim != 0:
            raise NotImplementedError("stand-in: abs(e) <= scalar only")
        k = e.k
        perm = np.arange(2 * k).reshape(2, k).T.ravel()          # (+e_0, -e_0, +e_1, -e_1, ...)
        both = Affine({v: np.concatenate([c, -c])[perm] for v, c in e.coefs.items()},
                      np.concatenate([e.const, -e.const])[perm] - float(bound))
        return Constraint("le", both, what="abs")


def abs(expr):                                     # noqa: A001 (the reference calls cvxpy.abs)
    if not isinstance(expr, Affine):
        raise NotImplementedError("stand-in: abs of a non-expression")
    return Abs(expr)


class Constraint:
    def __init__(self, kind, expr, what="plain"):
        self.kind, self.expr, self.what = kind, expr, what


class Quad:
    """sum_i e_i' M_i e_i  (+ a float): the only cost the reference builds."""

    def __init__(self, terms, const=0.0):
        self.terms, self.const = terms, float(const)

    def __add__(self, other):
        if isinstance(other, Quad):
            return Quad(self.terms + other.terms, self.const + other.const)
        if isinstance(other, (int, float)):
            return Quad(self.terms, self.const + other)
        return NotImplemented

    __radd__ = __add__


def quad_form(expr, M):
    M = np.asarray(M, dtype=np.float64)
    if not isinstance(expr, Affine) or M.shape != (expr.k, expr.k):
        raise NotImplementedError("stand-in: quad_form(expression, constant square matrix)")
    if not np.array_equal(M, M.T):
        raise NotImplementedError("stand-in: quad_form needs a symmetric matrix (cvxpy would raise too)")
    return Quad([(expr, M)])


class Minimize:
    def __init__(self, cost):
        if not isinstance(cost, Quad):
            raise NotImplementedError("stand-in: Minimize(sum of quad_forms)")
        self.cost = cost


class Problem:
    def __init__(self, objective, constraints):
        self.objective, self.constraints = objective, list(constraints)
        self.status = None

    # ---- assembly: what the reference emitted, nothing reordered ---------------------------------------------------------
    def _assemble(self):
        vs = {}
        for e, _ in self.objective.cost.terms:
            for v in e.coefs:
                vs[v.id] = v
        for c in self.constraints:
            for v in c.expr.coefs:
                vs[v.id] = v
        variables = [vs[i] for i in sorted(vs)]                  # creation order: x then u in the reference
        off, n = {}, 0
        for v in variables:
            off[v] = n
            n += v.size

        def dense(e):
            Mz = np.zeros((e.k, n))
            for v, c in e.coefs.items():
                Mz[:, off[v]:off[v] + v.size] += c
            return Mz

        P = np.zeros((n, n)); q = np.zeros(n); c0 = self.objective.cost.const
        for e, M in self.objective.cost.terms:                   # (Mz z + c)' M (Mz z + c)
            Mz = dense(e)
            P += Mz.T @ M @ Mz
            q += 2.0 * (Mz.T @ (M @ e.const))
            c0 += float(e.const @ M @ e.const)
        A, b, G, h, eq_src, in_src = [], [], [], [], [], []
        for ci, c in enumerate(self.constraints):
            Mz = dense(c.expr)
            if c.kind == "eq":
                A.append(Mz); b.append(-c.expr.const); eq_src += [ci] * c.expr.k
            else:
                G.append(Mz); h.append(-c.expr.const); in_src += [ci] * c.expr.k
        return dict(variables=variables, offsets=[off[v] for v in variables], n=n, P=P, q=q, c0=c0,
                    A=np.vstack(A) if A else np.zeros((0, n)), b=np.concatenate(b) if b else np.zeros(0),
                    G=np.vstack(G) if G else np.zeros((0, n)), h=np.concatenate(h) if h else np.zeros(0),
                    eq_src=np.array(eq_src, dtype=np.int32), in_src=np.array(in_src, dtype=np.int32))

    def solve(self, solver=None, verbose=False):
        if solver != ECOS:
            raise NotImplementedError("stand-in: the reference asks for ECOS")
        rec = self._assemble()
        sol = solve_qp(rec["P"], rec["q"], rec["A"], rec["b"], rec["G"], rec["h"])
        self.status = sol["status"]
        rec.update(status=sol["status"], z=sol["z"], lam=sol["lam"], ipm_iters=sol["ipm_iters"], kkt=sol["kkt"])
        if sol["status"] == OPTIMAL:
            for v, o in zip(rec["variables"], rec["offsets"]):
                r, c = v.shape
                v.value = sol["z"][o:o + v.size].reshape((c, r)).T.copy()
        rec.pop("variables")
        RECORDS.append(rec)
        return None if sol["status"] != OPTIMAL else float(sol["z"] @ rec["P"] @ sol["z"] + rec["q"] @ sol["z"] + rec["c0"])


# ---- the stand-in's solver: exact optimum of the emitted problem ----------------------------------------------------------
def _null_space(A, b):
    """Particular solution and an orthonormal null-space basis of A z = b (generic, by SVD)."""
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    r = int((s > 1e-12 * s[0]).sum()) if s.size else 0
    if r != A.shape[0]:
        raise RuntimeError("stand-in: dependent equality rows")
    zp = Vt[:r].T @ ((U.T @ b)[:r] / s[:r])
    return zp, Vt[r:].T


def _ipm(Q, c, G, h, max_iter=80):
    """Mehrotra predictor-corrector for  min 1/2 w'Qw + c'w  s.t.  G w <= h,  Q > 0."""
    n, m = Q.shape[0], G.shape[0]
    w = np.linalg.solve(Q, -c)
    s = np.maximum(h - G @ w, 1.0)
    lam = np.ones(m)
    for it in range(max_iter):
        rd = Q @ w + c + G.T @ lam
        rp = G @ w + s - h
        mu = float(s @ lam) / m
        if max(np.abs(rd).max(), np.abs(rp).max()) <= 1e-10 * (1 + np.abs(c).max()) and mu <= 1e-11:
            break
        D = lam / s
        K = Q + G.T @ (D[:, None] * G)
        try:
            Lc = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:                # s/lam spread beyond what the normal equations carry: the polish takes over
            break

        def newton(rc):                              # rc: the complementarity residual  s*lam - target
            rhs = -rd - G.T @ (D * rp - rc / s)
            dw = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
            ds = -rp - G @ dw
            dl = -(rc + lam * ds) / s
            return dw, ds, dl

        def step_len(v, dv):
            neg = dv < 0
            return min(1.0, float((-v[neg] / dv[neg]).min())) if neg.any() else 1.0

        dw, ds, dl = newton(s * lam)
        ap, ad = step_len(s, ds), step_len(lam, dl)
        mu_aff = float((s + ap * ds) @ (lam + ad * dl)) / m
        sigma = (mu_aff / mu) ** 3
        dw, ds, dl = newton(s * lam + ds * dl - sigma * mu)
        ap, ad = 0.995 * step_len(s, ds), 0.995 * step_len(lam, dl)
        a = min(ap, ad)
        w, s, lam = w + a * dw, s + a * ds, lam + a * dl
    return w, s, lam, it


def solve_qp(P, q, A, b, G, h):
    """Exact optimum of  min z'Pz + q'z  s.t.  A z = b,  G z <= h  (cost WITHOUT 1/2, as cvxpy's quad_form sums)."""
    zp, Z = _null_space(A, b)
    Gw, hw = G @ Z, h - G @ zp
    const = np.abs(Gw).max(axis=1, initial=0.0) <= 1e-13 * np.maximum(1.0, np.abs(G).max(axis=1))    # rows the equalities make constant
    if (hw[const] < -FEASTOL).any():
        return dict(status=INFEASIBLE, z=None, lam=None, ipm_iters=0, kkt=None)
    keep = np.flatnonzero(~const)
    if Z.shape[1] == 0:                              # the equalities fix every variable
        return dict(status=OPTIMAL, z=zp, lam=np.zeros(G.shape[0]), ipm_iters=0,
                    kkt=dict(stationarity=0.0, primal_eq=float(np.abs(A @ zp - b).max()), primal_in=0.0, dual_min=0.0, du_ipm=0.0))
    from scipy.optimize import linprog
    lp = linprog(np.zeros(Z.shape[1]), A_ub=Gw[keep], b_ub=hw[keep] + FEASTOL, bounds=(None, None), method="highs")
    if lp.status == 2:
        return dict(status=INFEASIBLE, z=None, lam=None, ipm_iters=0, kkt=None)
    Q = 2.0 * (Z.T @ P @ Z)
    Q = 0.5 * (Q + Q.T)
    c = Z.T @ (2.0 * (P @ zp) + q)
    Gk, hk = Gw[keep], hw[keep]
    w, s, lam_k, iters = _ipm(Q, c, Gk, hk)
    # polish: a KKT re-solve on the rows the interior-point iterate says are active, repaired one row at a time like a primal
    # active-set method; the multipliers come from a NON-NEGATIVE least-squares fit over every tight row, so that dependent
    # tight rows (non-unique multipliers) are certified like any other case
    from scipy.optimize import nnls
    nq = Q.shape[0]
    sc = max(1.0, np.abs(c).max())

    def eq_solve(rows):
        Ga = Gk[rows]
        K = np.block([[Q, Ga.T], [Ga, np.zeros((len(rows), len(rows)))]])
        rhs = np.concatenate([-c, hk[rows]])
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        for _r in range(3):                          # iterative refinement, residual in extended precision
            res = (rhs.astype(np.longdouble) - K.astype(np.longdouble) @ sol.astype(np.longdouble)).astype(np.float64)
            sol = sol + np.linalg.lstsq(K, res, rcond=None)[0]
        return sol[:nq], sol[nq:]

    act = np.flatnonzero(lam_k > s)
    for _ in range(400):
        wp, la = eq_solve(act)
        viol = Gk @ wp - hk
        out = np.setdiff1d(np.arange(len(hk)), act)
        if len(out) and viol[out].max() > 1e-9:
            act = np.union1d(act, [out[np.argmax(viol[out])]])
            continue
        tight = np.flatnonzero(viol >= -1e-9 * np.maximum(1.0, np.abs(hk)))
        lt, rnorm = nnls(Gk[tight].T, -(Q @ wp + c)) if len(tight) else (np.zeros(0), float(np.abs(Q @ wp + c).max()))
        if rnorm <= 1e-9 * sc:
            support = tight[lt > 0]
            wp2, la = eq_solve(support)               # NNLS supports are linearly independent: a unique, refined multiplier vector
            if np.abs(wp2 - wp).max() <= 1e-9 and (la > -1e-12 * sc).all():
                wp, act = wp2, support
                break
            la, act = lt[lt > 0], support             # (a support the re-solve moves away from: keep the certified NNLS pair)
            break
        if not len(act) or la.min() >= 0:
            raise RuntimeError("stand-in: polish has no row to drop but no multiplier certificate either")
        act = np.delete(act, np.argmin(la))
    else:
        raise RuntimeError("stand-in: polish did not settle on an active set")
    lam = np.zeros(G.shape[0])
    lam[keep[act]] = la
    z = zp + Z @ wp
    stat = 2.0 * (P @ z) + q + G.T @ lam
    # the equality multipliers absorb the range of A': stationarity is checked in the null space
    kkt = dict(stationarity=float(np.abs(Z.T @ stat).max()), primal_eq=float(np.abs(A @ z - b).max()),
               primal_in=float(max((G[keep] @ z - h[keep]).max(), 0.0)), dual_min=float(la.min()) if len(la) else 0.0,
               du_ipm=float(np.abs(Z @ (w - wp)).max()))
    return dict(status=OPTIMAL, z=z, lam=lam, ipm_iters=iters, kkt=kkt)

"""Independent numpy restatement of the QP the reference hands to cvxpy (main/lib/mpc.py:141-194),
in its ORIGINAL sparse form: variables x in R^{4x(T+1)}, u in R^{2xT}, dynamics as equalities.

Used only by tests to validate the oracle's condensed (H, g, G, h): the states are eliminated here
by generic dense linear algebra (no structural shortcuts), so a transcription error on either side
shows up as a mismatch.  Test infrastructure, never imported by the product.
"""
import numpy as np


def linear_model(v, phi, delta, dt, L):
    """main/lib/mpc.py:61-82"""
    A = np.eye(4)
    A[0, 2] = dt * np.cos(phi)
    A[0, 3] = -dt * v * np.sin(phi)
    A[1, 2] = dt * np.sin(phi)
    A[1, 3] = dt * v * np.cos(phi)
    A[3, 2] = dt * np.tan(delta) / L
    B = np.zeros((4, 2))
    B[2, 0] = dt
    B[3, 1] = dt * v / (L * np.cos(delta) ** 2)
    C = np.zeros(4)
    C[0] = dt * v * np.sin(phi) * phi
    C[1] = -dt * v * np.cos(phi) * phi
    C[3] = -dt * v * delta / (L * np.cos(delta) ** 2)
    return A, B, C


def proj(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c * c, c * s], [c * s, s * s]])


def build_sparse(cfg, T, dt, L, xref, xbar, x0, reaches_end, speed):
    """Returns (P, q, c0, Aeq, beq, Gin, hin) with z = [x(:,0), ..., x(:,T), u(:,0), ..., u(:,T-1)],
    cost = z'Pz + q'z + c0 (cvxpy's cost carries no 1/2), Aeq z = beq, Gin z <= hin.
    Row order of Gin follows the reference's constraint list (mpc.py:187-194), abs(e)<=b expanded to
    (+e<=b, -e<=b)."""
    nx, nu = 4 * (T + 1), 2 * T
    nz = nx + nu
    xi = lambda t, r: 4 * t + r
    ui = lambda t, c: nx + 2 * t + c
    P = np.zeros((nz, nz)); q = np.zeros(nz); c0 = 0.0
    R = np.diag(cfg["R"]); Rd = np.diag(cfg["Rd"]); Qvy = np.diag(cfg["Q_v_yaw"])
    Qf = np.diag(cfg["Qf"]) * T
    Aeq = []; beq = []
    for t in range(T + 1):
        if t > 0:
            Q = np.zeros((4, 4))
            if not reaches_end[t]:
                Q[:2, :2] = proj(xref[3, t] + 0.5 * np.pi) * cfg["w_perp"] + proj(xref[3, t]) * cfg["w_para"]
                Q[2:, 2:] = Qvy
            else:
                Q = Qf
            # (xref - x)' Q (xref - x)
            sl = slice(xi(t, 0), xi(t, 0) + 4)
            P[sl, sl] += Q
            q[sl] += -2.0 * Q @ xref[:, t]
            c0 += xref[:, t] @ Q @ xref[:, t]
        if t < T:
            A, B, Cv = linear_model(xbar[2, t], xbar[3, t], 0.0, dt, L)
            for r in range(4):
                row = np.zeros(nz)
                row[xi(t + 1, r)] = 1.0
                row[xi(t, 0):xi(t, 0) + 4] -= A[r]
                row[ui(t, 0):ui(t, 0) + 2] -= B[r]
                Aeq.append(row); beq.append(Cv[r])
            Ru = np.diag([10.0, 10.0]) if reaches_end[t] else R
            sl = slice(ui(t, 0), ui(t, 0) + 2)
            P[sl, sl] += Ru
        if t < T - 1:
            D = np.zeros((2, nz))
            D[0, ui(t + 1, 0)] = 1; D[0, ui(t, 0)] = -1
            D[1, ui(t + 1, 1)] = 1; D[1, ui(t, 1)] = -1
            P += D.T @ Rd @ D
    for r in range(4):
        row = np.zeros(nz); row[xi(0, r)] = 1.0
        Aeq.append(row); beq.append(x0[r])
    G = []; h = []
    dmax = np.deg2rad(cfg["MAX_DSTEER"]) * dt
    for t in range(T - 1):
        row = np.zeros(nz); row[ui(t + 1, 1)] = 1; row[ui(t, 1)] = -1
        G.append(row); h.append(dmax); G.append(-row); h.append(dmax)
    for t in range(T + 1):
        row = np.zeros(nz); row[xi(t, 2)] = 1; G.append(row); h.append(speed)
    for t in range(T + 1):
        row = np.zeros(nz); row[xi(t, 2)] = -1; G.append(row); h.append(5.0)  # x[2,:] >= MIN_SPEED=-5
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 0)] = 1; G.append(row); h.append(cfg["MAX_ACCEL"])
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 0)] = -1; G.append(row); h.append(-cfg["MAX_DECEL"])
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 1)] = 1
        G.append(row); h.append(np.deg2rad(45.0)); G.append(-row); h.append(np.deg2rad(45.0))
    return P, q, c0, np.array(Aeq), np.array(beq), np.array(G), np.array(h)


def condense(P, q, Aeq, beq, Gin, hin, T):
    """Eliminate x with the equalities: x = Phi u + phi.  Returns H, g (for 1/2 u'Hu + g'u), G, h."""
    nx = 4 * (T + 1)
    Ax, Au = Aeq[:, :nx], Aeq[:, nx:]
    Phi = -np.linalg.solve(Ax, Au)
    phi = np.linalg.solve(Ax, beq)
    Z = np.vstack([Phi, np.eye(2 * T)])     # z = Z u + z0
    z0 = np.concatenate([phi, np.zeros(2 * T)])
    H = 2.0 * Z.T @ P @ Z
    g = Z.T @ (2.0 * P @ z0 + q)
    G = Gin @ Z
    h = hin - Gin @ z0
    return 0.5 * (H + H.T), g, G, h, Phi, phi


def kkt_check(H, g, G, h, skip, u, lam):
    """Residuals of the KKT conditions of  min 1/2 u'Hu + g'u  s.t. Gu <= h."""
    stat = np.abs(H @ u + g + G.T @ lam).max()
    keep = ~skip.astype(bool)
    prim = np.maximum(G[keep] @ u - h[keep], 0.0).max()
    dual = np.maximum(-lam, 0.0).max()
    comp = np.abs(lam[keep] * (G[keep] @ u - h[keep])).max()
    return stat, prim, dual, comp


# ------------------------------------------------------------------------------------------------
# The acceleration-state variant, main/lib/mpc_jerk.py:144-199, in ITS sparse form: x in R^{5x(T+1)}
# ------------------------------------------------------------------------------------------------
JERK_CONFIG = {  # module constants of main/lib/mpc_jerk.py:16-39 (+ the weights hard-coded at :167,:171)
    "NX": 5, "w_perp": 10.0, "w_para": 1.0, "R": [0.01, 0.01], "Rd": [0.3, 1.0], "Q_v_yaw": [0.0, 0.5],
    "Qf": [1.0, 1.0, 0.0, 0.5], "GOAL_DIS": 1.5, "STOP_SPEED": 0.5 / 3.6, "MAX_ITER": 1, "MAX_DSTEER": 30.0,
    "MAX_ACCEL": 2.0, "MAX_DECEL": -5, "JERK_WEIGHT": 1.0,
}


def linear_model_jerk(v, phi, delta, dt, L):
    """main/lib/mpc_jerk.py:59-83"""
    A4, B4, C4 = linear_model(v, phi, delta, dt, L)
    A = np.eye(5); A[:4, :4] = A4
    A[2, 4] = dt
    B = np.zeros((5, 2)); B[:4] = B4
    B[4, 0] = dt
    C = np.zeros(5); C[:4] = C4
    return A, B, C


def build_sparse_jerk(cfg, T, dt, L, xref4, xbar4, x0, reaches_end, max_speed):
    """z = [x(:,0..T) (5 rows each), u(:,0..T-1)]; xref4 / xbar4 are the four stock rows (the fifth rows are zero in
    the reference).  Same return convention as build_sparse."""
    NX = 5
    nx, nu = NX * (T + 1), 2 * T
    nz = nx + nu
    xi = lambda t, r: NX * t + r
    ui = lambda t, c: nx + 2 * t + c
    xref = np.vstack([xref4, np.zeros((1, T + 1))])
    P = np.zeros((nz, nz)); q = np.zeros(nz); c0 = 0.0
    R = np.diag(cfg["R"]); Rd = np.diag(cfg["Rd"]); Qvy = np.diag(cfg["Q_v_yaw"])
    Qf = np.diag(list(cfg["Qf"]) + [0.0]) * T
    Aeq = []; beq = []
    for t in range(T + 1):
        if t > 0:
            Q = np.zeros((NX, NX))
            if not reaches_end[t]:
                Q[:2, :2] = proj(xref[3, t] + 0.5 * np.pi) * cfg["w_perp"] + proj(xref[3, t]) * cfg["w_para"]
                Q[2:4, 2:4] = Qvy
            else:
                Q = Qf
            sl = slice(xi(t, 0), xi(t, 0) + NX)
            P[sl, sl] += Q
            q[sl] += -2.0 * Q @ xref[:, t]
            c0 += xref[:, t] @ Q @ xref[:, t]
        if t < T:
            A, B, Cv = linear_model_jerk(xbar4[2, t], xbar4[3, t], 0.0, dt, L)
            for r in range(NX):
                row = np.zeros(nz)
                row[xi(t + 1, r)] = 1.0
                row[xi(t, 0):xi(t, 0) + NX] -= A[r]
                row[ui(t, 0):ui(t, 0) + 2] -= B[r]
                Aeq.append(row); beq.append(Cv[r])
            Ru = np.diag([10.0, 10.0]) if reaches_end[t] else R
            sl = slice(ui(t, 0), ui(t, 0) + 2)
            P[sl, sl] += Ru
        if t < T - 1:
            D = np.zeros((2, nz))
            D[0, ui(t + 1, 0)] = 1; D[0, ui(t, 0)] = -1
            D[1, ui(t + 1, 1)] = 1; D[1, ui(t, 1)] = -1
            P += D.T @ Rd @ D
            j = np.zeros(nz); j[xi(t + 1, 4)] = 1; j[xi(t, 4)] = -1          # :190
            P += cfg["JERK_WEIGHT"] * np.outer(j, j)
    for r in range(4):                                                      # x[:4, 0] == x0 (:193): x[4, 0] stays free
        row = np.zeros(nz); row[xi(0, r)] = 1.0
        Aeq.append(row); beq.append(x0[r])
    G = []; h = []
    dmax = np.deg2rad(cfg["MAX_DSTEER"]) * dt
    for t in range(T - 1):
        row = np.zeros(nz); row[ui(t + 1, 1)] = 1; row[ui(t, 1)] = -1
        G.append(row); h.append(dmax); G.append(-row); h.append(dmax)
    for t in range(T + 1):
        row = np.zeros(nz); row[xi(t, 2)] = 1; G.append(row); h.append(max_speed)
    for t in range(T + 1):
        row = np.zeros(nz); row[xi(t, 2)] = -1; G.append(row); h.append(5.0)
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 0)] = 1; G.append(row); h.append(cfg["MAX_ACCEL"])
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 0)] = -1; G.append(row); h.append(-cfg["MAX_DECEL"])
    for t in range(T):
        row = np.zeros(nz); row[ui(t, 1)] = 1
        G.append(row); h.append(np.deg2rad(45.0)); G.append(-row); h.append(np.deg2rad(45.0))
    return P, q, c0, np.array(Aeq), np.array(beq), np.array(G), np.array(h)


def condense_jerk(P, q, Aeq, beq, Gin, hin, T):
    """Eliminate every state except the free x[4, 0]: decision vector w = [u (2T), acc_0]."""
    NX = 5
    nx = NX * (T + 1)
    free = 4                                  # index of x[4, 0] inside x
    dep = [i for i in range(nx) if i != free]
    Ad = Aeq[:, dep]
    Aw = np.hstack([Aeq[:, nx:], Aeq[:, [free]]])
    Phi_d = -np.linalg.solve(Ad, Aw)
    phi_d = np.linalg.solve(Ad, beq)
    nw = 2 * T + 1
    Z = np.zeros((nx + 2 * T, nw)); z0 = np.zeros(nx + 2 * T)
    Z[dep] = Phi_d; z0[dep] = phi_d
    Z[free, 2 * T] = 1.0
    Z[nx:, :2 * T] = np.eye(2 * T)
    H = 2.0 * Z.T @ P @ Z
    g = Z.T @ (2.0 * P @ z0 + q)
    G = Gin @ Z
    h = hin - Gin @ z0
    return 0.5 * (H + H.T), g, G, h, Z[:nx], z0[:nx]

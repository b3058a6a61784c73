"""Readers for tests/golden/ref_qp_T*.npz -- the QP of stage S4 as the REFERENCE's own `_linear_mpc_control`
(main/lib/mpc.py:141-211) emitted it under the recording cvxpy stand-in (tests/golden/cvxpy_recorder.py,
tests/golden/make_golden_refqp.py).  Test infrastructure; imports nothing from oracle/ or the product."""
import numpy as np


def emitted_problem(g, i):
    """Dense (P, q, c0, A, b, G, h) of case i:  min z'Pz + q'z + c0  s.t.  A z = b,  G z <= h,
    z = [x(:,0), ..., x(:,T), u(:,0), ..., u(:,T-1)], rows in the order of the reference's `constraints` list."""
    n = int(g["n_z"])

    def dense(k, rows):
        lo, hi = int(g[f"{k}_ptr"][i]), int(g[f"{k}_ptr"][i + 1])
        M = np.zeros((rows, n))
        M[g[f"{k}_i"][lo:hi].astype(np.int64), g[f"{k}_j"][lo:hi].astype(np.int64)] = g[f"{k}_v"][lo:hi]
        return M

    return (dense("P", n), g["q"][i], float(g["c0"][i]), dense("A", g["b"].shape[1]), g["b"][i],
            dense("G", g["h"].shape[1]), g["h"][i])


def active_bits(words, m):
    """[B, ceil(m/32)] uint32 masks -> [B, m] bool."""
    words = np.ascontiguousarray(words).view(np.uint32)
    return ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(words.shape[0], -1)[:, :m].astype(bool)


def canonical_row_families(T):
    """(name, first row, count) in the canonical order of include/jsim_mpc.h == the order the reference's list emits."""
    return (("D", 0, 2 * T - 2), ("VU", 2 * T - 2, T + 1), ("VL", 3 * T - 1, T + 1), ("AU", 4 * T, T), ("AL", 5 * T, T),
            ("S", 6 * T, 2 * T))

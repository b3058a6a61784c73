"""Shared pieces of the GPU parity tests (tests/test_gpu_*.py)."""
import numpy as np
import torch


def engine(pkg, routes, batch, T, **kw):
    eng = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, device="cuda:0", smooth=False, **kw)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    return eng


def debug_bufs(eng):
    B, T = eng.B, eng.T
    f = dict(dtype=torch.float64, device=eng.device)
    return {"xbar": torch.zeros(B, 4, T + 1, **f), "ref_idx": torch.zeros(B, T + 1, dtype=torch.int64, device=eng.device),
            "H": torch.zeros(B, 2 * T, 2 * T, **f), "g": torch.zeros(B, 2 * T, **f), "lam": torch.zeros(B, 8 * T, **f)}


def oracle_batch(oracle, pkg, routes, batch, T, n_threads=1, **kw):
    p = oracle.make_params(T=T, **kw)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    return p, oracle.mpc_step_batch(p, batch.x0, batch.path_id, batch.path_len, batch.speed, cx, cy, cyaw, off,
                                    batch.target_ind, batch.oa, batch.od, n_threads=n_threads)


def kkt_check(eng, batch, dbg):
    """Size-independent property: the u* the kernel returned satisfies the KKT conditions of the condensed QP the kernel built
    (strictly convex => that IS the optimum), multipliers >= 0, complementary, active bits <=> positive multipliers."""
    B, T = eng.B, eng.T
    st = eng.status
    ok = st == 0
    H = dbg["H"]; H = torch.tril(H) + torch.tril(H, -1).transpose(1, 2)
    u = torch.stack([eng.oa, eng.od], dim=2).reshape(B, 2 * T)
    lam = dbg["lam"]
    n, m = 2 * T, 8 * T
    # G rows from the structure (canonical order), built once
    G = torch.zeros(m, n, dtype=torch.float64, device=eng.device)
    for t in range(T - 1):
        G[2 * t, 2 * t + 3] = 1; G[2 * t, 2 * t + 1] = -1; G[2 * t + 1] = -G[2 * t]
    for t in range(T + 1):
        G[2 * T - 2 + t, 0:2 * t:2] = eng.dt
        G[3 * T - 1 + t] = -G[2 * T - 2 + t]
    for t in range(T):
        G[4 * T + t, 2 * t] = 1; G[5 * T + t, 2 * t] = -1
        G[6 * T + 2 * t, 2 * t + 1] = 1; G[6 * T + 2 * t + 1, 2 * t + 1] = -1
    c = eng.config
    h = torch.zeros(B, m, dtype=torch.float64, device=eng.device)
    x0 = torch.from_numpy(batch.x0).to(eng.device)
    h[:, :2 * T - 2] = c.max_dsteer_rad * eng.dt
    h[:, 2 * T - 2:3 * T - 1] = (eng.speed - x0[:, 2])[:, None]
    h[:, 3 * T - 1:4 * T] = (x0[:, 2] - c.MIN_SPEED)[:, None]
    h[:, 4 * T:5 * T] = c.MAX_ACCEL
    h[:, 5 * T:6 * T] = -c.MAX_DECEL
    h[:, 6 * T:] = c.MAX_STEER_RAD
    stat = torch.einsum("bij,bj->bi", H, u) + dbg["g"] + lam @ G
    scale = dbg["g"].abs().amax(dim=1).clamp(min=1.0)
    assert float((stat.abs().amax(dim=1) / scale)[ok].max()) <= 1e-8
    slack = h - u @ G.T
    assert float((-slack)[ok].max()) <= 1e-8                       # primal feasible
    assert float(lam[ok].min()) >= 0.0                             # dual feasible
    assert float((lam * slack).abs()[ok].max() / float(scale.max())) <= 1e-8   # complementary
    # active bits <=> positive multipliers
    words = eng.active_mask.cpu().numpy().view(np.uint32)
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, -1)[:, :m].astype(bool)
    thr = (1e-9 * scale).cpu().numpy()[:, None]
    assert np.array_equal(bits[ok.cpu().numpy()], (lam.cpu().numpy() > thr)[ok.cpu().numpy()])

import sys, importlib, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import test_gpu_parity as tg
import oracle_py as oracle
pkg = importlib.import_module("av-simulation-at-intersections_amd")
T = int(sys.argv[1]) if len(sys.argv)>1 else 30
S=pkg.synth
routes = S.make_route_table()
for r in routes: S.smooth_yaw_inplace(r[:,2])
B=96
batch = S.make_ego_batch(routes, B, T, seed=0, truncate=True, near_end_frac=0.2)
eng = tg._engine(pkg, routes, batch, T)
dbg = tg._debug_bufs(eng)
eng.solve(torch.from_numpy(batch.x0).to(eng.device), debug=dbg)
torch.cuda.synchronize()
p, ref = tg._oracle_batch(oracle, pkg, routes, batch, T)
st = eng.status.cpu().numpy(); ok = st==0
oa, od = eng.oa.cpu().numpy(), eng.od.cpu().numpy()
eu = np.maximum(np.abs(oa-ref["oa"]).max(axis=1), np.abs(od-ref["od"]).max(axis=1))
print("bad egos", np.flatnonzero(eu>1e-4)[:20], "of", B, "max", eu.max())
print("n_iter gpu", eng.n_iter.cpu().numpy()[:16], "oracle", ref["n_iter"][:16])
N=2*T
H = dbg["H"].cpu().numpy().reshape(B,N,N); g = dbg["g"].cpu().numpy().reshape(B,N)
print("H sym err", np.abs(H-H.transpose(0,2,1)).max(), "H[0] diag", H[0].diagonal()[:6], "g[0]", g[0][:6])
# unconstrained optimum check for an ego with n_iter==0 in oracle
z = np.flatnonzero((ref["n_iter"]==0) & ok)
print("egos with 0 iters", z[:10])
for b in z[:3]:
    u0 = -np.linalg.solve(H[b], g[b])
    ug = np.stack([oa[b], od[b]],axis=1).reshape(-1)
    print(b, "err vs -H^-1 g", np.abs(u0-ug).max(), "err vs oracle", eu[b])
b = int(np.argmax(eu)); print("worst", b, "iters", eng.n_iter.cpu().numpy()[b], ref["n_iter"][b])
u0 = -np.linalg.solve(H[b], g[b]); ug = np.stack([oa[b], od[b]],axis=1).reshape(-1); uo=np.stack([ref["oa"][b], ref["od"][b]],axis=1).reshape(-1)
print(" gpu", ug[:8]); print(" ora", uo[:8]); print(" u0 ", u0[:8])

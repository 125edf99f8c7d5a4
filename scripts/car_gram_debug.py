"""Stage-by-stage check of the Gram route of the Caratheodory step (csrc/car_gram.inc) on the reference's level inputs:
G, the tridiagonalisation's reflectors, D / beta, P1, the LU's reflectors, Phi's span -- each against the numpy restatement
(scripts/gram_route_proto.py), then the kept sets / weights against the goldens.  python scripts/car_gram_debug.py [golden names]"""
import glob, os, sys
os.environ.setdefault("SOBER_CAR_GRAM", "1")            # the route is opt-in
os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sober_amd import _native as nat
from scripts.gram_route_proto import tridiag_householder, apply_qt
from tests.test_car_algorithm import nullspace_gebrd, pivots

dev = torch.device("cuda:0")
lib = nat.load()
NS, LD = 208, 128


def run(X, mu, verbose=True):
    N, n = X.shape
    m = n + 1
    Xd = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
    mud = torch.from_numpy(mu.copy()).to(dev)
    keep = torch.empty(N + 1, dtype=torch.int32, device=dev)
    ws_ = torch.empty(N, dtype=torch.float64, device=dev)
    mo = torch.empty(N, dtype=torch.float64, device=dev)
    nat._CAR_WS.pop(dev, None)
    nbytes = lib.sober_car_ws_bytes(N, m)
    ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
    nat._CAR_WS[dev] = ws
    nat.car_device(Xd, mud, keep, ws_, keep[N:], mo)
    torch.cuda.synchronize()
    nk = int(keep[N].item())
    w = ws.cpu().numpy()
    o = 0
    vws = w[o:o + m * NS].reshape(m, NS); o += m * NS
    taup = w[o:o + 128]; o += 128
    Phi = w[o:o + NS * 128].reshape(128, NS).T; o += NS * 128 + 512
    comm = w[o:o + 16].view(np.uint32); o += 16
    G = w[o:o + 128 * 128].reshape(128, 128); o += 128 * 128
    uws = w[o:o + 128 * 128].reshape(128, 128); o += 128 * 128
    tauq = w[o:o + 128]; o += 128
    Dq = w[o:o + 128]; o += 128
    betaq = w[o:o + 128]; o += 128
    P1 = w[o:o + NS * 128].reshape(NS, 128)
    A = np.vstack([np.ones(N), X.T])
    Gr = A @ A.T
    a, bs, U, tau = tridiag_householder(Gr)
    Dr = np.zeros(m); Dr[0] = a[0]
    for i in range(1, m):
        Dr[i] = a[i] - bs[i - 1] ** 2 / Dr[i - 1]
    route = comm[72 // 4] & 15
    res = dict(n_keep=nk, route=int(route), err=int(comm[0]), doneA=int(comm[80 // 4]), done=int(comm[56 // 4]))
    res["dG"] = np.abs(G[:m, :m] - Gr).max() / np.abs(Gr).max()
    res["dU"] = np.abs(uws[:max(m - 2, 0), :m] - U[:max(m - 2, 0)]).max() if m > 2 else 0.0
    res["dtauq"] = np.abs(tauq[:max(m - 2, 0)] - tau[:max(m - 2, 0)]).max() if m > 2 else 0.0
    res["dD"] = np.abs(Dq[:m] / Dr - 1).max()
    res["dbeta"] = np.abs(betaq[:m - 1] - bs).max() / np.abs(bs).max()
    if route == 1:
        C = apply_qt(U, tau, A)
        d = np.sqrt(Dr); e = bs / d[:-1]
        P1t = np.zeros((m, N)); P1t[0] = C[0] / d[0]
        for i in range(1, m):
            P1t[i] = (C[i] - e[i - 1] * P1t[i - 1]) / d[i]
        res["dP1"] = np.abs(P1[:N, :m] - P1t.T).max()
        res["orthP1"] = np.abs(P1[:N, :m].T @ P1[:N, :m] - np.eye(m)).max()
        # the LU's reflectors against the numpy restatement (scripts/gram_route_proto.py)
        Wk = P1t.T.copy(); V = np.zeros((N, m)); taus = np.zeros(m)
        for j in range(m):
            sj = Wk[j, j]; Dj = -1.0 if sj >= 0 else 1.0; piv = 1.0 + abs(sj)
            l = -Dj * Wk[j + 1:, j] / piv
            V[j, j] = 1.0; V[j + 1:, j] = l; taus[j] = piv
            Wk[j + 1:, j + 1:] -= np.outer(l, Wk[j, j + 1:])
        res["dV"] = np.abs(vws[:m, :N] - V.T).max()
        res["dtaup"] = np.abs(taup[:m] - taus).max()
        res["progress"] = int(comm[48 // 4] & 127)
        res["progressT"] = int(comm[64 // 4] & 127)
        st = Dq[104:113].view(np.uint64).astype(np.int64)                 # (s_memrealtime stamps: a -DCG_STAMPS build only)
        lu0 = int(taup[120:121].view(np.uint64)[0])
        names = ["start", "loaded", "tri_loop", "tri_done", "phaseA_seen", "-", "lu_done", "phaseB_last", "phaseA_last"]
        if st[0]:
          res["us"] = {nm: (int(v) - int(st[0])) / 100.0 for nm, v in zip(names, st) if nm != "-"}
          res["us"]["lu_loaded"] = (lu0 - int(st[0])) / 100.0
    P0 = nullspace_gebrd(A)
    res["dPhi"] = np.abs(Phi[:N, :N - m] - P0).max()
    w0, k0 = pivots(P0, mu)
    kr = keep[:N].cpu().numpy()
    kdev = np.flatnonzero(kr >= 0)
    res["same"] = bool(nk == len(k0) and np.array_equal(kdev, k0))
    if res["same"]:
        res["dw"] = np.abs(ws_[:nk].cpu().numpy() - w0).max() / np.abs(w0).max()
    if verbose:
        print("  ".join(f"{k}={v:.2e}" if isinstance(v, float) else f"{k}={v}" for k, v in res.items()))
    return res


if __name__ == "__main__":
    names = sys.argv[1:] or ["rbf_medium", "cfg1_rbf_ard", "matern_medium", "rbf_b30", "rbf_tiny_direct"]
    for nm in names:
        z = np.load(os.path.join(ROOT, "tests", "golden", f"recomb_{nm}.npz"))
        for i in range(int(z["n_levels"])):
            if f"L{i}_X_tmp" not in z.files:
                continue
            X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
            if X.shape[1] + 1 >= X.shape[0]:
                continue
            print(nm, i, X.shape, end="  ")
            run(X, mu)

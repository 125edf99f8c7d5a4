"""numpy prototype of the Gram-tridiagonalisation route to LAPACK's null-space basis (round 5):
   A = [1 | X]^T (m x n).  G = A A^T = Q T Q^T (Householder, Q e1 = e1), T = B B^T, P1^T = B^-1 Q^T A,
   Householder reconstruction (no-pivot LU of E1 - P1 D with D_jj = -sign(pivot candidate)) -> reflectors -> Phi.
Run on every golden level and on the level inputs of random fuzz cases, against dgebd2 (tests/test_car_algorithm.py)."""
import glob, os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_car_algorithm import nullspace_gebrd, pivots, larfg


def tridiag_householder(G):
    """G = Q T Q^T, Q e1 = e1; returns diag a, offdiag bsub, and the reflectors (U rows, tau)."""
    G = G.copy()
    m = G.shape[0]
    U = np.zeros((m, m)); tau = np.zeros(m)
    for i in range(m - 2):
        beta, t, v = larfg(G[i + 1, i], G[i + 2:, i])
        u = np.r_[1.0, v]
        U[i, i + 1:] = u; tau[i] = t
        S = G[i + 1:, i + 1:]
        p = t * (S @ u)
        w = p - 0.5 * t * (p @ u) * u
        G[i + 1:, i + 1:] = S - np.outer(u, w) - np.outer(w, u)
        G[i + 1, i] = G[i, i + 1] = beta
        G[i + 2:, i] = 0; G[i, i + 2:] = 0
    return np.diag(G).copy(), np.diag(G, -1).copy(), U, tau


def apply_qt(U, tau, A):
    """C = Q^T A, Q = H(0) H(1) ... (each column of A independently)."""
    C = A.copy()
    for i in range(U.shape[0] - 2):
        u = U[i]
        C -= tau[i] * np.outer(u, u @ C)
    return C


def nullspace_gram(A, variant="T"):
    m, n = A.shape
    G = A @ A.T
    a, bs, U, tau = tridiag_householder(G)
    C = apply_qt(U, tau, A)
    P1t = np.zeros((m, n))
    if variant == "T":            # B from T = B B^T (scalar recurrence), rows p_i = (c_i - e_{i-1} p_{i-1}) / d_i
        d = np.zeros(m); e = np.zeros(m)
        d[0] = np.sqrt(a[0])
        P1t[0] = C[0] / d[0]
        for i in range(1, m):
            e[i - 1] = bs[i - 1] / d[i - 1]
            d[i] = np.sqrt(a[i] - e[i - 1] ** 2)
            P1t[i] = (C[i] - e[i - 1] * P1t[i - 1]) / d[i]
    else:                         # Gram-Schmidt flavour: e = c_i . p_{i-1}, d = norm
        P1t[0] = C[0] / np.linalg.norm(C[0])
        for i in range(1, m):
            r = C[i] - (C[i] @ P1t[i - 1]) * P1t[i - 1]
            P1t[i] = r / np.linalg.norm(r)
    P1 = P1t.T                    # n x m, columns up to sign
    # Householder reconstruction: LU of (P1 D - E1) ... on W = E1 - P1 D; L = reflector vectors, U_jj = tau_j
    Wk = P1.copy()                # eliminated P1 (the s~ of the notes), W column j = e_j - D_jj * Wk[:, j]
    V = np.zeros((n, m)); taus = np.zeros(m)
    for j in range(m):
        s = Wk[j, j]
        Dj = -1.0 if s >= 0 else 1.0          # D_jj = -sign(candidate); pivot = 1 + |s|
        piv = 1.0 + abs(s)
        l = -Dj * Wk[j + 1:, j] / piv
        V[j, j] = 1.0; V[j + 1:, j] = l; taus[j] = piv
        # rows i > j of the remaining columns of P1:   p_ik -= l_i p_jk   (e_k untouched: k > j)
        Wk[j + 1:, j + 1:] -= np.outer(l, Wk[j, j + 1:])
    Phi = np.zeros((n, n - m)); Phi[m:, :] = np.eye(n - m)
    for i in range(m - 1, -1, -1):
        Phi -= taus[i] * np.outer(V[:, i], V[:, i] @ Phi)
    return Phi


def compare(A, mu, tag, stats):
    m, n = A.shape
    P0 = nullspace_gebrd(A)
    w0, k0 = pivots(P0, mu)
    out = []
    for var in ("T", "GS"):
        P1 = nullspace_gram(A, var)
        err = np.abs(P0 - P1).max()
        w1, k1 = pivots(P1, mu)
        same = np.array_equal(k0, k1)
        rw = np.abs(w1 - w0).max() / np.abs(w0).max() if same else np.nan
        out.append((err, same, rw))
    sv = np.linalg.svd(A, compute_uv=False)
    stats.append((tag, m, n, sv[0] / sv[-1], out))
    return out


if __name__ == "__main__":
    warnings.simplefilter("ignore")
    stats = []
    GOLD = os.path.join(ROOT, "tests", "golden")
    for path in sorted(glob.glob(os.path.join(GOLD, "recomb_*.npz"))):
        z = np.load(path)
        for i in range(int(z["n_levels"])):
            if f"L{i}_X_tmp" not in z.files:
                continue
            X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
            A = np.vstack([np.ones(len(X)), X.T])
            if A.shape[0] >= A.shape[1]:
                continue
            compare(A, mu, os.path.basename(path)[7:-4] + f":L{i}", stats)
    nfuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    if nfuzz:
        from oracle import sober_oracle as O
        from tests.tools.fuzz_parity import make_case, t, _obj
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        orig = O.tchernychova_lyons_car
        captured = []
        def spy(X, mu):
            captured.append((X.numpy().copy(), mu.numpy().copy()))
            return orig(X, mu)
        O.tchernychova_lyons_car = spy
        for ci in range(nfuzz):
            c = make_case(rng, False, batches=[5, 8, 9, 16, 17, 24, 32, 33, 40, 56, 64, 80, 100], n_factor=12)
            captured.clear()
            torch.manual_seed(c["seed"])
            O.recombination(t(c["X"]), t(c["Xn"]), c["b"], O.Kernel(c["spec"], c["mode"]), init_weights=t(c["mu0"].copy()), calc_obj=_obj(c))
            for li, (X, mu) in enumerate(captured):
                A = np.vstack([np.ones(len(X)), X.T])
                if A.shape[0] >= A.shape[1]:
                    continue
                compare(A, mu, f"fuzz{ci}(b={c['b']},d={c['d']},M={c['M']}):L{li}", stats)
    bad = {"T": 0, "GS": 0}
    worst = {"T": 0.0, "GS": 0.0}
    for tag, m, n, cond, out in stats:
        line = f"{tag:40s} m={m:3d} n={n:3d} cond(A)={cond:9.2e}"
        for var, (err, same, rw) in zip(("T", "GS"), out):
            line += f" | {var}: dPhi={err:8.1e} same={same} dw={rw:8.1e}"
            if not same or not (rw < 1e-7):
                bad[var] += 1
            worst[var] = max(worst[var], err)
        print(line)
    print("levels:", len(stats), "bad:", bad, "worst dPhi:", worst)

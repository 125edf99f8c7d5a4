"""The device Caratheodory kernel (sober_amd/csrc/car.hip) takes its null-space basis from the
right Householder reflectors of a Golub-Kahan bidiagonalisation instead of LAPACK's SVD.  This CPU
test pins the claim that makes that legitimate: on the reference's own per-level inputs (golden
fixtures) that basis equals torch.linalg.svd's Vh[m:, :] (MKL gesdd) to rounding, and the pivot
loop run on it selects the same sets with the same weights."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def larfg(alpha, x):
    xnorm = np.sqrt(np.dot(x, x))
    if xnorm == 0:
        return alpha, 0.0, np.zeros_like(x)
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    return beta, (beta - alpha) / beta, x / (alpha - beta)


def nullspace_gebrd(A):
    """numpy restatement of phases 1-2 of k_car: dgebd2 (m < N) + backward accumulation of
    P [0; I] -> Phi (N, N-m)."""
    A = A.copy()
    m, N = A.shape
    taup = np.zeros(m)
    for i in range(m):
        beta, tau, v = larfg(A[i, i], A[i, i + 1:])
        vv = np.r_[1.0, v]
        if i < m - 1:
            w = A[i + 1:, i:] @ vv
            A[i + 1:, i:] -= tau * np.outer(w, vv)
        A[i, i], A[i, i + 1:], taup[i] = beta, v, tau
        if i < m - 1:
            beta2, tauq, u = larfg(A[i + 1, i], A[i + 2:, i])
            uu = np.r_[1.0, u]
            w = uu @ A[i + 1:, i + 1:]
            A[i + 1:, i + 1:] -= tauq * np.outer(uu, w)
    Phi = np.zeros((N, N - m))
    Phi[m:, :] = np.eye(N - m)
    for i in range(m - 1, -1, -1):
        vt = np.zeros(N)
        vt[i] = 1.0
        vt[i + 1:] = A[i, i + 1:]
        Phi -= taup[i] * np.outer(vt, vt @ Phi)
    return Phi


def nullspace_merged(A):
    """numpy restatement of the MERGED bidiagonalisation of csrc/car.hip (car_bidiag2_block, round 3): the same
    reflectors as dgebd2 in exact arithmetic, in the operation order of the kernel -- row i+1 and column i+1 of
    A' = A G(i) are kept aside (rowg, colg); the next step's row and column follow from them and from z without
    touching the matrix; H(i)'s scalars come after the column sums Y = col'[i+2:]^T A'."""
    A = A.copy()
    m, N = A.shape
    V, taup = np.zeros((m, N)), np.zeros(m)
    rowg, colg = A[0].copy(), A[:, 0].copy()
    z, f, tauq = np.zeros(N), np.zeros(m), 0.0
    for i in range(m):
        if i > 0:
            A[i:, i:] -= np.outer(f[i:], z[i:])                      # H(i-1)'s update
        x = rowg[i + 1:] - tauq * z[i + 1:]                          # row i, rebuilt (u_i = 1)
        alpha = rowg[i] - tauq * z[i]
        cur = colg - f * z[i]                                        # column i, rebuilt
        beta, tau, v = larfg(alpha, x)
        sc = 0.0 if not np.any(x) else 1.0 / (alpha - beta)
        V[i, i], V[i, i + 1:], taup[i] = 1.0, v, tau
        if i == m - 1:
            break
        tG = tau * (cur[i + 1:] + sc * (A[i + 1:, i + 1:] @ x))      # tau * A [1; v]
        colp = cur[i + 1:] - tG                                      # column i after G(i)
        A[i + 1:, i + 1:] -= np.outer(tG, v)
        rowg, colg = A[i + 1].copy(), A[:, i + 1].copy()
        Y = colp[1:] @ A[i + 2:, :]
        beta2, tauq, u = larfg(colp[0], colp[1:])
        sc2 = 0.0 if not np.any(colp[1:]) else 1.0 / (colp[0] - beta2)
        z = np.zeros(N)
        z[i + 1:] = rowg[i + 1:] + sc2 * Y[i + 1:]
        f = np.zeros(m)
        f[i + 1], f[i + 2:] = tauq, tauq * sc2 * colp[1:]
    Phi = np.zeros((N, N - m))
    Phi[m:, :] = np.eye(N - m)
    for i in range(m - 1, -1, -1):
        Phi -= taup[i] * np.outer(V[i], V[i] @ Phi)
    return Phi


def pivots(Phi, mu):
    """phase 3 of k_car (q = prow/pp then one FMA, instead of (prow*col)/pp)."""
    Phi, mu = Phi.copy(), mu.copy()
    N, NC = Phi.shape
    for s in range(NC):
        col = Phi[:, s].copy()
        pos = col > 0
        if not pos.any():
            break
        a = np.where(pos, mu / np.where(pos, col, 1.0), np.inf)
        piv = int(np.argmin(a))
        mu = mu - a[piv] * col
        mu[piv] = 0.0
        q = Phi[piv, s + 1:] / col[piv]
        Phi[:, s + 1:] -= np.outer(col, q)
        Phi[piv, :] = 0.0
    keep = mu > 0
    return mu[keep], np.flatnonzero(keep)


CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "recomb_*.npz")) if "cfg2" not in p and "calc_obj" not in p)


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_gebrd_nullspace_reproduces_reference_car(path):
    z = np.load(path)
    for i in range(int(z["n_levels"])):
        X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
        A = np.vstack([np.ones(len(X)), X.T])
        m, N = A.shape
        Vh = torch.linalg.svd(torch.from_numpy(A))[2].numpy()
        Phi = nullspace_gebrd(A)
        np.testing.assert_allclose(Phi, Vh[m:].T, rtol=0, atol=1e-9)
        w, idx = pivots(Phi, mu)
        assert np.array_equal(idx, z[f"L{i}_idx_star"]), (path, i)
        np.testing.assert_allclose(w, z[f"L{i}_w_star"], rtol=1e-7)


def test_car_invariant_under_orthogonal_mixing():
    """The Caratheodory step sees the Nystrom test functions only through [1 | X]^T [1 | X]'s first row and
    Gram structure: X -> X O (O orthogonal) leaves the kept sets identical and the weights equal to rounding.
    This is what allows the device route to use ANY orthonormal basis of svd_lowrank's subspace (no small SVD)."""
    import glob
    from oracle import sober_oracle as O
    torch.manual_seed(0)
    n_checked = 0
    for f in sorted(glob.glob(os.path.join(GOLD, "recomb_*.npz"))):
        z = np.load(f)
        for k in [k for k in z.files if k.endswith("_X_tmp") and "cks" not in k and "shape" not in k]:
            X, mu = torch.from_numpy(z[k]), torch.from_numpy(z[k.replace("X_tmp", "tot_weights")])
            w1, i1 = O.tchernychova_lyons_car(X.clone(), mu.clone())
            Q, _ = torch.linalg.qr(torch.randn(X.shape[1], X.shape[1], dtype=torch.float64))
            w2, i2 = O.tchernychova_lyons_car((X @ Q).contiguous(), mu.clone())
            assert torch.equal(i1, i2), (f, k)
            np.testing.assert_allclose(w2.numpy(), w1.numpy(), rtol=1e-9)
            n_checked += 1
    assert n_checked >= 20


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_merged_bidiagonalisation_gives_the_dgebd2_basis(path):
    """The merged two-barrier form of the device kernel (round 3) against dgebd2 on the reference's own level inputs:
    the same null-space basis to rounding, the same kept sets, the same weights -- the parity bar of the contract
    (identical indices, weights well inside 1e-7), not bit-equality with LAPACK's rounding."""
    z = np.load(path)
    for i in range(int(z["n_levels"])):
        X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
        A = np.vstack([np.ones(len(X)), X.T])
        P0, P1 = nullspace_gebrd(A), nullspace_merged(A)
        assert np.abs(P0 - P1).max() < 1e-11, i
        w0, k0 = pivots(P0, mu)
        w1, k1 = pivots(P1, mu)
        assert np.array_equal(k0, k1), i
        np.testing.assert_allclose(w1, w0, rtol=1e-9)


def test_screened_ratio_test_accepts_only_the_exact_argmin():
    """The argument behind the pivot kernels' screened ratio test (csrc/car.hip sp_ratio_test, csrc/car_mc.hip ratio_test),
    restated in numpy: keys are the HIGH WORDS of approximate quotients q~ = mu * r~ with |r~ col - 1| <= e (the v_rcp_f64
    seed: e = 2^-24.4 measured on the device, the band is laid out for 2^-22), a lone candidate within BAND = 4 key steps of
    the minimum is accepted.  Property: whatever the errors do inside that bound, an accepted candidate IS the exact
    argmin of mu / col over col > 0 -- on random columns and on adversarial ones whose two smallest quotients lie
    2^-26 .. 2^-17 apart (where the screen must either pick the right one or decline)."""
    rng = np.random.default_rng(5)
    BAND, E = 4, 2.0 ** -22
    accepted = declined = 0
    for trial in range(4000):
        n = int(rng.integers(4, 257))
        col = rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, n))
        mu = rng.random(n) * np.exp(rng.uniform(-8, 2, n))
        pos = np.flatnonzero(col > 0)
        if len(pos) < 2:
            continue
        if trial % 2:                                                  # plant a near tie at the minimum
            q = mu[pos] / col[pos]
            i, j = pos[np.argsort(q)[:2]]
            mu[j] = (mu[i] / col[i]) * col[j] * (1.0 + 2.0 ** -rng.uniform(17, 26))
        exact = np.where(col > 0, mu / np.where(col > 0, col, 1.0), np.inf)
        want = int(np.argmin(exact))
        rt = (1.0 / np.where(col > 0, col, 1.0)) * (1.0 + rng.uniform(-E, E, n))      # the seed, anywhere inside its bound
        qa = mu * rt
        hi = (qa.view(np.uint64) >> np.uint64(32)).astype(np.uint64)
        key = np.where(col > 0, (hi + np.uint64(0x80100000)) & np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF))
        H = key.min()
        if not (0x80100000 <= H < 0xFFFFFFFF - BAND):
            declined += 1
            continue
        cand = np.flatnonzero(key <= H + BAND)
        if len(cand) != 1:
            declined += 1
            continue
        accepted += 1
        assert int(cand[0]) == want, (trial, cand, want)
    assert accepted > 1500 and declined > 200, (accepted, declined)

"""Multi-CU ladder probe (k_chol_mc) against the one-workgroup probe: info and smallest pivots on Gram matrices of
several sizes (rungs that fail early, late and not at all), then the time of both at 500 x 500 x 11 rungs."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def gram(M, d=10, ls=0.6, neg=0.0):
    X = rng.random((M, d))
    K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls)
    if neg:
        v = rng.standard_normal(M); v /= np.linalg.norm(v)
        K = K - neg * np.outer(v, v)
    return K


bad = 0
for M, ls, neg in ((500, 0.6, 0.0), (500, 3.0, 0.0), (500, 0.6, 3e-4), (536, 1.0, 1e-3), (257, 2.0, 0.0), (256, 0.5, 5e-5),
                   (100, 1.0, 0.0), (33, 1.0, 1e-4), (32, 1.0, 0.0), (480, 5.0, 1e-2)):
    K = gram(M, ls=ls, neg=neg)
    C = torch.from_numpy(K).to(dev)
    n_r = 11
    shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=torch.float64, device=dev)
    i1 = torch.zeros(n_r, dtype=torch.int32, device=dev); p1 = torch.zeros(n_r, dtype=torch.float64, device=dev)
    i2 = torch.full((n_r,), 99, dtype=torch.int32, device=dev); p2 = torch.zeros(n_r, dtype=torch.float64, device=dev)
    w1 = torch.empty(n_r * M * M, dtype=torch.float64, device=dev); w2 = torch.empty_like(w1)
    ws = torch.empty(nat.cholesky_probe_mc_ws_bytes(M, n_r), dtype=torch.uint8, device=dev)
    nat.cholesky_probe(C, shifts, w1, i1, p1)
    nat.cholesky_probe_mc(C, shifts, w2, i2, p2, ws)
    torch.cuda.synchronize()
    a, b = i1.cpu().tolist(), i2.cpu().tolist()
    pa, pb = p1.cpu().numpy(), p2.cpu().numpy()
    ok = a == b and np.array_equal(pa, pb)
    bad += not ok
    print(M, ls, neg, "OK" if ok else "MISMATCH", a, b, "max piv rel diff", np.max(np.abs(pa - pb) / np.maximum(np.abs(pa), 1e-300)))
    if M == 500 and neg == 0.0:
        print("   one workgroup per rung: %.1f us   eight: %.1f us" % (
            timed(lambda: nat.cholesky_probe(C, shifts, w1, i1, p1)), timed(lambda: nat.cholesky_probe_mc(C, shifts, w2, i2, p2, ws))))
print("mismatches", bad)

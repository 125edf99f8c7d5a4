"""Time the one-workgroup Cholesky kernels: the 11-rung probe of a 500 x 500 Gram, single factorisations at
n = 99 / 198 / 500, the blocked TRSM; with a -DCH_STAMPS build also the per-phase ticks of k_chol."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
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


M = 500
X = rng.random((M, 10))
K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.6)
C = torch.from_numpy(K).to(dev)
shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(11)], dtype=torch.float64, device=dev)
info = torch.zeros(11, dtype=torch.int32, device=dev)
work = torch.empty(11 * M * M, dtype=torch.float64, device=dev)
t = timed(lambda: nat.cholesky_probe(C, shifts, work, info))
print("cholesky_probe 500x500 x 11 rungs: %.1f us" % t, "info", info.cpu().tolist())
k = int(np.argmax(info.cpu().numpy() == 0))
Lk = np.tril(work[k * M * M:(k + 1) * M * M].view(M, M).cpu().numpy())
ref = np.linalg.cholesky(K + float(shifts[k]) * np.eye(M))
print("rung", k, "max |L - numpy|", np.abs(Lk - ref).max())
for n in (99, 128, 198, 500):
    A = rng.standard_normal((n, 2 * n)); S = A @ A.T / n + 0.5 * np.eye(n)
    Sd = torch.from_numpy(S).to(dev)
    W = Sd.clone(); inf1 = torch.zeros(1, dtype=torch.int32, device=dev); piv = torch.zeros(1, dtype=torch.float64, device=dev)
    def run():
        W.copy_(Sd); nat.cholesky(W, 0.0, inf1, piv)
    tc = timed(lambda: W.copy_(Sd))
    t = timed(run)
    run(); torch.cuda.synchronize()
    Wh = W.cpu().numpy()
    err = np.abs(np.tril(Wh) - np.linalg.cholesky(S)).max()
    print("k_chol n=%d: %.1f us (copy %.1f us subtracted), err %.2e, stamps(ticks: (a) load D, (b) diagonal block, (c) panel, (d) update + look-ahead, copy back) %s" % (n, t - tc, tc, err, Wh[0, 8:13].tolist()))

for m, q in ((500, 99), (500, 199)):
    Y = torch.from_numpy(rng.standard_normal((m, q))).to(dev)
    G = (Y.T @ Y).contiguous(); Gd = G.clone()
    inf1 = torch.zeros(1, dtype=torch.int32, device=dev); piv = torch.zeros(1, dtype=torch.float64, device=dev)
    xinv = torch.empty(((q + 31) // 32) * 1024, dtype=torch.float64, device=dev)
    nat.cholesky_inv(Gd, 0.0, inf1, piv, xinv)
    Q = torch.empty_like(Y)
    print("q=%d trsm_blocks %.1f us" % (q, timed(lambda: nat.trsm_blocks(Y, Gd, xinv, Q))))


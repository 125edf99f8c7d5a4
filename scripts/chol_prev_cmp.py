"""Pivots / info / L / X of the current library's Cholesky kernels against a previous build (SOBER_PREV_LIB)."""
import numpy as np, torch, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
out = {}
for n in (99, 100, 32, 33, 199, 500):
    A = rng.standard_normal((n, 2 * n)); S = A @ A.T / n + 0.5 * np.eye(n)
    W = torch.from_numpy(S).to(dev); inf1 = torch.zeros(1, dtype=torch.int32, device=dev); piv = torch.zeros(1, dtype=torch.float64, device=dev)
    xinv = torch.zeros(((n + 31) // 32) * 1024, dtype=torch.float64, device=dev)
    nat.cholesky_inv(W, 0.0, inf1, piv, xinv)
    torch.cuda.synchronize()
    out[n] = (np.tril(W.cpu().numpy()), xinv.cpu().numpy(), int(inf1), float(piv))
np.save(sys.argv[1], np.array([out], dtype=object), allow_pickle=True)
if len(sys.argv) > 2:
    other = np.load(sys.argv[2], allow_pickle=True)[0]
    for n in out:
        a, b = out[n], other[n]
        print(n, "L equal", np.array_equal(a[0], b[0]), "X equal", np.array_equal(a[1], b[1]), "info", a[2], b[2], "piv equal", a[3] == b[3])

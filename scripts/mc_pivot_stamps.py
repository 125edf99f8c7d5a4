"""When every pivot of k_mc_pivot is published (diagnostic build -DMC_TSTAMPS: make -C sober_amd/csrc BUILD=build_mts
EXTRA='-DSOBER_DIAG_BUILD -DMC_TSTAMPS' OUT=build_mts/libsober_hip_mts.so) at N = 400, m = 200: microseconds between consecutive publishes."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
N, m = 400, 200
X = rng.standard_normal((N, m - 1)) * np.exp(-0.02 * np.arange(m - 1))[None, :]
mu = rng.random(N) + 0.1
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
lib = nat.load()
acc = []
BC = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for it in range(8):
    nat.car_device(Xd, mud, kr, ws, nk, mo, multi_cu=True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 260)()
    assert lib.sober_debug_mc_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.int64)
    K = N - m
    if it >= 3:
        acc.append((t[:K] - t[256]) / 100.0)
        tail = (t[257] - t[K - 1]) / 100.0
a = np.mean(acc, 0)
d = np.diff(np.concatenate([[0.0], a]))
first = np.arange(len(d)) % BC == 0
print("pivots", len(d), "first publish %.2f us after wave 0 was elected; last publish at %.2f us; tail %.2f us" % (d[0], a[-1], tail))
print("in-block pivots : mean %.3f us (min %.3f max %.3f)" % (d[~first].mean(), d[~first].min(), d[~first].max()))
print("hand-over pivots: mean %.3f us (min %.3f max %.3f)  [%d of them]" % (d[first][1:].mean(), d[first][1:].min(), d[first][1:].max(), first.sum() - 1))
print("by position in the block:", [round(float(d[(np.arange(len(d)) % BC == j) & (np.arange(len(d)) >= BC)].mean()), 3) for j in range(BC)])
print("per block (us):", [round(float(d[b * BC:(b + 1) * BC].sum()), 2) for b in range((len(d) + BC - 1) // BC)])

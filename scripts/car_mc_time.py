"""Time the Caratheodory step on the device: one-CU kernels (car.hip) and multi-CU kernels (car_mc.hip) at batch 100,
multi-CU at batch 200 (N = 400, m = 200).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")


def bench(X, mu, mc, reps=50):
    N = X.shape[0]
    Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
    kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
    for it in range(3): nat.car_device(Xd, mud, kr, ws, nk, mo, multi_cu=mc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(reps): nat.car_device(Xd, mud, kr, ws, nk, mo, multi_cu=mc)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, int(nk.item())


z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
print("batch 100 (200 x 100), one CU  : %.4f ms, n_keep %d" % bench(X, mu, False))
print("batch 100 (200 x 100), multi CU: %.4f ms, n_keep %d" % bench(X, mu, True))
rng = np.random.default_rng(0)
for (N, m) in [(400, 200), (300, 150), (448, 224)]:
    X = rng.standard_normal((N, m - 1)) * np.exp(-0.02 * np.arange(m - 1))[None, :]
    mu = rng.random(N) + 0.1
    print("N %d m %d, multi CU: %.4f ms, n_keep %d" % ((N, m) + bench(X, mu, True)))

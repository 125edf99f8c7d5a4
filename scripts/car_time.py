"""Time sober_car_device on the reference's level-0 input of the matern_medium golden (N = 200, m = 100).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N, n = X.shape
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for it in range(3): nat.car_device(Xd, mud, kr, ws, nk, mo)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(50): nat.car_device(Xd, mud, kr, ws, nk, mo)
e1.record(); torch.cuda.synchronize(); print("avg ms per call", e0.elapsed_time(e1) / 50, "n_keep", int(nk.item()))

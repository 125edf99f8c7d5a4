"""Per-segment cycle counts of k_car_pivot, waves 0 and 1 (build car.hip with -DCAR_STAMPS)."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
z = np.load("tests/golden/recomb_matern_medium.npz")
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N, n = X.shape; m = n + 1
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for it in range(3):
    nat.car_device(Xd, mud, kr, ws, nk, mo)
    torch.cuda.synchronize()
    scratch = nat._CAR_WS[Xd.device]
    Phi = scratch[m * 208 + 128:]
    sub = Phi.view(torch.int64)[207 * 128 + 112: 207 * 128 + 128].cpu().numpy()
    print("pivot kernel ticks per pivot [read+mu, owner ratio (avg), non-owner elim, barrier wait]: wave0", sub[:4] / 100.0, " wave1", sub[8:12] / 100.0,
          " owner: iteration start -> published:", sub[4] / max(sub[5], 1), "ticks x", sub[5], " owner's own barrier wait:", sub[6] / max(sub[5], 1), " barrier wait + deferred block:", sub[7] / max(sub[5], 1))
    arr = Phi.view(torch.int64)[206 * 128 + 64: 206 * 128 + 80].cpu().numpy()
    print("per-wave mean arrival at the barrier (ticks after iteration start):", (arr / 100.0).round())
    a50 = Phi.view(torch.int64)[205 * 128 + 64: 205 * 128 + 80].cpu().numpy()
    print("arrival at step 50 (owner = wave 3):", a50)

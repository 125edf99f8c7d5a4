"""Where a bidiagonalisation step of the one-CU Caratheodory kernel (csrc/car.hip) spends its cycles: run with the
stamp build (`make -C sober_amd/csrc stamps`, SOBER_HIP_LIB=sober_amd/csrc/build_stamps/libsober_hip_stamps.so)."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import os, sys
os.environ.setdefault("SOBER_CAR_UNFUSED", "1")     # the stamps live in the stand-alone bidiagonalisation kernel
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N, n = X.shape
m = n + 1
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for it in range(3):
    nat.car_device(Xd, mud, kr, ws, nk, mo)
torch.cuda.synchronize()
buf = nat._CAR_WS[Xd.device]
off = m * 208 + 128 + 208 * 128
d = buf[off:off + 512].cpu().numpy().view(np.uint64).astype(np.float64)[:4 * 8 * 10].reshape(4, 8, 10)
names = ["top", "larfg2", "z,x,ss", "larfg", "q->tG", "G-upd+store+publish", "Y+writes", "bar1", "D", "bar2"]   # merged form (round 3)
tot = d.sum(1)
for w in range(4):
    t = tot[w] / m
    print("wave %d " % w + " ".join("%s %.0f" % (a, b) for a, b in zip(names, t)) + " | sum %.0f" % t.sum())
print("by 16-step block (wave 0): " + " ".join("%.0f" % (d[0, s].sum() / 16) for s in range(7)))

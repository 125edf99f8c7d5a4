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
ph = torch.zeros(N, N - m, dtype=torch.float64, device=dev)
for it in range(5):
    nat.car_device(Xd, mud, kr, ws, nk, mo, ph)
    torch.cuda.synchronize()
    st = ph.view(torch.int64).flatten()[:8].cpu().numpy()
    cyc = np.diff(st[0::2]); rt = np.diff(st[1::2])
    sub = ph.view(torch.int64).flatten()[16:22].cpu().numpy()
    if sub.sum() > 0: print("phase-1 sub-phases (wave 0 cycles): A+bar, B+bar, C+bar, D1+bar, D2+bar, D3:", sub, "per step", sub / 100.0)
    print("cycles phase1/2/3:", cyc, " realtime(100MHz ticks):", rt, " => us:", rt / 100.0, " clock GHz:", cyc.sum() / (rt.sum() / 100.0) / 1e3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(20): nat.car_device(Xd, mud, kr, ws, nk, mo, ph)
e1.record(); torch.cuda.synchronize(); print("avg ms per call", e0.elapsed_time(e1) / 20)

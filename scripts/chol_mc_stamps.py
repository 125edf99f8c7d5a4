"""Per-phase and per-step ticks of the multi-CU ladder probe (k_chol_mc), from a -DCM_STAMPS build:
  make -C sober_amd/csrc BUILD=build_cms EXTRA='-DSOBER_DIAG_BUILD -DCM_STAMPS' OUT=build_cms/libsober_hip_cms.so
  SOBER_HIP_LIB=.../build_cms/libsober_hip_cms.so python scripts/chol_mc_stamps.py"""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
M, n_r = 500, 11
X = rng.random((M, 10)); K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.6)
C = torch.from_numpy(K).to(dev)
shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=torch.float64, device=dev)
i2 = torch.zeros(n_r, dtype=torch.int32, device=dev); p2 = torch.zeros(n_r, dtype=torch.float64, device=dev)
w2 = torch.empty(n_r * M * M, dtype=torch.float64, device=dev)
ws = torch.empty(nat.cholesky_probe_mc_ws_bytes(M, n_r), dtype=torch.uint8, device=dev)
for _ in range(3): nat.cholesky_probe_mc(C, shifts, w2, i2, p2, ws)
torch.cuda.synchronize()
W = w2.view(n_r, M, M).cpu().numpy()
print("ticks (100 MHz -> x10 ns): wait+X | panel | FP wait | gather | update(+diag) | end sync")
for r in (1, 9):
    for g in range(8):
        print(r, g, (W[r, g, 400:406] / 100).round(1).tolist(), "us   sum %.1f" % (W[r, g, 400:406].sum() / 100))
r = 1
T = W[r, :8, 300:396].reshape(8, 16, 6)
t0 = T[T > 0].min()
own = lambda j: (j % 16) if (j % 16) < 8 else 15 - (j % 16)
print("per step (us since start): for each WG: X seen | panel done | FP seen | gathered | updated | synced")
for k in range(0, 15):
    print("step", k, "owner", own(k), "next owner", own(k + 1))
    for g in range(8):
        if T[g, k].max() > 0:
            print("    wg", g, ((T[g, k] - t0) / 100).round(1).tolist())
print("last ch_diag_block (ticks/100 = us): factor, invert:", (W[1, :8, 410:412] / 100).round(2).tolist())

"""Timeline of k_car_pivot_team from its own s_memrealtime stamps (-DTP_STAMPS build: SOBER_HIP_LIB=.../libsober_hip_tps.so):
publish time of every pivot (chain wave 0), kernel entry (258) and result out (257), 100 MHz clock."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N = X.shape[0]
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
lib = nat.load()
for it in range(3): nat.car_device(Xd, mud, kr, ws, nk, mo)
torch.cuda.synchronize()
ph = (ctypes.c_ulonglong * 32)()
lib.sober_debug_tp_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.sober_debug_tp_phases(ph, 1)
nat.car_device(Xd, mud, kr, ws, nk, mo)
torch.cuda.synchronize()
lib.sober_debug_tp_phases(ph, 0)
names = ["divisions+keys", "wave min", "ballots(+pick-up)", "record", "keys arrive", "winner data+publish", "weights+elim"]
for q in range(4):
    print("chain wave", q, "cycles per pivot:", {names[k]: int(ph[q * 8 + k]) // 100 for k in range(7)}, "sum", sum(int(ph[q * 8 + k]) for k in range(7)) // 100)
out = (ctypes.c_ulonglong * 520)()
lib.sober_debug_tp_stamps.argtypes = [ctypes.c_void_p]
rc = lib.sober_debug_tp_stamps(out)
t = np.array(out[:], dtype=np.float64) / 100.0     # us
K = N - (X.shape[1] + 1)
t0 = t[257]
p = t[:K] - t0
d = np.diff(p)
print("n_keep", int(nk.item()), "pivots", K, "first publish after entry %.2f us, last %.2f us, result out %.2f us" % (p[0], p[-1], t[256] - t0))
print("per pivot: mean %.3f us; by position in the block:" % d.mean(), [round(float(np.mean([d[i] for i in range(len(d)) if (i + 1) % 7 == j])), 3) for j in range(7)])
print("per block (us):", [round(float(p[min(7 * k + 6, K - 1)] - (p[7 * k - 1] if k else 0.0)), 2) for k in range((K + 6) // 7)])
print("blocks reach the near store at (us):", {i: round(float(t[260 + i] - t0), 2) for i in range(1, 16) if t[260 + i] > 0})
print("far waves done at (us):", {i: round(float(t[300 + i] - t0), 2) for i in range(5, 16) if t[300 + i] > 0})
print("fetcher (far workgroup 0) commits pivot s at (us):", [round(float(t[400 + i] - t0), 2) for i in range(0, 66, 5) if t[400 + i] > 0])
print("chain publishes pivot s at (us):", [round(float(p[i]), 2) for i in range(0, K, 5)])

"""Where a bidiagonalisation step of the multi-CU Caratheodory kernel spends its cycles: run with the stamp build
(`make -C sober_amd/csrc stamps`, SOBER_HIP_LIB=sober_amd/csrc/build_stamps/libsober_hip_stamps.so).  Prints, per wave role,
the mean cycles per step of every segment (shares, not absolute times: the stamps serialise the LDS traffic)."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
N, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 200)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, m - 1)) * np.exp(-0.02 * np.arange(m - 1))[None, :]
mu = rng.random(N) + 0.1
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for it in range(3):
    nat.car_device(Xd, mud, kr, ws, nk, mo, multi_cu=True)
torch.cuda.synchronize()
buf = nat._CAR_WS[Xd.device]
dbg = buf[-8192:].cpu().numpy().view(np.uint64).astype(np.float64)
G = 9
names = ["loop", "row+q", "lds+barA", "exchange", "barB", "G(i)+stores", "H(i)", "z", "update"]
d = dbg[:G * 4 * 4 * 12].reshape(G * 4, 4, 12)       # [wave][SL][segment]
tot = d.sum(1)                                          # per wave over all steps
print("N %d m %d G %d: mean s_memtime cycles per step and segment" % (N, m, G))
for role, sel in (("waves 0-1", [w for w in range(G * 4) if w % 4 < 2]),
                  ("waves 2-3", [w for w in range(G * 4) if w % 4 >= 2])):
    t = tot[sel].mean(0) / m
    print(role, " ".join("%s %.0f" % (n, v) for n, v in zip(names, t[:9])), "| sum %.0f" % t[:9].sum())
# pivot kernel: per wave [256 + gw][12]
pn = ["p.mask", "p.ratio", "p.publish", "p.mu+elim+rot", "c.loop", "c.wait", "c.handover", "c.apply"]
dp = dbg[256 * 12:(256 + 32) * 12].reshape(32, 12)
K = N - m
print("pivot kernel, cycles per wave (sum over its pivots); chain = sum of produce + hand-over waits")
for w in (0, 1, 2, 8, 16, 24):
    if w * 8 < K:
        print("wave %2d " % w + " ".join("%s %.0f" % (n, v) for n, v in zip(pn, dp[w][:8])))
nb = (K + 7) // 8
prod = dp[:nb, :4].sum()
hand = dp[1:nb, 6].sum()
print("blocks %d: produce total %.0f cycles (%.0f per pivot), hand-over waits %.0f (%.0f per block)" % (nb, prod, prod / K, hand, hand / max(nb - 1, 1)))

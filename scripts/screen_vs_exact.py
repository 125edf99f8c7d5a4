"""Screened vs exact ratio test of the pivot kernels, bit for bit, on many random Caratheodory steps (one-CU and multi-CU
sizes; a share with zero masses, tiny negative masses, near-tied masses, duplicated points):  python scripts/screen_vs_exact.py [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(2025)
sizes = [(200, 100), (199, 99), (150, 60), (128, 64), (100, 37), (64, 20), (40, 12), (208, 100), (190, 80), (20, 10), (33, 17),
         (400, 200), (300, 180), (448, 230), (260, 40), (320, 100)]
bad = 0
for k in range(n_cases):
    N, m = sizes[k % len(sizes)]
    X = rng.standard_normal((N, m - 1)) * np.exp(-0.05 * rng.random() * np.arange(m - 1))[None, :]
    mu = rng.random(N) * np.exp(rng.uniform(-3, 0, N)) + 1e-3
    kind = k % 5
    if kind == 1: mu[:: 3 + k % 5] = 0.0
    elif kind == 2: mu[1::13] = -1e-17
    elif kind == 3: mu[:] = 1.0 / N                                      # equal masses: near ties all over
    elif kind == 4: X[N // 2:N // 2 + 7] = X[:7]; mu[N // 2:N // 2 + 7] = mu[:7]
    Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
    outs = []
    for exact in (False, True):
        if exact: os.environ["SOBER_CAR_EXACT_RATIO"] = "1"
        else: os.environ.pop("SOBER_CAR_EXACT_RATIO", None)
        nat.reload_switches()
        keep = torch.empty(N + 1, dtype=torch.int32, device=dev); w = torch.zeros(N, dtype=torch.float64, device=dev)
        mo = torch.empty(N, dtype=torch.float64, device=dev)
        nat.car_device(Xd, mud, keep, w, keep[N:], mo)
        outs.append((keep.cpu().numpy(), w.cpu().numpy().view(np.int64), mo.cpu().numpy().view(np.int64)))
    same = all(np.array_equal(a, b) for a, b in zip(*outs))
    if not same:
        bad += 1
        print("DIFFERS: case", k, (N, m), "kind", kind)
os.environ.pop("SOBER_CAR_EXACT_RATIO", None); nat.reload_switches()
print("cases", n_cases, "differing", bad)

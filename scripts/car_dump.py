"""Dump the Caratheodory step's outputs (kept ranks, weights, full weight vector) for the reference's level inputs and a set
of random steps (one-CU and multi-CU sizes) to an .npz -- run once per library build (SOBER_HIP_LIB), then
`python scripts/car_dump.py --compare a.npz b.npz` tells whether two builds agree bit for bit."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.int64) if a[k].dtype == np.float64 else a[k], b[k].view(np.int64) if b[k].dtype == np.float64 else b[k])]
    print("arrays", len(a.files), "differing", len(bad), bad[:8])
    sys.exit(1 if bad else 0)
import torch
from sober_amd import _native as nat
dev = torch.device("cuda:0")
cases = []
for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "recomb_*.npz"))):
    z = np.load(p)
    if "L0_X_tmp" not in z.files:
        continue
    for i in range(int(z["n_levels"])):
        X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
        if X.shape[1] + 1 < X.shape[0] and nat.car_supported(X.shape[0], X.shape[1] + 1):
            cases.append((f"{os.path.basename(p)[:-4]}_L{i}", X, mu))
rng = np.random.default_rng(77)
for k, (N, m) in enumerate([(200, 100), (150, 60), (128, 64), (64, 20), (40, 12), (208, 100), (400, 200), (300, 180), (448, 224), (260, 40)] * 2):
    X = rng.standard_normal((N, m - 1)) * np.exp(-0.03 * rng.random() * np.arange(m - 1))[None, :]
    cases.append((f"rand{k}_{N}x{m}", X, rng.random(N) + 0.05))
out = {}
for name, X, mu in cases:
    N = X.shape[0]
    keep = torch.empty(N + 1, dtype=torch.int32, device=dev)
    w = torch.zeros(N, dtype=torch.float64, device=dev)
    mo = torch.empty(N, dtype=torch.float64, device=dev)
    nat.car_device(torch.from_numpy(np.ascontiguousarray(X)).to(dev), torch.from_numpy(np.ascontiguousarray(mu)).to(dev), keep, w, keep[N:], mo)
    out[name + "_keep"], out[name + "_w"], out[name + "_mu"] = keep.cpu().numpy(), w.cpu().numpy(), mo.cpu().numpy()
np.savez(sys.argv[1], **out)
print("cases", len(cases), "->", sys.argv[1])

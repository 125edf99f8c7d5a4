import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
from tests.test_car_algorithm import nullspace_gebrd, pivots
dev = torch.device("cuda:0")
for name, lv in (("rbf_basekernel", 0), ("rbf_b30", 0), ("matern_medium", 0)):
    z = np.load(f"tests/golden/recomb_{name}.npz")
    X, mu = z[f"L{lv}_X_tmp"], z[f"L{lv}_tot_weights"]
    N, n = X.shape; m = n + 1
    A = np.vstack([np.ones(N), X.T])
    Phi_ref = nullspace_gebrd(A)
    Vh = torch.linalg.svd(torch.from_numpy(A))[2].numpy()
    Xd, mud = torch.from_numpy(np.ascontiguousarray(X)).to(dev), torch.from_numpy(mu).to(dev)
    kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
    ph = torch.zeros(N, N - m, dtype=torch.float64, device=dev)
    nat.car_device(Xd, mud, kr, ws, nk, mo, ph)
    P = ph.cpu().numpy()
    print(name, "N", N, "m", m, "|Phi_gpu - Phi_np|max", np.abs(P - Phi_ref).max(), "|Phi_gpu - Vh|max", np.abs(P - Vh[m:].T).max(),
          "|np - Vh|", np.abs(Phi_ref - Vh[m:].T).max())
    w_np, i_np = pivots(P, mu)          # numpy pivots on the GPU basis
    k = int(nk.item())
    print("   w(gpu) vs w(np pivots on gpu Phi):", np.abs(ws.cpu().numpy()[:k] - w_np).max() / w_np.max(), " vs golden:",
          np.max(np.abs(ws.cpu().numpy()[:k] - z[f"L{lv}_w_star"]) / z[f"L{lv}_w_star"]))
    err_cols = np.abs(P - Phi_ref).max(axis=0); print("   per-col err (first 8)", err_cols[:8], "rows", np.abs(P - Phi_ref).max(axis=1)[:8])

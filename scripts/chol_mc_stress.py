"""k_chol_mc repeated: counts rungs without a verdict (info = -7) over many launches, alone and with another stream
keeping the GPU busy (the probe's workgroups must all find a seat: 16 per XCD)."""
import numpy as np, torch, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
M, n_r = 500, 11
X = rng.random((M, 10)); K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.6)
C = torch.from_numpy(K).to(dev)
shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=torch.float64, device=dev)
info = torch.zeros(n_r, dtype=torch.int32, device=dev); piv = torch.zeros(n_r, dtype=torch.float64, device=dev)
work = torch.empty(n_r * M * M, dtype=torch.float64, device=dev)
ws = torch.empty(nat.cholesky_probe_mc_ws_bytes(M, n_r), dtype=torch.uint8, device=dev)
acc = torch.zeros(n_r, dtype=torch.int64, device=dev)
for mode in ("alone", "beside a GEMM stream"):
    side = torch.cuda.Stream()
    A = torch.rand(4096, 4096, device=dev)
    acc.zero_(); t = time.perf_counter(); n = 2000
    for it in range(n):
        if mode != "alone" and it % 4 == 0:
            with torch.cuda.stream(side):
                A @ A
        nat.cholesky_probe_mc(C, shifts, work, info, piv, ws)
        acc += (info == nat.PROBE_NO_VERDICT)
    torch.cuda.synchronize()
    print(mode, ": launches", n, "rungs without verdict", int(acc.sum()), "info", info.cpu().tolist(), "us per launch %.1f" % ((time.perf_counter() - t) / n * 1e6))

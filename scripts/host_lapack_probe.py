"""Probe: host LAPACK cost of the parity-critical pieces vs torch thread count (run on the GPU box)."""
import time, torch, numpy as np, os
print("cpu count", os.cpu_count(), "torch threads", torch.get_num_threads())
torch.manual_seed(0)
A = torch.randn(500, 500, dtype=torch.double); G = A @ A.T / 500 + 1e-3 * torch.eye(500, dtype=torch.double)
X = torch.randn(100, 200, dtype=torch.double)
X4 = torch.randn(200, 400, dtype=torch.double)
def tm(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
for th in (1, 2, 4, 8, 16, 32, 128):
    torch.set_num_threads(th)
    r = dict(th=th,
             chol=tm(lambda: torch.linalg.cholesky(G)),
             eigvalsh=tm(lambda: torch.linalg.eigvalsh(G)),
             eig=tm(lambda: torch.linalg.eig(G), 1) if th in (1, 8, 128) else -1,
             svd_lowrank=tm(lambda: torch.svd_lowrank(G, q=99)),
             svd100x200=tm(lambda: torch.linalg.svd(X)),
             svd200x400=tm(lambda: torch.linalg.svd(X4)),
             absmul=tm(lambda: torch.sqrt(G * G.T)))
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items()})

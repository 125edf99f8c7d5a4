"""Where does the host part of the Nystrom basis go (run on the GPU box)?"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sober_oracle as O
from tests.golden.synth import synth, build_spec
from sober_amd._utils import SafeTensorOperator
from sober_amd._engine import host_lapack_threads, ker_svd_sparsify_host
case = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
inp = synth(case); spec = build_spec(case, inp)
Xn = torch.from_numpy(inp["X_nys"])
G = O.Kernel(spec)(Xn, Xn)
tm = SafeTensorOperator()
def T(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
warnings.simplefilter("ignore")
print("threads default", torch.get_num_threads())
for th in (1, 4):
    torch.set_num_threads(th)
    A = torch.sqrt(torch.nan_to_num(G) * torch.nan_to_num(G).T)
    print(f"--- threads={th}")
    print("sym check          %.2f ms" % T(lambda: bool((G == G.T).all())))
    print("abs (sqrt(c*c.T))  %.2f ms" % T(lambda: torch.sqrt(torch.nan_to_num(G) * torch.nan_to_num(G).T)))
    print("eigvalsh(|G|)      %.2f ms" % T(lambda: torch.linalg.eigvalsh(A)))
    print("cholesky(|G|+.02)  %.2f ms" % T(lambda: torch.linalg.cholesky(A + 5.0 * torch.eye(500, dtype=torch.double))))
    print("make_cov_psd       %.2f ms" % T(lambda: tm.make_cov_psd(G.clone())))
    B = tm.make_cov_psd(G.clone())
    print("svd_lowrank        %.2f ms" % T(lambda: torch.svd_lowrank(B, q=99)))
    R = torch.randn(500, 99, dtype=torch.double)
    print("  randn            %.2f ms" % T(lambda: torch.randn(500, 99, dtype=torch.double)))
    print("  matmul 500x500x99 %.2f ms" % T(lambda: B @ R))
    Xq = B @ R
    print("  qr 500x99        %.2f ms" % T(lambda: torch.linalg.qr(Xq)))
    Q = torch.linalg.qr(Xq).Q
    Bs = Q.T @ B
    print("  svd 99x500       %.2f ms" % T(lambda: torch.linalg.svd(Bs, full_matrices=False)))
torch.set_num_threads(128)
print("full ker_svd_sparsify_host (thread pinning inside) %.2f ms" % T(lambda: ker_svd_sparsify_host(G.clone(), 99)))


"""The jitter ladder's multi-workgroup probe (k_chol_mc) on the Gram matrices of the BASELINE configurations: duration and each rung's
verdict -- the kernel ends when its LAST rung does, so its time follows how many rungs run to the end:  python scripts/probe_by_config.py"""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, sober_amd
from sober_amd import _native as nat
dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
for c in (2, 3, 4, 5):
    cfg = bench.CONFIGS[c]
    if c == 4:
        cfg = dict(cfg, N=100000)
    X_cand, X_nys, mu0, spec, inp, N = bench.build_inputs(cfg, 0, 1, dev)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
    tr = {}
    torch.manual_seed(123)
    sober_amd.recombination(X_cand, X_nys, cfg["b"], sober_amd.Kernel(ks, cfg["mode"]), init_weights=mu0.clone(), _trace=tr)
    G = torch.as_tensor(tr["gram"]).to(dev).double().contiguous()
    M = G.shape[0]
    C = torch.empty_like(G); flag = torch.zeros(4, dtype=torch.int32, device=dev)
    nat.abs_sym(G, C, flag)
    shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(11)], dtype=torch.float64, device=dev)
    info = torch.zeros(11, dtype=torch.int32, device=dev); piv = torch.zeros(11, dtype=torch.float64, device=dev)
    work = torch.empty(11 * M * M, dtype=torch.float64, device=dev)
    ws = torch.zeros(nat.cholesky_probe_mc_ws_bytes(M, 11), dtype=torch.uint8, device=dev)
    for _ in range(3):
        nat.cholesky_probe_mc(C, shifts, work, info, piv, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        nat.cholesky_probe_mc(C, shifts, work, info, piv, ws)
    e1.record(); torch.cuda.synchronize()
    print("cfg-%d  M=%d  k_chol_mc + init %.1f us per call; info per rung (0 = positive definite, else the failing column)" % (c, M, e0.elapsed_time(e1) / 20 * 1e3), info.cpu().tolist(), flush=True)

"""Where do the jitter ladder's rungs fail?  info (failing column + 1, 0 = positive definite) and smallest pivot of every
rung of make_cov_psd's ladder for the Gram matrices of BASELINE configurations 1-4, and the range finder's pivot ratios."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd, bench
from sober_amd import _native as nat
from tests.golden.synth import synth, build_spec
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
for c in (1, 2, 3, 4):
    cfg = bench.CONFIGS[c]
    case = {k: v for k, v in cfg.items() if k not in ("name", "golden", "strong", "cpu_sample_N", "device_pool")}
    case["N"] = min(case["N"], 20000)
    inp = synth(case); spec = build_spec(case, inp)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
    Xn = torch.from_numpy(inp["X_nys"]).to(dev)
    G = sober_amd.Kernel(ks, case["mode"])(Xn, Xn)
    M = G.shape[0]
    C = torch.empty_like(G); flag = torch.zeros(1, dtype=torch.int32, device=dev)
    nat.abs_sym(G, C, flag)
    n_r = 11
    shifts = torch.tensor([1e-5 * (2 ** k - 1) for k in range(n_r)], dtype=torch.float64, device=dev)
    work = torch.empty(n_r * M * M, dtype=torch.float64, device=dev)
    info = torch.zeros(n_r, dtype=torch.int32, device=dev); piv = torch.zeros(n_r, dtype=torch.float64, device=dev)
    nat.cholesky_probe(C, shifts, work, info, piv)
    ev = torch.linalg.eigvalsh(C.cpu())
    print("cfg", c, "M", M, "symmetric", int(flag.item()), "eig(|cov|) min %.3e max %.3e" % (float(ev[0]), float(ev[-1])),
          "info per rung", info.cpu().tolist(), "diag range %.3f..%.3f" % (float(C.diag().min()), float(C.diag().max())))

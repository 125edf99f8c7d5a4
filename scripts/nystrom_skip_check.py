"""Does skipping intermediate CholeskyQR passes of the range finder (csrc/nystrom_exec.cpp) change the RESULT?  Full-size
BASELINE configurations, every pass taken (the default) against SOBER_NYSTROM_SKIP=1; the range finder's pivot ratios."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd, bench
from tests.golden.synth import SEED_CALL
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
os.environ["SOBER_NYSTROM_DEBUG"] = "1"
for c in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 1]:
    cfg = bench.CONFIGS[c]
    X, Xn, mu0, spec, inp, N = bench.build_inputs(cfg, 0, 1, dev)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
    kern = sober_amd.Kernel(ks, cfg["mode"])
    sober_amd.setting_parameters(device=dev, dtype=torch.double)
    res = []
    for no_skip in (True, False):
        if no_skip: os.environ.pop("SOBER_NYSTROM_SKIP", None)
        else: os.environ["SOBER_NYSTROM_SKIP"] = "1"
        mu = mu0.clone(); torch.manual_seed(SEED_CALL)
        idx, w = sober_amd.recombination(X, Xn, cfg["b"], kern, dev, torch.double, init_weights=mu)
        res.append((idx.cpu().numpy(), w.cpu().numpy()))
    same = np.array_equal(res[0][0], res[1][0])
    print("cfg", c, "indices equal", same, "max rel w", float(np.abs(res[0][1] - res[1][1]).max() / np.abs(res[0][1]).max()) if same else None)

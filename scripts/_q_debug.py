import os, sys, warnings, numpy as np, torch
sys.path.insert(0, os.getcwd())
import sober_amd
from sober_amd import _ops_hip
from tests.golden.synth import SEED_CALL, load_case
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
path = "tests/golden/recomb_rbf_noleft.npz"
case, inp, spec, z = load_case(path)
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
for rep in range(4):
    for q in (True, False):
        old = _ops_hip.HipOps.__init__
        def patched(self, *a, _old=old, **k):
            _old(self, *a, **k); self.queue_levels = q
        _ops_hip.HipOps.__init__ = patched
        mu = t(inp["mu0"].copy()).to(dev)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, w = sober_amd.recombination(t(inp["X_cand"]).to(dev), t(inp["X_nys"]).to(dev), case["b"], sober_amd.Kernel(ks), init_weights=mu)
        _ops_hip.HipOps.__init__ = old
        print(rep, q, np.array_equal(idx.cpu().numpy(), z["idx"]), idx.cpu().numpy()[:5], float(w.sum()))

"""GP prediction / pi over 100k candidates at several numbers of observations: the fused launch (from the triangular root of W) against
the materialised route:  python tests/tools/predict_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sober_amd
from oracle import sober_oracle as O
from tests.golden.synth import build_spec, synth
from tests.test_hip_round4 import kspec, _t
dev = torch.device("cuda:0")
for n_obs in [int(v) for v in os.environ.get("PT_OBS", "100,200,255,300,400,511").split(",")]:
    case = dict(kind=O.RBF, mode="predictive_covariance", N=100000, M=20, d=10, n_obs=n_obs, b=5, seed=1, ard=False)
    inp = synth(case); ks = kspec(build_spec(case, inp)).to(dev)
    X = _t(inp["X_cand"]).to(dev)
    out = []
    for env in (None, "SOBER_PREDICT_FROM_W", "SOBER_PREDICT_MATERIALISED"):
        for k in ("SOBER_PREDICT_FROM_W", "SOBER_PREDICT_MATERIALISED"):
            os.environ.pop(k, None)
        if env:
            os.environ[env] = "1"
        pi = sober_amd.PI(ks)
        for _ in range(3):
            pi(X)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            pi(X)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 10 * 1e3)
    print("n_obs %3d: from the root %.3f ms | from W %.3f ms | materialised %.3f ms" % (n_obs, *out), flush=True)

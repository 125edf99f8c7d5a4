"""recombination(..., calc_obj=...) at cfg-2's shapes: the queued chain of the level executor against the level-by-level route, and
the plain step beside them, alternating in one process:  python tests/tools/calc_obj_ab.py"""
import os, sys, time, warnings, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sober_amd
from oracle import sober_oracle as O
from sober_amd._ops_hip import HipOps
from tests.golden.synth import SEED_CALL, build_spec, calc_obj_fn, synth
from tests.test_hip_round4 import kspec, _t

dev = torch.device("cuda:0")
case = dict(kind=O.RBF, mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
inp = synth(case)
spec = build_spec(case, inp)
Xc, Xn = _t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev)
obj_cache = {}
def cobj(x):
    # (the acquisition itself is not what is timed: evaluated once)
    if "v" not in obj_cache:
        obj_cache["v"] = calc_obj_fn(x)
    return obj_cache["v"]
variants = {"queued": dict(queue_obj_levels=True), "level by level": dict(queue_obj_levels=False), "plain (no calc_obj)": None}
opss = {k: HipOps(dev) for k in variants}
for k, v in variants.items():
    if v:
        for a, b in v.items():
            setattr(opss[k], a, b)
ts = {k: [] for k in variants}
kern = sober_amd.Kernel(kspec(spec), case["mode"])
for rep in range(25):
    for k in variants:
        mu = _t(inp["mu0"].copy()).to(dev)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.cuda.synchronize(); t = time.perf_counter()
            sober_amd.recombination(Xc, Xn, 100, kern, init_weights=mu, calc_obj=None if variants[k] is None else cobj, _ops=opss[k])
            torch.cuda.synchronize(); ts[k].append((time.perf_counter() - t) * 1e3)
for k, v in ts.items():
    v = v[5:]
    print("%-22s median %.3f ms  min %.3f ms" % (k, statistics.median(v), min(v)))

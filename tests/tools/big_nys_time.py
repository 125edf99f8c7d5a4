"""The Nystrom phase at N_nys beyond 1024 (device route up to 2048 since round 6) against the host route:  python tests/tools/big_nys_time.py"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sober_amd
from oracle import sober_oracle as O
from sober_amd._ops_hip import HipOps
from tests.golden.synth import SEED_CALL, build_spec, synth
from tests.test_hip_round4 import kspec, _t

dev = torch.device("cuda:0")
for M in (1000, 1500, 2048):
    case = dict(kind=O.RBF, mode="predictive_covariance", N=100000, M=M, d=10, b=100, n_obs=200, seed=7, ard=True)
    inp = synth(case); spec = build_spec(case, inp)
    Xc, Xn = _t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev)
    for host in (False, True):
        ops = HipOps(dev)
        ts = []
        for rep in range(3):
            mu = _t(inp["mu0"].copy()).to(dev); torch.manual_seed(SEED_CALL); timers = {}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.cuda.synchronize(); t = time.perf_counter()
                if host:
                    from sober_amd._engine import RecombinationEngine
                    orig = RecombinationEngine.__init__
                    def init(self, *a, **k):
                        orig(self, *a, **k); self.force_host_nystrom = True
                    RecombinationEngine.__init__ = init
                try:
                    sober_amd.recombination(Xc, Xn, 100, sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _timers=timers, _ops=ops)
                finally:
                    if host:
                        RecombinationEngine.__init__ = orig
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print("N_nys %4d  %s Nystrom route: %8.2f ms/step  (%s)" % (M, "host  " if host else "device", min(ts), ", ".join("%s %.2f" % (k, v * 1e3) for k, v in timers.items())), flush=True)

"""Whole recombination steps at batches beyond the register-resident Caratheodory kernels (csrc/car_big.hip), device step vs the
host route of rounds 1-5 (force_host_car):  python tests/tools/big_batch_time.py"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sober_amd
from oracle import sober_oracle as O
from tests.golden.synth import SEED_CALL, build_spec, synth
from tests.test_hip_round4 import kspec, _t

dev = torch.device("cuda:0")
for b, M, N in [(200, 500, 100000), (250, 600, 100000), (300, 700, 100000), (512, 1000, 100000), (1000, 1500, 100000)]:
    case = dict(kind=O.RBF, mode="predictive_covariance", N=N, M=M, d=10, b=b, n_obs=100, seed=7, ard=True)
    inp = synth(case)
    spec = build_spec(case, inp)
    Xc, Xn = _t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev)
    for host in (False, True):
        if host and b > 512:
            continue
        ts, res = [], None
        from sober_amd._ops_hip import HipOps
        from sober_amd import _native as nat
        ops = HipOps(dev)
        if host:
            ops.car_mode = nat.CAR_HOST
        for rep in range(3):
            mu = _t(inp["mu0"].copy()).to(dev)
            torch.manual_seed(SEED_CALL)
            timers = {}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.cuda.synchronize(); t = time.perf_counter()
                idx, w = sober_amd.recombination(Xc, Xn, b, sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _timers=timers,
                                                 _ops=ops)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            res = (idx.cpu().numpy(), w.cpu().numpy())
        print("batch %4d  N_nys %4d  %s Caratheodory steps: %8.1f ms/step  (%s)  kept %d" % (
            b, M, "host  " if host else "device", min(ts), ", ".join("%s %.1f" % (k, v * 1e3) for k, v in timers.items()), len(res[0])), flush=True)
        if host:
            print("    same points: %s   max rel weight diff %.1e" % (np.array_equal(res[0], dev_res[0]),
                  float(np.max(np.abs(res[1] - dev_res[1]) / dev_res[1])) if np.array_equal(res[0], dev_res[0]) else float("nan")), flush=True)
        else:
            dev_res = res

"""Step time of the SHARDED code path (Python level loop + collectives through RCCL) on one GPU with a one-rank
process group, next to the unsharded path (level executor loop): what a rank pays per step for being one of N."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import sober_amd
from sober_amd._ops_hip import HipOps
from tests.golden.synth import SEED_CALL, build_spec, synth
CFG2 = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
inp = synth(CFG2); spec = build_spec(CFG2, inp)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
kernel = sober_amd.Kernel(ks, CFG2["mode"])
sober_amd.setting_parameters(device=dev, dtype=torch.double)
X, Xn, mu0 = t(inp["X_cand"]).to(dev), t(inp["X_nys"]).to(dev), t(inp["mu0"]).to(dev)
ops = HipOps(dev)
def step(group):
    mu = mu0.clone(); torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sober_amd.recombination(X, Xn, CFG2["b"], kernel, dev, torch.double, init_weights=mu, group=group, row_offset=0, _ops=ops)
for name, g in (("unsharded (executor loop)", None), ("sharded path, 1 rank (RCCL)", dist.group.WORLD), ("unsharded again", None)):
    for _ in range(4): step(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): idx, w = step(g)
    torch.cuda.synchronize()
    print("%-30s %.3f ms/step" % (name, (time.perf_counter() - t0) / 10 * 1e3))
dist.destroy_process_group()

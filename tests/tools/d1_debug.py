"""d = 1 case (tests/golden/d1_sensitivity.npz): the device run level by level against the oracle's trace."""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sober_amd
from tests.golden import make_golden as MG
from tests.golden.synth import synth, build_spec
from tests.test_hip_parity import kspec, SEED_CALL, _t
from oracle import sober_oracle as O

case = MG.D1_CASE
inp = synth(case); spec = build_spec(case, inp)
dev = torch.device("cuda")
tr_o = {}
torch.manual_seed(SEED_CALL)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    io, wo = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], O.Kernel(spec, case["mode"]),
                             init_weights=_t(inp["mu0"].copy()), trace=tr_o)
tr = {}
mu = _t(inp["mu0"].copy()).to(dev)
torch.manual_seed(SEED_CALL)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    idx, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                     sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _trace=tr)
print("idx equal oracle:", np.array_equal(idx.cpu().numpy(), io.numpy()))
print("levels", len(tr.get("levels", [])), len(tr_o["levels"]))
Uo, Ud = tr_o["U"].numpy(), tr["U"].cpu().numpy()
# subspace distance between the two bases
Qo, _ = np.linalg.qr(Uo.T); Qd, _ = np.linalg.qr(Ud.T)
print("U subspace sin(theta) max:", np.sqrt(max(0.0, 1 - np.linalg.svd(Qo.T @ Qd, compute_uv=False).min() ** 2)))
for i, (a, b) in enumerate(zip(tr["levels"], tr_o["levels"])):
    ia, ib = np.asarray(a["idx_star"]), np.asarray(b["idx_star"])
    same = np.array_equal(ia, ib)
    wa, wb = np.asarray(a["w_star"].cpu() if hasattr(a["w_star"], "cpu") else a["w_star"]), np.asarray(b["w_star"])
    rel = float(np.max(np.abs(wa - wb) / np.abs(wb))) if same else float("nan")
    print("level", i, "R", a.get("R"), "kept sets equal:", same, "max rel w_star diff %.3e" % rel,
          "min w_star (oracle) %.3e" % float(np.min(np.abs(wb))))
    if not same:
        print("  device:", ia.tolist()); print("  oracle:", ib.tolist())
        break

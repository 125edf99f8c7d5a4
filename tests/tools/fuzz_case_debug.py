"""Replay one case of the randomised sweep (tests/tools/fuzz_parity.py) with traces on both sides and print where the
device first leaves the oracle:  [FUZZ_KIND=tanimoto] python -m tests.tools.fuzz_case_debug <seed> <case index>"""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sober_amd
from oracle import sober_oracle as O
from tests.tools import fuzz_parity as F

warnings.simplefilter("ignore")
seed, which = int(sys.argv[1]), int(sys.argv[2])
tani = os.environ.get("FUZZ_KIND") == "tanimoto"
rng = np.random.default_rng(seed)
for i in range(which + 1):
    c = F.make_case(rng, tani)
print(F.describe(which, c))
dev = torch.device("cuda:0")
t = F.t
spec, b, mode = c["spec"], c["b"], c["mode"]
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
tr_o, tr_d = {}, {}
torch.manual_seed(c["seed"])
i_o, w_o = O.recombination(t(c["X"]), t(c["Xn"]), b, O.Kernel(spec, mode), init_weights=t(c["mu0"].copy()), calc_obj=F._obj(c), trace=tr_o)
torch.manual_seed(c["seed"])
mu = t(c["mu0"].copy()).to(dev)
timers = {}
i_d, w_d = sober_amd.recombination(t(c["X"]).to(dev), t(c["Xn"]).to(dev), b, sober_amd.Kernel(ks, mode), init_weights=mu, calc_obj=F._obj(c), _trace=tr_d, _timers=timers)
print("device routes (timers):", {k: round(v * 1e3, 2) for k, v in timers.items()})
print("indices equal:", np.array_equal(i_d.cpu().numpy(), i_o.numpy()), " max rel w:", float((w_d.cpu() - w_o).abs().max() / w_o.abs().max()))
rel = lambda a, b_: float((torch.as_tensor(a).double().cpu() - torch.as_tensor(b_).double().cpu()).abs().max() / max(float(torch.as_tensor(b_).double().abs().max()), 1e-300))
print("oracle trace keys:", sorted(tr_o.keys()), "| device trace keys:", sorted(tr_d.keys()))
if "gram" in tr_d and "psd" in tr_o:
    print("psd branch (oracle):", tr_o["psd"].get("branch"), "jitter rounds", tr_o["psd"].get("n_jitter"))
    G = torch.as_tensor(tr_d["gram"]).double().cpu()
    C = torch.sqrt(torch.nan_to_num(G) * torch.nan_to_num(G).T)
    ev = torch.linalg.eigvalsh(0.5 * (C + C.T))
    print("|cov|: smallest eigenvalues", [float(v) for v in ev[:3]], "largest", float(ev[-1]), "diag max", float(C.diag().max()))
    for k in range(4):
        sh = 1e-5 * (2 ** k - 1)
        try:
            torch.linalg.cholesky(C + sh * torch.eye(C.shape[0], dtype=C.dtype)); okc = True
        except Exception:  # noqa: BLE001
            okc = False
        print("   shift %.1e: LAPACK Cholesky %s" % (sh, "succeeds" if okc else "fails"))
for k in ("gram", "U"):
    if k in tr_d and k in tr_o:
        a, b_ = tr_d[k], tr_o[k]
        if k == "U":   # subspaces: compare projectors
            a, b_ = torch.as_tensor(a).double().cpu(), torch.as_tensor(b_).double().cpu()
            Pa, Pb = a.T @ torch.linalg.pinv(a.T), b_.T @ torch.linalg.pinv(b_.T)
            print("U: projector difference", float((Pa - Pb).abs().max()))
        else:
            print(k, "max rel diff", rel(a, b_))
for l, (ld, lo) in enumerate(zip(tr_d.get("levels", []), tr_o.get("levels", []))):
    out = ["level %d" % l]
    for k in sorted(set(ld.keys()) & set(lo.keys())):
        try:
            a, b_ = torch.as_tensor(ld[k]), torch.as_tensor(lo[k])
            if a.shape != b_.shape: out.append("%s: shapes %s %s" % (k, tuple(a.shape), tuple(b_.shape))); continue
            if a.dtype in (torch.int64, torch.int32): out.append("%s: %s" % (k, "equal" if torch.equal(a.cpu().long(), b_.long()) else "DIFFER"))
            else: out.append("%s: %.1e" % (k, rel(a, b_)))
        except Exception as e:  # noqa: BLE001
            out.append("%s: ? (%s)" % (k, type(e).__name__))
    print("  ".join(out))

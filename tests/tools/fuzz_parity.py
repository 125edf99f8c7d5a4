"""Random small recombinations on the device against the oracle (same seeded inputs): batch sizes across the size
instantiations of the Caratheodory kernels, leftovers of every kind, both continuous kernels, with and without the
posterior correction.  python tests/tools/fuzz_parity.py [n_cases=60] [seed=0]"""
import os, sys, warnings, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sober_amd
from oracle import sober_oracle as O
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
only = set(int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v)     # replay these cases only (+ diagnostics)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    b = int(rng.choice([5, 8, 9, 16, 17, 24, 32, 33, 40, 56, 57, 64, 65, 80, 100, 112, 120, 150]))
    d = int(rng.integers(2, 13))
    M = int(rng.integers(b + 8, max(b + 9, 4 * b)))
    N = int(rng.integers(max(2 * b + 1, M + 1), 40 * b + 50))
    n_obs = int(rng.integers(8, 60))
    kind = [O.RBF, O.MATERN52][int(rng.integers(0, 2))] if hasattr(O, "MATERN52") else O.RBF
    tani = os.environ.get("FUZZ_KIND") == "tanimoto"
    if tani:                                                 # fingerprints: 0/1 rows of 64..2048 bits
        kind, d = O.TANIMOTO, int(rng.choice([64, 100, 512, 1000, 2048]))
        b = min(b, 64)                                       # (a pool of random fingerprints has no more structure than that)
        M = int(rng.integers(b + 8, max(b + 9, 4 * b))); N = int(rng.integers(max(2 * b + 1, M + 1), 40 * b + 50))
    mode = ["predictive_covariance", "kernel", "weighted_predictive_covariance"][int(rng.integers(0, 3))]
    use_obj = bool(rng.random() < 0.25) and b <= 100
    obj = (lambda Z: (Z ** 2).sum(1)) if use_obj else None
    X = rng.random((N, d)); Xo = rng.random((n_obs, d)); mu0 = rng.random(N); mu0 /= mu0.sum()
    if tani:
        pbit = float(rng.choice([0.03, 0.1, 0.3]))
        X = (X < pbit).astype(np.float64); Xo = (Xo < pbit).astype(np.float64)
    Xn = X[rng.permutation(N)[:M]].copy()
    spec = O.make_spec(kind, t(Xo), t((0.25 + 0.5 * rng.random(d)) * np.sqrt(d)), outputscale=float(0.5 + 2 * rng.random()),
                       noise=1e-2, y_obs=t(rng.standard_normal(n_obs)))
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                              spec.mean_const, spec.alpha)
    seed = int(rng.integers(0, 1 << 30))
    mu_ref = t(mu0.copy())
    if only and case not in only:
        continue

    def well_posed():
        """How much the REFERENCE's result moves when candidates and Nystrom points move by one ulp (None: its indices
        change), and the spectrum of the Gram matrix the Nystrom functions come from."""
        torch.manual_seed(seed)
        i0, w0 = O.recombination(t(X), t(Xn), b, O.Kernel(spec, mode), init_weights=t(mu0.copy()), calc_obj=obj)
        torch.manual_seed(seed)
        i1, w1 = O.recombination(t(np.nextafter(X, 2.0)), t(np.nextafter(Xn, 2.0)), b, O.Kernel(spec, mode), init_weights=t(mu0.copy()),
                                 calc_obj=obj)
        eq = np.array_equal(i0.numpy(), i1.numpy())
        G = O.Kernel(spec, mode)(t(Xn), t(Xn))
        ev = torch.linalg.eigvalsh(0.5 * (G + G.T)).flip(0)
        move = float((w0 - w1).abs().max() / w0.abs().max()) if eq else None
        print("  the oracle against itself with candidates and Nystrom points one ulp up: idx equal %s, max rel w %s; Gram eigenvalues: largest %.2e, "
              "number b-1 = %.2e, smallest %.2e" % (eq, "%.1e" % move if eq else "-", float(ev[0]), float(ev[b - 2]), float(ev[-1])))
        return move

    try:
        torch.manual_seed(seed)
        idx_ref, w_ref = O.recombination(t(X), t(Xn), b, O.Kernel(spec, mode), init_weights=mu_ref, calc_obj=obj)
        mu = t(mu0.copy()).to(dev)
        torch.manual_seed(seed)
        idx, w = sober_amd.recombination(t(X).to(dev), t(Xn).to(dev), b, sober_amd.Kernel(ks, mode), init_weights=mu, calc_obj=obj)
        same = np.array_equal(idx.cpu().numpy(), idx_ref.numpy())
        relw = float((w.cpu() - w_ref).abs().max() / w_ref.abs().max()) if same else float("nan")
        ok = same and relw < 1e-6
        verdict = "ok "
        if not ok or only:
            # beyond the sweep's own bar (1e-6; the contract's is 1e-4): is the case well-posed in the reference at all?
            move = well_posed()
            if move is None:
                ok, verdict = True, "ill-posed (the reference's own indices change under a one-ulp move of its inputs)"
            elif move > 1e-5:
                ok, verdict = True, "ill-posed (the reference's own weights move by %.1e under a one-ulp move of its inputs)" % move
            elif same and relw <= max(1e-6, 20 * move):
                ok, verdict = True, "ok (the reference's own weights move by %.1e under a one-ulp move)" % move
            else:
                verdict = "BAD"
    except Exception as e:                                   # noqa: BLE001
        ok, same, relw, verdict = False, False, float("nan"), "BAD"
        print("  exception:", type(e).__name__, str(e)[:200])
    bad += 0 if ok else 1
    print("case %2d b=%3d N=%5d M=%3d d=%2d n_obs=%2d %s %s%s: %s idx_equal=%s max rel w %.1e" %
          (case, b, N, M, d, n_obs, spec.kind, mode, " +obj" if use_obj else "", verdict, same, relw), flush=True)
print("bad cases:", bad, "of", n_cases)

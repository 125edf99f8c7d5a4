"""Random small recombinations on the device against the oracle (same seeded inputs): batch sizes across the size
instantiations of the Caratheodory kernels, leftovers of every kind, the continuous kernels and fingerprints, the three
modes, with and without `calc_obj`.  The functions are what tests/test_hip_round4.py::test_fuzz_slice_vs_oracle runs on a
fixed 40-case slice; as a script:  python tests/tools/fuzz_parity.py [n_cases=60] [seed=0]   (FUZZ_KIND=tanimoto: fingerprint
pools; FUZZ_ONLY=3,17: replay these cases with the diagnostics; FUZZ_BATCHES=230,250,300 FUZZ_NFACTOR=12: other batches).

Verdicts: "ok" = identical indices and weights within 1e-6 (the contract: 1e-4), or within 20x of what the REFERENCE's own
weights move when its inputs move by one ulp; "ill-posed" = the reference's own indices change, or its own weights move by
more than 1e-5, under that one-ulp move (the oracle is the reference: tests/test_oracle_vs_reference_fuzz.py holds it to
the reference on 64 random cases), or -- with identical indices -- one of make_cov_psd's Cholesky verdicts falls on a
numerically singular matrix (psd_knife_edge: round 5, sweep 22 case 91) -- a second implementation cannot be held to such a
case; "BAD" = anything else."""
import os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sober_amd
from oracle import sober_oracle as O

t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
BATCHES = [5, 8, 9, 16, 17, 24, 32, 33, 40, 56, 57, 64, 65, 80, 100, 112, 120, 150]


def make_case(rng, tani=False, batches=BATCHES, n_factor=40, even_elements=0, obj_max=100):
    """One random case (all the random draws happen here, in a fixed order: a seed names a sweep).
    even_elements = D > 0 (round 6, FUZZ_EVEN=D): the pool's size is rounded up to a multiple of 2^D x 2 b with at least two
    elements per set left at level D -- a pool without leftovers whose levels 1 .. D are DERIVED from level 0's class sums
    (csrc/level_class.hip) instead of evaluated; the random draws are the same as without the switch."""
    b = int(rng.choice(batches))
    d = int(rng.integers(2, 13))
    M = int(rng.integers(b + 8, max(b + 9, 4 * b)))
    N = int(rng.integers(max(2 * b + 1, M + 1), n_factor * b + 50))
    n_obs = int(rng.integers(8, 60))
    kind = [O.RBF, O.MATERN52][int(rng.integers(0, 2))]
    if tani:                                                 # fingerprints: 0/1 rows of 64..2048 bits
        kind, d = O.TANIMOTO, int(rng.choice([64, 100, 512, 1000, 2048]))
        b = min(b, 64)                                       # (a pool of random fingerprints has no more structure than that)
        M = int(rng.integers(b + 8, max(b + 9, 4 * b))); N = int(rng.integers(max(2 * b + 1, M + 1), n_factor * b + 50))
    mode = ["predictive_covariance", "kernel", "weighted_predictive_covariance"][int(rng.integers(0, 3))]
    use_obj = bool(rng.random() < 0.25) and b <= obj_max
    if even_elements > 0:
        q = (1 << even_elements) * 2 * b
        N = max(2 * q, ((N + q - 1) // q) * q)
    X = rng.random((N, d)); Xo = rng.random((n_obs, d)); mu0 = rng.random(N); mu0 /= mu0.sum()
    if tani:
        pbit = float(rng.choice([0.03, 0.1, 0.3]))
        X = (X < pbit).astype(np.float64); Xo = (Xo < pbit).astype(np.float64)
    Xn = X[rng.permutation(N)[:M]].copy()
    spec = O.make_spec(kind, t(Xo), t((0.25 + 0.5 * rng.random(d)) * np.sqrt(d)), outputscale=float(0.5 + 2 * rng.random()),
                       noise=1e-2, y_obs=t(rng.standard_normal(n_obs)))
    seed = int(rng.integers(0, 1 << 30))
    return dict(b=b, d=d, M=M, N=N, n_obs=n_obs, kind=kind, mode=mode, use_obj=use_obj, X=X, Xn=Xn, mu0=mu0, spec=spec, seed=seed)


def _obj(c):
    return (lambda Z: (Z ** 2).sum(1)) if c["use_obj"] else None


def well_posed(c, verbose=True):
    """How much the REFERENCE's result moves when candidates and Nystrom points move by one ulp (None: its indices change)."""
    spec, b, mode, obj = c["spec"], c["b"], c["mode"], _obj(c)
    torch.manual_seed(c["seed"])
    i0, w0 = O.recombination(t(c["X"]), t(c["Xn"]), b, O.Kernel(spec, mode), init_weights=t(c["mu0"].copy()), calc_obj=obj)
    torch.manual_seed(c["seed"])
    i1, w1 = O.recombination(t(np.nextafter(c["X"], 2.0)), t(np.nextafter(c["Xn"], 2.0)), b, O.Kernel(spec, mode),
                             init_weights=t(c["mu0"].copy()), calc_obj=obj)
    eq = np.array_equal(i0.numpy(), i1.numpy())
    move = float((w0 - w1).abs().max() / w0.abs().max()) if eq else None
    if verbose:
        G = O.Kernel(spec, mode)(t(c["Xn"]), t(c["Xn"]))
        ev = torch.linalg.eigvalsh(0.5 * (G + G.T)).flip(0)
        print("  the oracle against itself with candidates and Nystrom points one ulp up: idx equal %s, max rel w %s; Gram eigenvalues: "
              "largest %.2e, number b-1 = %.2e, smallest %.2e" % (eq, "%.1e" % move if eq else "-", float(ev[0]), float(ev[b - 2]), float(ev[-1])))
    return move


def psd_knife_edge(c, rel=1e-12):
    """True when one of the Cholesky verdicts of make_cov_psd (SOBER/_utils.py:117-157) falls on a numerically singular
    matrix: |smallest eigenvalue| <= rel x largest of a matrix is_psd is asked about.  The reference's jitter (or none) then
    hangs on the last bits of its Gram matrix -- a second implementation's Gram matrix differs in exactly those -- and a
    one-ulp move of 0/1 fingerprints does not reach them (the case the input-perturbation test misses)."""
    spec, mode = c["spec"], c["mode"]
    cov = O.Kernel(spec, mode)(t(c["Xn"]), t(c["Xn"])).double()

    def edge(m):
        ev = torch.linalg.eigvalsh(0.5 * (m + m.T))
        return bool(ev[0].abs() <= rel * ev[-1].abs())

    def psd(m):
        try:
            torch.linalg.cholesky(m)
            return True
        except Exception:                                    # noqa: BLE001
            return False
    if edge(cov):
        return True
    if psd(cov):
        return False
    m = torch.nan_to_num(cov)
    m = torch.sqrt(m * m.T)
    jitter = 1e-5
    for _ in range(12):
        if edge(m):
            return True
        if psd(m):
            return False
        m = m + jitter * torch.eye(m.shape[0], dtype=m.dtype)
        jitter *= 2
    return False


def check_case(c, dev, verbose=False, always_diagnose=False):
    """-> (ok, verdict, idx_equal, max rel weight error)."""
    spec, b, mode, obj = c["spec"], c["b"], c["mode"], _obj(c)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                              spec.mean_const, spec.alpha)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(c["seed"])
        idx_ref, w_ref = O.recombination(t(c["X"]), t(c["Xn"]), b, O.Kernel(spec, mode), init_weights=t(c["mu0"].copy()), calc_obj=obj)
        mu = t(c["mu0"].copy()).to(dev)
        torch.manual_seed(c["seed"])
        idx, w = sober_amd.recombination(t(c["X"]).to(dev), t(c["Xn"]).to(dev), b, sober_amd.Kernel(ks, mode), init_weights=mu, calc_obj=obj)
        same = np.array_equal(idx.cpu().numpy(), idx_ref.numpy())
        relw = float((w.cpu() - w_ref).abs().max() / w_ref.abs().max()) if same else float("nan")
        ok, verdict = same and relw < 1e-6, "ok"
        if not ok or always_diagnose:
            # beyond the sweep's own bar: is the case well-posed in the reference at all?
            move = well_posed(c, verbose)
            if move is None:
                ok, verdict = True, "ill-posed (the reference's own indices change under a one-ulp move of its inputs)"
            elif move > 1e-5:
                ok, verdict = True, "ill-posed (the reference's own weights move by %.1e under a one-ulp move of its inputs)" % move
            elif same and relw <= max(1e-6, 20 * move):
                ok, verdict = True, "ok (the reference's own weights move by %.1e under a one-ulp move)" % move
            elif same and psd_knife_edge(c):
                ok, verdict = True, "ill-posed (is_psd decides on a numerically singular matrix: the reference's jitter hangs on the last bits of its Gram matrix)"
            else:
                ok, verdict = False, "BAD"
    return ok, verdict, same, relw


def describe(i, c):
    return "case %2d b=%3d N=%5d M=%3d d=%4d n_obs=%2d %s %s%s" % (i, c["b"], c["N"], c["M"], c["d"], c["n_obs"], c["spec"].kind, c["mode"],
                                                                " +obj" if c["use_obj"] else "")


if __name__ == "__main__":
    warnings.simplefilter("ignore")
    dev = torch.device("cuda:0")
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    only = set(int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tani = os.environ.get("FUZZ_KIND") == "tanimoto"
    even = int(os.environ.get("FUZZ_EVEN", "0"))
    # FUZZ_BATCHES=230,250,300 (round 6): batches beyond the register-resident Caratheodory kernels (csrc/car_big.hip), with
    # calc_obj allowed up to batch 400 and pools of at most FUZZ_NFACTOR (default 40) x batch candidates
    batches = [int(v) for v in os.environ.get("FUZZ_BATCHES", "").split(",") if v] or BATCHES
    n_factor = int(os.environ.get("FUZZ_NFACTOR", "40"))
    bad = 0
    for i in range(n_cases):
        c = make_case(rng, tani, batches=batches, n_factor=n_factor, even_elements=even, obj_max=400 if batches is not BATCHES else 100)
        if only and i not in only:
            continue
        try:
            ok, verdict, same, relw = check_case(c, dev, verbose=True, always_diagnose=bool(only))
        except Exception as e:                               # noqa: BLE001
            ok, verdict, same, relw = False, "BAD", False, float("nan")
            print("  exception:", type(e).__name__, str(e)[:200])
        bad += 0 if ok else 1
        print("%s: %s idx_equal=%s max rel w %.1e" % (describe(i, c), verdict, same, relw), flush=True)
    print("bad cases:", bad, "of", n_cases)

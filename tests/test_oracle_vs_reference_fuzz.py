"""The oracle against the REFERENCE ITSELF on seeded random cases (build container only: skipped where /root/reference
is absent, e.g. on the GPU box).  The reference has no tests of its own; the ~25 committed fixtures pin the oracle on
the shapes somebody thought of -- this sweep pins it on 64 it nobody looked at: kinds x modes x leftovers of every kind
x calc_obj x zero weights x ARD, through the same loader that wrote the fixtures (tests/golden/make_golden.py:
load_reference / run_reference -- SOBER/_rchq.py:5-270 with SOBER/_kernel.py, _gp.py, _utils.py, no stand-ins above the
gpytorch base kernel).  Same RNG seed in front of both calls (the randn draw of torch.svd_lowrank, _rchq.py:34-39).

Bar: identical indices, weights to 1e-12 relative (the two are the same operations in the same order on the same
machine; in practice they are the same bits).  This is what makes the device sweep's verdicts (tests/test_hip_parity.py::
test_fuzz_slice_vs_oracle, tests/tools/fuzz_parity.py) rest on the reference and not on the oracle alone."""
import os
import warnings

import numpy as np
import pytest
import torch

REF = "/root/reference/SOBER"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only in the build container")

from oracle import sober_oracle as O                                     # noqa: E402
from tests.golden.synth import synth, build_spec, calc_obj_fn, SEED_CALL  # noqa: E402


def _cases():
    rng = np.random.default_rng(20261003)
    kinds = [O.RBF, O.MATERN52, O.TANIMOTO]
    modes = ["predictive_covariance", "weighted_predictive_covariance", "kernel"]
    out = []
    for i in range(64):
        kind = kinds[i % 3] if i % 8 != 7 else O.TANIMOTO
        mode = modes[(i // 3) % 3]
        b = int(rng.integers(4, 25))
        # leftovers of every kind: R mod 2b in {0, 1, 2b-1, random}, pools below one level (straight to the final level),
        # pools of exactly one level
        shape = i % 6
        S = 2 * b
        E = int(rng.integers(2, 30))
        N = {0: E * S, 1: E * S + 1, 2: E * S + S - 1, 3: E * S + int(rng.integers(1, S)),
             4: int(rng.integers(b + 1, S + 1)), 5: S + int(rng.integers(1, S))}[shape]
        d = int(rng.integers(2, 9)) if kind != O.TANIMOTO else int(rng.choice([64, 96, 128, 200]))
        M = int(rng.integers(max(b + 2, 20), 90))
        M = min(M, N)
        case = dict(name=f"fuzz{i}", kind=kind, mode=mode, N=N, M=M, d=d, b=b, n_obs=int(rng.integers(5, 40)),
                    seed=1000 + i, ard=bool(rng.integers(0, 2)) and kind != O.TANIMOTO,
                    zero_frac=0.25 if i % 5 == 0 else 0.0, calc_obj=(i % 4 == 1),
                    outputscale=float(rng.uniform(0.5, 2.0)), mean_const=float(rng.uniform(0.2, 1.0)),
                    bit_p=float(rng.uniform(0.05, 0.3)))
        if M < b + 1:
            continue
        out.append(case)
    return out


CASES = _cases()


@pytest.fixture(scope="module")
def ref():
    from tests.golden import make_golden
    return make_golden.load_reference()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] + "_" + c["kind"] + "_" + c["mode"][:4] for c in CASES])
def test_oracle_equals_reference(ref, case):
    from tests.golden import make_golden
    inp = synth(case)
    try:
        rec = make_golden.run_reference(ref, case, inp, threads=1)
    except Exception as e:                       # the reference itself cannot run this input (e.g. a singular step)
        pytest.skip(f"reference raised {type(e).__name__}: {e}")
    spec = build_spec(case, inp)
    mu = torch.from_numpy(inp["mu0"].copy())
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, w = O.recombination(torch.from_numpy(inp["X_cand"].copy()), torch.from_numpy(inp["X_nys"].copy()), case["b"],
                                     O.Kernel(spec, case["mode"]), init_weights=mu,
                                     calc_obj=calc_obj_fn if case["calc_obj"] else None)
    finally:
        torch.set_num_threads(old)
    assert torch.equal(idx, rec["idx"]), (idx.tolist(), rec["idx"].tolist())
    assert torch.allclose(w, rec["w"], rtol=1e-12, atol=0.0), float(((w - rec["w"]).abs() / rec["w"].abs()).max())
    assert torch.allclose(mu, rec["mu_after"], rtol=1e-12, atol=0.0)          # Q3: init_weights is mutated in place

"""The identity behind csrc/level_class.hip (round 6), on the CPU against the oracle's own level traces.

Survivors are compacted element-major (SOBER/_rchq.py:198-221): with b of the 2b sets kept and no leftovers, the survivor of
element e in the kept set of rank k becomes element e div 2, set (e mod 2) b + k of the next level with weight
mu w*_k / tot_k.  So the next level's set sums are this level's sums over the elements of one parity, scaled -- and class sums
over e mod 2^D at level 0 give levels 1 .. D without evaluating the kernel again.  The numpy restatement below follows
k_class_sum / k_class_derive line for line; the ORACLE (reference-shaped: it multiplies every survivor's weight and sums
afresh) is the checker."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import sober_oracle as O
from tests.golden.synth import SEED_CALL, build_spec, load_case, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def class_sum(K, mu, idx, S, CL):
    """level 0: Gc[row, c S + s] = sum over elements e = c (mod CL) of K[row, idx[e S + s]] mu[idx[e S + s]]; totc alike."""
    E = len(idx) // S
    pos = idx.reshape(E, S)
    Gc = np.zeros((K.shape[0], CL * S))
    totc = np.zeros(CL * S)
    for c in range(CL):
        sel = pos[c::CL]                                            # elements of class c
        Gc[:, c * S:(c + 1) * S] = (K[:, sel] * mu[sel][None]).sum(1)
        totc[c * S:(c + 1) * S] = mu[sel].sum(0)
    return Gc, totc


def fold(Gc, totc, S, CL):
    return sum(Gc[:, c * S:(c + 1) * S] for c in range(CL)), sum(totc[c * S:(c + 1) * S] for c in range(CL))


def derive(Gc, totc, S, CL, idx_star, w_star, tot):
    """k_class_derive: Gn[row, c' S + par b + k] = (w*_k / tot_{s_k}) Gc[row, (2 c' + par) S + s_k]."""
    b, CN = S // 2, CL // 2
    scale = w_star / tot[idx_star]
    Gn, totn = np.zeros((Gc.shape[0], CN * S)), np.zeros(CN * S)
    for c in range(CN):
        for par in (0, 1):
            src = (2 * c + par) * S + idx_star
            dst = c * S + par * b + np.arange(b)
            Gn[:, dst] = Gc[:, src] * scale[None]
            totn[dst] = totc[src] * scale
    return Gn, totn


def oracle_levels(inp, spec, case):
    trace = []
    kern = O.Kernel(spec, case["mode"])
    Xc, Xn = _t(inp["X_cand"]), _t(inp["X_nys"])
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U = O.ker_svd_sparsify(Xn, case["b"] - 1, kern)[1]
        O.mod_tchernychova_lyons(Xc, U, Xn, kern, mu=_t(inp["mu0"].copy()), trace=trace)
        K = kern(Xn, Xc).numpy()
    return trace, U.numpy(), K


def check_case(inp, spec, case, min_levels):
    S = 2 * case["b"]
    trace, U, K = oracle_levels(inp, spec, case)
    mu0 = inp["mu0"]
    idx0 = np.flatnonzero(mu0 != 0)
    R = len(idx0)
    D = 0
    while D < 3 and R % (S << (D + 1)) == 0 and R // (S << (D + 1)) >= 2:
        D += 1
    assert D >= min_levels, (R, S, D)
    CL = 1 << D
    Gc, totc = class_sum(K, mu0, idx0, S, CL)
    checked = 0
    for l in range(D + 1):
        lv = trace[l]
        assert lv["kind"] == "level" and lv["r"] == 0
        G, tot = fold(Gc, totc, S, CL >> l)
        X_tmp = (U @ G / tot[None]).T
        np.testing.assert_allclose(tot, lv["tot_weights"].numpy(), rtol=1e-13)
        np.testing.assert_allclose(X_tmp, lv["X_tmp"].numpy(), rtol=1e-9, atol=1e-12)
        checked += 1
        if l == D:
            break
        idx_star, w_star = lv["idx_star"].numpy(), lv["w_star"].numpy()
        if len(idx_star) != S // 2:                                  # (the device stops the chain here: need_keep)
            break
        Gc, totc = derive(Gc, totc, S, CL >> l, idx_star, w_star, tot)
    return checked


def test_derived_levels_equal_the_oracles_evaluated_ones_depth_3():
    """N = 2^3 x 10 elements of S = 40: levels 1, 2 and 3 of the ORACLE (kernel evaluated on the updated weights) against
    gather-and-scale of level 0's sums by e mod 8."""
    case = dict(kind=O.RBF, mode="predictive_covariance", N=40 * 80, M=64, d=3, b=20, n_obs=30, seed=5)
    inp = synth(case)
    assert check_case(inp, build_spec(case, inp), case, 3) == 4


def test_derived_levels_weighted_mode_and_matern():
    case = dict(kind=O.MATERN52, mode="weighted_predictive_covariance", N=24 * 36, M=40, d=4, b=12, n_obs=25, seed=8, mean_const=0.2)
    inp = synth(case)
    assert check_case(inp, build_spec(case, inp), case, 2) >= 3


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "recomb_*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_identity_on_the_reference_goldens_with_even_element_counts(path):
    """Every golden whose first level has no leftovers and an even number of elements (the reference's own inputs): the
    derived levels reproduce the oracle's traces -- which tests/test_oracle_golden.py pins to the reference's."""
    case, inp, spec, z = load_case(path)
    if case["calc_obj"] or case["N"] > 20000:
        pytest.skip("acquisition-guided branch / full-size pool: not this identity's business on the CPU")
    R, S = int(np.count_nonzero(inp["mu0"])), 2 * case["b"]
    if R % (2 * S) != 0 or R // (2 * S) < 2:
        pytest.skip(f"R = {R}, S = {S}: leftovers or an odd element count at level 0")
    assert check_case(inp, spec, case, 1) >= 2

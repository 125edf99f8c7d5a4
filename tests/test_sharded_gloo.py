"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups drive the product's engine
(position sharding, one all-reduce per level, closed-form compaction, replicated Caratheodory step,
final all-gather) through the CPU test double of the device ops.  Every rank must return the
unsharded result: identical indices, weights to 1e-9, and its own shard of the mutated weights."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, cuts, outq):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sober_amd
        from tests._oracle_ops import OracleOps
        from tests.golden.synth import SEED_CALL, calc_obj_fn, load_case
        torch.set_num_threads(1)
        case, inp, spec, z = load_case(os.path.join(GOLD, f"recomb_{name}.npz"))
        ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                                  spec.noise, spec.mean_const, spec.alpha)
        lo, hi = cuts[rank], cuts[rank + 1]
        X = torch.from_numpy(inp["X_cand"][lo:hi].copy())
        mu = torch.from_numpy(inp["mu0"][lo:hi].copy())
        torch.manual_seed(SEED_CALL + 17 * rank)          # ranks deliberately disagree: U is broadcast
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if rank == 0:
                torch.manual_seed(SEED_CALL)
            idx, w = sober_amd.recombination(X, torch.from_numpy(inp["X_nys"]), case["b"],
                                             sober_amd.Kernel(ks, case["mode"]), init_weights=mu,
                                             calc_obj=calc_obj_fn if case["calc_obj"] else None,
                                             group=dist.group.WORLD, row_offset=lo, _ops=OracleOps())
        outq.put((rank, idx.numpy(), w.numpy(), mu.numpy()))
    finally:
        dist.destroy_process_group()


def _run(name, cuts):
    world = len(cuts) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return outs


@pytest.mark.parametrize("name,cuts", [
    ("cfg1_rbf_ard", [0, 1000, 2000]),            # even split
    ("rbf_b30", [0, 700, 3000]),                  # uneven: ranges never align with the 2b sets
    ("rbf_zero_weights", [0, 1300, 2500]),        # mu == 0 entries drop out of the list per rank
    ("matern_b20", [0, 900, 1800, 3000]),         # three ranks
    ("rbf_calc_obj", [0, 800, 2000]),             # acquisition-guided branch, sharded
])
def test_sharded_equals_unsharded(name, cuts):
    z = np.load(os.path.join(GOLD, f"recomb_{name}.npz"))
    outs = _run(name, cuts)
    mu_all = np.concatenate([o[3] for o in outs])
    for rank, idx, w, _ in outs:
        assert np.array_equal(idx, z["idx"]), (name, rank)           # global indices, every rank
        np.testing.assert_allclose(w, z["w"], rtol=1e-9)
    nz = np.flatnonzero(mu_all)
    assert np.array_equal(nz, z["mu_after_idx"])                      # Q3 across the shards
    np.testing.assert_allclose(mu_all[nz], z["mu_after_val"], rtol=1e-9)


def test_rank_with_no_live_candidates():
    """A rank whose whole shard has zero weight contributes nothing and must not dead-lock."""
    z = np.load(os.path.join(GOLD, "recomb_rbf_noleft.npz"))
    outs = _run("rbf_noleft", [0, 2560, 2560])
    for rank, idx, w, _ in outs:
        assert np.array_equal(idx, z["idx"])
        np.testing.assert_allclose(w, z["w"], rtol=1e-9)

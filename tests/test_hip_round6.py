"""GPU tests added in round 6 (run with -m gpu on an MI355X), all through the C ABI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sober_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need the MI355X"
    from sober_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _json_line(stdout):
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2` with NO torch.distributed.run in front and no WORLD_SIZE in the environment -- the form the
    driver uses for N = 1 -- used to die on `assert world == args.gpus`.  It now starts the two ranks as a child
    torch.distributed.run (gloo hook: both ranks on the one GPU), forwards rank 0's JSON line and returns 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SOBER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SOBER_PEER_ALLREDUCE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--check-unsharded"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and "row-sharded x2" in d["config"]["parallelism"]
    chk = d["parity_sharded_vs_unsharded"]
    assert chk["idx_equal_unsharded"] and chk["ranks_agree"], chk


def test_bench_default_line_carries_the_other_configs_and_the_acquisition_step(dev):
    """The ONE command the driver runs (`python bench.py`, here with few steps and without the CPU leg) prints ONE JSON line
    whose `other_configs` holds BASELINE.json configurations 1, 3, 4 and 5 -- each with ms_per_step, the dominant kernel's
    roofline fraction and a parity verdict -- and whose `acquisition_step` is `Sober.next_batch` at configuration 2's shapes."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep",
           "--other-steps", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["parity"]["idx_equal_reference"] and d["parity"]["max_rel_w_vs_reference"] < 1e-7
    assert sorted(d["other_configs"]) == ["1", "3", "4", "5"]
    for c, r in d["other_configs"].items():
        assert "error" not in r, (c, r)
        assert r["ms_per_step"] > 0 and 0 < r["roofline"]["frac"] < 1, (c, r)
        p = r["parity"]
        if "idx_equal_reference" in p:
            assert p["idx_equal_reference"] and p["max_rel_w_vs_reference"] < 1e-7, (c, p)
        else:
            assert p["by"] == "invariants" and all(v for k, v in p.items() if isinstance(v, bool)), (c, p)
            assert p["mass_error"] < 1e-12, (c, p)
    a = d["acquisition_step"]
    assert "error" not in a, a
    assert a["ms_per_step"] > 0 and a["parity"]["reference_fixture_sober_next_batch_equal"] is True, a

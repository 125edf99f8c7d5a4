"""Host-side pieces of the library that need no GPU (the C-ABI library loads on a CPU-only machine)."""
import numpy as np


def test_mt19937_uniforms_and_state_follow_torch_randn():
    """csrc/host_rng.cpp against the installed torch: the uniforms of torch.rand(float64), and the generator state
    torch.randn(float64) leaves behind -- multiples of 16 and ragged sizes, starting anywhere inside a twister block."""
    import torch
    from sober_amd import _native as nat, _rng
    g = torch.Generator()
    for seed, numel, skip in ((0, 500 * 99, 0), (1, 16, 1), (2, 17, 623), (3, 624 * 3 + 1, 624), (4, 100000, 77)):
        g.manual_seed(seed)
        if skip:
            torch.rand(skip, generator=g)
        s0 = g.get_state()
        mine, u = s0.clone(), torch.empty(numel + 16, dtype=torch.float64)
        nat.mt19937_uniform53(mine, numel, u)
        assert torch.equal(u[:numel], torch.rand(numel, dtype=torch.float64, generator=g))
        g.set_state(s0)
        want = torch.randn(numel, dtype=torch.float64, generator=g)
        assert torch.equal(g.get_state(), mine)
        # Box-Muller of those uniforms on the host (numpy's libm may differ from torch's in the last place)
        un = u.numpy()

        def bm(v):
            r, th = np.sqrt(-2.0 * np.log(1.0 - v[:8])), 2.0 * np.pi * v[8:16]
            return np.concatenate([r * np.cos(th), r * np.sin(th)])
        out = np.empty(numel)
        for k in range(numel // 16):
            out[16 * k:16 * k + 16] = bm(un[16 * k:16 * k + 16])
        if numel % 16:
            out[numel - 16:] = bm(un[numel:numel + 16])
        np.testing.assert_allclose(out, want.numpy(), rtol=0, atol=4e-15)
    assert _rng._self_check()


def test_bench_self_launch_builds_a_child_torchrun(monkeypatch):
    """bench.py --gpus N (N > 1) without a launcher: the ranks are started as a CHILD `python -m torch.distributed.run` on the
    loopback interface (never an exec: the parent may not replace itself once anything touched the GPU, and it has touched
    nothing); with WORLD_SIZE set, or N = 1, nothing is started."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    class _R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench._self_launch() == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in __import__("os").environ
    seen.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus=2"])
    assert bench._self_launch() == 7 and "--nproc-per-node=2" in seen["cmd"]
    seen.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    assert bench._self_launch() is None and not seen
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench._self_launch() is None and not seen

"""torch.randn(M, q, dtype=float64) of torch.svd_lowrank (SOBER/_rchq.py:37) split between host and device.

The reference draws the range finder's test matrix from the global CPU generator; the drop-in has to leave that
generator exactly where the reference leaves it and use the same draw.  torch.randn itself is 0.6 ms of host time at
(500, 99) -- longer than the Cholesky probes it is meant to hide behind -- and ~70 % of that is libm's log / cos / sin.
Here the host only steps the Mersenne twister (csrc/host_rng.cpp: same uniforms, same final generator state, bit for
bit) and the Box-Muller transform runs on the device (csrc/misc.hip:k_box_muller; normals equal to torch.randn's to an
ulp or two).

`device_randn` checks once per process that the installed torch still draws the way host_rng.cpp assumes -- the
uniforms against torch.rand on a private generator, the end state against torch.randn's -- and otherwise (or for a
generator state of another layout) falls back to torch.randn + a copy: slower, same numbers.
"""
import os

import torch

from . import _native as nat

_STATE_BYTES = 5056          # torch.get_rng_state() of the CPU generator (mt19937 + legacy normal cache)
_verified = None
_pin = {}


def _self_check() -> bool:
    try:
        g = torch.Generator()
        for seed, numel, skip in ((1234, 16 * 41, 3), (99, 16 * 40 + 5, 700)):
            g.manual_seed(seed)
            torch.rand(skip, generator=g)                      # start inside a block of the twister
            s0 = g.get_state()
            if s0.numel() != _STATE_BYTES:
                return False
            mine, u = s0.clone(), torch.empty(numel + 16, dtype=torch.float64)
            nat.mt19937_uniform53(mine, numel, u)
            want_u = torch.rand(numel, dtype=torch.float64, generator=g)
            if not torch.equal(u[:numel], want_u):
                return False
            g.set_state(s0)
            torch.randn(numel, dtype=torch.float64, generator=g)
            if not torch.equal(g.get_state(), mine):
                return False
        return True
    except Exception:
        return False


def device_randn(M: int, q: int, device):
    """The (M, q) float64 draw of torch.randn on the global CPU generator, delivered on `device`; the generator ends
    where torch.randn leaves it.  The device work is enqueued on the current stream."""
    global _verified
    numel = M * q
    if _verified is None:
        _verified = _self_check()
    state = torch.get_rng_state() if _verified and numel >= 16 else None
    if state is None or state.numel() != _STATE_BYTES:
        return torch.randn(M, q, dtype=torch.float64).to(device)
    device = torch.device(device)
    u = _pin.get((numel, device))
    if u is None:
        u = _pin[(numel, device)] = torch.empty(numel + 16, dtype=torch.float64, pin_memory=True)
    else:
        # the previous copy out of this staging buffer must have left before it is overwritten
        ev = _pin.get(("ev", numel, device))
        if ev is not None:
            ev.synchronize()
    nat.mt19937_uniform53(state, numel, u)
    torch.set_rng_state(state)
    # the upload travels on a stream of its own: on the caller's stream it would queue behind whatever is running
    # there (the Cholesky probes of the Nystrom chain) and the 0.4 MB would cross the bus only afterwards
    main = torch.cuda.current_stream(device)
    side = _pin.get(("stream", device))
    if side is None:
        side = _pin[("stream", device)] = torch.cuda.Stream(device)
    ud = _pin.get(("dev", numel, device))
    if ud is None:
        ud = _pin[("dev", numel, device)] = torch.empty(numel + 16, dtype=torch.float64, device=device)
    used = _pin.get(("used", numel, device))
    if used is not None:
        side.wait_event(used)                                  # (the previous Box-Muller pass has read `ud`)
    with torch.cuda.stream(side):
        ud.copy_(u, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    _pin[("ev", numel, device)] = ev
    main.wait_event(ev)
    R = torch.empty(M, q, dtype=torch.float64, device=device)
    nat.box_muller(ud, numel, R)
    used = torch.cuda.Event()
    used.record(main)
    _pin[("used", numel, device)] = used
    return R

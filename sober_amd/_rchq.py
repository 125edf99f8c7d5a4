"""`recombination` with the reference's signature (SOBER/_rchq.py:5-31) on the MI355X.

    idx_star, w_star = recombination(pts_rec, pts_nys, num_pts, kernel, device, dtype,
                                     init_weights=None, calc_obj=None)

`calc_obj(samp) -> (N,)` (the acquisition-guided branch, :67-69) is supported: the extra objective row is summed on
the device and both eliminations of a level run there too (Caratheodory step with one more function, k_second_elim).
`kernel` is a `sober_amd.Kernel` (RBF / Matern-5/2 / Tanimoto posterior covariance, weighted or
raw): the whole step then runs on the fused HIP path -- the (E, M, S) kernel tensor of
SOBER/_rchq.py:124 is never materialised.  As in the reference, `device`/`dtype` are accepted and
ignored (the reference builds a SafeTensorOperator from the module globals, :30), `init_weights`
is modified in place (Q3), and the result depends on the global CPU RNG state through
`torch.svd_lowrank` (seed it before the call for repeatability).
"""
from __future__ import annotations

import torch

from ._engine import DistComm, RecombinationEngine, SoloComm
from ._kernel import MODES, Kernel
from ._settings import setting_parameters


_OPS = {}


def _default_ops(dev, fused, HipOps, CallableKernelOps):
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev, bool(fused))
    ops = _OPS.get(key)
    if ops is None:
        ops = _OPS[key] = HipOps(dev) if fused else CallableKernelOps(dev)
    return ops


def _device_of(t, fallback):
    return t.device if t.is_cuda else fallback


def recombination(pts_rec, pts_nys, num_pts, kernel, device=None, dtype=None, init_weights=None,
                  calc_obj=None, *, group=None, row_offset=0, _ops=None, _trace=None, _timers=None):
    """Kernel recombination of the weighted empirical measure (pts_rec, init_weights) down to at
    most `num_pts` points with positive weights that preserve the integrals of num_pts-1 Nystrom
    test functions built on `pts_nys`.  Returns (idx_star int64, w_star float64).

    Keyword-only extensions (not in the reference):
      group       a torch.distributed group: `pts_rec`/`init_weights` are then this rank's ROW SHARD
                  of the pool (global row index = row_offset + local index) and every rank returns
                  the same global (idx_star, w_star); one small all-reduce per level.
    """
    fused = isinstance(kernel, Kernel)
    if not fused and not callable(kernel):
        raise TypeError(f"kernel must be a sober_amd.Kernel or a callable kernel(x, y); got {type(kernel).__name__}")
    if fused and kernel.mode not in MODES:
        raise ValueError('mode should be from ["predictive_covariance", '
                         '"weighted_predictive_covariance", "kernel"]')
    if _ops is None:
        from . import _native as nat
        from ._ops_hip import CallableKernelOps, HipOps
        glob_dev, _ = setting_parameters()
        dev = _device_of(pts_rec, glob_dev)
        if fused and not nat.fused_dim_supported(nat.KIND_BY_NAME[kernel.spec(dev).kind], pts_rec.shape[1]):
            # beyond the fused level kernels' tile set (continuous inputs with d > 32, fingerprints with more than
            # 2048 bits): the reference accepts any dimension, so the same Kernel is used as a plain callable -- its
            # matrix K(X_cand, X_nys) is built once by the any-length pairwise kernel and stays in HBM
            fused = False
        # raises when there is no HIP device / library.  A sober_amd.Kernel takes the fused path; any other
        # callable (the reference's kernel protocol, e.g. BASQ's gspace_kernel) is evaluated by the caller's
        # own torch code and only the kernel matrix itself is outside the HIP path
        # one backend object per device and kind, kept across calls: pinned staging buffers, event pools and the
        # bit-packed form of a fingerprint pool that comes back unchanged at the next BO iteration
        _ops = _default_ops(dev, fused, HipOps, CallableKernelOps)
    dev = _ops.device

    N = pts_rec.shape[0]
    X_cand = pts_rec.detach().to(dev, torch.float64)
    X_nys = pts_nys.detach().to(dev, torch.float64)
    caller_mu = init_weights
    if init_weights is None:
        mu = torch.ones(N, dtype=torch.float64, device=dev) / N          # :60-61
    elif init_weights.device == dev and init_weights.dtype == torch.float64 and init_weights.is_contiguous():
        mu = init_weights                                                 # mutated in place (Q3)
    else:
        mu = init_weights.detach().to(dev, torch.float64).contiguous()

    comm = DistComm(group) if group is not None else SoloComm()
    eng = RecombinationEngine(_ops, comm, row_offset=row_offset)
    eng.trace = _trace
    plan = _ops.build_plan(kernel.spec(dev), kernel.mode, X_nys, X_cand, pool_owner=pts_rec) if fused else \
        _ops.build_plan(kernel, "callable", X_nys, X_cand, pool_owner=pts_rec)
    obj = None
    if calc_obj is not None:                                              # :67-69, once on all candidates
        obj = (-1 * calc_obj(X_cand)).detach().to(dev, torch.float64).reshape(-1).contiguous()
    idx_star, w_star = eng.run(plan, mu, int(num_pts), obj)
    if _timers is not None:
        for k, v in eng.timers.items():
            _timers[k] = _timers.get(k, 0.0) + v

    if caller_mu is not None and mu is not caller_mu:
        caller_mu.copy_(mu.to(caller_mu.device, caller_mu.dtype))         # keep Q3 for foreign tensors
    return idx_star, w_star


def rc_kernel_svd(samp, pt, s, kernel, tm=None, mu=None, calc_obj=None):
    """SOBER/_rchq.py:42-48."""
    return recombination(samp, pt, s, kernel, init_weights=mu, calc_obj=calc_obj)

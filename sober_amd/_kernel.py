"""`Kernel(model, mode)` with the reference's interface (SOBER/_kernel.py:4-47) on top of
the HIP library.

The reference's kernel callable hides a gpytorch model; the hot path touches only
``model.covar_module`` (RBF / Matern-5/2 / Tanimoto behind a ScaleKernel),
``model.train_inputs[0]``, ``model.likelihood.noise``,
``model.prediction_strategy.covar_cache`` and, in weighted mode, the posterior mean
(SOBER/_gp.py:212-305).  `KernelSpec` is that information reduced to tensors; it is
extracted from a duck-typed model (no gpytorch import) or given directly.

``kernel(x, y)`` keeps working as in the reference (2-D or 3-D second argument,
materialised result); `recombination` recognises a `Kernel` and runs the fused path
instead of calling it per level.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _native as nat

MODES = ("predictive_covariance", "weighted_predictive_covariance", "kernel")


@dataclass
class KernelSpec:
    """kind in {'rbf','matern52','tanimoto'}; lengthscale (d,) or (1,); W = S_cache S_cache^T is the
    `woodbury_inv` of SOBER/_gp.py:277; alpha/mean_const give the posterior mean
    (mean_const + k(x, X_obs) alpha) used by the weighted mode."""
    kind: str
    lengthscale: torch.Tensor
    outputscale: float
    X_obs: torch.Tensor
    S_cache: torch.Tensor
    noise: float = 0.0
    mean_const: float = 0.0
    alpha: Optional[torch.Tensor] = None

    def to(self, device):
        f = lambda t: None if t is None else t.detach().to(device=device, dtype=torch.float64).contiguous()
        return KernelSpec(self.kind, f(self.lengthscale), float(self.outputscale), f(self.X_obs),
                          f(self.S_cache), float(self.noise), float(self.mean_const), f(self.alpha))


def same_device(device) -> torch.device:
    """torch.device with the index made explicit (torch.device("cuda") and cuda:0 are the same device here)."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def _kernel_family(k):
    name = type(k).__name__
    if name == "RBFKernel":
        return "rbf", k
    if name == "MaternKernel":
        nu = float(getattr(k, "nu", 2.5))
        if nu != 2.5:
            raise ValueError(f"MaternKernel nu={nu} is not supported (only 2.5)")
        return "matern52", k
    if name in ("TanimotoKernel", "BitKernel"):
        return "tanimoto", k
    raise ValueError(f"unsupported covar_module {name}: expected (ScaleKernel of) RBFKernel, "
                     "MaternKernel(nu=2.5) or TanimotoKernel")


def spec_from_model(model) -> KernelSpec:
    """Read a KernelSpec off a (duck-typed) gpytorch exact-GP model, touching only the attributes
    the reference's hot path touches (SOBER/_gp.py:268-276,292-294)."""
    if isinstance(model, KernelSpec):
        return model
    if hasattr(model, "kernel_spec"):
        return model.kernel_spec() if callable(model.kernel_spec) else model.kernel_spec
    cm = model.covar_module
    outputscale = 1.0
    if type(cm).__name__ == "ScaleKernel":
        outputscale = float(cm.outputscale.detach().reshape(-1)[0])
        cm = cm.base_kernel
    kind, base = _kernel_family(cm)
    ls = torch.ones(1, dtype=torch.float64) if kind == "tanimoto" else base.lengthscale.detach().reshape(-1)
    X_obs = model.train_inputs[0]
    try:
        S = model.prediction_strategy.covar_cache
    except Exception:                                   # SOBER/_gp.py:272-276
        model.eval()
        model(X_obs[0].unsqueeze(0))
        S = model.prediction_strategy.covar_cache
    noise = float(torch.as_tensor(model.likelihood.noise).detach().reshape(-1)[0])
    alpha, mean_const = None, 0.0
    ps = getattr(model, "prediction_strategy", None)
    if ps is not None and hasattr(ps, "mean_cache"):
        alpha = ps.mean_cache.detach().reshape(-1)
        mm = getattr(model, "mean_module", None)
        const = getattr(mm, "constant", None) if mm is not None else None
        mean_const = 0.0 if const is None else float(torch.as_tensor(const).detach().reshape(-1)[0])
    return KernelSpec(kind, ls, outputscale, X_obs, S, noise, mean_const, alpha)


class PointSet:
    """Device-resident points in the layout the HIP kernels read: `data` (n, dt) float64 scaled by
    1/lengthscale and zero padded (continuous kinds) or bit-packed words viewed as float64
    (Tanimoto), plus `norm` (popcounts) for Tanimoto."""

    def __init__(self, data, norm, dt):
        self.data, self.norm, self.dt = data, norm, dt

    def __len__(self):
        return self.data.shape[0]

    def rows(self, lo, hi):
        return PointSet(self.data[lo:hi], None if self.norm is None else self.norm[lo:hi], self.dt)


def prepare_points(spec: KernelSpec, X: torch.Tensor) -> PointSet:
    """x / lengthscale (gpytorch RBF/Matern prologue) or 0/1 -> packed bits."""
    X = X.detach().to(dtype=torch.float64)
    if X.stride(-1) != 1:
        X = X.contiguous()
    n, d = X.shape
    kind = nat.KIND_BY_NAME[spec.kind]
    if kind == nat.KIND_TANIMOTO:
        nw = nat.bit_words(d, generic=True)
        words = torch.empty(n, nw, dtype=torch.int64, device=X.device)
        norms = torch.empty(n, dtype=torch.float64, device=X.device)
        bad = torch.zeros(1, dtype=torch.int32, device=X.device)
        nat.pack_bits(X, words, norms, bad)
        if int(bad.item()) != 0:
            raise ValueError("Tanimoto kernel: inputs must be 0/1 fingerprints")
        return PointSet(words.view(torch.float64), norms, nw)
    dt = nat.padded_dim(d, generic=True)
    out = torch.empty(n, dt, dtype=torch.float64, device=X.device)
    nat.scale_points(X, spec.lengthscale, out)
    return PointSet(out, None, dt)


class Kernel:
    """SOBER/_kernel.py:4-30.  `model` is a gpytorch-like model or a KernelSpec."""

    def __init__(self, model, mode="predictive_covariance"):
        self.model = model
        self.mode = mode
        self._spec_dev = None

    # -- spec handling ------------------------------------------------------
    def spec(self, device) -> KernelSpec:
        """The model's hyperparameters and caches on `device`.  The reference reads the LIVE gpytorch model on
        every call (SOBER/_gp.py:268-276,292-294), so a model retrained in place must not be served from a stale
        snapshot: a live model is read again on every call (a handful of small tensors); only a KernelSpec --
        a snapshot by construction -- is cached per device."""
        dev = same_device(device)
        if not isinstance(self.model, KernelSpec):
            return spec_from_model(self.model).to(dev)
        if self._spec_dev is None or same_device(self._spec_dev.X_obs.device) != dev:
            self._spec_dev = self.model.to(dev)
        return self._spec_dev

    def update_model(self, model):
        self.model, self._spec_dev = model, None

    # -- reference call protocol ---------------------------------------------
    def __call__(self, x, y):
        """Gram matrix of the chosen kernel: (M, N) for 2-D `y`, (E, M, S) for 3-D `y`
        (the broadcasting the reference relies on at SOBER/_rchq.py:124)."""
        if self.mode not in MODES:
            raise ValueError('mode should be from ["predictive_covariance", '
                             '"weighted_predictive_covariance", "kernel"]')
        if y.dim() == 3:
            E, S, d = y.shape
            flat = self._call2d(x, y.reshape(E * S, d))
            return flat.reshape(x.shape[0], E, S).permute(1, 0, 2)
        return self._call2d(x, y)

    def _call2d(self, x, y):
        spec = self.spec(x.device)
        kind = nat.KIND_BY_NAME[spec.kind]
        px, py = prepare_points(spec, x), prepare_points(spec, y)
        m, n = len(px), len(py)
        Kxy = torch.empty(m, n, dtype=torch.float64, device=x.device)
        nat.pairwise(kind, px.data, px.norm, py.data, py.norm, None, n, px.dt, spec.outputscale, Kxy)
        if self.mode == "kernel":
            return Kxy
        cov = posterior_correction(spec, px, py, Kxy)
        if self.mode == "predictive_covariance":
            return cov
        mu_x = posterior_mean(spec, px)
        mu_y = posterior_mean(spec, py)
        return mu_x.unsqueeze(1) * cov * mu_y.unsqueeze(0)        # SOBER/_kernel.py:44


def woodbury(spec: KernelSpec) -> torch.Tensor:
    """W = S @ S.T (SOBER/_gp.py:277) on the matrix cores."""
    n_obs = spec.S_cache.shape[0]
    W = torch.empty(n_obs, n_obs, dtype=torch.float64, device=spec.S_cache.device)
    nat.dgemm(spec.S_cache, spec.S_cache, W, transb=True)
    return W


def posterior_correction(spec, px: PointSet, py: PointSet, Kxy, W=None):
    """Kxy - (KxX @ W) @ KXy, left to right like SOBER/_gp.py:295."""
    kind = nat.KIND_BY_NAME[spec.kind]
    dev = Kxy.device
    pobs = prepare_points(spec, spec.X_obs)
    m, n, n_obs = len(px), len(py), len(pobs)
    KxX = torch.empty(m, n_obs, dtype=torch.float64, device=dev)
    KXy = torch.empty(n_obs, n, dtype=torch.float64, device=dev)
    nat.pairwise(kind, px.data, px.norm, pobs.data, pobs.norm, None, n_obs, px.dt, spec.outputscale, KxX)
    nat.pairwise(kind, pobs.data, pobs.norm, py.data, py.norm, None, n, px.dt, spec.outputscale, KXy)
    if W is None:
        W = woodbury(spec)
    T = torch.empty(m, n_obs, dtype=torch.float64, device=dev)
    nat.dgemm(KxX, W, T)
    cov = Kxy.clone()
    nat.dgemm(T, KXy, cov, alpha=-1.0, beta=1.0)
    return cov


def posterior_mean(spec, pts: PointSet):
    """mean_const + k(x, X_obs) @ alpha  (predict_mean, SOBER/_gp.py:240-253)."""
    if spec.alpha is None:
        raise ValueError("weighted_predictive_covariance needs the posterior-mean cache "
                         "(KernelSpec.alpha / model.prediction_strategy.mean_cache)")
    kind = nat.KIND_BY_NAME[spec.kind]
    pobs = prepare_points(spec, spec.X_obs)
    out = torch.empty(len(pts), dtype=torch.float64, device=pts.data.device)
    if not nat.fused_dim_supported(kind, spec.X_obs.shape[1]):
        # beyond the register-tiled kernels (d > 32, > 2048 bits): K(X_obs, pts) by the any-length pairwise kernel,
        # then alpha^T K on the matrix cores
        n_obs, n = len(pobs), len(pts)
        for lo in range(0, n, 1 << 16):
            hi = min(n, lo + (1 << 16))
            sub = pts.rows(lo, hi)
            KX = torch.empty(n_obs, hi - lo, dtype=torch.float64, device=out.device)
            nat.pairwise(kind, pobs.data, pobs.norm, sub.data, sub.norm, None, hi - lo, pts.dt, spec.outputscale, KX)
            row = torch.empty(1, hi - lo, dtype=torch.float64, device=out.device)
            nat.dgemm(spec.alpha.reshape(1, -1).contiguous(), KX, row)
            out[lo:hi] = row[0] + spec.mean_const
        return out
    nat.kernel_matvec(kind, pobs.data, pobs.norm, spec.alpha, pts.data, pts.norm, pts.dt,
                      spec.outputscale, spec.mean_const, out)
    return out

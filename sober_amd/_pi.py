"""GP prediction over a candidate pool and the LFI weight pi(x):
`predict` (SOBER/_gp.py:212-238) and `PI` (SOBER/_pi.py:5-56) with the reference's names.

    mean(x) = m + k(x, X_obs) alpha
    var(x)  = k(x, x) - k(x, X_obs) W k(X_obs, x) + noise          (exact GP; LOVE is upstream-only)
    pi(x)   = Phi((mean(x) - eta) / sqrt(var(x))),   eta = max_i mean(X_obs_i)

One launch per chunk of the pool (`sober_predict_fused`, csrc/predict.hip; n_obs <= 511): the K(X_obs, x) columns of 32
candidates are evaluated once into LDS, W k runs on the FP64 matrix cores, mean / variance / pi leave the kernel --
nothing of size n_obs x N is ever in memory.  With a square root S of W = S S^T (SOBER/_gp.py:277) the kernel works from S^T
itself -- var = k(x, x) - |S^T k|^2 + noise -- and skips the tile products above the diagonal when S^T is lower triangular (the
inverse Cholesky factor gpytorch caches: round 6).  Beyond 511 observations (or an input dimension outside the register-tiled
set) the materialised route of round 4 stays: KX = k(X_obs, chunk) by `sober_pairwise`, V = W KX (`sober_dgemm`), the
column-wise quadratic form and Phi in `sober_predict_finish`.
"""
import os
import math

import torch

from . import _native as nat
from ._kernel import KernelSpec, posterior_mean, prepare_points, same_device, spec_from_model, woodbury

CHUNK = 1 << 18


def _kxx_const(spec):
    if spec.kind == "rbf":
        return spec.outputscale
    if spec.kind == "matern52":                           # r = sqrt(clamp_min(0, 1e-30)) = 1e-15
        r = 1e-15
        return (math.sqrt(5.0) * r + 1.0 + (5.0 / 3.0) * r * r) * math.exp(-math.sqrt(5.0) * r) * spec.outputscale
    return 0.0                                            # tanimoto: per point, from the popcounts


def _model_side(spec):
    """What a prediction needs of the MODEL alone: the prepared observations, W = S S^T (SOBER/_gp.py:277) and alpha -- two
    launches and a handful of host calls that `PI` makes once per model snapshot instead of once per call (the reference's
    gpytorch model caches its prediction strategy in the same way)."""
    pobs = prepare_points(spec, spec.X_obs)
    alpha = spec.alpha if spec.alpha is None or (spec.alpha.dtype == torch.float64 and spec.alpha.is_contiguous()) \
        else spec.alpha.to(torch.float64).contiguous()
    S = spec.S_cache
    if S.dim() == 2 and S.shape[0] == S.shape[1] and not os.environ.get("SOBER_PREDICT_FROM_W"):
        # a square root (gpytorch's covar_cache of an exact GP): the fused kernel works from S^T itself -- W = S S^T
        # (SOBER/_gp.py:277) is not formed -- and is told ON THE DEVICE whether it is lower triangular (it is, for the inverse
        # Cholesky factor: the tile products above the diagonal are then skipped); no read-back
        St = S.to(torch.float64).t().contiguous()
        tri = (torch.count_nonzero(torch.triu(St, 1)) == 0).to(torch.int32).reshape(1)
        return pobs, St, alpha, tri
    W = woodbury(spec)
    if W.dtype != torch.float64 or W.stride(-1) != 1:      # (the fused kernel reads raw rows: csrc/predict.hip takes W's row stride only)
        W = W.to(torch.float64).contiguous()
    return pobs, W, alpha, None


def _predict(spec, X, eta=None, log=False, eta_dev=None, model_side=None):
    dev = X.device
    kind = nat.KIND_BY_NAME[spec.kind]
    pobs, W, alpha, root_tri = model_side if model_side is not None else _model_side(spec)
    n_obs, N = len(pobs), X.shape[0]
    mean = torch.empty(N, dtype=torch.float64, device=dev)
    var = torch.empty(N, dtype=torch.float64, device=dev)
    fused = (nat.predict_fused_supported(kind, n_obs, pobs.dt) and spec.alpha is not None
             and nat.fused_dim_supported(kind, spec.X_obs.shape[1]) and not os.environ.get("SOBER_PREDICT_MATERIALISED"))
    if not fused and root_tri is not None:                # (the materialised route wants W itself)
        W, root_tri = woodbury(spec).to(torch.float64).contiguous(), None
    if eta is None and eta_dev is not None:
        # the threshold is still in device memory: the fused kernel reads it there, the materialised route needs the number
        eta = 0.0 if fused else float(eta_dev.item())
        eta_dev = eta_dev if fused else None
    lfi = torch.empty(N, dtype=torch.float64, device=dev) if eta is not None else None
    for lo in range(0, N, CHUNK):
        hi = min(N, lo + CHUNK)
        pts = prepare_points(spec, X[lo:hi])
        n = hi - lo
        if fused:
            nat.predict_fused(kind, pobs.data, pobs.norm, pts.data, pts.norm, n, pts.dt, spec.outputscale, W, alpha,
                              spec.mean_const, _kxx_const(spec), spec.noise, mean[lo:hi], var[lo:hi],
                              0.0 if eta is None else eta, None if lfi is None else lfi[lo:hi], log,
                              eta_dev=eta_dev if eta is not None else None, root_tri=root_tri)
            continue
        mean[lo:hi] = posterior_mean(spec, pts)
        KX = torch.empty(n_obs, n, dtype=torch.float64, device=dev)
        nat.pairwise(kind, pobs.data, pobs.norm, pts.data, pts.norm, None, n, pts.dt, spec.outputscale, KX)
        V = torch.empty_like(KX)
        nat.dgemm(W, KX, V)
        nat.predict_finish(KX, V, mean[lo:hi], _kxx_const(spec), pts.norm, spec.outputscale, spec.noise,
                           var[lo:hi], 0.0 if eta is None else eta, None if lfi is None else lfi[lo:hi], log)
    return mean, var, lfi


def predict(test_x, model):
    """SOBER/_gp.py:212-238 -> (pred.mean, pred.variance) on the device."""
    spec = spec_from_model(model).to(test_x.device)
    x = test_x.reshape(-1, test_x.shape[-1])
    mean, var, _ = _predict(spec, x.to(torch.float64))
    return mean.reshape(test_x.shape[:-1]), var.reshape(test_x.shape[:-1])


def predict_mean(test_x, model):
    """SOBER/_gp.py:240-253."""
    return predict(test_x, model)[0]


class PI:
    """SOBER/_pi.py:5-56."""

    def __init__(self, model, label="lfi"):
        self.model = model
        self.label = label
        self.Xobs = spec_from_model(model).X_obs
        self._spec = None
        self._eta_dev = None                              # max posterior mean at the observations, ON the device
        self._model_side = None                           # prepared observations, W, alpha of the snapshot in self._spec
        self._side_key = None                             # what the model side was derived from (a live model: see _prepare)
        self._eta_set_by_caller = False

    def _prepare(self, device):
        """A live model is read again on every call (the reference evaluates `self.model` each time, :20-38); a
        KernelSpec snapshot is prepared once per device."""
        dev = same_device(device)
        live = not isinstance(self.model, KernelSpec)
        if live or self._spec is None or same_device(self._spec.X_obs.device) != dev:
            spec = spec_from_model(self.model).to(dev)
            # a live model is READ on every call, but what follows from it -- the prepared observations, the root of W, alpha,
            # the threshold: three launches, a prediction over the observations and their host side -- is redone only when
            # what was read has changed: the tensors' identity and version counters (an optimiser step or a new prediction
            # cache changes either) and the scalars.  (Sound because self._spec keeps the tensors it was derived from ALIVE: no
            # new tensor can come back at one of their addresses unless it shares their storage -- and then the version counter
            # tells.  A model whose attributes are recomputed on every access, like gpytorch's constrained lengthscale, simply
            # never hits the cache.)
            key = (spec.kind, float(spec.outputscale), float(spec.noise), float(spec.mean_const)) + tuple(
                (t.data_ptr(), t._version, tuple(t.shape), t.dtype) if t is not None else None
                for t in (spec.X_obs, spec.S_cache, spec.alpha, spec.lengthscale))
            if self._spec is None or key != self._side_key or self._eta_set_by_caller:
                self._spec, self._side_key, self._eta_set_by_caller = spec, key, False
                self._model_side = _model_side(spec)
                m_obs, _, _ = _predict(spec, spec.X_obs, model_side=self._model_side)
                self._eta_dev = m_obs.max().reshape(1)    # current maximum (:17): stays on the device -- no read-back per call
        return self._spec

    @property
    def eta(self):
        """The reference's `self.eta` (SOBER/_pi.py:17) as a number (reads the device value back)."""
        return None if self._eta_dev is None else float(self._eta_dev.item())

    @eta.setter
    def eta(self, value):
        """The reference's eta is a plain attribute a caller may assign (SOBER/_pi.py:17): the value goes to the device word the
        kernel reads; a live model's next call re-derives it from the model, like the reference's own `lfi`."""
        dev = self._eta_dev.device if self._eta_dev is not None else (self._spec.X_obs.device if self._spec is not None else None)
        if dev is None:
            from . import _settings
            dev = _settings._device
        self._eta_dev = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self._eta_set_by_caller = not isinstance(self.model, KernelSpec)   # (a live model's next call re-derives it)

    def lfi(self, X_cand, log=False):
        spec = self._prepare(X_cand.device)
        _, _, out = _predict(spec, X_cand.to(torch.float64), eta_dev=self._eta_dev, log=log, model_side=self._model_side)
        return out

    def __call__(self, X_cand, log=False):
        if self.label == "ts":
            raise NotImplementedError("Not implemented yet")
        elif self.label == "lfi":
            return self.lfi(X_cand, log=log)
        raise ValueError("Label should be either 'ts' or 'lfi'.")

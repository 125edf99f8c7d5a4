"""BASQ quadrature kernel provider: the g-space side of `ScaleMmltGP`
(SOBER/BASQ/_scale_mmlt.py:190-275) and `BASQ.quadrature` (SOBER/BASQ/_basq.py:43-81).

    mu_g(x)   = exp(mu_h(x) + var_h(x) / 2) - 1                       (mu_h, var_h) = predict(x, model)
    K_g(x, y) = mu_g(x) mu_g(y) (exp(C_h(x, y)) - 1)                  C_h = predictive_covariance

`exp(C_h) - 1` is non-linear in the posterior covariance, so the fused level kernel's sum-first
shortcut (G1 - T G2) does not apply: the posterior correction has to be applied per candidate.  On
the MI355X that is ONE pass over the pool per recombination call: K = k(cand, X_nys) and
KX = k(X, cand) from `sober_pairwise`, corr = KX^T (T^T) on the FP64 matrix cores (`sober_dgemm`, the
(N x n_obs) x (n_obs x M) contraction that dominates the reference's arithmetic), the epilogue
`sober_gspace_finish`; the (N, M) result stays in HBM and every halving level is the gather-sum
`sober_level_gather` (see `_ops_hip.MatrixKernelOps`).

Fitting / warping of the GP (update_gp, process_y_warping_with_scaling) is the control plane and
stays with the caller: this class is built from a fitted h-space model (duck-typed like
`sober_amd.Kernel`) or a `KernelSpec` with the posterior-mean cache.
"""
from __future__ import annotations

import math

import torch

from . import _native as nat
from ._kernel import KernelSpec, posterior_mean, prepare_points, same_device, spec_from_model, woodbury
from ._pi import _predict
from ._rchq import recombination

CHUNK = 1 << 15          # sober_pairwise: at most 65535 rows on the a side


class GspaceKernel:
    """`ScaleMmltGP.gspace_kernel` as an object: callable like the reference's bound method
    (2-D `x`, 2-D or 3-D `y`) and, for `recombination`, a native provider (`materialise`)."""

    def __init__(self, model, jitter=0.0):
        self._model = model
        self._spec = None
        self.jitter = float(jitter)                        # ScaleMmltGP.jitter (:73), 0 in the reference

    def spec(self, device):
        """(a live model is read again on every call, a KernelSpec snapshot once per device: sober_amd/_kernel.py)"""
        dev = same_device(device)
        live = not isinstance(self._model, KernelSpec)
        if live or self._spec is None or same_device(self._spec.X_obs.device) != dev:
            self._spec = spec_from_model(self._model).to(dev)
            if self._spec.alpha is None:
                raise ValueError("the g-space kernel needs the posterior-mean cache (KernelSpec.alpha / "
                                 "model.prediction_strategy.mean_cache)")
            if self._spec.kind == "tanimoto":
                raise NotImplementedError("g-space kernel: continuous base kernels only")
            self._W = woodbury(self._spec)
        return self._spec

    # -- pieces ---------------------------------------------------------------------------------
    def mean_predict(self, x):
        """gspace_mean_predict, SOBER/BASQ/_scale_mmlt.py:236-245 (any leading shape)."""
        return self.predict(x)[0]

    def predict(self, x):
        """gspace_predict, SOBER/BASQ/_scale_mmlt.py:206-219 -> (mu_g, var_g)."""
        if not x.is_cuda:
            raise nat.SoberHipError("GspaceKernel: inputs must live on the HIP device")
        spec = self.spec(x.device)
        flat = x.reshape(-1, x.shape[-1]).to(torch.float64)
        mu_h, var_h, _ = _predict(spec, flat)
        mu_g = (mu_h + 0.5 * var_h).exp() - 1
        var_g = (mu_g ** 2) * (var_h.exp() - 1)
        return mu_g.reshape(x.shape[:-1]), var_g.reshape(x.shape[:-1])

    def materialise(self, X_cand, X_nys, gram=False):
        """(N, M) matrix K_g(cand_c, x_r), candidate-major, in HBM.  gram=True: the (:273-274) jitter goes on
        the diagonal (X_cand is X_nys then)."""
        dev = X_cand.device
        spec = self.spec(dev)
        kind = nat.KIND_BY_NAME[spec.kind]
        Xn = X_nys.to(torch.float64)
        Xc = X_cand.to(torch.float64)
        N, M = Xc.shape[0], Xn.shape[0]
        pn = prepare_points(spec, Xn)
        pobs = prepare_points(spec, spec.X_obs)
        n_obs = len(pobs)
        # T^T = W k(X, X_nys): (n_obs, M)
        KXn = torch.empty(n_obs, M, dtype=torch.float64, device=dev)
        nat.pairwise(kind, pobs.data, pobs.norm, pn.data, pn.norm, None, M, pn.dt, spec.outputscale, KXn)
        Tt = torch.empty(n_obs, M, dtype=torch.float64, device=dev)
        nat.dgemm(self._W, KXn, Tt)
        mug_n = self.mean_predict(Xn)
        K = torch.empty(N, M, dtype=torch.float64, device=dev)
        for lo in range(0, N, CHUNK):
            hi = min(N, lo + CHUNK)
            n = hi - lo
            pc = prepare_points(spec, Xc[lo:hi])
            Kc = K[lo:hi]
            nat.pairwise(kind, pc.data, pc.norm, pn.data, pn.norm, None, M, pn.dt, spec.outputscale, Kc)
            KX = torch.empty(n_obs, n, dtype=torch.float64, device=dev)
            nat.pairwise(kind, pobs.data, pobs.norm, pc.data, pc.norm, None, n, pc.dt, spec.outputscale, KX)
            corr = torch.empty(n, M, dtype=torch.float64, device=dev)
            nat.dgemm(KX, Tt, corr, transa=True)                       # k(c, X) W k(X, x_r)
            # mu_g of the chunk from the same KX (var = kxx - diag(KX^T W KX) + noise)
            mean = posterior_mean(spec, pc)
            V = torch.empty_like(KX)
            nat.dgemm(self._W, KX, V)
            var = torch.empty(n, dtype=torch.float64, device=dev)
            from ._pi import _kxx_const
            nat.predict_finish(KX, V, mean, _kxx_const(spec), pc.norm, spec.outputscale, spec.noise, var, 0.0,
                               None, False)
            mug_c = (mean + 0.5 * var).exp() - 1
            nat.gspace_finish(Kc, corr, mug_c, mug_n)
        if gram and self.jitter != 0.0:
            K.diagonal().add_(self.jitter)
        return K

    def __call__(self, x, y):
        """The reference's protocol (SOBER/BASQ/_scale_mmlt.py:256-275): (len(x), len(y)) for 2-D `y`,
        (E, len(x), S) for `y` of shape (E, S, d)."""
        if y.dim() == 3:
            E, S, d = y.shape
            K = self.materialise(y.reshape(E * S, d), x)                  # (E*S, M)
            out = K.view(E, S, -1).transpose(1, 2).contiguous()
            if self.jitter != 0.0:                                        # :273-274 as written (first two dims)
                dd = min(len(x), len(y))
                out[range(dd), range(dd)] = out[range(dd), range(dd)] + self.jitter
            return out
        K = self.materialise(y, x).T.contiguous()
        if self.jitter != 0.0:
            dd = min(len(x), len(y))
            K[range(dd), range(dd)] = K[range(dd), range(dd)] + self.jitter
        return K


class ScaleMmlt:
    """The prediction/kernel surface of `ScaleMmltGP` that BASQ uses (`BASQ.update_model`,
    SOBER/BASQ/_basq.py:28-41): gspace_kernel, gspace_mean_predict, gspace_predict, hspace_predict,
    hspace_kernel, beta -- on a fitted h-space GP."""

    def __init__(self, model, beta, jitter=0.0):
        self.model = model
        self.beta = beta if torch.is_tensor(beta) else torch.tensor(float(beta), dtype=torch.float64)
        self.gspace_kernel = GspaceKernel(model, jitter)
        self.is_bq = True

    def hspace_predict(self, x):
        from ._pi import predict
        return predict(x, self.model)

    def hspace_mean_predict(self, x):
        return self.hspace_predict(x)[0]

    def gspace_predict(self, x):
        return self.gspace_kernel.predict(x)

    def gspace_mean_predict(self, x):
        return self.gspace_kernel.mean_predict(x)

    def hspace_kernel(self, x, y):
        from ._kernel import Kernel
        return Kernel(self.model, "predictive_covariance")(x, y)


def quadrature(X_cand, n_nys_quad, n_res_quad, model: ScaleMmlt, init_weights=None):
    """`BASQ.quadrature` (SOBER/BASQ/_basq.py:43-81) from the prior sample on: kernel recombination of
    the uniformly weighted sample with the g-space kernel, then
        EML   = w . mu_g(x)           ELML = log EML + beta   (beta and EML = exp(beta) when EML <= 0)
        AVLML = log |w K_g(x, x) w|
    Returns (ELML, AVLML, EML, idx, w)."""
    n = X_cand.shape[0]
    w_IS = torch.ones(n, dtype=torch.float64, device=X_cand.device) / n if init_weights is None else init_weights
    idx, w = recombination(X_cand, X_cand[:n_nys_quad], n_res_quad, model.gspace_kernel, init_weights=w_IS)
    x = X_cand[idx]
    EML = w @ model.gspace_mean_predict(x)
    beta = float(model.beta)
    if float(EML) <= 0:
        ELML, EML = beta, torch.tensor(math.exp(beta), dtype=torch.float64, device=X_cand.device)
    else:
        ELML = math.log(float(EML)) + beta
    AVLML = float((w @ model.gspace_kernel(x, x) @ w).abs().log())
    return ELML, AVLML, float(EML), idx, w

"""Weighted Gaussian KDE prior: `WeightedKernelDensityEstimation` with the
reference's constructor and `pdf` / `logpdf` (SOBER/_wkde.py:11-160).

    pdf(x) = sum_j w_j N(x - X_j; 0, Sigma)            (zero outside the bounds)

The reference materialises the (n_X * n_kde, d) differences and calls MultivariateNormal on them
(4096 components x 1e5 points = 3.3 GB at d = 10).  Here the points are whitened with the Cholesky
factor of Sigma (one small GEMM on the matrix cores), after which the mixture is the RBF
"kernel x vector" product of `sober_kernel_matvec` -- nothing is materialised.

Fitting (component selection, Scott/Silverman bandwidth, weighted covariance, PSD repair) is small
host-side tensor algebra and follows the reference line by line; like the reference it draws from
the global generator of the tensors' device (torch.multinomial).  `sample` batches all components
of a rejection round into one launch of `sober_wkde_draw`."""
import copy
import math

import torch

from . import _native as nat
from ._utils import SafeTensorOperator
from ._weights import WeightsStabiliser


class _HostWeights(WeightsStabiliser):
    """cleansing_weights with eps = 0 on whatever device the tensor lives (fit-time bookkeeping on
    n_kde-sized vectors; the pool-sized scrub is WeightsStabiliser.cleansing_weights in HIP)."""

    def cleansing_weights(self, weights):
        weights[weights < self.eps_weights] = 0
        weights[weights.isinf()] = self.eps_weights
        weights[weights.isnan()] = self.eps_weights
        if not weights.sum() == 0:
            weights /= weights.sum()
        else:
            weights = torch.ones_like(weights) / len(weights)
        return weights.detach()


class WeightedKernelDensityEstimation(_HostWeights, SafeTensorOperator):
    def __init__(self, X, W, n_dims, bounds=None, n_kde=4096, bw_method="scott", compute_cdf=False):
        """SOBER/_wkde.py:16-50."""
        WeightsStabiliser.__init__(self, eps=0, thresh=n_kde)
        SafeTensorOperator.__init__(self)
        if compute_cdf:
            raise NotImplementedError("compute_cdf=True (truncated-MVN normalising constants) is not supported")
        self.device, self.dtype = X.device, X.dtype       # follow the data, not the module globals
        self.n_dims = n_dims
        self.bounds = bounds
        self.n_kde_init = min([n_kde, len(X)])
        self.bw_method = bw_method
        self.compute_cdf = compute_cdf
        self.type = "continuous"
        self.initialisation(X, W)

    def initialise_n_kde(self):
        self.n_kde = copy.deepcopy(self.n_kde_init)

    def initialisation(self, X, Y):
        """SOBER/_wkde.py:52-80."""
        self.initialise_n_kde()
        if self.check_weights(Y):
            idx_accept = self.deweighted_resampling(Y, self.n_kde)
        else:
            idx_accept = torch.arange(Y.size(0), device=Y.device)[self.cleansing_weights(Y) > 0]
            self.n_kde = len(idx_accept)
            if self.n_kde < 1:
                raise ValueError("Invalid weights")
            elif self.n_kde > self.n_kde_init:
                self.initialise_n_kde()
                idx_accept = self.deweighted_resampling(Y, self.n_kde)
        self.Xobs = X[idx_accept]
        self.weights = self.cleansing_weights(Y[idx_accept])
        self.n_kde = self.Xobs.size(0)
        self.set_bandwidth()
        self._compute_covariance()
        self._whitened = None

    def to(self, device):
        """Move the fitted mixture (components, weights, covariance, bounds) to `device`."""
        self.Xobs, self.weights = self.Xobs.to(device), self.weights.to(device)
        self.covariance = self.covariance.to(device)
        if self.bounds is not None:
            self.bounds = self.bounds.to(device)
        self.device = self.Xobs.device
        self._whitened = None
        return self

    def set_bandwidth(self):
        """SOBER/_wkde.py:87-96."""
        self.neff = 1.0 / (self.weights ** 2).sum()
        if self.bw_method == "scott":
            self.bw = self.neff.pow(-1.0 / (self.n_dims + 4))
        elif self.bw_method == "silverman":
            self.bw = (self.neff * (self.n_dims + 2.0) / 4.0).pow(-1.0 / (self.n_dims + 4))

    def _compute_covariance(self):
        """SOBER/_wkde.py:98-108; the PSD repair runs on the host like everywhere else."""
        _mean = self.weights @ self.Xobs
        _residual = self.Xobs - _mean.unsqueeze(0)
        self._data_covariance = (_residual.T * self.weights.unsqueeze(0)) @ _residual
        self._data_covariance /= 1 - self.weights.pow(2).sum()
        cov = self.make_cov_psd((self._data_covariance * self.bw.pow(2)).cpu())
        self.covariance = cov.to(self.Xobs.device)

    # -- the pool-sized part ------------------------------------------------------------------
    def _prepare(self, device):
        """Whitening x~ = L^-1 x with L L^T = Sigma: N(x - X_j; 0, Sigma) = c exp(-|x~ - X~_j|^2 / 2),
        c = (2 pi)^(-d/2) / prod(diag L)."""
        if self._whitened is None or self._whitened[0].device != torch.device(device):
            d = self.n_dims
            cov = self.make_cov_psd(self.covariance.cpu().to(torch.float64).clone())   # safe_mvn_register, _utils.py:168
            L = torch.linalg.cholesky(cov)
            Linv_T = torch.linalg.solve_triangular(L, torch.eye(d, dtype=torch.float64), upper=False).T.contiguous()
            const = (2.0 * math.pi) ** (-0.5 * d) / float(torch.diagonal(L).prod())
            Linv_T = Linv_T.to(device)
            Xo = self.Xobs.to(device, torch.float64).contiguous()
            Xo_w = torch.empty_like(Xo)
            nat.dgemm(Xo, Linv_T, Xo_w)                                               # rows x~ = L^-1 x
            self._whitened = (Xo_w, Linv_T, const, self.weights.to(device, torch.float64).contiguous())
        return self._whitened

    def pdf(self, X):
        """SOBER/_wkde.py:109-145."""
        if not X.is_cuda:
            raise nat.SoberHipError("WeightedKernelDensityEstimation.pdf: X must live on the HIP device")
        Xo_w, Linv_T, const, w = self._prepare(X.device)
        d = self.n_dims
        Xd = X.detach().to(torch.float64).contiguous()
        Xw = torch.empty_like(Xd)
        nat.dgemm(Xd, Linv_T, Xw)
        dt = nat.padded_dim(d)
        ones = torch.ones(1, dtype=torch.float64, device=X.device)
        a = torch.empty(Xo_w.shape[0], dt, dtype=torch.float64, device=X.device)
        b = torch.empty(Xw.shape[0], dt, dtype=torch.float64, device=X.device)
        nat.scale_points(Xo_w, ones, a)
        nat.scale_points(Xw, ones, b)
        out = torch.empty(Xd.shape[0], dtype=torch.float64, device=X.device)
        nat.kernel_matvec(nat.KIND_RBF, a, None, w, b, None, dt, const, 0.0, out)
        if self.bounds is not None:                                                    # :127-131
            bd = self.bounds.to(X.device)
            outside = (X < bd[0]).any(axis=1) | (X > bd[1]).any(axis=1)
            out = torch.where(outside, torch.zeros_like(out), out)
        return out.to(X.dtype)

    def logpdf(self, X):
        """SOBER/_wkde.py:147-160."""
        return self.pdf(X).log()

    def sample(self, N_rec, stream="device", n_repeat=10):
        """SOBER/_wkde.py:221-248 with sample_from_Gaussian (:189-219) and rejection_sampling (:162-187).

        Component i contributes int(w_i N_rec) draws (2x when the truncated counts fall short) of
        N(X_i, Sigma); with bounds each round draws 10x the missing count and keeps the first
        accepted ones, at most `n_repeat` rounds; a surplus is thinned by multinomial without
        replacement.  The reference loops over up to 4096 components in Python; here all components
        of a round are ONE batch: the normals, one `sober_wkde_draw` launch (affine map + bounds
        test) and a segmented prefix count that keeps the first `cnt_i` accepted rows of each
        component.

        stream="device": normals and the final thinning come from the device generator (what the
        reference does when its tensors live on the GPU).  stream="reference": they come from the
        global CPU generator in the reference's call order (one (cnt_i, d) / (10 cnt_i, d) draw per
        component), so a CPU run of the reference with the same seed gives the same samples -- as
        long as every component fills up in its first rejection round (the reference redraws a
        short component before moving on, which this batched order cannot follow; `last_sample_exact`
        records whether that happened)."""
        if stream not in ("device", "reference"):
            raise ValueError("stream must be 'device' or 'reference'")
        dev = self.Xobs.device
        if not self.Xobs.is_cuda:
            raise nat.SoberHipError("WeightedKernelDensityEstimation.sample: the KDE must live on the HIP device")
        d = self.n_dims
        cnts = self.weights * N_rec                                                    # :231-234
        cnt_kde = cnts.type(torch.int)
        if cnt_kde.sum() < N_rec:
            cnt_kde = (2 * cnts).type(torch.int)
        cnt = cnt_kde.cpu().to(torch.int64)
        self.last_sample_exact = True
        empty = torch.empty(0, d, dtype=self.Xobs.dtype, device=dev)
        if bool((self.covariance == 0).all()) or int(cnt.sum()) == 0:                  # :203-208
            return empty
        cov = self.make_cov_psd(self.covariance.cpu().to(torch.float64).clone())
        L = torch.linalg.cholesky(cov).contiguous().to(dev)
        Xo = self.Xobs.to(torch.float64).contiguous()
        bounded = self.bounds is not None
        lo = hi = None
        if bounded:
            bd = self.bounds.to(dev, torch.float64)
            lo, hi = bd[0].contiguous(), bd[1].contiguous()

        def draw(comp_ids, per_comp):
            """rows of components `comp_ids` (host int64), `per_comp` rows each -> x, inside, comp_of_row"""
            total = int(per_comp.sum())
            if stream == "reference":
                eps = torch.cat([torch.randn(int(k), d, dtype=torch.float64) for k in per_comp.tolist()]).to(dev)
            else:
                eps = torch.randn(total, d, dtype=torch.float64, device=dev)
            comp = torch.repeat_interleave(comp_ids.to(dev), per_comp.to(dev)).to(torch.int32)
            x = torch.empty(total, d, dtype=torch.float64, device=dev)
            inside = torch.empty(total, dtype=torch.int32, device=dev) if bounded else None
            nat.wkde_draw(eps, comp, Xo, L, lo, hi, x, inside)
            return x, inside, comp

        active = torch.nonzero(cnt > 0).flatten()
        if not bounded:
            samples, _, _ = draw(active, cnt[active])
        else:
            have = torch.zeros_like(cnt)
            parts, keys = [], []
            for rnd in range(n_repeat):                                               # :177-187
                need = cnt[active] - have[active]
                per = cnt[active] * n_repeat                                          # int(n_repeat * cnt) rows per round
                x, inside, comp = draw(active, per)
                starts = torch.cumsum(per, 0) - per                                   # first row of each component
                csum = torch.cumsum(inside, 0, dtype=torch.int64)
                starts_d = starts.to(dev)
                before = torch.where(starts_d > 0, csum[(starts_d - 1).clamp(min=0)], torch.zeros_like(starts_d))
                rank = csum - inside - torch.repeat_interleave(before, per.to(dev))   # accepted rows ahead of me
                need_row = torch.repeat_interleave(need.to(dev), per.to(dev))
                keep = (inside > 0) & (rank < need_row)
                parts.append(x[keep])
                keys.append(comp[keep].to(torch.int64))
                ends = torch.cumsum(per, 0).to(dev) - 1
                accepted = (csum[ends] - before).cpu()
                have[active] += torch.minimum(accepted, need)
                active = active[have[active] < cnt[active]]
                if active.numel() == 0:
                    break
                self.last_sample_exact = False                                        # a component needed a redraw
            samples = torch.cat(parts)
            if len(parts) > 1:                                                        # component-major like the reference
                samples = samples[torch.sort(torch.cat(keys), stable=True).indices]
        if len(samples) > N_rec:                                                      # :245-247
            if stream == "reference":
                indice = torch.multinomial(torch.ones(len(samples), dtype=self.Xobs.dtype), N_rec).to(dev)
            else:
                indice = torch.multinomial(torch.ones(len(samples), dtype=self.Xobs.dtype, device=dev), N_rec)
            samples = samples[indice]
        return samples.to(self.Xobs.dtype)

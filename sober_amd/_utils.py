"""TensorManager / SafeTensorOperator with the reference's names and behaviour
(SOBER/_utils.py:20-157).  Allocation helpers follow the global device/dtype; the PSD
repair (`make_cov_psd`) runs on the HOST: it is a 500x500 LAPACK problem whose branch
decisions (and the subsequent randomised SVD) must match the reference's CPU LAPACK
(SURVEY.md App. C), so the device only produces the Gram matrix."""
import warnings

import torch

from ._settings import setting_parameters


class TensorManager:
    """SOBER/_utils.py:20-78."""

    def __init__(self, device=None, dtype=None):
        _device, _dtype = setting_parameters()
        self.device = _device if device is None else device
        self.dtype = _dtype if dtype is None else dtype

    def standardise_tensor(self, tensor):
        return tensor.to(self.device, self.dtype)

    def standardise_device(self, tensor):
        return tensor.to(self.device)

    def ones(self, n_samples, n_dims=None):
        shape = (n_samples,) if n_dims is None else (n_samples, n_dims)
        return torch.ones(*shape, device=self.device, dtype=self.dtype)

    def zeros(self, n_samples, n_dims=None):
        shape = (n_samples,) if n_dims is None else (n_samples, n_dims)
        return torch.zeros(*shape, device=self.device, dtype=self.dtype)

    def arange(self, length):
        return torch.arange(length, device=self.device)

    def null(self):
        return torch.tensor([], device=self.device)

    def tensor(self, x):
        return torch.tensor(x, device=self.device, dtype=self.dtype)

    def randperm(self, length):
        return self.standardise_device(torch.randperm(length))

    def multinomial(self, weights, n):
        return self.standardise_device(torch.multinomial(weights, n))

    def numpy(self, x):
        return x.detach().cpu().numpy()

    def is_cuda(self):
        return torch.device(self.device).type == "cuda"


class SafeTensorOperator(TensorManager):
    """SOBER/_utils.py:81-157 (the parts the recombination path uses)."""

    def __init__(self):
        super().__init__()
        self.max_iter = 10

    @staticmethod
    def is_psd(mat) -> bool:
        """SOBER/_utils.py:117-129: Cholesky succeeds AND exactly symmetric AND no negative
        eigenvalue.  Same verdict, cheaper order: the symmetry test first (an asymmetric matrix is
        rejected whatever Cholesky says), and the symmetric eigen-solver instead of the reference's
        non-symmetric `linalg.eig` (0.2 s at 500x500) -- by then `mat` is exactly symmetric, so both
        solvers agree whenever the verdict is well defined (|lambda_min| above rounding)."""
        if not bool((mat == mat.T).all()):
            return False
        try:
            torch.linalg.cholesky(mat)
        except Exception:
            return False
        return bool((torch.linalg.eigvalsh(mat) >= 0).all())

    def make_cov_psd(self, cov):
        """SOBER/_utils.py:131-157 on a HOST tensor (quirk Q2: sqrt(cov*cov.T) = |cov|; jitter
        1e-5*2^k added to the diagonal until PSD; diagonal fallback after more than max_iter
        rounds).

        The reference probes each rung of the jitter ladder with a Cholesky + eig (up to 12 of
        each).  Adding t to the diagonal shifts every eigenvalue by t, so ONE symmetric
        eigen-decomposition of |cov| tells which rung is the first PSD one; the rungs are then
        applied with the reference's own sequence of diagonal additions.  If lambda_min + t lands
        within 1e-9 of zero for the deciding rungs the literal ladder runs instead."""
        if self.is_psd(cov):
            return cov
        warnings.warn("Estimated covariance matrix was not positive semi-definite. Conveting...")
        cov = torch.nan_to_num(cov)
        cov = torch.sqrt(cov * cov.T)
        n_dim = cov.size(0)
        k_first = None
        try:
            lam_min = float(torch.linalg.eigvalsh(cov)[0])
            shifts = [1e-5 * (2 ** k - 1) for k in range(self.max_iter + 1)]     # after k additions
            psd = [lam_min + t > 0 for t in shifts]
            clear = all(abs(lam_min + t) > 1e-9 * max(1.0, abs(lam_min)) for t in shifts)
            if clear and lam_min == lam_min:
                k_first = psd.index(True) if any(psd) else self.max_iter + 1
        except Exception:
            k_first = None
        if k_first is None:
            return self._make_cov_psd_ladder(cov)
        jitter = torch.ones(n_dim, dtype=cov.dtype, device=cov.device) * 1e-5
        diag = cov.diagonal()
        for _ in range(k_first):                      # the reference's additions, one rung at a time
            diag += jitter
            jitter *= 2
        if k_first > self.max_iter:
            cov = cov.diag().diag()
        return cov

    def _make_cov_psd_ladder(self, cov):
        """The literal loop of SOBER/_utils.py:145-156 (borderline spectra only)."""
        if not self.is_psd(cov):
            n_dim = cov.size(0)
            jitter = torch.ones(n_dim, dtype=cov.dtype, device=cov.device) * 1e-5
            n_iter = 0
            while not self.is_psd(cov):
                cov[range(n_dim), range(n_dim)] += jitter
                jitter *= 2
                n_iter += 1
                if n_iter > self.max_iter:
                    cov = cov.diag().diag()
                    break
        return cov


class Utils(SafeTensorOperator):
    pass

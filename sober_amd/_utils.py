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
        eigenvalue.  The reference asks the NON-symmetric solver (`linalg.eig`, 0.2 s at
        500x500) for the last test; by then `mat` is exactly symmetric, so the symmetric
        solver gives the same verdict whenever the verdict is well defined (|lambda_min| above
        rounding, which the 1e-5 jitter ladder guarantees)."""
        try:
            torch.linalg.cholesky(mat)
        except Exception:
            return False
        if not bool((mat == mat.T).all()):
            return False
        return bool((torch.linalg.eigvalsh(mat) >= 0).all())

    def make_cov_psd(self, cov):
        """SOBER/_utils.py:131-157 on a HOST tensor (quirk Q2: sqrt(cov*cov.T) = |cov|;
        jitter 1e-5*2^k; diagonal fallback after more than max_iter rounds)."""
        if self.is_psd(cov):
            return cov
        warnings.warn("Estimated covariance matrix was not positive semi-definite. Conveting...")
        cov = torch.nan_to_num(cov)
        cov = torch.sqrt(cov * cov.T)
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

"""`RecombinationSampler` (SOBER/_sampler.py:11-59): the funnel from `Sober.next_batch` into
`recombination`."""
from ._rchq import recombination
from ._utils import TensorManager
from ._weights import WeightsStabiliser


class RecombinationSampler(WeightsStabiliser, TensorManager):
    def __init__(self, kernel, thresh=5):
        WeightsStabiliser.__init__(self, thresh=thresh)
        TensorManager.__init__(self)
        self.kernel = kernel

    def sampling_recombination(self, X_cand, X_nys, weights, batch_size, calc_obj=None):
        """SOBER/_sampler.py:27-59 -> (idx_rchq, w_rchq)."""
        idx_rchq, w_rchq = recombination(
            X_cand, X_nys, batch_size, self.kernel, self.device, self.dtype,
            init_weights=weights, calc_obj=calc_obj)
        return idx_rchq, w_rchq

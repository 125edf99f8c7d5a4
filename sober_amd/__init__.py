"""sober_amd: the kernel-recombination hot path of SOBER (ma921/SOBER) on the AMD Instinct MI355X.

Same names as the reference for this path: `recombination`, `Kernel`, `WeightsStabiliser`,
`KMeans`, `RecombinationSampler`, `EmpiricalSampler`, `Sober`, `TensorManager`, `SafeTensorOperator`,
`setting_parameters`.
The compute is hand-written HIP (gfx950) behind the C ABI of include/sober_hip.h; importing the
package does not need a GPU, running it does (there is no CPU fallback)."""
from ._settings import setting_parameters
from ._utils import SafeTensorOperator, TensorManager, Utils
from ._kernel import Kernel, KernelSpec, spec_from_model
from ._rchq import rc_kernel_svd, recombination
from ._weights import KMeans, WeightsStabiliser
from ._sampler import EmpiricalSampler, RecombinationSampler, adaptive_pruning
from ._pi import PI, predict, predict_mean
from ._sober import Sober
from ._wkde import WeightedKernelDensityEstimation
from ._basq import GspaceKernel, ScaleMmlt, quadrature as basq_quadrature

__all__ = ["setting_parameters", "TensorManager", "SafeTensorOperator", "Utils", "Kernel", "KernelSpec",
           "spec_from_model", "recombination", "rc_kernel_svd", "WeightsStabiliser", "KMeans",
           "RecombinationSampler", "EmpiricalSampler", "Sober", "adaptive_pruning", "PI", "predict", "predict_mean",
           "WeightedKernelDensityEstimation", "GspaceKernel", "ScaleMmlt", "basq_quadrature"]
__version__ = "0.1.0"

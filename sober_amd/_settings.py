"""Global device/dtype switch with the reference's semantics (SOBER/_settings.py:3-22).

The reference picks cuda when available and torch.double; on ROCm ``torch.device('cuda')``
IS the HIP device.  ``setting_parameters`` mutates the module globals exactly like the
reference's (no other config system exists there)."""
import torch

_device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
_dtype = torch.double


def setting_parameters(device=None, dtype=None):
    """SOBER/_settings.py:11-22 -> (device, dtype)."""
    global _device, _dtype
    if device:
        _device = device
    if dtype:
        _dtype = dtype
    return _device, _dtype

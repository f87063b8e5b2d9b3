"""Numerics half of the reference's LRPtools/utils.py (constants :7-14, safe_divide :16-18).
The visualisation half (heatmap/gamma/project, :34-184) is out of scope (image rendering)."""
import torch

from .. import _lib, ops

LOWEST = -1
HIGHEST = 1
EPSILON = 0.01          # LRPtools/utils.py:10
Z_EPSILON = 1e-7        # :11
LOGIT_BETA = 4
RELEVANCE_RECT = -1e-6  # :14
ALPHA = 1.
BETA = 0.


def safe_divide(numerator, divisor):
    """numerator / (divisor + Z_EPSILON * [divisor == 0])  (LRPtools/utils.py:16-18), on the device."""
    if numerator.shape != divisor.shape:
        divisor = divisor.expand_as(numerator)
    n = numerator.contiguous().view(1, -1)
    pad = (-n.shape[1]) % 4
    if pad:
        n = torch.nn.functional.pad(n, (0, pad))
        d = torch.nn.functional.pad(divisor.contiguous().view(1, -1), (0, pad), value=1.0)
    else:
        d = divisor.contiguous().view(1, -1)
    out = ops.divide_stab(n, d, None, _lib.STAB_SAFE)
    return out[:, :numerator.numel()].view(numerator.shape)

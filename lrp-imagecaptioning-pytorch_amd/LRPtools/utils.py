"""The reference's LRPtools/utils.py on the device: constants (:7-14), safe_divide (:16-18) and the heat-map
rendering of relevance maps (`gamma` :97-145 + `heatmap` :67-90 with `project` :34-52) as one kernel
(`relevance_heatmap`), so that maps are coloured without leaving HBM.  `visuallize_attention` (:150-184) needs
skimage's `pyramid_expand` and is not provided."""
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


_LUT_CACHE = {}


def colormap_lut(cmap_type="seismic", device="cuda"):
    """The 256-entry RGB table of a matplotlib colour map (what `plt.cm.get_cmap(cmap_type)` indexes, utils.py:68,84)."""
    key = (cmap_type, str(device))
    if key not in _LUT_CACHE:
        import numpy as np
        import matplotlib
        cm = matplotlib.colormaps[cmap_type] if hasattr(matplotlib, "colormaps") else __import__("matplotlib.pyplot").pyplot.cm.get_cmap(cmap_type)
        lut = np.asarray(cm(np.arange(256)))[:, :3].astype(np.float32)
        _LUT_CACHE[key] = torch.from_numpy(lut).to(device).contiguous()
    return _LUT_CACHE[key]


def relevance_heatmap(maps, gamma=0.7, cmap_type="seismic", lut=None):
    """`LRPutil.heatmap(LRPutil.gamma(hm))` of the explainers' `visualize_explanations` (models/gridTDmodel.py:1196-1198)
    for a batch of maps: (N,C,H,W) relevance on the device -> (N,H,W,3) float32 colours in [0,1].  Every map is
    normalised on its own (the reference renders one map per call); `lut`: optional (256,3) table instead of a
    matplotlib colour map."""
    if not maps.is_cuda:
        raise _lib.LrpxError("relevance_heatmap runs on the device: pass a CUDA tensor")
    maps = maps.to(torch.float32).contiguous()
    n, c, h, w = maps.shape
    lut = colormap_lut(cmap_type, maps.device) if lut is None else lut.to(maps.device, torch.float32).contiguous()
    tmp = torch.empty(n, h * w, device=maps.device, dtype=torch.float32)
    out = torch.empty(n, h, w, 3, device=maps.device, dtype=torch.float32)
    _lib.check(_lib.load().lrpx_heatmap(_lib.ptr(maps), n, c, h * w, float(gamma), _lib.ptr(lut), lut.shape[0],
                                        _lib.ptr(tmp), _lib.ptr(out), _lib.stream_ptr()))
    return out

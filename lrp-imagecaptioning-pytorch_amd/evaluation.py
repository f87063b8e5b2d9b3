"""Device-side consumers of relevance maps: the reductions of the reference's evaluation experiments
(evaluation.py:57-80 `block_image`, :313-336 `_calculate_overlaped_pixels`, :338-343 `_project_maxabs`, :124-134 /
:406-412 channel reductions, :506-513 tpfp statistics) on (N,C,H,W) maps that stay in HBM.
Host logic only: every function sequences HIP kernels of liblrpx (no CPU fallback)."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _dev(t):
    if not t.is_cuda:
        raise _lib.LrpxError("relevance-map consumers run on the device: pass CUDA tensors")
    return t.to(torch.float32).contiguous()


def spatial_relevance(maps, mode="mean"):
    """(N,C,H,W) -> (N,H,W): "mean" (evaluation.py:134), "pos" = mean_c(max(x,0)) (:410-412), "neg" = mean_c(max(-x,0))."""
    maps = _dev(maps)
    n, c, h, w = maps.shape
    out = torch.empty(n, h, w, device=maps.device, dtype=torch.float32)
    check(_lib.load().lrpx_spatial_reduce(ptr(maps), n, c, h * w, {"mean": 0, "pos": 1, "neg": 2}[mode], ptr(out),
                                          stream_ptr()))
    return out


def project_maxabs(x):
    """`_project_maxabs` (evaluation.py:338-343) per map; returns a new tensor."""
    x = _dev(x).clone()
    check(_lib.load().lrpx_project_maxabs(ptr(x), x.shape[0], x[0].numel(), stream_ptr()))
    return x


def block_image(spatial, patch_size=8, num_delete_patches=20):
    """`block_image` (evaluation.py:57-80) for every map: (N,H,W) -> mask (N,H,W) with zeros on the
    `num_delete_patches` squares of `patch_size` pixels that carry the largest relevance sums."""
    spatial = _dev(spatial)
    n, h, w = spatial.shape
    if h % patch_size or w % patch_size:
        raise AssertionError("map size must be a multiple of the patch size")          # evaluation.py:59-60
    mask = torch.empty_like(spatial)
    check(_lib.load().lrpx_patch_mask(ptr(spatial), n, h, w, patch_size, num_delete_patches, ptr(mask), stream_ptr()))
    return mask


def overlapped_pixels(spatial, boxes, thresholds=(0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)):
    """`_calculate_overlaped_pixels` (evaluation.py:313-336) for every map and threshold: spatial (N,H,W) (already
    projected), boxes (N,4) int [x0,y0,x1,y1] -> (N, len(thresholds)) correctness ratios."""
    spatial = _dev(spatial)
    n, h, w = spatial.shape
    boxes = boxes.to(spatial.device, torch.int32).contiguous()
    thr = torch.tensor(list(thresholds), dtype=torch.float32, device=spatial.device)
    out = torch.empty(n, thr.numel(), device=spatial.device, dtype=torch.float32)
    check(_lib.load().lrpx_bbox_ratio(ptr(spatial), n, h, w, ptr(boxes), ptr(thr), thr.numel(), ptr(out), stream_ptr()))
    return out


def map_statistics(spatial):
    """tpfp statistics (evaluation.py:506-513) per map: (N,H,W) -> (N,4) = mean, mean |x|, mean of positives, max."""
    spatial = _dev(spatial)
    out = torch.empty(spatial.shape[0], 4, device=spatial.device, dtype=torch.float32)
    check(_lib.load().lrpx_map_stats(ptr(spatial), spatial.shape[0], spatial[0].numel(), ptr(out), stream_ptr()))
    return out


def map_quantiles(spatial, points=None):
    """The quantile row of the tpfp statistics (evaluation.py:451 `quantile_point`, :510, :543): np.quantile(map, points)
    per map, (N,H,W) -> (N, len(points)); default points i/100, i = 0..99 like the reference."""
    spatial = _dev(spatial)
    lib = _lib.load()
    pts = [i / 100 for i in range(100)] if points is None else [float(p) for p in points]
    if not pts or min(pts) < 0 or max(pts) > 1:
        raise ValueError("Quantiles must be in the range [0, 1]")       # numpy's own message
    q = torch.tensor(pts, dtype=torch.float64, device=spatial.device)
    n, per = spatial.shape[0], spatial[0].numel()
    need = lib.lrpx_map_quantiles_workspace(n, per)
    if need == 0:
        raise ValueError(f"map_quantiles: {n} maps of {per} values are more than one call can sort (2^31 values)")
    ws = torch.empty(need, dtype=torch.uint8, device=spatial.device)
    out = torch.empty(n, len(pts), device=spatial.device, dtype=torch.float32)
    check(lib.lrpx_map_quantiles(ptr(spatial), n, per, ptr(q), len(pts), ptr(out), ptr(ws), need, stream_ptr()))
    return out

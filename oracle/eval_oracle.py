"""CPU restatement (numpy) of the relevance-map consumers of the reference's evaluation experiments - TEST
INFRASTRUCTURE ONLY (tests/ import it; the product path never does).  Pinned against the reference's own methods:
tests/golden/eval_consumers.npz is produced by calling `EvaluationExperiments.block_image`,
`._calculate_overlaped_pixels` and `._project_maxabs` of /root/reference/evaluation.py on seeded maps
(tests/golden/make_golden.py --only eval)."""
import numpy as np


def spatial_relevance(m, mode="mean"):
    """evaluation.py:134 `torch.mean(relevance_img, dim=(0, 1))`; :410-412 positive part; :406-408 negative part.
    m: (C,H,W) -> (H,W)"""
    if mode == "pos":
        m = np.maximum(m, 0)
    elif mode == "neg":
        m = np.maximum(-m, 0)
    return m.mean(axis=0, dtype=np.float64).astype(np.float32)


def project_maxabs(x):
    """evaluation.py:338-343"""
    absmax = np.max(np.abs(x))
    if absmax == 0:
        return np.zeros(x.shape, x.dtype)
    return (1.0 * x / absmax).astype(x.dtype)


def block_image(rel, patch_size=8, num_delete_patches=20):
    """evaluation.py:57-80: (H,W) relevance -> (H,W) mask, 0 on the `num_delete_patches` patches with the largest sums."""
    h, w = rel.shape
    assert h % patch_size == 0 and w % patch_size == 0
    nph, npw = h // patch_size, w // patch_size
    sums = rel.reshape(nph, patch_size, npw, patch_size).sum(axis=(1, 3), dtype=np.float64).reshape(-1)
    top = np.argsort(-sums, kind="stable")[:num_delete_patches]
    sel = np.zeros(nph * npw, bool)
    sel[top] = True
    return np.where(np.repeat(np.repeat(sel.reshape(nph, npw), patch_size, 0), patch_size, 1), 0.0, 1.0).astype(np.float32), sums


def overlapped_pixels(bbox, rel, threshold):
    """evaluation.py:313-336; bbox = [x0, y0, x1, y1]; `rel` is not modified (the reference zeroes it in place, which for
    its increasing threshold list is the same as thresholding a fresh copy)."""
    rel = np.where(rel <= threshold, 0, rel)
    total = rel.sum(dtype=np.float64)
    if total == 0:
        return 0.0
    inside = rel[bbox[1]:bbox[3], bbox[0]:bbox[2]].sum(dtype=np.float64)
    return min(1.0, float(inside / total))


def map_statistics(rel):
    """evaluation.py:506-513: mean, mean |x|, mean of the positive entries (0 if none), max"""
    pos = rel > 0
    mean_pos = 0.0 if pos.sum() == 0 else float(np.maximum(rel, 0).sum(dtype=np.float64) / pos.sum())
    return np.array([rel.mean(dtype=np.float64), np.abs(rel).mean(dtype=np.float64), mean_pos, rel.max()], np.float64)


def map_quantiles(rel, points=None):
    """evaluation.py:451, :510: np.quantile of the (H,W) map at i/100, i = 0..99"""
    return np.quantile(rel, [i / 100 for i in range(100)] if points is None else points)


def relevance_heatmap(m, lut, gamma=0.7):
    """LRPtools/utils.py `gamma` (:97-145, minamp 0, maxamp = max|X|) then `heatmap` (:67-90) with `project` (:34-52):
    m (C,H,W) -> (H,W,3) colours from the 256-entry table `lut`."""
    x = m.astype(np.float32)
    maxamp = np.abs(x).max()
    if maxamp != 0:
        xs = x / maxamp
        y = np.where(xs >= 0, np.abs(xs) ** np.float32(gamma), -(np.abs(xs) ** np.float32(gamma))).astype(np.float32) * maxamp
    else:
        y = x
    t = y.sum(axis=0)
    absmax = np.abs(t).max()
    if absmax != 0:
        t = t / absmax
    t = np.clip((t + 1) / 2, 0, 1) * 255
    return lut[t.astype(np.int64)]

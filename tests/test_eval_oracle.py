"""The CPU restatement of the relevance-map consumers (oracle/eval_oracle.py) against the outputs of the reference's own
methods (tests/golden/eval_consumers.npz: EvaluationExperiments.block_image / _calculate_overlaped_pixels /
_project_maxabs of /root/reference/evaluation.py on seeded maps)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import eval_oracle as E


def golden_maps(seed):
    rs = np.random.RandomState(seed)
    maps = (rs.standard_normal((2, 3, 224, 224)) * np.exp(2 * rs.standard_normal((2, 3, 224, 224)))).astype(np.float32)
    maps[1, :, :16, :] = 0
    maps[1, :, :, -24:] = 0
    return maps


def test_consumers_vs_reference():
    g = np.load(os.path.join(GOLDEN, "eval_consumers.npz"))
    maps = golden_maps(int(g["seed"]))
    for i in range(2):
        mask, sums = E.block_image(E.spatial_relevance(maps[i]))
        top = np.sort(sums)[::-1]
        assert top[19] - top[20] > 1e-3 * abs(top[19])          # the fixture has no near-tie at the selection boundary
        assert np.array_equal(mask.astype(np.uint8), g[f"mask_{i}"])
        assert int((mask == 0).sum()) == 20 * 64
        proj = E.project_maxabs(E.spatial_relevance(maps[i], "pos"))
        assert np.abs(proj - g[f"proj_{i}"]).max() < 1e-6
        for j, t in enumerate(g["thresholds"]):
            r = E.overlapped_pixels(list(g["boxes"][i]), g[f"proj_{i}"], float(t))
            assert abs(r - g[f"ratio_{i}"][j]) < 1e-6, (i, j)
    assert np.array_equal(E.project_maxabs(np.zeros((4, 4), np.float32)), np.zeros((4, 4), np.float32))
    st = E.map_statistics(np.array([[1.0, -3.0], [0.0, 2.0]], np.float32))
    assert np.allclose(st, [0.0, 1.5, 1.5, 2.0])


def test_heatmap_vs_reference():
    # LRPutil.heatmap(LRPutil.gamma(hm)) of /root/reference/LRPtools/utils.py on the seeded maps (every 2nd pixel stored)
    g = np.load(os.path.join(GOLDEN, "eval_consumers.npz"))
    maps = golden_maps(int(g["seed"]))
    for i in range(2):
        hm = E.relevance_heatmap(maps[i], g["lut"])
        d = np.abs(hm[::2, ::2] - g[f"heatmap_sub2_{i}"]).max(axis=-1)
        assert (d > 0).mean() < 1e-3 and d.max() < 0.05            # at most a LUT step on a few rounding boundaries

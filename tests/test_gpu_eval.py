"""GPU parity of the device-side relevance-map consumers (csrc/lrpx_eval.hip) through the C ABI,
against the outputs of the reference's own evaluation methods (tests/golden/eval_consumers.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN
from test_eval_oracle import golden_maps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import evaluation
    return evaluation


def test_block_image_and_overlap_vs_reference(ev):
    g = np.load(os.path.join(GOLDEN, "eval_consumers.npz"))
    maps = torch.from_numpy(golden_maps(int(g["seed"]))).cuda()
    masks = ev.block_image(ev.spatial_relevance(maps, "mean"), 8, 20).cpu().numpy()
    proj = ev.project_maxabs(ev.spatial_relevance(maps, "pos"))
    ratios = ev.overlapped_pixels(proj, torch.from_numpy(g["boxes"]), g["thresholds"].tolist()).cpu().numpy()
    for i in range(2):
        assert np.array_equal(masks[i].astype(np.uint8), g[f"mask_{i}"])        # bit-exact: integer-valued output
        assert np.abs(proj[i].cpu().numpy() - g[f"proj_{i}"]).max() < 1e-6
        assert np.abs(ratios[i] - g[f"ratio_{i}"]).max() < 1e-5


def test_statistics_and_edge_cases_vs_oracle(ev):
    from oracle import eval_oracle as E
    g = torch.Generator().manual_seed(5)
    maps = torch.randn(5, 3, 64, 96, generator=g) * torch.exp(torch.randn(5, 3, 64, 96, generator=g))
    maps[3] = 0                                     # all-zero map: project -> zeros, ratio 0, mean_pos 0
    maps[4] = -maps[4].abs()                        # no positive entry
    d = maps.cuda()
    for mode in ("mean", "pos", "neg"):
        sp = ev.spatial_relevance(d, mode).cpu().numpy()
        for i in range(5):
            assert np.abs(sp[i] - E.spatial_relevance(maps[i].numpy(), mode)).max() < 1e-5
    sp = ev.spatial_relevance(d, "mean")
    st = ev.map_statistics(sp).cpu().numpy()
    pr = ev.project_maxabs(sp).cpu().numpy()
    boxes = torch.tensor([[3, 5, 40, 60], [0, 0, 96, 64], [10, 10, 11, 11], [0, 0, 5, 5], [20, 0, 90, 30]])
    rt = ev.overlapped_pixels(ev.project_maxabs(ev.spatial_relevance(d, "pos")), boxes, [0, 0.25, 0.5]).cpu().numpy()
    mk = ev.block_image(sp, 8, 7).cpu().numpy()
    for i in range(5):
        spi = sp[i].cpu().numpy()
        assert np.allclose(st[i], E.map_statistics(spi), rtol=1e-5, atol=1e-6)
        assert np.abs(pr[i] - E.project_maxabs(spi)).max() < 1e-6
        rel = E.project_maxabs(E.spatial_relevance(maps[i].numpy(), "pos"))
        for j, t in enumerate([0, 0.25, 0.5]):
            assert abs(rt[i, j] - E.overlapped_pixels(boxes[i].tolist(), rel, t)) < 1e-5
        want, sums = E.block_image(spi, 8, 7)
        if i != 3:                                   # (an all-zero map is all ties: lower index first on both sides)
            assert np.array_equal(mk[i], want)
        assert int((mk[i] == 0).sum()) == 7 * 64
    assert rt[3].max() == 0.0 and st[3].max() == 0.0
    assert st[4, 2] == 0.0                           # mean_pos with no positive entry (evaluation.py:506-507)
    with pytest.raises(AssertionError):
        ev.block_image(torch.zeros(1, 30, 30, device="cuda"), 8, 2)             # evaluation.py:59-60 asserts
    with pytest.raises(ValueError):
        ev.block_image(torch.zeros(1, 16, 16, device="cuda"), 8, 5)             # more patches than exist (:65 assert)


def test_relevance_heatmap_vs_reference():
    """gamma + heatmap of LRPtools/utils.py in one kernel: colours equal the reference's except where powf rounding moves
    a value across an integer boundary of the 256-entry table (one table step, < 0.2 % of the pixels)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd.LRPtools import utils as U
    g = np.load(os.path.join(GOLDEN, "eval_consumers.npz"))
    maps = torch.from_numpy(golden_maps(int(g["seed"]))).cuda()
    hm = U.relevance_heatmap(maps, gamma=0.7, lut=torch.from_numpy(g["lut"])).cpu().numpy()
    assert hm.shape == (2, 224, 224, 3)
    for i in range(2):
        d = np.abs(hm[i, ::2, ::2] - g[f"heatmap_sub2_{i}"]).max(axis=-1)
        assert (d > 0).mean() < 2e-3 and d.max() < 0.05, ((d > 0).mean(), d.max())
    zero = U.relevance_heatmap(torch.zeros(1, 3, 8, 8, device="cuda"), lut=torch.from_numpy(g["lut"])).cpu().numpy()
    assert np.allclose(zero, g["lut"][127])            # all-zero map: (0+1)/2*255 = 127.5 -> entry 127 everywhere
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        return
    assert np.allclose(U.colormap_lut("seismic").cpu().numpy(), g["lut"], atol=1e-6)


def test_map_quantiles_vs_numpy():
    """100-point quantiles of the tpfp statistics (evaluation.py:451, :510): per-map sort + numpy's linear
    interpolation; maps with ties, a constant map, negative values, a non-square size, points 0 and 1"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import evaluation as ev
    from oracle import eval_oracle as E
    g = torch.Generator().manual_seed(11)
    sp = torch.randn(6, 224, 224, generator=g) * torch.logspace(-6, 3, 6).view(6, 1, 1)
    sp[1] = sp[1].round()                      # many ties
    sp[2] = 0.25                               # constant
    sp[3] = -sp[3].abs()                       # no positive entry
    sp[4, :100] = 0.0                          # a block of exact zeros (signed-zero keys)
    q = ev.map_quantiles(sp.cuda()).cpu().numpy()
    assert q.shape == (6, 100)
    for i in range(6):
        want = E.map_quantiles(sp[i].numpy())
        assert np.abs(q[i] - want).max() <= 2e-6 * sp[i].abs().max().item(), i
    pts = [0.0, 0.5, 0.999, 1.0]
    small = torch.randn(3, 7, 13, generator=g)
    got = ev.map_quantiles(small.cuda(), pts).cpu().numpy()
    for i in range(3):
        assert np.allclose(got[i], E.map_quantiles(small[i].numpy(), pts), rtol=2e-6, atol=1e-7)
    assert got[0, 0] == small[0].min().item() and got[0, 3] == small[0].max().item()
    with pytest.raises(ValueError):
        ev.map_quantiles(small.cuda(), [1.5])

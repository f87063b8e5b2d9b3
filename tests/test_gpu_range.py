"""Dynamic range of the forward trace (the "Z+ flush hole").

The reference stabilises EXACT zeros only (LRPtools/utils.py:16-18): a receptive field of tiny but non-zero inputs has a tiny
Z+ and its incoming relevance is REDISTRIBUTED over it - S = R / Z+ is finite and x W+ S sums back to R
(LRPtools/lrp_modules.py:124-150, utils.py:21-31).  A forward trace on fp16 split products behind one power-of-two scale per
image flushes inputs more than ~2^29 below the image's maximum: Z+ becomes 0 (or keeps a few bits), `div_safe0` then sets
S := 0 and the relevance of the region is DROPPED.  These tests build exactly that state - positive biases, an image patch
scaled by 2^-33 (so the layer above the patch is alive through its bias and carries relevance), relevance placed over the
patch - and run the GPU's OWN forward (nothing injected) in every conv mode against the CPU oracle.

Since round 6 the forward trace runs on the exact kernels in EVERY conv mode by default (fp32 MFMA for conv1_1, exact bf16
splits above: lrpx_set_forward_f16(0)); the fp16 forward is an explicit opt-in (`forward_f16 = 1`) and the last test documents
what it does to this state."""
import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu
PATCH = (slice(64, 160), slice(48, 176))      # rows, columns of the tiny region
SHIFT = 2.0 ** -33


def _state(variant):
    """VGG16 with strictly positive biases; variant 'conv1_2': conv1_1's bias is zero, so its OUTPUT is tiny over the patch too and
    the tiny operands reach the split-product forward kernel of conv1_2 (variant 'conv1_1': they stop at the first layer)"""
    from lrp_amd import weights
    sd = weights.make_gridtd_state(seed=21, vocab_size=64, vgg_bias_std=0.05)
    for k in sd:
        if k.startswith("img_encoder.encoder.") and k.endswith(".bias"):
            sd[k] = (np.abs(sd[k]) + 0.01).astype(np.float32)
    if variant == "conv1_2":
        sd["img_encoder.encoder.0.bias"] = np.zeros_like(sd["img_encoder.encoder.0.bias"])
    img = torch.from_numpy(weights.make_images(22, 1)).clone()
    img[:, :, PATCH[0], PATCH[1]] *= SHIFT
    return sd, img


def _target():
    """relevance at the encoder output over the feature positions that look at the patch (16 pixels per position)"""
    torch.manual_seed(3)
    r = torch.zeros(2, 512, 14, 14)
    r[0, :, 5:9, 4:10] = torch.rand(512, 4, 6)
    r[1, :, 4:10, 3:11] = torch.rand(512, 6, 8) * torch.randn(512, 6, 8).abs()
    return r


def _run(mode, variant, forward_f16=None):
    import test_gpu_vgg as TV
    from lrp_amd import ops
    from oracle import lrp_oracle as O
    sd, img = _state(variant)
    sdt = O.state_to_torch(sd)
    r_feat = _target()
    _, _, saved = O.vgg_forward(sdt, img)
    want = torch.cat([O.vgg_lrp(sdt, saved, r_feat[i:i + 1]) for i in range(2)])
    vgg = TV._vgg(ops, sd)
    vgg.conv_mode = mode
    vgg.forward_f16 = forward_f16
    vgg.forward(img.cuda())
    maps = vgg.relevance(TV.to_nhwc(r_feat).cuda(), torch.zeros(2, dtype=torch.int32, device="cuda")).cpu()
    return maps, want


def _patch_mass(m):
    return m[:, PATCH[0], PATCH[1]].abs().double().sum().item()


@pytest.mark.parametrize("variant", ["conv1_1", "conv1_2"])
@pytest.mark.parametrize("mode", [1, 0])
def test_tiny_region_keeps_its_relevance_in_the_exact_modes(mode, variant):
    """the default mode (1, exact bf16 splits) and the fp32 MFMA: operands keep fp32's exponent range end to end"""
    maps, want = _run(mode, variant)
    for i in range(2):
        # the oracle puts a large share of the map over the patch: the state is what the docstring says it is
        share = _patch_mass(want[i]) / want[i].abs().double().sum().item()
        assert share > 0.2, share
        ratio = _patch_mass(maps[i]) / _patch_mass(want[i])
        d = (maps[i] - want[i]).abs() / want[i].abs().max()
        inside = d[:, PATCH[0], PATCH[1]]
        print(f"mode {mode} {variant} map {i}: patch share of |R| {share:.2f}, GPU / oracle patch mass {ratio:.6f}, "
              f"worst pixel {d.max().item():.2e}, pixels over the patch off by > 1e-4: {(inside > 1e-4).float().mean().item():.2e}")
        assert abs(ratio - 1.0) < 1e-4, (mode, variant, i, ratio)
        assert (inside > 1e-4).float().mean().item() < 1e-3
        assert_close_modulo_pool_ties(maps[i:i + 1], want[i:i + 1], what=(mode, variant, i), frac=1e-2, hard=8e-3, l2=1.5e-3)


@pytest.mark.parametrize("variant", ["conv1_1", "conv1_2"])
@pytest.mark.parametrize("mode", [2, 3])
def test_speed_modes_keep_the_mass_but_not_1e4_on_a_2p33_range(mode, variant):
    """RANGE CONTRACT of the opt-in speed modes, pinned: their relevance pass scales S = R / Z+ by ONE power of two per map into the
    fp16 range (csrc/conv_f16x3.h:10-15), so entries more than ~2^29 below the map's maximum flush.  Here S over the patch is 2^33
    above S elsewhere: with the exact forward (the default) the patch's relevance is no longer dropped - its mass is right to 2e-3 -
    but the rest of the map is rounded against the patch's scale and the 1e-4 contract does NOT hold on this state (the exact modes
    above hold it).  A layer whose activations span less than ~2^16 within an image - every natural image - is inside the contract;
    include/lrpx.h (lrpx_set_conv_mode) states it."""
    maps, want = _run(mode, variant)
    worst = 0.0
    for i in range(2):
        ratio = _patch_mass(maps[i]) / _patch_mass(want[i])
        d = (maps[i] - want[i]).abs() / want[i].abs().max()
        worst = max(worst, d.max().item())
        print(f"mode {mode} {variant} map {i}: GPU / oracle patch mass {ratio:.6f}, worst pixel {d.max().item():.2e}")
        assert abs(ratio - 1.0) < 5e-3, (mode, variant, i, ratio)
        assert torch.isfinite(maps[i]).all()
    assert worst > 1e-4, "the speed modes now hold 1e-4 on a 2^33 spatial range: tighten this test and the contract in include/lrpx.h"


def test_opt_in_fp16_forward_rounds_the_region_away():
    """`forward_f16 = 1` (opt-in since round 6) on the same state: Z+ over the patch keeps a handful of bits (fp16 subnormals behind the
    per-image scale; 2^-40 instead of 2^-33 would flush it to zero and drop the relevance) - the documented range contract of that
    switch (include/lrpx.h, lrpx_set_forward_f16).  Pinned so that a change of the default cannot go unnoticed."""
    maps, want = _run(3, "conv1_1", forward_f16=1)
    d = (maps[0] - want[0]).abs() / want[0].abs().max()
    inside = d[:, PATCH[0], PATCH[1]]
    print(f"opt-in fp16 forward, mode 3: worst pixel over the patch {inside.max().item():.2e}")
    assert inside.max().item() > 1e-3

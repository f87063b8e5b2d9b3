"""GPU parity of the drop-in API surface: LRPtools hook API (add_lrp / compute_lrp, per-layer rule classes) and
the ExplainGridTDAttention class, against the reference's golden vectors."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, cosine, assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    g = np.load(os.path.join(GOLDEN, "gridtd_T3.npz"))
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    return g, sd, img


def _torch_vgg(sd):
    from lrp_amd.LRPtools.lrp_wrapper import VGG16_FEATURES
    mods, cin, idx = [], 3, 0
    for v in VGG16_FEATURES:
        if v == 'M':
            mods.append(nn.MaxPool2d(2, 2)); idx += 1
        else:
            c = nn.Conv2d(cin, v, 3, padding=1)
            c.weight.data = torch.from_numpy(sd[f"img_encoder.encoder.{idx}.weight"])
            c.bias.data = torch.from_numpy(sd[f"img_encoder.encoder.{idx}.bias"])
            mods += [c, nn.ReLU(inplace=True)]; idx += 2; cin = v
    return nn.Sequential(*mods).cuda().eval()


def test_add_lrp_compute_lrp_accumulates_like_reference(gold):
    """three compute_lrp calls on the SAME sample tensor: the reference returns running sums (sample.grad)."""
    from lrp_amd.LRPtools import lrp_wrapper
    g, sd, img = gold
    enc = _torch_vgg(sd)
    lrp_wrapper.add_lrp(enc)
    lrp_wrapper.add_lrp(enc)                      # idempotent (the reference would stack hooks)
    sample = img.cuda()
    for t in range(3):
        out = enc.compute_lrp(sample, target=torch.from_numpy(g[f"r_feat_{t}"]).cuda())
        assert out.shape == (1, 3, 224, 224) and out.device.type == "cuda"
        assert_close_modulo_pool_ties(out[..., ::4, ::4].cpu(), g[f"map_sub4_{t}"], what=t)
    assert_close_modulo_pool_ties(out.cpu(), g["map_full_2"], what="full")
    out2, logits = enc.compute_lrp(img.cuda(), target=torch.from_numpy(g["r_feat_0"]).cuda(), return_output=True)
    assert rel_err(logits.cpu(), g["features"]) < 1e-4
    with pytest.raises(AssertionError):           # lrp_wrapper.py:81 — all-zero relevance
        enc.compute_lrp(img.cuda(), target=torch.zeros(1, 512, 14, 14).cuda())


def test_rule_classes_on_reference_fixtures():
    """Conv2d (signed input, exact-zero region) and Pool2d (tie, zero window) rule classes against the outputs of
    the reference's own classes (layers.npz), embedded in a 14x14 / 28x28 zero canvas."""
    from lrp_amd.LRPtools import lrp_modules, lrp_wrapper
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    G = np.load(os.path.join(GOLDEN, "layers.npz"))
    params = lrp_wrapper.SequentialPresetA().lrp_params
    conv = nn.Conv2d(4, 6, 3, padding=1).cuda()
    conv.weight.data = torch.from_numpy(G["conv_w"]).cuda()
    X = torch.zeros(2, 4, 14, 14); X[:, :, :6, :6] = torch.from_numpy(G["conv_x"])
    R = torch.zeros(2, 6, 14, 14); R[:, :, :6, :6] = torch.from_numpy(G["conv_rout"])
    conv.input = (X.cuda(),)
    out = lrp_modules.get_lrp_module(conv).propagate_relevance(conv, (None, None, None), (R.cuda(),), "alpha_beta", params)
    assert len(out) == 3
    assert rel_err(out[0][:, :, :6, :6].cpu(), G["conv_rin"]) < 1e-4
    pool = nn.MaxPool2d(2, 2)
    pool.input = (torch.from_numpy(G["pool_x"]).cuda(),)
    rp = lrp_modules.get_lrp_module(pool).propagate_relevance(pool, None, (torch.from_numpy(G["pool_rout"]).cuda(),),
                                                              "alpha_beta", params)[0]
    assert torch.equal(rp.cpu(), torch.from_numpy(G["pool_rin"]))
    relu = nn.ReLU()
    r = torch.randn(2, 3, device="cuda")
    assert lrp_modules.get_lrp_module(relu).propagate_relevance(relu, None, (r,), "identity", params)[0] is r


def test_mini_network_layer_by_layer(gold):
    """conv-relu-conv-relu-pool-conv-relu of the reference fixture, walked backwards with the rule classes
    (what the reference's hooks do), 16x16 input embedded in 28x28."""
    from lrp_amd.LRPtools import lrp_modules, lrp_wrapper
    import torch.nn.functional as F
    G = np.load(os.path.join(GOLDEN, "layers.npz"))
    params = lrp_wrapper.SequentialPresetA().lrp_params
    T = lambda k: torch.from_numpy(G[k])
    def emb(x, hw):
        out = torch.zeros(x.shape[0], x.shape[1], hw, hw); out[:, :, :x.shape[2], :x.shape[3]] = x
        return out.cuda()
    x = T("mini_x")
    a0 = F.relu(F.conv2d(x, T("mini_w0"), T("mini_b0"), padding=1))
    a1 = F.relu(F.conv2d(a0, T("mini_w2"), T("mini_b2"), padding=1))
    p = F.max_pool2d(a1, 2, 2)
    def conv(w, inp):
        c = nn.Conv2d(w.shape[1], w.shape[0], 3, padding=1).cuda(); c.weight.data = w.cuda(); c.input = (inp,); return c
    c5, c2, c0 = conv(T("mini_w5"), emb(p, 14)), conv(T("mini_w2"), emb(a0, 28)), conv(T("mini_w0"), emb(x, 28))
    pool = nn.MaxPool2d(2, 2); pool.input = (emb(a1, 28),)
    r = emb(T("mini_target"), 14)
    r = lrp_modules.Conv2d().propagate_relevance(c5, (None, None, None), (r,), "alpha_beta", params)[0]
    r = lrp_modules.Pool2d().propagate_relevance(pool, None, (r,), "alpha_beta", params)[0]
    r = lrp_modules.Conv2d().propagate_relevance(c2, (None, None, None), (r,), "alpha_beta", params)[0]
    r = lrp_modules.Conv2d().propagate_relevance(c0, (None, None, None), (r,), "alpha_beta", params)[0]
    assert rel_err(r[:, :, :16, :16].cpu(), G["mini_r"]) < 1e-4


def _fixture_net(G):
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(inplace=True),
                        nn.MaxPool2d(2, 2), nn.Conv2d(8, 16, 3, padding=1), nn.ReLU(inplace=True))
    for i in (0, 2, 5):
        net[i].weight.data = torch.from_numpy(G[f"mini_w{i}"])
        net[i].bias.data = torch.from_numpy(G[f"mini_b{i}"])
    return net.cuda().eval()


def test_generic_add_lrp_on_the_reference_fixture_net():
    """`add_lrp(net)` + `net.compute_lrp(x, target=...)` on the reference's own 7-layer fixture net
    (make_golden.py:gen_layers, 16x16 / 8x8 maps, in-place ReLUs) - a NON-VGG leaf sequence - reproduces the reference's
    result `mini_r` (layers.npz) through the generic driver (recorded forward + reverse walk over the rule classes)."""
    from lrp_amd.LRPtools import lrp_wrapper
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    G = np.load(os.path.join(GOLDEN, "layers.npz"))
    net = _fixture_net(G)
    lrp_wrapper.add_lrp(net)
    lrp_wrapper.add_lrp(net)
    x = torch.from_numpy(G["mini_x"]).cuda()
    r = net.compute_lrp(x, target=torch.from_numpy(G["mini_target"]).cuda())
    assert r.shape == x.shape and r.device.type == "cuda"
    e = rel_err(r.cpu(), G["mini_r"])
    print(f"generic add_lrp, fixture net: {e:.2e}")
    assert e < 1e-4
    with pytest.raises(RuntimeError, match="Mismatch in shape"):
        net.compute_lrp(x, target=torch.zeros(1, 16, 4, 4).cuda())


def test_generic_add_lrp_on_a_residual_toy_net():
    """Conv-BN-ReLU, a skip connection through the explicit Add module, MaxPool, Flatten, Linear: every M4 rule reached
    through `add_lrp` / `compute_lrp`, against the reference's own add_lrp on the same net (toy_resnet.npz).  The output of
    relu1 feeds conv2 AND the Add: its relevance is the sum of both (autograd's accumulation in the reference); two calls
    on the same sample return the `.grad` running sum (lrp_wrapper.py:64-82)."""
    import sys
    from lrp_amd.LRPtools import lrp_wrapper, lrp_modules
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, GOLDEN)
    from make_golden import toy_resnet
    G = np.load(os.path.join(GOLDEN, "toy_resnet.npz"))
    net = toy_resnet(np.random.RandomState(int(G["seed"])), lrp_modules.resAdd, lrp_modules.resFlatten).cuda()
    lrp_wrapper.add_lrp(net)
    xs = torch.from_numpy(G["x"]).cuda()
    r1, logits = net.compute_lrp(xs, target=torch.from_numpy(G["target"]).cuda(), return_output=True)
    assert rel_err(logits.cpu(), G["logits"]) < 1e-5
    r2 = net.compute_lrp(xs, target=torch.from_numpy(G["target2"]).cuda())
    e1, e2 = rel_err(r1.cpu(), G["r1"]), rel_err(r2.cpu(), G["r2"])
    print(f"generic add_lrp, residual toy net: {e1:.2e} (first call), {e2:.2e} (running sum of two calls)")
    assert e1 < 1e-4 and e2 < 1e-4
    assert cosine(r1.cpu(), G["r1"]) > 0.99999

    class Functional(nn.Module):                 # a functional `+` between leaves: the rules cannot see it -> refused, not dropped
        def __init__(self):
            super().__init__()
            self.a, self.b, self.relu = nn.Conv2d(3, 8, 3, padding=1), nn.Conv2d(3, 8, 3, padding=1), nn.ReLU()

        def forward(self, x):
            return self.relu(self.a(x) + self.b(x))
    f = Functional().cuda().eval()
    lrp_wrapper.add_lrp(f)
    with pytest.raises(ValueError, match="functional"):
        f.compute_lrp(xs, target=torch.rand(2, 8, 14, 14).cuda())


def test_explainer_class_drop_in(gold):
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import ExplainGridTDAttention
    g, sd, img = gold
    V = int(g["V"])
    args = types.SimpleNamespace(embed_dim=512, hidden_dim=512, encoder='vgg16', weight='', save_path='/tmp',
                                 dataset='synthetic', height=224, width=224)
    ex = ExplainGridTDAttention(args, weights.make_word_map(V), model={k: torch.from_numpy(v) for k, v in sd.items()})
    maps, rws = ex.explain_caption(img, caption_encode=[int(c) for c in g["caption"]])
    assert len(maps) == 3 and maps[0].shape == (1, 3, 224, 224) and rws[2].shape == (3,)
    assert ex.caption_length == 3 and ex.predictions.shape == (3, V) and ex.alphas.shape == (3, 196)
    for t in range(3):
        assert np.abs(rws[t].cpu().numpy() - g[f"r_words_{t}"]).max() < 1e-5
        assert_close_modulo_pool_ties(maps[t][..., ::4, ::4].cpu(), g[f"map_sub4_{t}"], what=t)
        rf, rw = ex.explain_caption_wordt(t)
        assert rel_err(rf.cpu(), g[f"r_feat_{t}"]) < 1e-4 and rf.shape == (1, 512, 14, 14)
    # explain_cnn accumulates over calls on the same image exactly like the reference's sample.grad
    ex._img_grad = None
    for t in range(3):
        m = ex.explain_cnn(ex.explain_caption_wordt(t)[0])
    assert rel_err(m.cpu(), maps[2].cpu()) < 1e-5
    pred = ex.teacherforce_forward(img, [int(c) for c in g["caption"][:3]])
    assert rel_err(pred[:, ::97].cpu(), g["tr_predictions"]) < 1e-4

"""The drop-in explainer classes called the way the reference's evaluation experiments call them:

    evaluation.py:107,350,471   relevance_imgs, relevance_words = self.explainer.explain_caption(img_filepath)          (gridTD)
    evaluation.py:637           ... = self.explainer.explain_caption(img_filepath, head_idx)                            (AoA)
    evaluation.py:266,437,702   self.explainer.teacherforce_forward(self.explainer.img.detach().clone(), beam_caption_encoded)
    evaluation.py:767           self.explainer.get_hidden_parameters(img_filepath); self.explainer.alphas[t][head_idx]
    models/aoamodel.py:1183     explain_caption_words(img_filepath)

An image FILE goes in: a seeded uint8 picture written to tmp_path as PNG.  `preprocess_img` (models/gridTDmodel.py:767-771,
models/aoamodel.py:864-868: PIL -> Resize -> ToTensor -> Normalize) is checked against the same arithmetic done by hand (torchvision
is not installed here, so the reference's own transform cannot run: the Resize of a PIL image is PIL's bilinear resize, which is what
both sides call); the explanation of the file equals the explanation of the preprocessed tensor bit for bit, and meets the CPU
oracle on that tensor.  `teacherforce_forward` of the four explainer families against the reference's own outputs
(tests/golden/teacherforce.npz)."""
import os
import types

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu
MEAN, STD = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)


def _args(**kw):
    d = dict(embed_dim=512, hidden_dim=512, encoder='vgg16', weight='', save_path='/tmp', dataset='synthetic', height=224,
             width=224, num_head=8)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _picture(path, seed, h=224, w=224):
    """a seeded, smooth-ish uint8 RGB picture saved as PNG; returns the (1,3,224,224) tensor `preprocess_img` must produce"""
    from PIL import Image
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    im = Image.fromarray(base, "RGB").resize((w, h), Image.BICUBIC)
    im.save(path)
    im = Image.open(path).convert("RGB").resize((224, 224), Image.BILINEAR)      # transforms.Resize((224, 224)) on a PIL image
    x = np.asarray(im, dtype=np.float32) / 255.0                                   # transforms.ToTensor
    x = (x - MEAN) / STD                                                           # transforms.Normalize
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).unsqueeze(0)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_gridtd_explain_caption_of_an_image_file(gpu, tmp_path):
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import ExplainGridTDAttention, ExplainiGridTDGuidedGradient
    from oracle import lrp_oracle as O
    V = 467
    sd = weights.make_gridtd_state(seed=5, vocab_size=V)
    wm = weights.make_word_map(V)
    path = str(tmp_path / "picture.png")
    want_img = _picture(path, 11, 300, 260)                      # not 224 x 224: the Resize runs
    ex = ExplainGridTDAttention(_args(), wm, model=sd)
    assert torch.equal(ex.preprocess_img(path).cpu(), want_img)
    # evaluation.py:107 - the explainer captions the file itself (beam 2, up to 50 words) and explains that caption
    maps, words = ex.explain_caption(path)
    n = ex.caption_length
    assert ex.img_filepath == path and torch.equal(ex.img.cpu(), want_img)
    assert len(maps) == n == len(words) == len(ex.beam_caption_encode) - 1 and ex.beam_caption_encode[0] == wm['<start>']
    cap = list(ex.beam_caption_encode)
    maps = [m.clone() for m in maps]
    if n:
        assert maps[0].shape == (1, 3, 224, 224) and words[-1].shape == (n,) and ex.predictions.shape == (n, V)
        pred = ex.teacherforce_forward(ex.img.detach().clone(), ex.beam_caption_encode)          # evaluation.py:266
        assert pred.shape == (n + 1, V) and rel_err(pred[:n].cpu(), ex.predictions.cpu()) < 1e-4
    # the same picture handed over as a tensor: the same caption, the same maps, bit for bit
    ex2 = ExplainGridTDAttention(_args(), wm, model=sd)
    maps2, words2 = ex2.explain_caption(want_img)
    assert ex2.beam_caption_encode == cap and all(torch.equal(a, b) for a, b in zip(maps, maps2))
    # a given 3-word caption on the file against the CPU oracle on the preprocessed tensor
    cap3 = [wm['<start>']] + [int(c) for c in weights.make_captions(12, 1, 3, V)[0][1:]]
    maps3, words3 = ex.explain_caption(path, caption_encode=cap3)
    w_maps, w_rw = O.gridtd_explain_caption(O.state_to_torch(sd), want_img, np.array(cap3))
    for t in range(3):
        assert_close_modulo_pool_ties(maps3[t].cpu(), w_maps[t], what=("file", t))
        assert np.abs(words3[t].cpu().numpy() - w_rw[t].numpy()).max() < 1e-4
    # the guided-backprop explainer through the same door (evaluation.py:639 `explain_caption(img_filepath)`)
    gb = ExplainiGridTDGuidedGradient(_args(), wm, model=sd)
    gmaps, _ = gb.explain_caption(path, caption_encode=cap3)
    gmaps2, _ = ExplainiGridTDGuidedGradient(_args(), wm, model=sd).explain_caption(want_img, caption_encode=cap3)
    assert len(gmaps) == 3 and all(torch.equal(a, b) for a, b in zip(gmaps, gmaps2))


def test_aoa_explain_caption_of_an_image_file(gpu, tmp_path):
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import ExplainAOAAttention, ExplainAOAGradient
    from oracle import lrp_oracle as O
    V, head = 479, 3
    sd = weights.make_aoa_state(seed=6, vocab_size=V)
    wm = weights.make_word_map(V)
    path = str(tmp_path / "picture.png")
    want_img = _picture(path, 13)
    ex = ExplainAOAAttention(_args(), wm, model=sd)
    assert torch.equal(ex.preprocess_img(path).cpu(), want_img)
    maps, words = ex.explain_caption(path, head)                                  # evaluation.py:637, literally
    n = ex.caption_length
    assert ex.img_filepath == path and torch.equal(ex.img.cpu(), want_img) and len(maps) == n == len(words)
    cap = list(ex.beam_caption_encode)
    maps = [m.clone() for m in maps]
    if n:
        assert ex.alphas.shape == (n, 8, 196) and ex.predictions.shape == (n, V)
        pred = ex.teacherforce_forward(ex.img.detach().clone(), ex.beam_caption_encode)          # evaluation.py:702
        assert pred.shape == (n + 1, V) and rel_err(pred[:n].cpu(), ex.predictions.cpu()) < 1e-4
    ex2 = ExplainAOAAttention(_args(), wm, model=sd)
    maps2, _ = ex2.explain_caption(want_img, head)
    assert ex2.beam_caption_encode == cap and all(torch.equal(a, b) for a, b in zip(maps, maps2))
    # evaluation.py:767-776: the trace alone, then the attention of one head
    ex2.get_hidden_parameters(path)
    assert ex2.beam_caption_encode == cap and ex2.img_filepath == path
    if n:
        assert ex2.alphas[n - 1][head].shape == (196,) and abs(ex2.alphas[n - 1][head].sum().item() - 1.0) < 1e-4
    # models/aoamodel.py:1183: linguistic relevance of the file, head 0
    rw = ex.explain_caption_words(path)
    assert len(rw) == n
    # a given caption on the file against the CPU oracle on the preprocessed tensor
    cap3 = [wm['<start>']] + [int(c) for c in weights.make_captions(14, 1, 3, V)[0][1:]]
    maps3, words3 = ex.explain_caption(path, head, caption_encode=cap3)
    w_maps, w_rw = O.aoa_explain_caption(O.state_to_torch(sd), want_img, np.array(cap3), head)
    for t in range(3):
        assert_close_modulo_pool_ties(maps3[t].cpu(), w_maps[t], what=("file", t))
        assert np.abs(words3[t].cpu().numpy() - w_rw[t].numpy()).max() < 1e-4
    rw3 = ex.explain_caption_words(path, caption_encode=cap3)
    assert len(rw3) == 3
    sdt = O.state_to_torch(sd)
    feats, _, _ = O.vgg_forward(sdt, want_img)
    tr = O.aoa_trace(sdt, feats[0].reshape(512, -1).t().contiguous(), np.array(cap3))
    for t in range(3):
        assert np.abs(rw3[t].cpu().numpy() - O.aoa_explain_wordt(sdt, tr, t, 0)[1].numpy()).max() < 1e-4
    # the gradient explainer through the same door
    g = ExplainAOAGradient(_args(), wm, model=sd)
    gm, _ = g.explain_caption(path, head, caption_encode=cap3)
    gm2, _ = ExplainAOAGradient(_args(), wm, model=sd).explain_caption(want_img, head, caption_encode=cap3)
    assert len(gm) == 3 and all(torch.equal(a, b) for a, b in zip(gm, gm2)) and g.img_filepath == path


def test_teacherforce_forward_of_the_four_explainer_families_vs_reference(gpu):
    """tests/golden/teacherforce.npz: the reference's `teacherforce_forward` (models/gridTDmodel.py:892-931, :1282-1321;
    models/aoamodel.py:952-988, :1377-1413) after `get_hidden_parameters`, caption incl. <start> (evaluation.py:702): scores within
    1e-4 of their maximum, arg-max ids bit-exact.  The LRP explainers add bias_ih twice in the LanguageLSTM, the gradient family
    does not: each class must reproduce ITS forward."""
    from lrp_amd import weights
    from lrp_amd.explainers import aoa, gridtd
    g = np.load(os.path.join(GOLDEN, "teacherforce.npz"))
    seed = int(g["seed"])
    img = torch.from_numpy(weights.make_images(seed, 1))
    for tag, make_state, classes in (("grid", weights.make_gridtd_state, (gridtd.ExplainGridTDAttention, gridtd.ExplainGridTDGradient)),
                                     ("aoa", weights.make_aoa_state, (aoa.ExplainAOAAttention, aoa.ExplainAOAGradient))):
        V = int(g[f"{tag}_lrp_V"])
        sd = make_state(seed=seed, vocab_size=V)
        wm = weights.make_word_map(V)
        got = {}
        for fam, cls in zip(("lrp", "grad"), classes):
            k = f"{tag}_{fam}"
            ex = cls(_args(), wm, model=sd)
            cap = [int(c) for c in g[f"{k}_caption"]]
            ex.get_hidden_parameters(img, caption_encode=cap)
            pred = ex.teacherforce_forward(ex.img.detach().clone(), ex.beam_caption_encode).cpu()
            got[fam] = pred
            assert tuple(pred.shape) == (len(cap), V)
            scale = float(g[f"{k}_absmax"])
            e = max((pred[:, ::97] - torch.from_numpy(g[f"{k}_pred_sub"])).abs().max().item(),
                    (pred[-1] - torch.from_numpy(g[f"{k}_pred_last"])).abs().max().item()) / scale
            print(f"teacherforce_forward {k}: {e:.2e} of the largest score")
            assert e < 1e-4, (k, e)
            assert np.array_equal(pred.argmax(-1).numpy(), g[f"{k}_argmax"]), k
        assert (got["lrp"] - got["grad"]).abs().max().item() > 1e-3

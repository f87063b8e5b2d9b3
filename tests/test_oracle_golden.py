"""Pin the CPU oracle against END-TO-END outputs of the reference explainers
(tests/golden/{gridtd_T3,aoa_T3,aoa_bu_T3,greedy_cfg1}.npz, made by make_golden.py from
models/gridTDmodel.py:1141 `explain_caption`, models/aoamodel.py:1165, and
models/gridTDmodel.py:480 `greedy_search`).  Weights/inputs are regenerated from the seeded
generator (lrp_amd.weights) — only inputs' seeds and the reference's outputs are stored."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from lrp_amd import weights
from conftest import GOLDEN, rel_err, cosine
from oracle import lrp_oracle as O

TOL_REL = 1e-4      # BASELINE.json: max|dR|/max|R_ref| <= 1e-4
TOL_WORDS = 1e-5    # r_words max abs diff


@pytest.fixture(scope="module")
def gridtd():
    g = np.load(os.path.join(GOLDEN, "gridtd_T3.npz"))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    torch.set_num_threads(8)
    maps, rws, rfs, tr = O.gridtd_explain_caption(sd, img, g["caption"], return_feat=True)
    return g, maps, rws, rfs, tr


def test_gridtd_trace(gridtd):
    g, maps, rws, rfs, tr = gridtd
    pairs = dict(h1t="h1", c1t="c1", h2t="h2", c2t="c2", g1t="g1", g2t="g2", i1t_act="i1", f1t_act="f1",
                 i2t_act="i2", f2t_act="f2", st="s", context="ctx", context_hat="ctx_hat", alphas="alpha",
                 betas="beta")
    for ref_name, mine in pairs.items():
        assert rel_err(tr[mine], g["tr_" + ref_name]) < 2e-5, ref_name
    assert rel_err(tr["pred"][:, ::97], g["tr_predictions"]) < 2e-5


def test_gridtd_feature_relevance_and_words(gridtd):
    g, maps, rws, rfs, tr = gridtd
    for t in range(3):
        assert rel_err(rfs[t], g[f"r_feat_{t}"]) < TOL_REL
        assert cosine(rfs[t], g[f"r_feat_{t}"]) > 0.99999
        assert np.abs(rws[t].numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS


def test_gridtd_pixel_maps(gridtd):
    g, maps, rws, rfs, tr = gridtd
    for t in range(3):
        sub = maps[t][..., ::4, ::4]
        scale = g[f"map_stats_{t}"][1]
        assert np.abs(sub.numpy() - g[f"map_sub4_{t}"]).max() / scale < TOL_REL
        assert abs(maps[t].double().sum().item() - g[f"map_stats_{t}"][0]) < 1e-3 * abs(g[f"map_stats_{t}"][0]) + 1e-9
    assert rel_err(maps[2], g["map_full_2"]) < TOL_REL
    assert cosine(maps[2], g["map_full_2"]) > 0.99999
    assert (maps[2] - torch.from_numpy(g["map_full_2"])).abs().max() < 1e-4      # BASELINE absolute bound


def test_aoa_heads():
    g = np.load(os.path.join(GOLDEN, "aoa_T3.npz"))
    sd = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    for hd in g["heads"]:
        maps, rws, rfs, tr = O.aoa_explain_caption(sd, img, g["caption"], int(hd), return_feat=True)
        for t in range(3):
            assert rel_err(rfs[t], g[f"h{hd}_r_feat_{t}"]) < TOL_REL
            assert np.abs(rws[t].numpy() - g[f"h{hd}_r_words_{t}"]).max() < TOL_WORDS
            scale = g[f"h{hd}_map_stats_{t}"][1]
            assert np.abs(maps[t][..., ::4, ::4].numpy() - g[f"h{hd}_map_sub4_{t}"]).max() / scale < TOL_REL
        if f"h{hd}_map_full_2" in g:
            assert rel_err(maps[2], g[f"h{hd}_map_full_2"]) < TOL_REL
    assert rel_err(tr["alpha"], g["tr_alphas"]) < 2e-5
    assert rel_err(tr["h"], g["tr_ht"]) < 2e-5


def test_aoa_bottom_up_regions():
    # config 5: 36x2048 region features, no CNN stage (SURVEY §8(a) row A-BU)
    g = np.load(os.path.join(GOLDEN, "aoa_bu_T3.npz"))
    sd = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"]), feat_dim=2048,
                                                 with_encoder=False))
    feats = torch.from_numpy(weights.make_bu_features(int(g["seed"]), 1)[0])
    tr = O.aoa_trace(sd, feats, g["caption"])
    for t in range(3):
        rf, rw = O.aoa_explain_wordt(sd, tr, t, int(g["head"]))
        assert rel_err(rf, g[f"r_feat_{t}"]) < TOL_REL
        assert np.abs(rw.numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS


def test_greedy_tokens_bit_exact():
    # config 1: token ids are integers -> bit-exact
    g = np.load(os.path.join(GOLDEN, "greedy_cfg1.npz"))
    V = int(g["V"])
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    toks = O.gridtd_model_greedy(sd, img, max_cap_length=len(g["tokens"]), start_id=V - 2, end_id=V - 1)
    assert toks == [int(x) for x in g["tokens"]]


def test_sample_lrp_tokens_bit_exact():
    # LRP-inference decoding, `GridTDModel.sample_lrp` greedy (models/gridTDmodel.py:631-702): token ids bit-exact,
    # log-probabilities to 1e-4 absolute.  Case 2 makes one word a stop word (exempt from the re-weighting) and
    # lets another play <end> (zero padding after it).
    g = np.load(os.path.join(GOLDEN, "sample_lrp.npz"))
    V, L = int(g["V"]), int(g["max_len"])
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 3, g["seq"].shape[0]))
    wm = weights.make_word_map(V)
    for seq_k, lp_k, skip_k, end_id in (("seq", "logprobs", "skip", wm['<end>']),
                                        ("seq2", "logprobs2", "skip2", int(g["end2"]))):
        for b in range(imgs.shape[0]):
            seq, lps = O.gridtd_sample_lrp(sd, imgs[b:b + 1], L, wm['<start>'], end_id, set(g[skip_k].tolist()))
            assert seq == g[seq_k][b].tolist()
            assert np.abs(np.array(lps) - g[lp_k][b]).max() < 1e-4


def test_aoa_sample_lrp_tokens_bit_exact():
    # `AOAModel.sample_lrp` greedy (models/aoamodel.py:679-745; the rule sees log-softmax scores, :721-723)
    g = np.load(os.path.join(GOLDEN, "aoa_sample_lrp.npz"))
    V, L = int(g["V"]), int(g["max_len"])
    sd = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 3, g["seq"].shape[0]))
    wm = weights.make_word_map(V)
    for seq_k, lp_k, skip_k, end_id in (("seq", "logprobs", "skip", wm['<end>']),
                                        ("seq2", "logprobs2", "skip2", int(g["end2"]))):
        seq, lps = O.aoa_sample_lrp(sd, imgs[:1], L, wm['<start>'], end_id, set(g[skip_k].tolist()))
        assert seq == g[seq_k][0].tolist()
        assert np.abs(np.array(lps) - g[lp_k][0]).max() < 1e-4


def test_guided_backprop_vs_reference():
    # ExplainiGridTDGuidedGradient (models/gridTDmodel.py:1585-1723): decoder BPTT + guided backprop through VGG16
    g = np.load(os.path.join(GOLDEN, "guided_T3.npz"))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    maps, rws, dfs, tr = O.gridtd_guided_explain_caption(sd, img, g["caption"], return_feat=True)
    assert rel_err(tr["sen_gate"], g["tr_sen_gate"]) < 2e-5 and rel_err(tr["o2"], g["tr_o2t_act"]) < 2e-5
    assert rel_err(tr["h2"], g["tr_h2t"]) < 2e-5
    for t in range(3):
        assert rel_err(dfs[t], g[f"d_feat_{t}"]) < TOL_REL
        assert np.abs(rws[t].numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS
        scale = g[f"map_stats_{t}"][1]
        assert np.abs(maps[t][..., ::4, ::4].numpy() - g[f"map_sub4_{t}"]).max() / scale < TOL_REL
    assert rel_err(maps[2], g["map_full_2"]) < TOL_REL


def test_plain_gradient_vs_reference():
    # ExplainGridTDGradient (models/gridTDmodel.py:1214-1539): decoder BPTT without the guided gates + autograd
    # gradient through VGG16
    g = np.load(os.path.join(GOLDEN, "gradient_T3.npz"))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    maps, rws, dfs, tr = O.gridtd_gradient_explain_caption(sd, img, g["caption"], return_feat=True)
    for t in range(3):
        assert rel_err(dfs[t], g[f"d_feat_{t}"]) < TOL_REL
        assert np.abs(rws[t].numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS
        scale = g[f"map_stats_{t}"][1]
        assert np.abs(maps[t][..., ::4, ::4].numpy() - g[f"map_sub4_{t}"]).max() / scale < TOL_REL
    assert rel_err(maps[2], g["map_full_2"]) < TOL_REL


def test_grad_cam_vs_reference():
    # ExplainGridTDGradCam (models/gridTDmodel.py:1752-1771): (1,196) heat map per word, in [0,1]; word 1 of the
    # fixture is the all-negative case (clamp -> all zeros)
    g = np.load(os.path.join(GOLDEN, "gradcam_T3.npz"))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    cams, rws = O.gridtd_gradient_explain_caption(sd, img, g["caption"], cam=True)
    for t in range(3):
        assert tuple(cams[t].shape) == tuple(g[f"cam_{t}"].shape) == (1, 196)
        assert np.abs(cams[t].numpy() - g[f"cam_{t}"]).max() < 1e-4
        assert np.abs(rws[t].numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS
    assert g["cam_1"].max() == 0.0 and cams[1].max().item() == 0.0


def test_aoa_gradient_family_vs_reference():
    # ExplainAOAGradient / ExplainAOAGuidedGradient / ExplainAOAGradCam (models/aoamodel.py:1257-1711), one head
    g = np.load(os.path.join(GOLDEN, "aoa_gradient_T3.npz"))
    head = int(g["head"])
    sd = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    maps, rws, dfs, tr = O.aoa_gradient_explain_caption(sd, img, g["caption"], head, "gradient", return_feat=True)
    assert rel_err(tr["o"], g["tr_ot_act"]) < 2e-5 and rel_err(tr["h"], g["tr_ht"]) < 2e-5
    for t in range(3):
        assert rel_err(dfs[t], g[f"d_feat_{t}"]) < TOL_REL
        assert np.abs(rws[t].numpy() - g[f"r_words_{t}"]).max() < TOL_WORDS
        scale = g[f"grad_map_stats_{t}"][1]
        assert np.abs(maps[t][..., ::4, ::4].numpy() - g[f"grad_map_sub4_{t}"]).max() / scale < TOL_REL
    gmaps, _ = O.aoa_gradient_explain_caption(sd, img, g["caption"], head, "guided")
    cams, _ = O.aoa_gradient_explain_caption(sd, img, g["caption"], head, "gradcam")
    for t in range(3):
        scale = g[f"guided_map_stats_{t}"][1]
        assert np.abs(gmaps[t][..., ::4, ::4].numpy() - g[f"guided_map_sub4_{t}"]).max() / scale < TOL_REL
        assert np.abs(cams[t].numpy() - g[f"cam_{t}"]).max() < 1e-4


def _t20_rows(g, prefix, k, T, stride, get, tol_words, layout):
    for t in range(T):
        rf, rw = get(t)                                               # (P,C), (t+1,)
        st = g[f"{prefix}{k}_r_feat_stats_{t}"]
        sub = torch.from_numpy(g[f"{prefix}{k}_r_feat_sub_{t}"]).double()
        want = sub.reshape(sub.shape[0], -1).t() if layout == "chw" else sub
        assert ((rf.double()[:, (t % stride)::stride] - want).abs().max() / st[1]).item() < TOL_REL, (prefix, k, t)
        assert abs(rf.double().norm().item() - st[2]) <= 1e-4 * st[2]
        assert np.abs(rw.numpy() - g[f"{prefix}{k}_r_words_{t}"]).max() < tol_words, (prefix, k, t)


def test_t20_decoder_relevance_at_headline_length():
    # T = 20 (BASELINE configs 2 / 3 / 5): the reference's explain_caption_wordt for every word of two images per model
    # (tests/golden/t20.npz; models/gridTDmodel.py:1014-1135, models/aoamodel.py:1064-1156) against the oracle's decoder
    # relevance; the VGG16 forward is the oracle's own (same oneDNN convs as the reference's)
    g = np.load(os.path.join(GOLDEN, "t20.npz"))
    T, n_img = int(g["T"]), int(g["n_img"])
    torch.set_num_threads(8)
    imgs = torch.from_numpy(weights.make_images(int(g["img_seed"]), n_img))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["grid_V"])))
    sda = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["aoa_V"])))
    for k in range(n_img):
        feats, avg, _ = O.vgg_forward(sd, imgs[k:k + 1])
        tr = O.gridtd_trace(sd, feats[0], avg[0], g["grid_caption"][k])
        _t20_rows(g, "grid", k, T, 32, lambda t: O.gridtd_explain_wordt(sd, tr, t), TOL_WORDS, "chw")
        F_pix = feats[0].reshape(512, -1).t().contiguous()            # the two models share the seeded VGG16 weights
        tra = O.aoa_trace(sda, F_pix, g["aoa_caption"][k])
        gh = {key.replace(f"aoa{k}_h0_", f"aoa{k}_"): g[key] for key in g.files if key.startswith(f"aoa{k}_h0_")}
        # AoA r_words at T = 20: the normalising entry is a 512-term sum with heavy cancellation; the REFERENCE's own fp32
        # result is up to 4.3e-5 away from the fp64 evaluation of its formula on the same trace (image 0, words 17 / 19:
        # reference 1.0e-5 / 4.3e-5, this oracle 1.3e-5 / 4.3e-5 from fp64; 2.4e-5 / 5e-6 from each other), so 1e-5 is
        # not a property of the reference at this length: bound 1e-4 (gridTD stays at 1e-5: worst 1.7e-6)
        _t20_rows(gh, "aoa", k, T, 32, lambda t: O.aoa_explain_wordt(sda, tra, t, 0), 1e-4, "chw")
    sdb = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["aoa_V"]), feat_dim=2048,
                                                  with_encoder=False))
    bu = weights.make_bu_features(int(g["img_seed"]), n_img)
    for k in range(n_img):
        trb = O.aoa_trace(sdb, torch.from_numpy(bu[k]), g["bu_caption"][k])
        _t20_rows(g, "bu", k, T, 64, lambda t: O.aoa_explain_wordt(sdb, trb, t, 0), 1e-4, "pc")


def test_forwardlrp_context_forward_values():
    # `forwardlrp_context` of both models (models/gridTDmodel.py:579-630, models/aoamodel.py:628-677):
    # raw and LRP-reweighted scores of every teacher-forced step; case "2" has a stop word (weights of 1)
    g = np.load(os.path.join(GOLDEN, "forwardlrp.npz"))
    B, L = int(g["batch"]), int(g["grid_L"])
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 5, B))
    for tag, mk, fn in (("grid", weights.make_gridtd_state, O.gridtd_forwardlrp_context),
                        ("aoa", weights.make_aoa_state, O.aoa_forwardlrp_context)):
        sd = O.state_to_torch(mk(seed=int(g["seed"]), vocab_size=int(g[f"{tag}_V"])))
        for sfx in ("", "2") if tag == "grid" else ("",):
            for b in range(B):
                p, wp = fn(sd, imgs[b:b + 1], g[f"{tag}_caption"][b], L, set(g[f"{tag}_skip{sfx}"].tolist()))
                assert rel_err(p[:, ::13], g[f"{tag}_pred_sub{sfx}"][b]) < 2e-5
                assert rel_err(wp[:, ::13], g[f"{tag}_wpred_sub{sfx}"][b]) < 2e-5
                assert p.argmax(-1).tolist() == g[f"{tag}_pred_argmax{sfx}"][b].tolist()
                assert wp.argmax(-1).tolist() == g[f"{tag}_wpred_argmax{sfx}"][b].tolist()
            assert rel_err(p[L - 1], g[f"{tag}_pred_row{sfx}"]) < 2e-5 and rel_err(wp[L - 1], g[f"{tag}_wpred_row{sfx}"]) < 2e-5
    # the stop-word case: where the arg-max word is exempt the two score sets coincide
    assert np.array_equal(g["grid_pred_sub2"][g["grid_pred_argmax2"] == g["grid_skip2"][-1]],
                          g["grid_wpred_sub2"][g["grid_pred_argmax2"] == g["grid_skip2"][-1]])


def test_guided_grad_cam_vs_reference():
    # ExplainGridTDGuidedGradCam (models/gridTDmodel.py:1796-1836) / ExplainAOAGuidedGradCam (models/aoamodel.py:1714-1751):
    # guided backprop x expanded Grad-CAM.  The fixture was made by the reference's classes with the one skimage call served
    # by O.pyramid_expand (skimage is not installed: that step is "parity unpinned"); word 1 of gridTD is the all-negative
    # Grad-CAM case (zero map)
    g = np.load(os.path.join(GOLDEN, "guided_gradcam_T3.npz"))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["grid_V"])))
    feats, _, _ = O.vgg_forward(sd, img)
    maps, rws, dfs, _ = O.gridtd_guided_explain_caption(sd, img, g["grid_caption"], return_feat=True)
    sda = O.state_to_torch(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["aoa_V"])))
    amaps, arws, adfs, _ = O.aoa_gradient_explain_caption(sda, img, g["aoa_caption"], int(g["head"]), "guided", return_feat=True)
    for tag, mm, rr, dd in (("grid", maps, rws, dfs), ("aoa", amaps, arws, adfs)):
        for t in range(int(g["T"])):
            got = O.guided_grad_cam(feats, dd[t], mm[t])
            scale = g[f"{tag}_map_stats_{t}"][1]
            if scale == 0:
                assert got.abs().max().item() == 0
                continue
            assert np.abs(got[..., ::4, ::4].numpy() - g[f"{tag}_map_sub4_{t}"]).max() / scale < TOL_REL, (tag, t)
            assert np.abs(rr[t].numpy() - g[f"{tag}_r_words_{t}"]).max() < TOL_WORDS
        assert rel_err(O.guided_grad_cam(feats, dd[2], mm[2]), g[f"{tag}_map_full_2"]) < TOL_REL
    assert g["grid_map_stats_1"][1] == 0


def test_pyramid_expand_restatement_properties():
    # no skimage output exists to pin O.pyramid_expand against; what can be checked: shape, that a constant map stays that
    # constant (both steps are normalised interpolations), symmetry under flips / transposes (mirror borders), linearity, and
    # that the one-matrix form the GPU uses (ops.pyramid_expand_matrix: E = M cam M^T) is the same operator
    from lrp_amd import ops
    rs = np.random.RandomState(4)
    cam = torch.from_numpy(np.maximum(rs.standard_normal((14, 14)), 0).astype(np.float32))
    e = O.pyramid_expand(cam, 16)
    assert tuple(e.shape) == (224, 224)
    assert (O.pyramid_expand(torch.full((14, 14), 0.37), 16) - 0.37).abs().max() < 1e-6
    assert (O.pyramid_expand(cam.flip(0), 16) - e.flip(0)).abs().max() < 1e-6
    assert (O.pyramid_expand(cam.t(), 16) - e.t()).abs().max() < 1e-6
    assert (O.pyramid_expand(2.5 * cam, 16) - 2.5 * e).abs().max() < 1e-6
    M = ops.pyramid_expand_matrix(14, 16, "cpu")
    assert ((M @ cam @ M.t()) - e).abs().max() < 1e-6
    assert e.min() >= 0 and e.max() <= cam.max()           # smoothing + interpolation never overshoot


def _beam_cases(g, tag):
    V = int(g[f"{tag}_V"])
    wm = weights.make_word_map(V)
    cases = [("sen", wm)]
    for key, tok, name in (("sen_end", "end2", '<end>'), ("sen_end0", "end0", '<end>'), ("sen_unk", "unk2", '<unk>')):
        w2 = dict(wm)
        w2[name] = int(g[f"{tag}_{tok}"])
        cases.append((key, w2))
    return V, cases


def test_beam_search_captions_vs_reference():
    # the captions the explainers explain when none is given (models/gridTDmodel.py:935 beam 2 / 50 steps; aoamodel.py:992
    # beam 3 / 20 steps): token ids of the reference's own beam_search, bit-exact, incl. complete sequences (<end> reached),
    # an empty caption (<end> first) and the dropped <unk>
    g = np.load(os.path.join(GOLDEN, "beam.npz"))
    img = torch.from_numpy(weights.make_images(int(g["seed"]) + 7, 1))
    for tag, mk, fn in (("grid", weights.make_gridtd_state, O.gridtd_beam_caption), ("aoa", weights.make_aoa_state, O.aoa_beam_caption)):
        V, cases = _beam_cases(g, tag)
        sd = O.state_to_torch(mk(seed=int(g["seed"]), vocab_size=V))
        for key, wm in cases:
            cap, _ = fn(sd, img, int(g[f"{tag}_beam"]), int(g[f"{tag}_steps"]), wm)
            assert cap[1:] == g[f"{tag}_{key}"].tolist(), (tag, key)
    assert len(g["grid_sen_end0"]) == 0 and len(g["grid_sen"]) == 19


def test_t20_guided_decoder_vs_reference():
    # Guided-Backprop decoder BPTT at the headline caption length (tests/golden/t20_guided.npz: the reference's
    # ExplainiGridTDGuidedGradient.explain_caption_wordt, models/gridTDmodel.py:1588-1675, every word of two images): the
    # oracle's restatement on its own (oneDNN) forward
    g = np.load(os.path.join(GOLDEN, "t20_guided.npz"))
    T, n_img = int(g["T"]), int(g["n_img"])
    torch.set_num_threads(8)
    imgs = torch.from_numpy(weights.make_images(int(g["img_seed"]), n_img))
    sd = O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"])))
    for k in range(n_img):
        feats, avg, _ = O.vgg_forward(sd, imgs[k:k + 1])
        tr = O.gridtd_grad_trace(sd, feats[0], avg[0], g["caption"][k])
        for t in range(T):
            df, rw = O.gridtd_guided_wordt(sd, tr, t)                      # (P,C), (t+1,)
            st = g[f"gb{k}_d_feat_stats_{t}"]
            sub = torch.from_numpy(g[f"gb{k}_d_feat_sub_{t}"]).double()
            want = sub.reshape(sub.shape[0], -1).t()
            assert ((df.double()[:, (t % 32)::32] - want).abs().max() / st[1]).item() < TOL_REL, (k, t)
            assert abs(df.double().norm().item() - st[2]) <= 1e-4 * st[2]
            assert np.abs(rw.numpy() - g[f"gb{k}_r_words_{t}"]).max() < 5e-5, (k, t)


def test_t20_fp64_fixture_is_the_reference_in_double():
    # tests/golden/t20_f64.npz = the rows of t20.npz with the reference's classes in float64 (make_golden.py:gen_t20_f64).
    # The oracle evaluated in float64 on an fp64 forward must land on it (1e-9: same formula, same precision), and the
    # reference's own fp32 rows must sit where the GPU test's bound assumes: <= 7e-5 on the two ill-conditioned AoA rows,
    # <= 1e-5 everywhere else
    g, g64 = np.load(os.path.join(GOLDEN, "t20.npz")), np.load(os.path.join(GOLDEN, "t20_f64.npz"))
    T = int(g["T"])
    torch.set_num_threads(8)
    imgs = torch.from_numpy(weights.make_images(int(g["img_seed"]), 1)).double()
    sd = {k: v.double() for k, v in O.state_to_torch(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["grid_V"]))).items()}
    feats, avg, _ = O.vgg_forward(sd, imgs)
    assert np.abs(feats[0, ::4].reshape(128, 196).float().numpy() - g64["features64_0"]).max() <= 2e-7 * float(g64["features64_absmax_0"])
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        tr = O.gridtd_trace(sd, feats[0], avg[0], g["grid_caption"][0])
        for t in (0, 7, 19):
            _, rw = O.gridtd_explain_wordt(sd, tr, t)
            assert np.abs(rw.numpy() - g64[f"grid0_r_words64_{t}"]).max() < 1e-9, t
    finally:
        torch.set_default_dtype(old)
    worst = {}
    for pre in ("grid0", "grid1", "aoa0_h0", "aoa1_h0", "aoa1_h3", "bu0", "bu1"):
        for t in range(T):
            worst[(pre, t)] = float(np.abs(g[f"{pre}_r_words_{t}"].astype(np.float64) - g64[f"{pre}_r_words64_{t}"]).max())
    ill = {("aoa0_h0", 17), ("aoa0_h0", 19)}
    assert all(v <= (7e-5 if k in ill else 1e-5) for k, v in worst.items()), max(worst.items(), key=lambda kv: kv[1])
    assert worst[("aoa0_h0", 17)] > 3e-5 and worst[("aoa0_h0", 19)] > 3e-5       # (the rows the anchoring exists for)


def test_teacherforce_forward_vs_reference():
    """`teacherforce_forward` of the LRP and the gradient explainers of both models (tests/golden/teacherforce.npz: the reference's
    own methods called as evaluation.py:702 calls them): scores to 2e-5 of their maximum, arg-max ids bit-exact; the two
    families differ (the LRP explainers add bias_ih twice), so a wrong bias shows"""
    g = np.load(os.path.join(GOLDEN, "teacherforce.npz"))
    seed = int(g["seed"])
    img = torch.from_numpy(weights.make_images(seed, 1))
    for tag, make_state, fn in (("grid", weights.make_gridtd_state, O.gridtd_teacherforce), ("aoa", weights.make_aoa_state, O.aoa_teacherforce)):
        sd = O.state_to_torch(make_state(seed=seed, vocab_size=int(g[f"{tag}_lrp_V"])))
        preds = {}
        for fam, grad in (("lrp", False), ("grad", True)):
            k = f"{tag}_{fam}"
            pred = fn(sd, img, g[f"{k}_caption"], gradient_family=grad)
            preds[fam] = pred
            assert tuple(pred.shape) == (int(g["T"]) + 1, int(g[f"{k}_V"]))
            assert (pred[:, ::97] - torch.from_numpy(g[f"{k}_pred_sub"])).abs().max().item() < 2e-5 * float(g[f"{k}_absmax"]), k
            assert (pred[-1] - torch.from_numpy(g[f"{k}_pred_last"])).abs().max().item() < 2e-5 * float(g[f"{k}_absmax"]), k
            assert np.array_equal(pred.argmax(-1).numpy(), g[f"{k}_argmax"]), k
        assert (preds["lrp"] - preds["grad"]).abs().max().item() > 1e-3     # the bias quirk is visible in the scores

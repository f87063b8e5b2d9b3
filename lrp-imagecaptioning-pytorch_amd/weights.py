"""Deterministic synthetic weights / inputs for the LRP hot path.

The reference loads a checkpoint `state_dict` (models/gridTDmodel.py:717-718,
models/aoamodel.py:762-763).  There is no network here, so BASELINE.md asks for
random-init weights of the same architecture.  Everything is drawn from a seeded
`numpy.random.RandomState` (a frozen legacy stream: identical numbers on every numpy
version / machine), keyed by the reference's own `state_dict` names, so the very same
dict can be loaded into the reference's `GridTDModel` / `AOAModel` (golden generation)
and into this package's engine.

Distributions follow the reference's initialisers:
  * VGG16 conv: kaiming-normal fan_out, zero bias          (models/vgg.py:48-53)
  * nn.Linear / nn.Conv2d 1x1 / nn.LSTMCell: U(-1/sqrt(fan), 1/sqrt(fan)) (PyTorch defaults)
  * nn.Embedding: N(0, 1)
"""
from collections import OrderedDict
import math

import numpy as np

# models/vgg.py:81 cfgs['D'] with the trailing pool removed (models/gridTDmodel.py:34)
VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
SPECIAL_TOKENS = ("<pad>", "<unk>", "<start>", "<end>")


def vgg16_layers():
    """[(kind, sequential_index, cin, cout)] in forward order; kind in {'conv','pool'}.
    Sequential indices match `vgg.make_layers` (models/vgg.py:62-75): conv, relu, ..., pool."""
    layers, idx, cin = [], 0, 3
    for v in VGG16_CFG:
        if v == 'M':
            layers.append(('pool', idx, cin, cin))
            idx += 1
        else:
            layers.append(('conv', idx, cin, v))
            idx += 2  # conv + relu
            cin = v
    return layers


def _uniform(rs, shape, bound):
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def _normal(rs, shape, std):
    return (rs.standard_normal(size=shape) * std).astype(np.float32)


def _vgg_state(rs, prefix, bias_std):
    sd = OrderedDict()
    for kind, idx, cin, cout in vgg16_layers():
        if kind != 'conv':
            continue
        std = math.sqrt(2.0 / (cout * 9))
        sd[f"{prefix}{idx}.weight"] = _normal(rs, (cout, cin, 3, 3), std)
        if bias_std > 0:
            sd[f"{prefix}{idx}.bias"] = _normal(rs, (cout,), bias_std)
        else:
            sd[f"{prefix}{idx}.bias"] = np.zeros((cout,), np.float32)
    return sd


def _linear(rs, sd, name, out_f, in_f, bias=True):
    b = 1.0 / math.sqrt(in_f)
    sd[name + ".weight"] = _uniform(rs, (out_f, in_f), b)
    if bias:
        sd[name + ".bias"] = _uniform(rs, (out_f,), b)


def _lstm(rs, sd, name, in_f, hid):
    b = 1.0 / math.sqrt(hid)
    sd[name + ".weight_ih"] = _uniform(rs, (4 * hid, in_f), b)
    sd[name + ".weight_hh"] = _uniform(rs, (4 * hid, hid), b)
    sd[name + ".bias_ih"] = _uniform(rs, (4 * hid,), b)
    sd[name + ".bias_hh"] = _uniform(rs, (4 * hid,), b)


def make_gridtd_state(seed=0, vocab_size=9586, embed_dim=512, hidden_dim=512, feat_dim=512,
                      num_pixels=196, vgg_bias_std=0.0):
    """state_dict (numpy float32) for the reference `GridTDModel` (models/gridTDmodel.py:111-130)."""
    rs = np.random.RandomState(seed)
    sd = _vgg_state(rs, "img_encoder.encoder.", vgg_bias_std)
    b = 1.0 / math.sqrt(feat_dim)
    sd["img_projector.weight"] = _uniform(rs, (hidden_dim, feat_dim, 1, 1), b)
    sd["img_projector.bias"] = _uniform(rs, (hidden_dim,), b)
    _linear(rs, sd, "global_img_feature_proj", embed_dim, feat_dim)
    _lstm(rs, sd, "LanguageLSTM", 2 * hidden_dim, hidden_dim)
    _lstm(rs, sd, "AdaLSTM.lstm_cell", 2 * embed_dim + hidden_dim, hidden_dim)
    _linear(rs, sd, "AdaLSTM.x_gate", hidden_dim, 2 * embed_dim + hidden_dim)
    _linear(rs, sd, "AdaLSTM.h_gate", hidden_dim, hidden_dim)
    _linear(rs, sd, "AdaAttention.W_v_proj", num_pixels, hidden_dim)
    _linear(rs, sd, "AdaAttention.W_s_proj", num_pixels, hidden_dim)
    _linear(rs, sd, "AdaAttention.W_g_proj", num_pixels, hidden_dim, bias=False)
    _linear(rs, sd, "AdaAttention.w_h", 1, num_pixels, bias=False)
    sd["embedding.weight"] = _normal(rs, (vocab_size, embed_dim), 1.0)
    _linear(rs, sd, "fc", vocab_size, hidden_dim)
    return sd


def make_aoa_state(seed=0, vocab_size=11027, embed_dim=512, hidden_dim=512, feat_dim=512,
                   vgg_bias_std=0.0, with_encoder=True):
    """state_dict (numpy float32) for the reference `AOAModel` (models/aoamodel.py:116-142).
    `with_encoder=False` gives the bottom-up variant's decoder-side tensors only (feat_dim=2048,
    models/aoamodel.py:1795-1797: `img_projector` is then a Linear(feat_dim, hidden))."""
    rs = np.random.RandomState(seed)
    sd = _vgg_state(rs, "img_encoder.encoder.", vgg_bias_std) if with_encoder else OrderedDict()
    b = 1.0 / math.sqrt(feat_dim)
    sd["img_projector.weight"] = _uniform(rs, (hidden_dim, feat_dim, 1, 1), b)
    sd["img_projector.bias"] = _uniform(rs, (hidden_dim,), b)
    sd["embedding.weight"] = _normal(rs, (vocab_size, embed_dim), 1.0)
    _lstm(rs, sd, "LanguageLSTM", hidden_dim + embed_dim, hidden_dim)
    _linear(rs, sd, "decoder_k_proj", hidden_dim, hidden_dim)
    _linear(rs, sd, "decoder_v_proj", hidden_dim, hidden_dim)
    _linear(rs, sd, "decoder_multihead_attention.q_proj", hidden_dim, hidden_dim)
    _linear(rs, sd, "decoder_aoa_linear_gate", hidden_dim, hidden_dim)
    _linear(rs, sd, "decoder_aoa_linear", hidden_dim, hidden_dim)
    _linear(rs, sd, "fc", vocab_size, hidden_dim)
    return sd


def make_word_map(vocab_size):
    """Synthetic word map with the reference's special-token layout (dataset/wordmap_*.json:
    <pad>=0, <unk>=V-3, <start>=V-2, <end>=V-1)."""
    wm = OrderedDict()
    wm["<pad>"] = 0
    for i in range(1, vocab_size - 3):
        wm[f"w{i}"] = i
    wm["<unk>"] = vocab_size - 3
    wm["<start>"] = vocab_size - 2
    wm["<end>"] = vocab_size - 1
    return wm


def make_images(seed, batch, height=224, width=224):
    """Synthetic pre-processed images ~N(0,1), NCHW float32."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal(size=(batch, 3, height, width)).astype(np.float32)


def make_captions(seed, batch, length, vocab_size):
    """Teacher-forced token ids: column 0 is <start>, then `length` words ~U[1, V-4]
    (BASELINE.md §3).  Returns int64 (batch, length+1) — the layout of the reference's
    `beam_caption_encode` (models/gridTDmodel.py:937)."""
    rs = np.random.RandomState(seed)
    cap = rs.randint(1, vocab_size - 3, size=(batch, length + 1)).astype(np.int64)
    cap[:, 0] = vocab_size - 2
    return cap


def make_bu_features(seed, batch, regions=36, feat_dim=2048):
    """Synthetic bottom-up region features ~relu(N(0,1))."""
    rs = np.random.RandomState(seed)
    return np.maximum(rs.standard_normal(size=(batch, regions, feat_dim)), 0).astype(np.float32)

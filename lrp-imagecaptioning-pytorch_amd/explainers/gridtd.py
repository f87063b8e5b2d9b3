"""Batched gridTD LRP engine + the drop-in `ExplainGridTDAttention` explainer.

Reference: models/gridTDmodel.py:705-1211 (`ExplainGridTDAttention`).  The reference explains one
image at a time with ~50k tiny tensor ops per image; here a batch of B images x T words is traced
and explained in lock-step by HIP kernels behind the lrpx C ABI (see csrc/lrpx_decoder.hip).
The host logic below only sequences kernel launches; torch is used for device memory."""
import ctypes as C

import numpy as np
import torch

from .. import _lib, ops
from .._lib import (EPI_PLAIN, EPI_REL, GridGradState, GridRelState, GridStepArgs, GridTrace, PACK_DENSE, PACK_DENSE_T, check, ptr,
                    ptr_at, stream_ptr)
from .ragged import ragged

VGG_PREFIX = "img_encoder.encoder."


def _t(v, dev):
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.detach().to(device=dev, dtype=torch.float32).contiguous()


IMAGENET_MEAN, IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def load_image(img_filepath, height, width, mean, std, device):
    """`preprocess_img` of every reference explainer (models/gridTDmodel.py:767-771, models/aoamodel.py:864-868): PIL open ->
    RGB -> `transforms.Resize((height, width))` (PIL bilinear) -> `ToTensor` (/255, CHW) -> `Normalize(mean, std)` -> (1,3,H,W)
    on the device.  Host side, as in the reference (image decoding is outside the path)."""
    from PIL import Image
    im = Image.open(img_filepath).convert('RGB').resize((width, height), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
    x = (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    return x.unsqueeze(0).contiguous().to(device)


class GridTDEngine:
    """Device-resident gridTD model + trace/relevance pipelines.  `state` is the reference model's
    `state_dict` (torch tensors or numpy arrays, names of models/gridTDmodel.py:111-130)."""

    def __init__(self, state, device="cuda"):
        _lib.load()   # fail loudly without the HIP library
        if not torch.cuda.is_available():
            raise _lib.LrpxError("the LRP hot path needs an MI355X; there is no CPU fallback")
        dev = torch.device(device)
        self.device = dev
        sd = {k: _t(v, dev) for k, v in state.items() if not k.startswith(VGG_PREFIX)}
        names = [k for k in state if k.startswith(VGG_PREFIX) and k.endswith(".weight")]
        self.vgg = ops.Vgg16([_t(state[k], dev) for k in names],
                             [_t(state[k.replace(".weight", ".bias")], dev) for k in names])
        self.sd = sd
        self.V, self.E = sd["embedding.weight"].shape
        self.H = sd["fc.weight"].shape[1]
        self.C = sd["img_projector.weight"].shape[1]
        self.P = sd["AdaAttention.W_v_proj.weight"].shape[0]
        H, E, Cc = self.H, self.E, self.C
        assert H == 512 and E == 512, "kernels are built for hidden=embed=512 (config.py:123-124)"
        # --- forward weights
        a, l = "AdaLSTM.lstm_cell.", "LanguageLSTM."
        self.Wcat1 = torch.cat([torch.cat([sd[a + "weight_ih"], sd[a + "weight_hh"]], 1),
                                torch.cat([sd["AdaLSTM.x_gate.weight"], sd["AdaLSTM.h_gate.weight"]], 1)], 0).contiguous()
        self.bcat1 = torch.cat([sd[a + "bias_ih"] + sd[a + "bias_hh"],
                                sd["AdaLSTM.x_gate.bias"] + sd["AdaLSTM.h_gate.bias"]]).contiguous()
        self.Wcat2 = torch.cat([sd[l + "weight_ih"], sd[l + "weight_hh"]], 1).contiguous()
        self.bcat2_explainer = (sd[l + "bias_ih"] + sd[l + "bias_ih"]).contiguous()   # quirk: gridTDmodel.py:789
        self.bcat2_model = (sd[l + "bias_ih"] + sd[l + "bias_hh"]).contiguous()       # nn.LSTMCell
        # gate rows interleaved for the fused decoder step of the teacher-forced trace (lrpx_gridtd_fwd_steps): row 16 j + 4 gate + u = row
        # gate * H + 4 j + u, so that one workgroup of a gate linear holds the i, f, g, o pre-activations of the hidden units 4j .. 4j+3
        self.fused_steps = H % 16 == 0 and E % 16 == 0          # False: the 7-launch step (A/B; the decoding loops always use it)
        if self.fused_steps:
            jj, qq, uu = torch.meshgrid(torch.arange(H // 4), torch.arange(4), torch.arange(4), indexing="ij")
            il = (qq * H + 4 * jj + uu).reshape(-1).to(self.Wcat1.device)
            self.Wcat1_il, self.bcat1_il = self.Wcat1[:4 * H][il].contiguous(), self.bcat1[:4 * H][il].contiguous()
            self.Wcat2_il = self.Wcat2[il].contiguous()
            self.bcat2_explainer_il, self.bcat2_model_il = self.bcat2_explainer[il].contiguous(), self.bcat2_model[il].contiguous()
        self.w_proj2d = sd["img_projector.weight"].reshape(H, Cc).contiguous()
        kc = ops.conv_kc(0, 1, Cc)
        self.p_proj_fwd = ops.pack_weights(self.w_proj2d, H, Cc, 1, PACK_DENSE, kc)
        self.p_attv_fwd = ops.pack_weights(sd["AdaAttention.W_v_proj.weight"], self.P, H, 1, PACK_DENSE, kc)
        self.p_fc_fwd = ops.pack_weights(sd["fc.weight"], self.V, H, 1, PACK_DENSE, kc)
        # fp16 split-product packs (csrc/dense_f16x3.hip): built always, USED only while ops.decoder_f16() says so (`_f16()`: conv modes 2 / 3)
        self.force_f16 = None            # True / False: this engine's decoder GEMMs on / off the fp16 split products whatever the conv mode (A/B and tests)
        self.p_fc_fwd_h = ops.pack_weights_f16x2(sd["fc.weight"], self.V, H, _lib.PACK_FWD, taps=1) if H % 64 == 0 else None
        # --- guided-backprop weights: full gate matrices, contraction over the 4H gate rows (gridTDmodel.py:1637-1659)
        self.p_g2 = ops.pack_weights(self.Wcat2, 4 * H, 3 * H, 1, PACK_DENSE_T, kc)
        self.p_g1 = ops.pack_weights(sd[a + "weight_ih"].contiguous(), 4 * H, 2 * E + H, 1, PACK_DENSE_T, kc)
        self.p_gp_grad = self.p_gp_rel = ops.pack_weights(sd["global_img_feature_proj.weight"], E, Cc, 1, PACK_DENSE_T, kc)
        # --- relevance weights: g-gate rows of the LSTMs, [W_ih^g | W_hh^g]  (gridTDmodel.py:1019-1024)
        wg1 = torch.cat([sd[a + "weight_ih"][2 * H:3 * H], sd[a + "weight_hh"][2 * H:3 * H]], 1).contiguous()
        wg2 = torch.cat([sd[l + "weight_ih"][2 * H:3 * H], sd[l + "weight_hh"][2 * H:3 * H]], 1).contiguous()
        self.p_wg1 = ops.pack_weights(wg1, H, 2 * E + 2 * H, 1, PACK_DENSE_T, kc)
        self.p_wg2 = ops.pack_weights(wg2, H, 3 * H, 1, PACK_DENSE_T, kc)
        self.p_gp_rel = ops.pack_weights(sd["global_img_feature_proj.weight"], E, Cc, 1, PACK_DENSE_T, kc)
        self.p_proj_rel = ops.pack_weights(self.w_proj2d, H, Cc, 1, PACK_DENSE_T, kc)
        # the projector rule runs over every (word, pixel) row: split products on the fp16 matrix cores (csrc/dense_f16x3.hip)
        self.p_proj_rel_h = ops.pack_weights_f16x2(self.w_proj2d, H, Cc, _lib.PACK_BWD_PLAIN, taps=1) if H % 64 == 0 else None
        # ... in the default (exact) arithmetic: the same tile on the bf16 matrix cores with operands split exactly into three bf16 parts
        # (dense_f16x3.hip, B6: six products, fp32 range - what conv mode 1 is for the VGG16 chains); the fp32 MFMA where the sizes do not fit
        self.p_proj_rel_6 = ops.pack_weights_bf16x3(self.w_proj2d, H, Cc, _lib.PACK_BWD_PLAIN, taps=1) if (H % 32 == 0 and Cc % 4 == 0) else None
        self.dense_bf16x6 = True         # False: those rules on the fp32 MFMA kernel (A/B and tests)
        # ... and the lock-step gate rules (rows = images x words): the few-row kernel of the same file.  `lockstep_f16 = False`
        # puts them back on the fp32 MFMA (csrc/dense_small.hip; A/B: tools/phase_times.py --lockstep-fp32)
        self.lockstep_f16 = H % 16 == 0 and E % 16 == 0
        self.p_wg1_h = ops.pack_weights_f16x2(wg1, H, 2 * E + 2 * H, _lib.PACK_BWD_PLAIN, taps=1) if self.lockstep_f16 else None
        self.p_wg2_h = ops.pack_weights_f16x2(wg2, H, 3 * H, _lib.PACK_BWD_PLAIN, taps=1) if self.lockstep_f16 else None
        self.p_gp_rel_h = ops.pack_weights_f16x2(sd["global_img_feature_proj.weight"], E, Cc, _lib.PACK_BWD_PLAIN, taps=1) if self.lockstep_f16 else None
        torch.cuda.synchronize()
        self._idx_cache = {}

    # ------------------------------------------------------------------------------------------
    def _alloc_trace(self, B, T, grad=False):
        dev, H, E, P = self.device, self.H, self.E, self.P
        shapes = {"xh1": (B, T, 2 * E + 2 * H), "xh2": (B, T, 3 * H), "alpha": (B, T, P), "beta": (B, T)}
        for k in ("h1", "c1", "h2", "c2"):
            shapes[k] = (B, T + 1, H)
        for k in ("g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "hc"):
            shapes[k] = (B, T, H)
        names = ["xh1", "xh2", "h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "hc",
                 "alpha", "beta"]
        if grad:     # the gradient explainers also keep the output gates and the sentinel gate (:1323-1422)
            for k in ("o1", "o2", "sgate"):
                shapes[k] = (B, T, H)
            names += ["o1", "o2", "sgate"]
        tr = dict(B=B, T=T)
        tr.update(ops.zeros_arena(dev, shapes))            # one allocation, one fill
        c = GridTrace()
        c.B, c.T, c.H, c.E, c.P = B, T, H, E, P
        for k in names:
            setattr(c, k, ptr(tr[k]))
        tr["_c"] = c
        return tr

    def encode(self, images):
        """VGG16 forward + the image-side constants of get_hidden_parameters (gridTDmodel.py:941-950)."""
        lib = _lib.load()
        st = stream_ptr()
        B = images.shape[0]
        H, E, Cc, P = self.H, self.E, self.C, self.P
        images = images.to(self.device, torch.float32).contiguous()
        feats = self.vgg.forward(images)                                   # (B,P,C) NHWC view into the trace
        enc = dict(B=B, feats=feats)
        enc["avg"] = torch.empty(B, Cc, device=self.device)
        check(lib.lrpx_mean_pixels(ptr(feats), ptr(enc["avg"]), B, P, Cc, st))
        enc["proj_pre"] = torch.empty(B, P, H, device=self.device)
        ops.conv_mfma(feats, self.p_proj_fwd, B, 0, Cc, H, 1, EPI_PLAIN, pix_per_map=P, oc_split=H,
                      bias=self.sd["img_projector.bias"], out0=enc["proj_pre"])
        enc["Vp"] = torch.empty_like(enc["proj_pre"])
        check(lib.lrpx_relu(ptr(enc["proj_pre"]), ptr(enc["Vp"]), enc["Vp"].numel(), st))
        enc["glob_pre"] = torch.empty(B, E, device=self.device)
        check(lib.lrpx_linear_small(ptr(enc["avg"]), Cc, ptr(self.sd["global_img_feature_proj.weight"]),
                                    ptr(self.sd["global_img_feature_proj.bias"]), ptr(enc["glob_pre"]), E, B, Cc, E, 0, st))
        enc["glob"] = torch.empty_like(enc["glob_pre"])
        check(lib.lrpx_relu(ptr(enc["glob_pre"]), ptr(enc["glob"]), enc["glob"].numel(), st))
        # time-invariant part of the attention scores: W_v_proj(V) + b  (gridTDmodel.py:79)
        enc["att_img"] = torch.empty(B, P, P, device=self.device)
        ops.conv_mfma(enc["Vp"], self.p_attv_fwd, B, 0, H, -(-P // 32) * 32, 1, EPI_PLAIN, pix_per_map=P, oc_split=P,
                      bias=self.sd["AdaAttention.W_v_proj.bias"], out0=enc["att_img"])
        return enc

    def _step(self, tr, enc, t, tokens, model_bias, after_lstm1=None):
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E = tr["B"], tr["T"], self.H, self.E
        c = C.byref(tr["_c"])
        sd = self.sd
        check(lib.lrpx_gridtd_fwd_pre(c, t, ptr(enc["glob"]), ptr(sd["embedding.weight"]), ptr(tokens),
                                      tokens.shape[1], st))
        W1 = 2 * E + 2 * H
        if "_zz1" not in tr:
            tr["_zz1"] = torch.empty(B, 5 * H, device=self.device)
            tr["_zz2"] = torch.empty(B, 4 * H, device=self.device)
            tr["_att_scr"] = torch.empty(B, 3 * self.P, device=self.device)
        zz1 = tr["_zz1"]
        check(lib.lrpx_linear_small(ptr_at(tr["xh1"], t * W1), T * W1, ptr(self.Wcat1), ptr(self.bcat1), ptr(zz1),
                                    5 * H, B, W1, 5 * H, 0, st))
        check(lib.lrpx_gridtd_fwd_lstm(c, t, ptr(zz1), 5 * H, 1, st))
        if after_lstm1 is not None:
            after_lstm1(t)
        aa = "AdaAttention."
        scr = tr["_att_scr"]
        check(lib.lrpx_gridtd_fwd_attention(c, t, ptr(enc["Vp"]), ptr(enc["att_img"]), ptr(sd[aa + "W_g_proj.weight"]),
                                            ptr(sd[aa + "W_s_proj.weight"]), ptr(sd[aa + "W_s_proj.bias"]),
                                            ptr(sd[aa + "w_h.weight"]), ptr(scr), st))
        zz2 = tr["_zz2"]
        b2 = self.bcat2_model if model_bias else self.bcat2_explainer
        check(lib.lrpx_linear_small(ptr_at(tr["xh2"], t * 3 * H), T * 3 * H, ptr(self.Wcat2), ptr(b2), ptr(zz2), 4 * H,
                                    B, 3 * H, 4 * H, 0, st))
        check(lib.lrpx_gridtd_fwd_lstm(c, t, ptr(zz2), 4 * H, 2, st))

    def trace(self, enc, captions, model_bias=False, predictions=True, grad=False):
        """get_hidden_parameters (gridTDmodel.py:952-1012) for B images under teacher forcing.
        captions: (B,T+1) int64 on device, column 0 = <start>.  grad=True: the gradient explainers' trace
        (:1323-1422: correct LSTM bias, output + sentinel gates kept)."""
        lib = _lib.load()
        B, T = captions.shape[0], captions.shape[1] - 1
        captions = captions.to(self.device, torch.int64).contiguous()      # (token ids index the embedding table: never another width)
        tr = self._alloc_trace(B, T, grad)
        model_bias = model_bias or grad
        # the T steps in one native call (lrpx_gridtd_fwd_steps: the launches of `_step`, its host loop in C)
        H, P, dev = self.H, self.P, self.device
        tr["_zz1"], tr["_zz2"], tr["_att_scr"] = torch.empty(B, 5 * H, device=dev), torch.empty(B, 4 * H, device=dev), torch.empty(B, 3 * P, device=dev)
        sa, sd, aa = GridStepArgs(), self.sd, "AdaAttention."
        sa.glob, sa.emb, sa.tok, sa.tok_ld = ptr(enc["glob"]), ptr(sd["embedding.weight"]), ptr(captions), captions.shape[1]
        sa.w_cat1, sa.b_cat1, sa.w_cat2 = ptr(self.Wcat1), ptr(self.bcat1), ptr(self.Wcat2)
        sa.b_cat2 = ptr(self.bcat2_model if model_bias else self.bcat2_explainer)
        if self.fused_steps:         # gate rows interleaved: gate linear + LSTM cell in one launch (lrpx_gridtd_fwd_steps)
            sa.w_il1, sa.b_il1, sa.w_il2 = ptr(self.Wcat1_il), ptr(self.bcat1_il), ptr(self.Wcat2_il)
            sa.b_il2 = ptr(self.bcat2_model_il if model_bias else self.bcat2_explainer_il)
        sa.Vp, sa.att_img = ptr(enc["Vp"]), ptr(enc["att_img"])
        sa.Wg, sa.Ws, sa.bs, sa.wh = ptr(sd[aa + "W_g_proj.weight"]), ptr(sd[aa + "W_s_proj.weight"]), ptr(sd[aa + "W_s_proj.bias"]), ptr(sd[aa + "w_h.weight"])
        sa.zz1, sa.zz2, sa.att_scratch = ptr(tr["_zz1"]), ptr(tr["_zz2"]), ptr(tr["_att_scr"])
        check(lib.lrpx_gridtd_fwd_steps(C.byref(tr["_c"]), 0, T, C.byref(sa), stream_ptr()))
        tr["captions"] = captions
        tr["logit"] = torch.empty(B * T, device=self.device)
        check(lib.lrpx_target_logit(ptr(tr["hc"]), ptr(self.sd["fc.weight"]), ptr(self.sd["fc.bias"]), ptr(captions),
                                    T + 1, ptr(tr["logit"]), B, T, self.H, stream_ptr()))
        if predictions:
            tr["pred"] = self.logits(tr["hc"].view(B * T, self.H), fast=True).view(B, T, self.V)
        return tr

    def logits(self, hc_rows, fast=False):
        """fc(context_hat + h2) (gridTDmodel.py:990) for R rows -> (R,V).  fast=True (the (T,V) block a trace keeps, not the
        decisions of a decoding loop): split products on the fp16 matrix cores (csrc/dense_f16x3.hip, fp32-grade)"""
        R = hc_rows.shape[0]
        out = torch.empty(R, self.V, device=self.device)
        if fast and R >= 128 and self.p_fc_fwd_h is not None and self._f16():
            hc_rows = hc_rows.contiguous()
            ops.conv_mfma(hc_rows, self.p_fc_fwd_h, R, 0, self.H, -(-self.V // 32) * 32, 1, EPI_PLAIN, pix_per_map=1, oc_split=self.V,
                          bias=self.sd["fc.bias"], out0=out, f16x3=1, in_amax=ops.amax_maps(hc_rows, R))
            return out
        ops.conv_mfma(hc_rows, self.p_fc_fwd, R, 0, self.H, -(-self.V // 32) * 32, 1, EPI_PLAIN, pix_per_map=1,
                      oc_split=self.V, bias=self.sd["fc.bias"], out0=out)
        return out

    def greedy(self, enc, max_cap_length, start_id, end_id, model_bias=True):
        """GridTDModel.greedy_search (gridTDmodel.py:480-520): argmax per step; after the first <end> the
        sequence is padded with 0.  Returns int64 (B, max_cap_length) incl. <start>."""
        lib = _lib.load()
        B = enc["B"]
        T = max_cap_length - 1
        toks = torch.zeros(B, T + 1, dtype=torch.int64, device=self.device)
        toks[:, 0] = start_id
        tr = self._alloc_trace(B, T)
        nxt = torch.empty(B, dtype=torch.int64, device=self.device)
        unfinished = torch.ones(B, dtype=torch.bool, device=self.device)
        for t in range(T):
            self._step(tr, enc, t, toks, model_bias)
            lg = self.logits(tr["hc"][:, t].contiguous())
            check(lib.lrpx_argmax_rows(ptr(lg), self.V, B, self.V, ptr(nxt), stream_ptr()))
            unfinished = unfinished & (nxt != end_id)          # token bookkeeping (integers), :500-505
            toks[:, t + 1] = nxt * unfinished
        return toks

    def beam_search(self, enc, beam_size, max_cap_length, start_id, end_id):
        """`GridTDModel.beam_search` (models/gridTDmodel.py:400-478) for ONE image (enc of a single image, as the
        reference asserts :411): returns the chosen token sequence incl. <start> (`seq`, :469-472)."""
        from .beam import run_beam_search
        assert enc["B"] == 1, "beam search captions one image (models/gridTDmodel.py:411)"
        nb = int(beam_size)
        encb = {k: (v.expand(nb, *v.shape[1:]).contiguous() if torch.is_tensor(v) else v) for k, v in enc.items()}
        encb["B"] = nb
        T = int(max_cap_length)
        tr = self._alloc_trace(nb, T)
        toks = torch.zeros(nb, T + 1, dtype=torch.int64, device=self.device)

        def step(t, prev):
            toks[:, t] = prev
            self._step(tr, encb, t, toks, True)

        def reorder(t, src):
            sel = torch.tensor(src, dtype=torch.int64, device=self.device)
            for k in ("h1", "c1", "h2", "c2"):
                tr[k][:len(src), t + 1] = tr[k][sel, t + 1]

        return run_beam_search(step, lambda t: self.logits(tr["hc"][:, t].contiguous()), reorder, self.V, nb, T,
                               start_id, end_id, self.device)

    def sample_lrp(self, enc, max_length, start_id, end_id, skip_ids):
        """GridTDModel.sample_lrp, greedy (gridTDmodel.py:631-702): LRP-inference decoding.  Every step's logits are
        recomputed from the fc input re-weighted by the predicted word's relevance (`get_lrp_weight_step` :548-577)
        before the next word is taken.  `skip_ids`: ids exempt from the re-weighting (the reference's STOP_WORDS and
        special tokens).  Returns (seq int64 (B,max_length), seq_logprobs float32 (B,max_length)); like the reference,
        tokens after <end> are 0 and nothing is written once every sequence has finished (:699-700)."""
        lib = _lib.load()
        B, T, H, E = enc["B"], max_length, self.H, self.E
        W1 = 2 * E + 2 * H
        dev = self.device
        skip = torch.zeros(self.V, dtype=torch.uint8, device=dev)
        skip[torch.as_tensor(sorted(int(i) for i in skip_ids), dtype=torch.int64, device=dev)] = 1
        toks = torch.zeros(B, T + 1, dtype=torch.int64, device=dev)
        toks[:, 0] = start_id
        lps = torch.zeros(B, T, dtype=torch.float32, device=dev)
        tr = self._alloc_trace(B, T)
        c = C.byref(tr["_c"])
        xg = torch.empty(B, W1, device=dev)
        zg = torch.empty(B, H, device=dev)
        hcw = torch.empty(B, H, device=dev)
        nxt = torch.empty(B, dtype=torch.int64, device=dev)
        lp = torch.empty(B, dtype=torch.float32, device=dev)
        w_gate, b_gate = self.Wcat1[4 * H:], self.bcat1[4 * H:]          # [x_gate | h_gate] rows of the fused weight
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)

        def sentinel_new_h(t):
            st = stream_ptr()
            check(lib.lrpx_gridtd_fwd_gate_input(c, t, ptr(xg), st))
            check(lib.lrpx_linear_small(ptr(xg), W1, ptr(w_gate), ptr(b_gate), ptr(zg), H, B, W1, H, 0, st))
            check(lib.lrpx_gridtd_fwd_sentinel(c, t, ptr(zg), H, st))

        for t in range(T):
            self._step(tr, enc, t, toks, True, after_lstm1=sentinel_new_h)
            st = stream_ptr()
            pred = self.logits(tr["hc"][:, t].contiguous())
            check(lib.lrpx_gridtd_lrp_reweight(c, t, ptr(pred), self.V, self.V, ptr(self.sd["fc.weight"]), ptr(skip),
                                               ptr(hcw), st))
            wpred = self.logits(hcw)
            check(lib.lrpx_argmax_logprob_rows(ptr(wpred), self.V, B, self.V, ptr(nxt), ptr(lp), st))
            alive = unfinished.any()                                        # the reference's `break` (:699-700)
            unfinished = unfinished & (nxt != end_id)
            toks[:, t + 1] = torch.where(alive, nxt * unfinished, torch.zeros_like(nxt))
            lps[:, t] = torch.where(alive, lp, torch.zeros_like(lp))
        return toks[:, 1:].contiguous(), lps

    def forwardlrp_context(self, enc, captions, caption_lengths, skip_ids):
        """The forward half of `GridTDModel.forwardlrp_context` (models/gridTDmodel.py:579-630; LRP-inference fine-tuning,
        SURVEY §8(f) row 2): teacher-forced decoding with the model's own forward (sentinel gate on the NEW h1, :617) where
        every step's scores are recomputed from the fc input re-weighted by the relevance of the step's arg-max word
        (`get_lrp_weight_step`, :548-577).  captions (B, >= L) int64 incl. <start> in column 0; L = max(caption_lengths) - 1.
        Returns (predictions (B,L,V), weighted_predictions (B,L,V), L).  The loss and its gradients (train.py:211-263)
        are training and stay outside the path."""
        lib = _lib.load()
        B, H, E = enc["B"], self.H, self.E
        L = int(max(caption_lengths)) - 1
        W1 = 2 * E + 2 * H
        dev = self.device
        captions = captions.to(dev, torch.int64).contiguous()
        assert captions.shape[0] == B and captions.shape[1] >= L
        skip = torch.zeros(self.V, dtype=torch.uint8, device=dev)
        skip[torch.as_tensor(sorted(int(i) for i in skip_ids), dtype=torch.int64, device=dev)] = 1
        toks = captions[:, :L + 1].contiguous() if captions.shape[1] > L else torch.cat(
            [captions, captions.new_zeros(B, 1)], 1).contiguous()          # column t is the input of step t
        tr = self._alloc_trace(B, L)
        c = C.byref(tr["_c"])
        xg, zg, hcw = torch.empty(B, W1, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
        w_gate, b_gate = self.Wcat1[4 * H:], self.bcat1[4 * H:]
        preds = torch.empty(B, L, self.V, device=dev)
        wpreds = torch.empty(B, L, self.V, device=dev)

        def sentinel_new_h(t):
            st = stream_ptr()
            check(lib.lrpx_gridtd_fwd_gate_input(c, t, ptr(xg), st))
            check(lib.lrpx_linear_small(ptr(xg), W1, ptr(w_gate), ptr(b_gate), ptr(zg), H, B, W1, H, 0, st))
            check(lib.lrpx_gridtd_fwd_sentinel(c, t, ptr(zg), H, st))

        for t in range(L):
            self._step(tr, enc, t, toks, True, after_lstm1=sentinel_new_h)
            pred = self.logits(tr["hc"][:, t].contiguous())
            check(lib.lrpx_gridtd_lrp_reweight(c, t, ptr(pred), self.V, self.V, ptr(self.sd["fc.weight"]), ptr(skip),
                                               ptr(hcw), stream_ptr()))
            preds[:, t] = pred
            wpreds[:, t] = self.logits(hcw)
        return preds, wpreds, L

    # ------------------------------------------------------------------------------------------
    def _row_index(self, B, T):
        key = (B, T)
        if key not in self._idx_cache:
            b = torch.arange(B, device=self.device).view(B, 1)
            t = torch.arange(T, device=self.device).view(1, T)
            s = torch.arange(T, device=self.device).view(T, 1, 1)
            idx = (b * T + (t - s).clamp(min=0)).to(torch.int32).reshape(T, B * T).contiguous()
            row2img = (b + 0 * t).to(torch.int32).reshape(B * T).contiguous()
            self._idx_cache[key] = (idx, row2img)
        return self._idx_cache[key]

    def relevance(self, enc, tr, lens=None, want_r_feat=True):
        """explain_caption_wordt (gridTDmodel.py:1014-1135) for every (image, word) row at once.
        Returns r_feat (B*T, P, C) relevance of the encoder output (NHWC), r_words (B*T, T) and the row -> image table.
        lens (one caption length per image, list / array / tensor; explainers/ragged.py): words past an image's length are
        skipped by the lock-step kernels (their r_words rows stay zero), and the (word, pixel) rules run on the VALID rows only -
        r_feat is then COMPACT, (sum(lens), P, C) in image-major order, with the matching row -> image table."""
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E, P, Cc = tr["B"], tr["T"], self.H, self.E, self.P, self.C
        rows = B * T
        dev = self.device
        rg = ragged(lens, B, T, dev)
        e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        rs = dict(r_h2n=e(rows, H), r_c2=e(rows, H), r_c1=e(rows, H), r_ch0=e(rows, H), r_h2p=e(rows, H),
                  r_glob=e(rows, E), A=e(rows, H), rx=e(rows, 2 * E + 2 * H), wacc=ops.zeros(rows, T, H, device=dev),
                  r_words=e(rows, T))
        c = GridRelState()
        c.lens = ptr(rg.lens) if rg is not None else None
        for k, v in rs.items():
            setattr(c, k, ptr(v))
        ctr, crs = C.byref(tr["_c"]), C.byref(c)
        idx, row2img = self._row_index(B, T)
        check(lib.lrpx_gridtd_rel_init(ctr, crs, ptr(self.sd["fc.weight"]), ptr(tr["logit"]), ptr(tr["captions"]),
                                       T + 1, st))
        W1 = 2 * E + 2 * H
        f16 = 1 if (self.lockstep_f16 and self._f16()) else 0
        # the T lock-steps in one native call (lrpx_gridtd_rel_steps): phase 0, LanguageLSTM dense rule, phase 1, AdaLSTM dense rule, phase 2
        d2 = ops.conv_desc(rs["A"], self.p_wg2_h if f16 else self.p_wg2, rows, 0, H, 3 * H, 1, EPI_REL, pix_per_map=1, oc_split=3 * H,
                           x=tr["xh2"], map2img=idx[0], out0=rs["rx"], f16x3=f16)
        d1 = ops.conv_desc(rs["A"], self.p_wg1_h if f16 else self.p_wg1, rows, 0, H, W1, 1, EPI_REL, pix_per_map=1, oc_split=W1,
                           x=tr["xh1"], map2img=idx[0], out0=rs["rx"], f16x3=f16)
        check(lib.lrpx_gridtd_rel_steps(ctr, crs, T, C.byref(d2), C.byref(d1), ptr(idx), idx.shape[1], st))
        # global feature path (:1116-1124) and projector (:1125-1128)
        a_glob = e(rows, E)
        check(lib.lrpx_gridtd_rel_glob(ctr, crs, ptr(enc["glob_pre"]), ptr(a_glob), st))
        r_avg = e(rows, Cc)
        ops.conv_mfma(a_glob, self.p_gp_rel_h if f16 else self.p_gp_rel, rows, 0, E, -(-Cc // 32) * 32 if f16 else Cc, 1, EPI_REL,
                      pix_per_map=1, oc_split=Cc, x=enc["avg"], map2img=row2img, out0=r_avg, f16x3=f16)
        U = e(rows, Cc)
        check(lib.lrpx_rel_avg_u(ptr(r_avg), ptr(enc["avg"]), ptr(U), rows, T, Cc, P, st))
        check(lib.lrpx_rel_words_norm(ptr(rs["r_words"]), rows, T, st))
        n, rowlist = rows, None
        if rg is not None and not rg.full:       # unequal lengths: the (word, pixel) rule on the valid rows only, compact
            n, rowlist, row2img = rg.n, rg.rows, rg.row2img
            if n == 0:
                return e(0, P, Cc), rs["r_words"], row2img
            U = ops.gather_rows(U, rowlist)
        a_proj = e(n, P, H)
        check(lib.lrpx_gridtd_rel_pix_rows(ctr, crs, ptr(enc["Vp"]), ptr(enc["proj_pre"]), ptr(a_proj), ptr(rowlist), n, st))
        r_feat = e(n, P, Cc)
        if self.p_proj_rel_h is not None and self._f16():
            ops.conv_mfma(a_proj, self.p_proj_rel_h, n, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc,
                          x=enc["feats"], u=U, map2img=row2img, out0=r_feat, f16x3=1, in_amax=ops.amax_maps(a_proj, n))
        elif self.p_proj_rel_6 is not None and self.dense_bf16x6:
            ops.conv_mfma(a_proj, self.p_proj_rel_6, n, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc,
                          x=enc["feats"], u=U, map2img=row2img, out0=r_feat, bf16x6=1)
        else:
            ops.conv_mfma(a_proj, self.p_proj_rel, n, 0, H, Cc, 1, EPI_REL, pix_per_map=P, oc_split=Cc,
                          x=enc["feats"], u=U, map2img=row2img, out0=r_feat)
        return r_feat, rs["r_words"], row2img

    def explain_batch_graph(self, images, captions, accumulate=False, predictions=False):
        """`explain_batch` replayed from a captured HIP graph (one graph per (B,T) shape): the ~450 kernel launches
        of a step (decoder trace and lock-step relevance are launch-bound) are issued by one hipGraphLaunch instead
        of the Python loop.  Inputs are copied into the graph's static buffers; the returned tensors are the graph's
        static outputs (overwritten by the next call with the same shape)."""
        images = images.to(self.device, torch.float32)
        captions = captions.to(self.device, torch.int64)
        key = (tuple(images.shape), tuple(captions.shape), bool(accumulate), bool(predictions), self._f16())
        g = self._graphs.get(key) if hasattr(self, "_graphs") else None
        if g is None:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            st_img, st_cap = images.clone(), captions.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up outside capture (kernel attributes, caches)
                self.explain_batch(st_img, st_cap, accumulate=accumulate, predictions=predictions)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.explain_batch(st_img, st_cap, accumulate=accumulate, predictions=predictions)
            g = self._graphs[key] = (graph, st_img, st_cap, out)
        graph, st_img, st_cap, out = g
        st_img.copy_(images)
        st_cap.copy_(captions)
        graph.replay()
        return out

    def explain_batch_replay(self, images, captions, accumulate=False, predictions=False):
        """`explain_batch` as a RECORDED step (lrp_amd._lib.Recording): the first call with a given input shape runs the step eagerly
        on static copies of the inputs and keeps its library calls - functions, arguments, and every buffer they point at; later calls
        copy the inputs into those static buffers and issue the same calls again: the same kernels in the same order on the same
        stream, as ordinary launches (they overlap with other streams' kernels like any launch; a HIP graph replay did not), without the
        interpreter's ~9 us per launch.  Bit-identical to `explain_batch` by construction.  Like a graph's, the returned tensors are
        the recording's static outputs: overwritten by the next call of the same shape on this engine (take `replica()`s for batches
        in flight).  Captions of equal length only (`lens` makes the launch sequence data-dependent)."""
        src = images
        src = src.to(self.device, torch.float32)
        captions = captions.to(self.device, torch.int64)
        key = (tuple(src.shape), tuple(captions.shape), bool(accumulate), bool(predictions), _lib.stream_ptr().value,
               self.vgg.conv_mode if self.vgg is not None else None, self._f16())
        if not hasattr(self, "_recordings"):
            self._recordings = {}
        rec = self._recordings.get(key)
        if rec is None:
            st_src, st_cap = src.clone(), captions.clone()
            # warm-up outside the recording: one-time work (kernel attributes, index caches, workspace allocations) must not be replayed
            self.explain_batch(st_src, st_cap, accumulate=accumulate, predictions=predictions)
            rec = _lib.Recording()
            with rec:
                rec.result = self.explain_batch(st_src, st_cap, accumulate=accumulate, predictions=predictions)
            rec.inputs = (st_src, st_cap)
            self._recordings[key] = rec
            return rec.result
        st_src, st_cap = rec.inputs
        st_src.copy_(src)
        st_cap.copy_(captions)
        return rec.replay()

    def guided_gradient(self, enc, tr, lens=None, mask_features=True):
        """ExplainiGridTDGuidedGradient.explain_caption_wordt (gridTDmodel.py:1588-1675) for every (image, word)
        row: decoder BPTT with alpha/beta constant.  `tr` must be a grad=True trace.
        Returns d_feat (B*T, P, C), r_words (B*T, T), row2img.  With `lens` d_feat holds the valid rows only (as `relevance`)."""
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E, P, Cc = tr["B"], tr["T"], self.H, self.E, self.P, self.C
        rows = B * T
        rg = ragged(lens, B, T, self.device)
        lens = rg.lens if rg is not None else None
        e = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        gs = dict(d_h2n=e(rows, H), d_c2=e(rows, H), d_c1=e(rows, H), d_ch0=e(rows, H), d_h2p=e(rows, H),
                  d_glob=e(rows, E), gates=e(rows, 4 * H), dx=e(rows, 3 * H), wacc=ops.zeros(rows, T, H, device=self.device),
                  r_words=e(rows, T))
        c = GridGradState()
        c.lens = ptr(lens)
        for k, v in gs.items():
            setattr(c, k, ptr(v))
        ctr, cgs = C.byref(tr["_c"]), C.byref(c)
        _, row2img = self._row_index(B, T)
        check(lib.lrpx_gridtd_grad_init(ctr, cgs, ptr(self.sd["fc.weight"]), ptr(tr["captions"]), T + 1, st))
        for s in range(T):
            check(lib.lrpx_gridtd_grad_step(ctr, cgs, s, 0, st))
            ops.conv_mfma(gs["gates"], self.p_g2, rows, 0, 4 * H, 3 * H, 1, EPI_PLAIN, pix_per_map=1, oc_split=3 * H,
                          out0=gs["dx"])
            check(lib.lrpx_gridtd_grad_step(ctr, cgs, s, 1, st))
            ops.conv_mfma(gs["gates"], self.p_g1, rows, 0, 4 * H, 2 * E + H, 1, EPI_PLAIN, pix_per_map=1,
                          oc_split=2 * E + H, out0=gs["dx"])
            check(lib.lrpx_gridtd_grad_step(ctr, cgs, s, 2, st))
        d_avg = e(rows, Cc)
        ops.conv_mfma(gs["d_glob"], self.p_gp_grad, rows, 0, E, Cc, 1, EPI_PLAIN, pix_per_map=1, oc_split=Cc, out0=d_avg)
        U = e(rows, Cc)
        check(lib.lrpx_scale(ptr(d_avg), ptr(U), d_avg.numel(), 1.0 / P, st))                    # :1667
        check(lib.lrpx_rel_words_norm(ptr(gs["r_words"]), rows, T, st))
        rowlist = None
        if rg is not None and not rg.full:
            rows, rowlist, row2img = rg.n, rg.rows, rg.row2img
            if rows == 0:
                return e(0, P, Cc), gs["r_words"], row2img
            U = ops.gather_rows(U, rowlist)
        a_proj = e(rows, P, H)
        check(lib.lrpx_spread_pixels_rows(ptr(gs["wacc"]), ptr(tr["alpha"]), ptr(lens), ptr(a_proj), B, T, H, P, ptr(rowlist),
                                          rows, st))
        if mask_features:
            mask = e(B, P, Cc)
            check(lib.lrpx_positive_mask(ptr(enc["feats"]), ptr(mask), mask.numel(), st))         # :1674
        else:   # ExplainGridTDGradient.explain_caption_wordt (:1424-1505): same BPTT, no `features <= 0` gate
            mask = torch.ones(B, P, Cc, device=self.device, dtype=torch.float32)
        d_feat = e(rows, P, Cc)
        if self.p_proj_rel_h is not None and self._f16():      # the split-fp16 GEMM of the relevance path (fp32-grade, 4x the fp32 MFMA's rate)
            ops.conv_mfma(a_proj, self.p_proj_rel_h, rows, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc, x=mask,
                          u=U, map2img=row2img, out0=d_feat, f16x3=1, in_amax=ops.amax_maps(a_proj, rows))
        elif self.p_proj_rel_6 is not None and self.dense_bf16x6:
            ops.conv_mfma(a_proj, self.p_proj_rel_6, rows, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc, x=mask,
                          u=U, map2img=row2img, out0=d_feat, bf16x6=1)
        else:
            ops.conv_mfma(a_proj, self.p_proj_rel, rows, 0, H, Cc, 1, EPI_REL, pix_per_map=P, oc_split=Cc, x=mask, u=U,
                          map2img=row2img, out0=d_feat)                                          # :1668, :1674
        return d_feat, gs["r_words"], row2img

    def explain_batch_guided(self, images, captions, lens=None, return_features=False, gradcam=False):
        """Batched `ExplainiGridTDGuidedGradient.explain_caption`: guided-backprop maps (B,T,3,224,224) and word
        scores (B,T,T).  (No running sums here: the reference zeroes the image gradient per word, :1717.)
        gradcam=True: `ExplainGridTDGuidedGradCam` (:1796-1836) - every map times the 16x expanded Grad-CAM heat map of the
        same (guided) decoder gradient."""
        images = images.to(self.device, torch.float32).contiguous()
        captions = captions.to(self.device, torch.int64).contiguous()
        B, T = captions.shape[0], captions.shape[1] - 1
        enc = self.encode(images)
        tr = self.trace(enc, captions, predictions=False, grad=True)
        rg = ragged(lens, B, T, self.device)
        d_feat, r_words, row2img = self.guided_gradient(enc, tr, rg)
        if d_feat.shape[0] == 0:                  # every caption empty
            maps = d_feat.new_zeros(0, 3, 224, 224)
        else:
            maps = self.vgg.guided_backprop(d_feat, row2img)
            if gradcam:
                maps = ops.guided_gradcam(maps, self.grad_cam(enc, d_feat, row2img), int(round(self.P ** 0.5)))
        if rg is not None and not rg.full:        # back to the padded (image, word) layout, zeros behind the last word
            maps = ops.scatter_maps(maps, rg)
            if return_features:
                d_feat = ops.scatter_maps(d_feat, rg)
        out = (maps.view(B, T, 3, 224, 224), r_words.view(B, T, T))
        if return_features:
            out = out + (d_feat.view(B, T, self.P, self.C), tr, enc)
        return out

    def explain_batch_gradient(self, images, captions, lens=None, cam=False, return_features=False):
        """Batched `ExplainGridTDGradient.explain_caption` (models/gridTDmodel.py:1214-1539): plain
        decoder gradient + autograd gradient through the encoder -> maps (B,T,3,224,224), word scores (B,T,T).
        cam=True: `ExplainGridTDGradCam` (:1752-1771) - the per-word result is the Grad-CAM heat map (B,T,196)."""
        images = images.to(self.device, torch.float32).contiguous()
        captions = captions.to(self.device, torch.int64).contiguous()
        B, T = captions.shape[0], captions.shape[1] - 1
        enc = self.encode(images)
        tr = self.trace(enc, captions, predictions=False, grad=True)
        rg = ragged(lens, B, T, self.device)
        d_feat, r_words, row2img = self.guided_gradient(enc, tr, rg, mask_features=False)
        if d_feat.shape[0] == 0:
            maps = d_feat.new_zeros((0, self.P) if cam else (0, 3, 224, 224))
        elif cam:
            maps = self.grad_cam(enc, d_feat, row2img)
        else:
            maps = self.vgg.gradient(d_feat, row2img)
        if rg is not None and not rg.full:
            maps = ops.scatter_maps(maps, rg)
            if return_features:
                d_feat = ops.scatter_maps(d_feat, rg)
        maps = maps.view(B, T, self.P) if cam else maps.view(B, T, 3, 224, 224)
        out = (maps, r_words.view(B, T, T))
        if return_features:
            out = out + (d_feat.view(B, T, self.P, self.C), tr, enc)
        return out

    def grad_cam(self, enc, d_feat, row2img):
        """`grad_cam` (models/gridTDmodel.py:1760-1771) for every (image, word) row: (rows,P,C) gradients -> (rows,P)."""
        rows = d_feat.shape[0]
        cam = torch.empty(rows, self.P, device=self.device, dtype=torch.float32)
        check(_lib.load().lrpx_gradcam(ptr(enc["feats"]), ptr(d_feat.contiguous()), ptr(row2img), ptr(cam), rows, self.P,
                                       self.C, stream_ptr()))
        return cam

    def explain_batch(self, images, captions, lens=None, accumulate=False, return_features=False, predictions=False):
        """Batched `explain_caption` (gridTDmodel.py:1141-1156): images (B,3,224,224), captions (B,T+1) int64.
        Returns maps (B,T,3,224,224) and r_words (B,T,T) (row t holds t+1 valid entries).
        accumulate=True reproduces the running sums the reference returns (lrp_wrapper.py:64-82 quirk).
        predictions=True also computes the (B,T,V) scores the reference's explainer keeps (`self.predictions`, :1011; read
        by evaluation.py:109) and returns them as a third tensor."""
        images = images.to(self.device, torch.float32).contiguous()
        captions = captions.to(self.device, torch.int64).contiguous()
        B, T = captions.shape[0], captions.shape[1] - 1
        enc = self.encode(images)
        tr = self.trace(enc, captions, predictions=predictions)
        rg = ragged(lens, B, T, self.device)
        r_feat, r_words, row2img = self.relevance(enc, tr, rg)
        if rg is not None and not rg.full:
            # unequal caption lengths (models/gridTDmodel.py:1147-1153 explains `caption_length` words): the chain runs on
            # the sum(lens) valid maps; the result goes back to the padded layout (running sums per image over ITS words)
            maps = self.vgg.relevance(r_feat, row2img) if rg.n else r_feat.new_zeros(0, 3, 224, 224)
            maps = ops.scatter_maps(maps, rg, accumulate=accumulate)
            if return_features:
                r_feat = ops.scatter_maps(r_feat, rg)
        else:
            maps = self.vgg.relevance(r_feat, row2img)
            if accumulate:
                maps = ops.cumsum_maps(maps, B, T)
        out = (maps.view(B, T, 3, 224, 224), r_words.view(B, T, T))
        if predictions:
            out = out + (tr["pred"],)
        if return_features:
            out = out + (r_feat.view(B, T, self.P, self.C), tr, enc)
        return out

    def _f16(self):
        """the decoder GEMMs on the fp16 split products?  (ops.decoder_f16: with conv modes 2 / 3 only - the engine's own `vgg.conv_mode` or the
        process default; `force_f16` overrides per engine)"""
        if self.force_f16 is not None:
            return bool(self.force_f16)
        return ops.decoder_f16(self.vgg.conv_mode if self.vgg is not None else None)

    def replica(self):
        """A second execution context over the SAME weights (device tensors and packed blobs are shared): own VGG16
        trace / workspace buffers, so two batches can be in flight on two HIP streams."""
        import copy
        r = copy.copy(self)
        r.vgg = self.vgg.replica()
        r._idx_cache = {}
        for k in ("_graphs", "_replicas", "_streams", "_recordings"):     # a replica never shares another engine's streams / buffer sets
            r.__dict__.pop(k, None)
        return r

    def explain_stream(self, batches, depth=3, accumulate=False):
        """Explain an iterable of independent (images, captions) batches with `depth` batches in flight, each on its own
        HIP stream and buffer set.  Batches are independent (no exchange step), and roughly a seventh of
        a batch's time is the decoder's lock-step chain of small latency-bound kernels: it overlaps the MFMA-bound CNN
        relevance chain of the neighbouring batch.  Yields (maps, r_words) in input order; each result is complete
        (its stream has been synchronised) when it is yielded.  Results are bit-identical to `explain_batch`."""
        depth = max(1, int(depth))
        if not hasattr(self, "_replicas"):
            self._replicas, self._streams = [self], [torch.cuda.Stream(device=self.device)]
        while len(self._replicas) < depth:
            self._replicas.append(self.replica())
            self._streams.append(torch.cuda.Stream(device=self.device))
        pending = []
        for i, batch in enumerate(batches):
            images, captions = batch[0], batch[1]
            lens = batch[2] if len(batch) > 2 else None                 # (images, captions[, lens])
            k = i % depth
            st = self._streams[k]
            st.wait_stream(torch.cuda.current_stream(self.device))     # inputs produced on the caller's stream
            with torch.cuda.stream(st):
                out = self._replicas[k].explain_batch(images, captions, lens=lens, accumulate=accumulate)
                ev = torch.cuda.Event()
                ev.record(st)
            for t in out:
                t.record_stream(torch.cuda.current_stream(self.device))
            # the side stream reads the caller's tensors (`.to()` copies nothing when they already are device fp32 / int64):
            # keep them alive until the batch's event has completed, or the caching allocator could hand their memory to
            # the caller's next batch while this one is still queued
            pending.append((out, ev, images, captions))
            if len(pending) >= depth:
                o, e, _, _ = pending.pop(0)
                e.synchronize()
                yield o
        for o, e, _, _ in pending:
            e.synchronize()
            yield o


# ------------------------------------------------------------------------------------------------
# drop-in explainer (models/gridTDmodel.py:705-1211)
# ------------------------------------------------------------------------------------------------
class ExplainGridTDAttention(object):
    """Same constructor, attributes and methods as the reference's `ExplainGridTDAttention`
    (models/gridTDmodel.py:705-1156): `explain_caption(img_filepath) -> (relevance_imgs, relevance_preceeding_words)`,
    `explain_caption_wordt(t)`, `explain_cnn(R)`, `teacherforce_forward(img, ids)`; attributes `.model .word_map .img
    .beam_caption .beam_caption_encode .predictions .alphas .betas .args`.

    `model` may be the reference's `GridTDModel` (any nn.Module with that `state_dict`), a `state_dict`, or None
    (then `args.weight` is loaded like :717-718).  Without `caption_encode=` the image is captioned as the reference does
    it (`beam_search(beam_size=2, max_cap_length=50)`, :935-937; `GridTDEngine.beam_search`), so the same caption is
    explained.  Nothing is written to disk (visualisation is out of scope)."""
    EPS = 0.01
    EX_TYPE = 'lrp'

    def __init__(self, args, word_map, model=None):
        self.args = args
        self.word_map = word_map
        self.vocab_size = len(word_map)
        # one device engine per weight set (explainers/engine_cache.py): evaluation.py:806-838 builds an explainer per image
        from . import engine_cache
        key = engine_cache.fingerprint("gridtd", args.weight if model is None else model)

        def build():
            if model is None:
                state = torch.load(args.weight, map_location="cpu")['state_dict']
            elif hasattr(model, "state_dict"):
                state = model.state_dict()
            else:
                state = model
            return GridTDEngine(state)
        self.model = model
        # the weights are shared, the trace / workspace buffers are this explainer's own: two live explainers never see each other's image
        self.engine = engine_cache.get(key, build, hold=engine_cache.source_tensors(model)).replica()
        self.mean = list(IMAGENET_MEAN)
        self.std = list(IMAGENET_STD)
        self.rev_word_map = {v: k for k, v in word_map.items()}

    def preprocess_img(self, img_filepath):
        """Resize -> ToTensor -> Normalize (models/gridTDmodel.py:767-771), host side."""
        return load_image(img_filepath, getattr(self.args, "height", 224), getattr(self.args, "width", 224), self.mean, self.std,
                          self.engine.device)

    def get_hidden_parameters(self, img, caption_encode=None, max_cap_length=50):
        """Forward trace (:933-1012).  `img`: file path or a (1,3,224,224) tensor."""
        self.img = self.preprocess_img(img) if isinstance(img, str) else img.to(self.engine.device, torch.float32)
        eng = self.engine
        # a caption that is handed over goes to the device BEFORE the encoder is enqueued: the copy of a pageable host list waits for the
        # stream, and behind the VGG16 forward it stalled the host for 1 ms per call (the device then idled until the decoder was issued)
        cap_dev = None if caption_encode is None else torch.tensor([[int(c) for c in caption_encode]], dtype=torch.int64, device=eng.device)
        self._enc = eng.encode(self.img)
        if caption_encode is None:
            from .beam import caption_from_sequence
            seq = eng.beam_search(self._enc, 2, max_cap_length, self.word_map['<start>'], self.word_map['<end>'])   # :935
            caption_encode = caption_from_sequence(seq, self.word_map)
        self.beam_caption_encode = [int(c) for c in caption_encode]
        special = {self.word_map[k] for k in ('<start>', '<end>', '<unk>', '<pad>') if k in self.word_map}
        self.beam_caption = [' '.join(self.rev_word_map.get(c, str(c)) for c in self.beam_caption_encode[1:]
                                      if c not in special)]
        self.caption_length = len(self.beam_caption_encode) - 1
        self.num_pixels = eng.P
        self._rel = None
        if self.caption_length == 0:
            return
        cap = cap_dev if cap_dev is not None else torch.tensor([self.beam_caption_encode], dtype=torch.int64, device=eng.device)
        self._cap_dev = cap
        self._tr = eng.trace(self._enc, cap, predictions=True)
        self.image_features = ops.nhwc_to_nchw(self._enc["feats"].contiguous(), eng.C, 14, 14)
        self.num_pixels = eng.P
        self.predictions = self._tr["pred"][0]
        self.alphas = self._tr["alpha"][0]
        self.betas = self._tr["beta"][0]
        self._rel = None

    def _relevance(self):
        if self._rel is None:
            self._rel = self.engine.relevance(self._enc, self._tr)
        return self._rel

    def explain_caption_wordt(self, t):
        """(:1014-1135) -> (r_img_feature (1,C,h,w), r_words (t+1,))"""
        assert t < self.caption_length
        r_feat, r_words, _ = self._relevance()
        r = ops.nhwc_to_nchw(r_feat[t:t + 1].contiguous(), self.engine.C, 14, 14)
        return r, r_words[t, :t + 1].clone()

    def explain_cnn(self, r_img_feature):
        """(:1137-1139) — `compute_lrp` on `self.img`; like the reference the result accumulates over calls on
        the same image (`sample.grad`, lrp_wrapper.py:64-82)."""
        t_nhwc = ops.nchw_to_nhwc(r_img_feature.to(torch.float32))
        r = self.engine.vgg.relevance(t_nhwc, torch.zeros(r_img_feature.shape[0], dtype=torch.int32,
                                                          device=self.engine.device))
        if getattr(self, "_img_grad", None) is None:
            self._img_grad = r
        else:
            check(_lib.load().lrpx_accumulate(ptr(self._img_grad), ptr(r), r.numel(), stream_ptr()))
        ops.check_relevance(self._img_grad, finite=True, nonzero=True)
        return self._img_grad.clone()

    def explain_caption(self, img_filepath, t_list=None, caption_encode=None):
        """(:1141-1156) -> ([T] x (1,3,H,W), [T] x (t+1,)); maps are the reference's running sums."""
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:          # the beam search produced <end> first: nothing to explain (empty lists, :1147-1156)
            return [], []
        self._img_grad = None
        r_feat, r_words, row2img = self._relevance()
        maps = self.engine.vgg.relevance(r_feat, row2img)
        maps = ops.cumsum_maps(maps, 1, self.caption_length)
        ops.check_relevance(maps, finite=True, nonzero=True)
        relevance_imgs = [maps[t:t + 1] for t in range(self.caption_length)]
        relevance_preceeding_words = [r_words[t, :t + 1] for t in range(self.caption_length)]
        assert len(relevance_imgs) == self.caption_length
        return relevance_imgs, relevance_preceeding_words

    TF_MODEL_BIAS = False      # the LRP explainer's LanguageLSTM forward adds bias_ih twice (:789); the gradient family's is correct (:1265)

    def teacherforce_forward(self, img, beam_caption_encode):
        """(:892-931; gradient family :1282-1321) -> predictions (len(beam_caption_encode), V) under teacher forcing: step t reads
        token t - evaluation.py:266,437 hands the caption WITH <start> - with this explainer's own forward."""
        eng = self.engine
        if isinstance(img, str):
            img = self.preprocess_img(img)
        enc = eng.encode(img.to(eng.device, torch.float32))
        cap = torch.tensor([[int(c) for c in beam_caption_encode] + [0]], dtype=torch.int64, device=eng.device)
        n = cap.shape[1] - 1
        tr = eng.trace(enc, cap, model_bias=self.TF_MODEL_BIAS, predictions=False)
        return eng.logits(tr["hc"].view(n, eng.H))       # the fp32 kernel of the decoding loops, at any caption length


class ExplainiGridTDGuidedGradient(ExplainGridTDAttention):
    """Drop-in for the reference's `ExplainiGridTDGuidedGradient` (models/gridTDmodel.py:1585-1723; the spelling is
    the reference's): guided backprop instead of LRP, same `explain_caption` surface."""
    EX_TYPE = 'GuidedBackpropagate'
    TF_MODEL_BIAS = True

    def get_hidden_parameters(self, img, caption_encode=None, max_cap_length=50):
        super().get_hidden_parameters(img, caption_encode, max_cap_length)
        if self.caption_length == 0:
            return
        self._tr = self.engine.trace(self._enc, self._cap_dev, predictions=True, grad=True)     # :1323-1422 (correct LSTM bias)
        self.predictions = self._tr["pred"][0]
        self.alphas, self.betas = self._tr["alpha"][0], self._tr["beta"][0]

    def _relevance(self):
        if self._rel is None:
            self._rel = self.engine.guided_gradient(self._enc, self._tr)
        return self._rel

    def explain_cnn(self, d_img_feature):
        t_nhwc = ops.nchw_to_nhwc(d_img_feature.to(torch.float32))
        return self.engine.vgg.guided_backprop(t_nhwc, torch.zeros(d_img_feature.shape[0], dtype=torch.int32,
                                                                   device=self.engine.device))

    def explain_caption(self, img_filepath, t_list=None, caption_encode=None):
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:          # the beam search produced <end> first: nothing to explain (empty lists, :1147-1156)
            return [], []
        d_feat, r_words, row2img = self._relevance()
        maps = self.engine.vgg.guided_backprop(d_feat, row2img)
        return ([maps[t:t + 1] for t in range(self.caption_length)],
                [r_words[t, :t + 1] for t in range(self.caption_length)])


class ExplainGridTDGuidedGradCam(ExplainiGridTDGuidedGradient):
    """Drop-in for `ExplainGridTDGuidedGradCam` (models/gridTDmodel.py:1796-1836): the guided-backprop map of every word
    times the Grad-CAM heat map of the same decoder gradient expanded 16x by `skimage.transform.pyramid_expand` (:1826).
    (`d_img_feature[self.image_features < 0] = 0`, :1815, never selects anything: the features are ReLU outputs.)"""
    EX_TYPE = 'GuidedGradCam'

    def grad_cam(self, img_feature, grads):
        """(1,C,h,w) features and gradients -> (h, w) heat map (:1799-1810)"""
        f = ops.nchw_to_nhwc(img_feature.to(torch.float32))
        g = ops.nchw_to_nhwc(grads.to(torch.float32))
        P, Cc = f.shape[1], f.shape[2]
        cam = torch.empty(1, P, device=self.engine.device, dtype=torch.float32)
        check(_lib.load().lrpx_gradcam(ptr(f), ptr(g), ptr(None), ptr(cam), 1, P, Cc, stream_ptr()))
        return cam.view(img_feature.shape[-2], img_feature.shape[-1])

    def explain_cnn(self, d_img_feature):
        guided = super().explain_cnn(d_img_feature)
        cam = self.grad_cam(self.image_features, d_img_feature).reshape(1, -1)
        return ops.guided_gradcam(guided, cam, self.image_features.shape[-1])

    def explain_caption(self, img_filepath, t_list=None, caption_encode=None):
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:          # the beam search produced <end> first: nothing to explain (empty lists, :1147-1156)
            return [], []
        d_feat, r_words, row2img = self._relevance()
        eng = self.engine
        maps = ops.guided_gradcam(eng.vgg.guided_backprop(d_feat, row2img), eng.grad_cam(self._enc, d_feat, row2img),
                                  int(round(eng.P ** 0.5)))
        return ([maps[t:t + 1] for t in range(self.caption_length)],
                [r_words[t, :t + 1] for t in range(self.caption_length)])


class ExplainGridTDGradient(ExplainiGridTDGuidedGradient):
    """Drop-in for the reference's `ExplainGridTDGradient` (models/gridTDmodel.py:1214-1539): plain gradient - the same
    hand-written decoder BPTT as the guided explainer without its three gates, and the autograd gradient through the
    encoder (`explain_cnn`, :1507-1521).  (In the reference the guided class derives from this one; here the
    inheritance runs the other way, the surface is the same.)"""
    EX_TYPE = 'gradient'

    def _relevance(self):
        if self._rel is None:
            self._rel = self.engine.guided_gradient(self._enc, self._tr, mask_features=False)
        return self._rel

    def explain_cnn(self, d_img_feature):
        t_nhwc = ops.nchw_to_nhwc(d_img_feature.to(torch.float32))
        return self.engine.vgg.gradient(t_nhwc, torch.zeros(d_img_feature.shape[0], dtype=torch.int32,
                                                            device=self.engine.device))

    def explain_caption(self, img_filepath, t_list=None, caption_encode=None):
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:          # the beam search produced <end> first: nothing to explain (empty lists, :1147-1156)
            return [], []
        d_feat, r_words, row2img = self._relevance()
        maps = self._maps(d_feat, row2img)
        return ([maps[t:t + 1] for t in range(self.caption_length)],
                [r_words[t, :t + 1] for t in range(self.caption_length)])

    def _maps(self, d_feat, row2img):
        return self.engine.vgg.gradient(d_feat, row2img)


class ExplainGridTDGradCam(ExplainGridTDGradient):
    """Drop-in for `ExplainGridTDGradCam` (models/gridTDmodel.py:1752-1771): `explain_caption` returns per word the
    (1, 196) Grad-CAM heat map of the plain decoder gradient."""
    EX_TYPE = 'GradCam'

    def grad_cam(self, img_feature, grads):
        """(1,C,h,w) features and gradients -> (h*w,) heat map, as the reference's method of the same name."""
        f = ops.nchw_to_nhwc(img_feature.to(torch.float32))
        g = ops.nchw_to_nhwc(grads.to(torch.float32))
        P, Cc = f.shape[1], f.shape[2]
        cam = torch.empty(1, P, device=self.engine.device, dtype=torch.float32)
        check(_lib.load().lrpx_gradcam(ptr(f), ptr(g), ptr(None), ptr(cam), 1, P, Cc, stream_ptr()))
        return cam.view(-1)

    def explain_cnn(self, d_img_feature):
        return self.grad_cam(self.image_features, d_img_feature).unsqueeze(0)

    def _maps(self, d_feat, row2img):
        return self.engine.grad_cam(self._enc, d_feat, row2img)

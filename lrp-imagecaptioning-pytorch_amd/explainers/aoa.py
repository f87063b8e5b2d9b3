"""Batched AoA (attention-on-attention, 8-head) LRP engine + the drop-in `ExplainAOAAttention` explainer.

Reference: models/aoamodel.py:748-1254.  Also serves the bottom-up variant (36 x 2048 region features, no CNN
stage; SURVEY §8(a) row A-BU): pass `features=` instead of images.  Host logic only sequences HIP kernels."""
import ctypes as C

import numpy as np
import torch

from .. import _lib, ops
from .._lib import (AoaGradState, AoaRelState, AoaStepArgs, AoaTrace, EPI_PLAIN, EPI_REL, PACK_DENSE, PACK_DENSE_T, STAB_EPS, check, ptr, ptr_at,
                    stream_ptr)
from .gridtd import IMAGENET_MEAN, IMAGENET_STD, VGG_PREFIX, _t, load_image
from .ragged import ragged


class AOAEngine:
    """`state`: the reference `AOAModel` state_dict (models/aoamodel.py:116-142).  Without encoder weights
    (bottom-up model, :1795-1797) only `features=` inputs are accepted."""

    def __init__(self, state, num_head=8, device="cuda"):
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LrpxError("the LRP hot path needs an MI355X; there is no CPU fallback")
        dev = torch.device(device)
        self.device = dev
        sd = {k: _t(v, dev) for k, v in state.items() if not k.startswith(VGG_PREFIX)}
        names = [k for k in state if k.startswith(VGG_PREFIX) and k.endswith(".weight")]
        self.vgg = None
        if names:
            self.vgg = ops.Vgg16([_t(state[k], dev) for k in names],
                                 [_t(state[k.replace(".weight", ".bias")], dev) for k in names])
        self.sd = sd
        self.NH = num_head
        self.V, self.E = sd["embedding.weight"].shape
        self.H = sd["fc.weight"].shape[1]
        w_proj = sd["img_projector.weight"]
        self.C = w_proj.shape[1]
        H, E, Cc = self.H, self.E, self.C
        assert H == 512, "kernels are built for hidden=512 (config.py:188)"
        l = "LanguageLSTM."
        self.Wcat = torch.cat([sd[l + "weight_ih"], sd[l + "weight_hh"]], 1).contiguous()          # (4H, E+2H)
        self.bcat_explainer = (sd[l + "bias_ih"] + sd[l + "bias_ih"]).contiguous()                  # quirk :873
        self.bcat_model = (sd[l + "bias_ih"] + sd[l + "bias_hh"]).contiguous()
        # gate rows interleaved for the fused decoder step (lrpx_aoa_fwd_steps): row 16 j + 4 gate + u = row gate * H + 4 j + u, so
        # that one workgroup of the gate linear holds the i, f, g, o pre-activations of the hidden units 4j .. 4j+3
        jj, qq, uu = torch.meshgrid(torch.arange(H // 4), torch.arange(4), torch.arange(4), indexing="ij")
        il = (qq * H + 4 * jj + uu).reshape(-1).to(self.device)
        self.Wcat_il = self.Wcat[il].contiguous()
        self.bcat_model_il, self.bcat_explainer_il = self.bcat_model[il].contiguous(), self.bcat_explainer[il].contiguous()
        # the decoupled teacher-forced trace (lrpx_aoa_fwd_recurrence, include/lrpx.h): input part and recurrent part of the gate rows
        self.decoupled = H % 16 == 0 and E % 16 == 0          # False: the stepwise kernels (A/B; the decoding loops always use them)
        self.W_ih_il = self.Wcat_il[:, :E + H].contiguous()
        self.W_hh_il = self.Wcat_il[:, E + H:].contiguous()
        self.Wqg = torch.cat([sd["decoder_multihead_attention.q_proj.weight"], sd["decoder_aoa_linear_gate.weight"]], 0).contiguous()
        self.bqg = torch.cat([sd["decoder_multihead_attention.q_proj.bias"], sd["decoder_aoa_linear_gate.bias"]]).contiguous()
        self.w_proj2d = w_proj.reshape(H, Cc).contiguous()
        kc = ops.conv_kc(0, 1, Cc)
        self.p_proj_fwd = ops.pack_weights(self.w_proj2d, H, Cc, 1, PACK_DENSE, kc)
        self.p_k_fwd = ops.pack_weights(sd["decoder_k_proj.weight"], H, H, 1, PACK_DENSE, kc)
        self.p_v_fwd = ops.pack_weights(sd["decoder_v_proj.weight"], H, H, 1, PACK_DENSE, kc)
        self.p_fc_fwd = ops.pack_weights(sd["fc.weight"], self.V, H, 1, PACK_DENSE, kc)
        # fp16 split-product packs (csrc/dense_f16x3.hip): built always, USED only while ops.decoder_f16() says so (`_f16()`: conv modes 2 / 3)
        self.force_f16 = None            # True / False: this engine's decoder GEMMs on / off the fp16 split products whatever the conv mode (A/B and tests)
        self.p_fc_fwd_h = ops.pack_weights_f16x2(sd["fc.weight"], self.V, H, _lib.PACK_FWD, taps=1) if H % 64 == 0 else None
        # plain GEMMs over all (image, word) rows of the decoupled trace: (pack for the fp16 split-product kernel, fp32 pack)
        self._plain = {}
        self.tok_table = None
        if self.decoupled:
            for name, w in (("ih", self.W_ih_il), ("qg", self.Wqg), ("lin", sd["decoder_aoa_linear.weight"])):
                n, k = w.shape
                self._plain[name] = (ops.pack_weights_f16x2(w, n, k, _lib.PACK_FWD, taps=1) if k % 64 == 0 else None,
                                     ops.pack_weights(w, n, k, 1, PACK_DENSE, kc), n, k)
            # The embedding part of the gate pre-activations depends on the TOKEN alone: tok_table[v] = embedding[v] W_ie^T (V x 4H, once per
            # model, exact fp32 products on the fp32 matrix cores); the image part is one small linear per trace (W_ig, bias).  The trace then
            # needs no GEMM over its B*T rows for the LSTM input at all (lrpx_aoa_fwd_recurrence_tab).
            w_ie = self.W_ih_il[:, :E].contiguous()
            self.W_ig_il = self.W_ih_il[:, E:].contiguous()
            self.tok_table = torch.empty(self.V, 4 * H, device=self.device)
            ops.conv_mfma(sd["embedding.weight"], ops.pack_weights(w_ie, 4 * H, E, 1, PACK_DENSE, kc), self.V, 0, E, 4 * H, 1, EPI_PLAIN,
                          pix_per_map=1, oc_split=4 * H, out0=self.tok_table)
        wg = torch.cat([sd[l + "weight_ih"][2 * H:3 * H], sd[l + "weight_hh"][2 * H:3 * H]], 1).contiguous()
        self.p_wg = ops.pack_weights(wg, H, E + 2 * H, 1, PACK_DENSE_T, kc)
        self.p_lin_rel = ops.pack_weights(sd["decoder_aoa_linear.weight"], H, H, 1, PACK_DENSE_T, kc)
        self.p_v_rel = ops.pack_weights(sd["decoder_v_proj.weight"], H, H, 1, PACK_DENSE_T, kc)
        self.p_proj_rel = ops.pack_weights(self.w_proj2d, H, Cc, 1, PACK_DENSE_T, kc)
        # the v_proj / projector rules run over every (word, pixel) row: split products on the fp16 matrix cores
        # (csrc/dense_f16x3.hip)
        self.p_v_rel_h = self.p_proj_rel_h = None
        self.p_v_rel_head = None
        if H % 64 == 0:
            self.p_v_rel_h = ops.pack_weights_f16x2(sd["decoder_v_proj.weight"], H, H, _lib.PACK_BWD_PLAIN, taps=1)
            self.p_proj_rel_h = ops.pack_weights_f16x2(self.w_proj2d, H, Cc, _lib.PACK_BWD_PLAIN, taps=1)
            dk = H // self.NH
            if dk % 64 == 0:      # `lrp_mha` passes one head: the v_proj rule contracts over that head's dk rows of W_v only
                self.p_v_rel_head = [ops.pack_weights_f16x2(sd["decoder_v_proj.weight"][h * dk:(h + 1) * dk].contiguous(), dk, H,
                                                            _lib.PACK_BWD_PLAIN, taps=1) for h in range(self.NH)]
        # ... in the default (exact) arithmetic: the same tiles on the bf16 matrix cores with operands split exactly into three bf16 parts
        # (dense_f16x3.hip, B6: six products, fp32 range - what conv mode 1 is for the VGG16 chains); the fp32 MFMA where the sizes do not fit
        self.p_v_rel_6 = self.p_proj_rel_6 = self.p_v_rel_head6 = None
        if H % 32 == 0 and Cc % 4 == 0:
            self.p_v_rel_6 = ops.pack_weights_bf16x3(sd["decoder_v_proj.weight"], H, H, _lib.PACK_BWD_PLAIN, taps=1)
            self.p_proj_rel_6 = ops.pack_weights_bf16x3(self.w_proj2d, H, Cc, _lib.PACK_BWD_PLAIN, taps=1)
            dk = H // self.NH
            if dk % 32 == 0:
                self.p_v_rel_head6 = [ops.pack_weights_bf16x3(sd["decoder_v_proj.weight"][h * dk:(h + 1) * dk].contiguous(), dk, H,
                                                              _lib.PACK_BWD_PLAIN, taps=1) for h in range(self.NH)]
        self.dense_bf16x6 = True         # False: those rules on the fp32 MFMA kernel (A/B and tests)
        self.head_only = True            # (False: the v_proj rule over all H columns, 7/8 of them zero; A/B and tests)
        # ... and the lock-step gate rule / the aoa_linear rule (rows = images x words): the few-row kernel of the same file
        self.fused_steps = True          # decoder steps as 4 launches instead of 7 (False: the unfused kernels; A/B and tests)
        self.fused_rel = True            # relevance lock-steps as ONE launch each (False: GEMM + point-wise kernel; A/B and tests)
        self.fused_rel_exact = True      # ... also while the decoder GEMMs are exact (modes 0 / 1): dense_ks_kernel<REL, FUSE> (False: two launches; A/B and tests)
        self.lockstep_f16 = H % 16 == 0 and E % 16 == 0
        self.p_wg_h = ops.pack_weights_f16x2(wg, H, E + 2 * H, _lib.PACK_BWD_PLAIN, taps=1) if self.lockstep_f16 else None
        self.p_lin_rel_h = ops.pack_weights_f16x2(sd["decoder_aoa_linear.weight"], H, H, _lib.PACK_BWD_PLAIN, taps=1) if self.lockstep_f16 else None
        # gradient explainers (:1435-1499): contraction over the 4H gate rows / over the outputs of the two aoa linears
        self.p_gates_grad = ops.pack_weights(self.Wcat, 4 * H, E + 2 * H, 1, PACK_DENSE_T, kc)
        self.p_gate_grad = ops.pack_weights(sd["decoder_aoa_linear_gate.weight"], H, H, 1, PACK_DENSE_T, kc)
        torch.cuda.synchronize()
        self._idx_cache = {}

    # ------------------------------------------------------------------------------------------
    def encode(self, images=None, features=None):
        """(:999-1009) image-side constants; `features`: (B,P,C) region/pixel features instead of images."""
        lib = _lib.load()
        st = stream_ptr()
        H, Cc = self.H, self.C
        if features is None:
            if self.vgg is None:
                raise ValueError("this AoA model has no encoder: pass features=(B,P,C)")
            feats = self.vgg.forward(images.to(self.device, torch.float32).contiguous())
        else:
            feats = features.to(self.device, torch.float32).contiguous()
        B, P, _ = feats.shape
        e = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        enc = dict(B=B, P=P, feats=feats)
        enc["proj_pre"] = e(B, P, H)
        ops.conv_mfma(feats, self.p_proj_fwd, B, 0, Cc, H, 1, EPI_PLAIN, pix_per_map=P, oc_split=H,
                      bias=self.sd["img_projector.bias"], out0=enc["proj_pre"])
        enc["Vp"] = e(B, P, H)
        check(lib.lrpx_relu(ptr(enc["proj_pre"]), ptr(enc["Vp"]), enc["Vp"].numel(), st))
        enc["glob"] = e(B, H)
        check(lib.lrpx_mean_pixels(ptr(enc["Vp"]), ptr(enc["glob"]), B, P, H, st))
        enc["key"], enc["value"] = e(B, P, H), e(B, P, H)
        ops.conv_mfma(enc["Vp"], self.p_k_fwd, B, 0, H, H, 1, EPI_PLAIN, pix_per_map=P, oc_split=H,
                      bias=self.sd["decoder_k_proj.bias"], out0=enc["key"])
        ops.conv_mfma(enc["Vp"], self.p_v_fwd, B, 0, H, H, 1, EPI_PLAIN, pix_per_map=P, oc_split=H,
                      bias=self.sd["decoder_v_proj.bias"], out0=enc["value"])
        return enc

    def _alloc_trace(self, B, T, P, grad=False):
        dev, H, E, NH = self.device, self.H, self.E, self.NH
        shapes = {"xh": (B, T, E + 2 * H), "h": (B, T + 1, H), "c": (B, T + 1, H), "alpha": (B, T, NH, P)}
        for k in ("g", "i", "f", "ctx", "lin", "c_aoa", "hc"):
            shapes[k] = (B, T, H)
        if grad:      # the gradient explainers also keep the output gate and sigmoid(aoa gate)  (:1309-1376)
            shapes["o"], shapes["sg"] = (B, T, H), (B, T, H)
        shapes["_amax"] = (3, B * T)      # row maxima (float bits) the decoupled trace records for its GEMMs' operand scales
        tr = dict(B=B, T=T, P=P)
        tr.update(ops.zeros_arena(dev, shapes))            # one allocation, one fill
        c = AoaTrace()
        c.B, c.T, c.H, c.E, c.P, c.NH = B, T, H, E, P, NH
        names = ["xh", "h", "c", "g", "i", "f", "ctx", "lin", "c_aoa", "hc", "alpha"]
        if grad:
            names += ["o", "sg"]
        for k in names:
            setattr(c, k, ptr(tr[k]))
        tr["_c"] = c
        return tr

    def _step(self, tr, enc, t, captions, bias):
        """one decoder step of `get_hidden_parameters` (models/aoamodel.py:1020-1052) for all B images"""
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E = tr["B"], tr["T"], self.H, self.E
        c = C.byref(tr["_c"])
        W = E + 2 * H
        sd = self.sd
        if "_zz" not in tr:
            tr["_zz"] = torch.empty(B, 4 * H, device=self.device)
            tr["_qg"] = torch.empty(B, 2 * H, device=self.device)
            tr["_lin"] = torch.empty(B, H, device=self.device)
        zz, qg, lin = tr["_zz"], tr["_qg"], tr["_lin"]
        check(lib.lrpx_aoa_fwd_pre(c, t, ptr(enc["glob"]), ptr(sd["embedding.weight"]), ptr(captions), captions.shape[1], st))
        check(lib.lrpx_linear_small(ptr_at(tr["xh"], t * W), T * W, ptr(self.Wcat), ptr(bias), ptr(zz), 4 * H, B, W,
                                    4 * H, 0, st))
        check(lib.lrpx_aoa_fwd_lstm(c, t, ptr(zz), 4 * H, st))
        check(lib.lrpx_linear_small(ptr_at(tr["h"], (t + 1) * H), (T + 1) * H, ptr(self.Wqg), ptr(self.bqg), ptr(qg),
                                    2 * H, B, H, 2 * H, 0, st))
        check(lib.lrpx_aoa_fwd_attention(c, t, ptr(qg), 2 * H, ptr(enc["key"]), ptr(enc["value"]), st))
        check(lib.lrpx_linear_small(ptr_at(tr["ctx"], t * H), T * H, ptr(sd["decoder_aoa_linear.weight"]),
                                    ptr(sd["decoder_aoa_linear.bias"]), ptr(lin), H, B, H, H, 0, st))
        check(lib.lrpx_aoa_fwd_post(c, t, ptr(qg), 2 * H, ptr(lin), st))

    def sample_lrp(self, enc, max_length, start_id, end_id, skip_ids):
        """AOAModel.sample_lrp, greedy (models/aoamodel.py:679-745): LRP-inference decoding.  `get_lrp_weight_step`
        (:597-626) is handed the log-softmax of the scores (:721-723), unlike the gridTD model.  Returns (seq int64
        (B,max_length), seq_logprobs (B,max_length)); tokens after <end> are 0, nothing is written once every sequence
        has finished (:742-744)."""
        lib = _lib.load()
        B, T, H = enc["B"], max_length, self.H
        dev = self.device
        skip = torch.zeros(self.V, dtype=torch.uint8, device=dev)
        skip[torch.as_tensor(sorted(int(i) for i in skip_ids), dtype=torch.int64, device=dev)] = 1
        toks = torch.zeros(B, T + 1, dtype=torch.int64, device=dev)
        toks[:, 0] = start_id
        lps = torch.zeros(B, T, dtype=torch.float32, device=dev)
        tr = self._alloc_trace(B, T, enc["P"])
        hcw = torch.empty(B, H, device=dev)
        nxt = torch.empty(B, dtype=torch.int64, device=dev)
        lp = torch.empty(B, dtype=torch.float32, device=dev)
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        for t in range(T):
            self._step(tr, enc, t, toks, self.bcat_model)
            st = stream_ptr()
            pred = self.logits(tr["hc"][:, t].contiguous())
            check(lib.lrpx_lrp_reweight_rows(ptr(pred), self.V, self.V, ptr_at(tr["h"], (t + 1) * H), (T + 1) * H,
                                             ptr_at(tr["c_aoa"], t * H), T * H, ptr(self.sd["fc.weight"]), ptr(skip),
                                             ptr(hcw), B, H, 1, st))
            wpred = self.logits(hcw)
            check(lib.lrpx_argmax_logprob_rows(ptr(wpred), self.V, B, self.V, ptr(nxt), ptr(lp), st))
            alive = unfinished.any()
            unfinished = unfinished & (nxt != end_id)
            toks[:, t + 1] = torch.where(alive, nxt * unfinished, torch.zeros_like(nxt))
            lps[:, t] = torch.where(alive, lp, torch.zeros_like(lp))
        return toks[:, 1:].contiguous(), lps

    def beam_search(self, enc, beam_size, max_cap_length, start_id, end_id):
        """`AOAModel.beam_search` (models/aoamodel.py, the algorithm of models/gridTDmodel.py:400-478 on the AoA step) for
        ONE image: returns the chosen token sequence incl. <start>."""
        from .beam import run_beam_search
        assert enc["B"] == 1, "beam search captions one image"
        nb = int(beam_size)
        encb = {k: (v.expand(nb, *v.shape[1:]).contiguous() if torch.is_tensor(v) else v) for k, v in enc.items()}
        encb["B"] = nb
        T = int(max_cap_length)
        tr = self._alloc_trace(nb, T, enc["P"])
        toks = torch.zeros(nb, T + 1, dtype=torch.int64, device=self.device)

        def step(t, prev):
            toks[:, t] = prev
            self._step(tr, encb, t, toks, self.bcat_model)

        def reorder(t, src):
            sel = torch.tensor(src, dtype=torch.int64, device=self.device)
            for k in ("h", "c"):
                tr[k][:len(src), t + 1] = tr[k][sel, t + 1]

        return run_beam_search(step, lambda t: self.logits(tr["hc"][:, t].contiguous()), reorder, self.V, nb, T,
                               start_id, end_id, self.device)

    def forwardlrp_context(self, enc, captions, caption_lengths, skip_ids):
        """The forward half of `AOAModel.forwardlrp_context` (models/aoamodel.py:628-677): teacher-forced decoding with the
        model's own forward; every step's scores are recomputed from the fc input re-weighted by the relevance of the
        step's arg-max word (`get_lrp_weight_step`, :597-626 - handed the RAW scores here, unlike `sample_lrp`).  Dropout is
        the identity (evaluation mode).  Returns (predictions (B,L,V), weighted_predictions (B,L,V), L)."""
        lib = _lib.load()
        B, H = enc["B"], self.H
        L = int(max(caption_lengths)) - 1
        dev = self.device
        captions = captions.to(dev, torch.int64).contiguous()
        assert captions.shape[0] == B and captions.shape[1] >= L
        skip = torch.zeros(self.V, dtype=torch.uint8, device=dev)
        skip[torch.as_tensor(sorted(int(i) for i in skip_ids), dtype=torch.int64, device=dev)] = 1
        toks = captions[:, :L + 1].contiguous() if captions.shape[1] > L else torch.cat(
            [captions, captions.new_zeros(B, 1)], 1).contiguous()
        tr = self._alloc_trace(B, L, enc["P"])
        hcw = torch.empty(B, H, device=dev)
        preds = torch.empty(B, L, self.V, device=dev)
        wpreds = torch.empty(B, L, self.V, device=dev)
        for t in range(L):
            self._step(tr, enc, t, toks, self.bcat_model)
            pred = self.logits(tr["hc"][:, t].contiguous())
            check(lib.lrpx_lrp_reweight_rows(ptr(pred), self.V, self.V, ptr_at(tr["h"], (t + 1) * H), (L + 1) * H,
                                             ptr_at(tr["c_aoa"], t * H), L * H, ptr(self.sd["fc.weight"]), ptr(skip),
                                             ptr(hcw), B, H, 0, stream_ptr()))
            preds[:, t] = pred
            wpreds[:, t] = self.logits(hcw)
        return preds, wpreds, L

    def _plain_rows(self, x, name, bias, amax=None):
        """x (R, K) @ W^T + bias -> (R, N) for one of the trace's plain linears over all (image, word) rows: split products on the
        fp16 matrix cores (as `logits(fast=True)`: <= 2e-7 of a row's maximum; operand scale per ROW), the fp32 MFMA kernel where K is
        no multiple of 64.  The kernel never depends on the number of rows: an image's TRACE is the same in every batch (the recurrence in
        slices of <= 64 images; `tests/test_gpu_aoa.py::test_trace_is_the_same_in_every_batch`: B = 1 / 64 / 65 bit for bit).  The (T,V)
        `pred` block is the exception: `logits(fast=True)` takes the fp32 kernel below 128 rows (scores equal to rounding)."""
        p_h, p_f, n, k = self._plain[name]
        if not self._f16():
            p_h = None
        R = x.shape[0]
        out = torch.empty(R, n, device=self.device)
        if p_h is not None:
            ops.conv_mfma(x, p_h, R, 0, k, -(-n // 32) * 32, 1, EPI_PLAIN, pix_per_map=1, oc_split=n, bias=bias, out0=out, f16x3=1,
                          in_amax=amax if amax is not None else ops.amax_maps(x, R))      # (amax: recorded by the kernel that wrote x)
        else:
            ops.conv_mfma(x, p_f, R, 0, k, -(-n // 32) * 32, 1, EPI_PLAIN, pix_per_map=1, oc_split=n, bias=bias, out0=out)
        return out

    def _trace_decoupled(self, tr, enc, captions, model_bias):
        """get_hidden_parameters' loop (models/aoamodel.py:1019-1052) with the recurrence decoupled (include/lrpx.h,
        lrpx_aoa_fwd_recurrence): one GEMM for the input part of all gate pre-activations, T launches of K = H for the recurrence,
        then q / gate linear, attention, decoder_aoa_linear and the gated sum once over all B*T rows."""
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E = tr["B"], tr["T"], self.H, self.E
        R = B * T
        c = C.byref(tr["_c"])
        sd = self.sd
        check(lib.lrpx_aoa_fwd_inputs(c, ptr(enc["glob"]), ptr(sd["embedding.weight"]), ptr(captions), captions.shape[1], None, st))
        bias_il = self.bcat_model_il if model_bias else self.bcat_explainer_il
        gimg = torch.empty(B, 4 * H, device=self.device)                     # glob W_ig^T + bias: the image part of every step's z
        for b0 in range(0, B, 64):          # slices of <= 64 rows: the same (MFMA) kernel whatever the batch size - an image's trace never depends on its batch
            nb = min(64, B - b0)
            check(lib.lrpx_linear_small(ptr_at(enc["glob"], b0 * H), H, ptr(self.W_ig_il), ptr(bias_il), ptr_at(gimg, b0 * 4 * H), 4 * H, nb, H, 4 * H, 0, st))
        check(lib.lrpx_aoa_fwd_recurrence_tab(c, ptr(self.W_hh_il), ptr(self.tok_table), ptr(gimg), ptr(captions), captions.shape[1], st))
        hn = torch.empty(R, H, device=self.device)
        am = tr["_amax"].view(torch.int32)                                    # [3][R] row maxima of hn / ctx / hc (ctx: zeroed with the trace)
        check(lib.lrpx_aoa_fwd_gather_h(c, ptr(hn), ptr(am[0]), st))
        qg = self._plain_rows(hn, "qg", self.bqg, amax=am[0])
        check(lib.lrpx_aoa_fwd_attention_all(c, ptr(qg), 2 * H, ptr(enc["key"]), ptr(enc["value"]), ptr(am[1]), st))
        lin = self._plain_rows(tr["ctx"].view(R, H), "lin", sd["decoder_aoa_linear.bias"], amax=am[1])
        check(lib.lrpx_aoa_fwd_post_all(c, ptr(qg), 2 * H, ptr(lin), ptr(am[2]), st))

    def trace(self, enc, captions, model_bias=False, predictions=True, grad=False):
        """grad=True: the trace of the gradient explainers (:1309-1376): correct LSTM bias, output gate and aoa gate kept."""
        model_bias = model_bias or grad
        lib = _lib.load()
        st = stream_ptr()
        B, T = captions.shape[0], captions.shape[1] - 1
        H, E = self.H, self.E
        captions = captions.to(self.device, torch.int64).contiguous()
        tr = self._alloc_trace(B, T, enc["P"], grad)
        c = C.byref(tr["_c"])
        W = E + 2 * H
        bias = self.bcat_model if model_bias else self.bcat_explainer
        sd = self.sd
        if self.decoupled and T > 0:          # (any B: the recurrence runs in slices of <= 64 images inside the library)
            self._trace_decoupled(tr, enc, captions, model_bias)
            tr["captions"] = captions
            tr["logit"] = torch.empty(B * T, device=self.device)
            check(lib.lrpx_target_logit(ptr(tr["hc"]), ptr(sd["fc.weight"]), ptr(sd["fc.bias"]), ptr(captions), T + 1,
                                        ptr(tr["logit"]), B, T, H, st))
            if predictions:
                tr["pred"] = self.logits(tr["hc"].view(B * T, H), fast=True, amax=tr["_amax"].view(torch.int32)[2]).view(B, T, self.V)
            return tr
        # the T decoder steps (:1019-1052) in one native call (the host loop of `_step` in C: the bottom-up path is bound by the
        # launch rate of the interpreter otherwise)
        tr["_zz"], tr["_qg"], tr["_lin"] = (torch.empty(B, n * H, device=self.device) for n in (4, 2, 1))
        sa = AoaStepArgs()
        sa.glob, sa.emb, sa.tok, sa.tok_ld = ptr(enc["glob"]), ptr(sd["embedding.weight"]), ptr(captions), captions.shape[1]
        sa.w_cat, sa.b_cat, sa.w_qg, sa.b_qg = ptr(self.Wcat), ptr(bias), ptr(self.Wqg), ptr(self.bqg)
        if self.fused_steps:
            sa.w_cat_il, sa.b_cat_il = ptr(self.Wcat_il), ptr(self.bcat_model_il if model_bias else self.bcat_explainer_il)
        sa.w_lin, sa.b_lin = ptr(sd["decoder_aoa_linear.weight"]), ptr(sd["decoder_aoa_linear.bias"])
        sa.key, sa.value, sa.zz, sa.qg, sa.lin = ptr(enc["key"]), ptr(enc["value"]), ptr(tr["_zz"]), ptr(tr["_qg"]), ptr(tr["_lin"])
        check(lib.lrpx_aoa_fwd_steps(c, 0, T, C.byref(sa), st))
        tr["captions"] = captions
        tr["logit"] = torch.empty(B * T, device=self.device)
        check(lib.lrpx_target_logit(ptr(tr["hc"]), ptr(sd["fc.weight"]), ptr(sd["fc.bias"]), ptr(captions), T + 1,
                                    ptr(tr["logit"]), B, T, H, st))
        if predictions:
            tr["pred"] = self.logits(tr["hc"].view(B * T, H), fast=True).view(B, T, self.V)
        return tr

    def gradient(self, enc, tr, head_idx, lens=None):
        """`ExplainAOAGradient.explain_caption_wordt` (models/aoamodel.py:1435-1499; the guided and Grad-CAM classes
        inherit it unchanged) for every (image, word) row in lock-step.  `tr` must be a grad=True trace.
        Returns d_feat (B*T, P, C), r_words (B*T, T), row2img.  With `lens` (explainers/ragged.py) d_feat holds the valid
        (image, word) rows only, compact, with the matching row -> image table."""
        lib = _lib.load()
        st = stream_ptr()
        B, T, H, E, P, Cc = tr["B"], tr["T"], self.H, self.E, tr["P"], self.C
        rows = B * T
        rg = ragged(lens, B, T, self.device)
        lens = rg.lens if rg is not None else None
        e = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        gs = dict(d_h=e(rows, H), d_c=e(rows, H), dA=e(rows, H), dB=e(rows, H), gates=e(rows, 4 * H),
                  dx=e(rows, E + 2 * H), d_glob=e(rows, H), r_words=e(rows, T))
        c = AoaGradState()
        c.lens = ptr(lens)
        for k, v in gs.items():
            setattr(c, k, ptr(v))
        ctr, cgs = C.byref(tr["_c"]), C.byref(c)
        _, row2img, _ = self._row_index(B, T)
        check(lib.lrpx_aoa_grad_init(ctr, cgs, ptr(self.sd["fc.weight"]), ptr(tr["captions"]), T + 1, st))

        def dense(a, wp, n):      # (rows, K) @ packed (K, n) -> (rows, n), fp32 MFMA
            out = e(rows, n)
            ops.conv_mfma(a, wp, rows, 0, a.shape[1], -(-n // 32) * 32, 1, EPI_PLAIN, pix_per_map=1, oc_split=n, out0=out)
            return out
        d_ctx = dense(gs["dA"], self.p_lin_rel, H)                                   # :1469  d_A @ W_aoa_linear
        t1 = dense(gs["dB"], self.p_gate_grad, H)                                    # :1470  d_B @ W_aoa_linear_gate
        check(lib.lrpx_accumulate(ptr(gs["d_h"]), ptr(t1), rows * H, st))
        dk = H // self.NH
        check(lib.lrpx_keep_cols(ptr(d_ctx), rows, H, head_idx * dk, (head_idx + 1) * dk, st))   # gradient_mha :1428
        for s in range(T):
            check(lib.lrpx_aoa_grad_step(ctr, cgs, s, 0, st))
            ops.conv_mfma(gs["gates"], self.p_gates_grad, rows, 0, 4 * H, E + 2 * H, 1, EPI_PLAIN, pix_per_map=1,
                          oc_split=E + 2 * H, out0=gs["dx"])
            check(lib.lrpx_aoa_grad_step(ctr, cgs, s, 1, st))
        # pixels: d_value = alpha (x) d_ctx[head]; through v_proj and the projector both stay rank-1 in the pixel index
        u = dense(d_ctx, self.p_v_rel, H)                                            # :1489  d_value @ W_v
        v1 = dense(u, self.p_proj_rel, Cc)                                           # :1492
        v2 = dense(gs["d_glob"], self.p_proj_rel, Cc)
        check(lib.lrpx_scale(ptr(v2), ptr(v2), v2.numel(), 1.0 / P, st))             # :1491  d_glob / P
        check(lib.lrpx_rel_words_norm(ptr(gs["r_words"]), rows, T, st))
        n, rowlist = rows, None
        if rg is not None and not rg.full:
            n, rowlist, row2img = rg.n, rg.rows, rg.row2img
        d_feat = e(n, P, Cc)
        if n:
            check(lib.lrpx_aoa_grad_pix_rows(ctr, head_idx, ptr(v1), ptr(v2), ptr(d_feat), Cc, ptr(rowlist), n, st))
        return d_feat, gs["r_words"], row2img

    def explain_batch_gradient(self, captions, head_idx, images, kind="gradient", lens=None, return_features=False):
        """Batched `explain_caption` of ExplainAOAGradient (kind="gradient", :1501-1534), ExplainAOAGuidedGradient
        ("guided", :1621-1640), ExplainAOAGradCam ("gradcam", :1669-1689; result (B,T,P) heat maps) and
        ExplainAOAGuidedGradCam ("guided_gradcam", :1714-1751)."""
        enc = self.encode(images)
        captions = captions.to(self.device, torch.int64).contiguous()
        B, T = captions.shape[0], captions.shape[1] - 1
        tr = self.trace(enc, captions, predictions=False, grad=True)
        rg = ragged(lens, B, T, self.device)
        d_feat, r_words, row2img = self.gradient(enc, tr, head_idx, rg)
        n = d_feat.shape[0]                      # B*T, or the valid rows of captions of unequal length
        if n == 0:
            maps = d_feat.new_zeros((0, enc["P"]) if kind == "gradcam" else (0, 3, 224, 224))
        elif kind == "gradcam":
            maps = torch.empty(n, enc["P"], device=self.device, dtype=torch.float32)
            check(_lib.load().lrpx_gradcam(ptr(enc["feats"]), ptr(d_feat), ptr(row2img), ptr(maps), n, enc["P"], self.C,
                                           stream_ptr()))
        elif kind in ("guided", "guided_gradcam"):
            maps = self.vgg.guided_backprop(d_feat, row2img)
            if kind == "guided_gradcam":         # ExplainAOAGuidedGradCam (:1714-1751): x the expanded Grad-CAM map
                cam = torch.empty(n, enc["P"], device=self.device, dtype=torch.float32)
                check(_lib.load().lrpx_gradcam(ptr(enc["feats"]), ptr(d_feat), ptr(row2img), ptr(cam), n, enc["P"],
                                               self.C, stream_ptr()))
                maps = ops.guided_gradcam(maps, cam, int(round(enc["P"] ** 0.5)))
        else:
            maps = self.vgg.gradient(d_feat, row2img)
        if rg is not None and not rg.full:       # back to the padded layout, zeros behind an image's last word
            maps = ops.scatter_maps(maps, rg)
            if return_features:
                d_feat = ops.scatter_maps(d_feat, rg)
        maps = maps.view(B, T, enc["P"]) if kind == "gradcam" else maps.view(B, T, 3, 224, 224)
        out = (maps, r_words.view(B, T, T))
        if return_features:
            out = out + (d_feat.view(B, T, enc["P"], self.C), tr, enc)
        return out

    def logits(self, hc_rows, fast=False, amax=None):
        """fc scores for R rows -> (R,V).  fast=True (the (T,V) block a trace keeps, not the decisions of a decoding loop): split
        products on the fp16 matrix cores (csrc/dense_f16x3.hip, fp32-grade: <= 2e-7 of a row's maximum)"""
        R = hc_rows.shape[0]
        out = torch.empty(R, self.V, device=self.device)
        if fast and R >= 128 and self.p_fc_fwd_h is not None and self._f16():
            hc_rows = hc_rows.contiguous()
            ops.conv_mfma(hc_rows, self.p_fc_fwd_h, R, 0, self.H, -(-self.V // 32) * 32, 1, EPI_PLAIN, pix_per_map=1, oc_split=self.V,
                          bias=self.sd["fc.bias"], out0=out, f16x3=1, in_amax=amax if amax is not None else ops.amax_maps(hc_rows, R))
            return out
        ops.conv_mfma(hc_rows, self.p_fc_fwd, R, 0, self.H, -(-self.V // 32) * 32, 1, EPI_PLAIN, pix_per_map=1,
                      oc_split=self.V, bias=self.sd["fc.bias"], out0=out)
        return out

    def _row_index(self, B, T):
        key = (B, T)
        if key not in self._idx_cache:
            b = torch.arange(B, device=self.device).view(B, 1)
            t = torch.arange(T, device=self.device).view(1, T)
            s = torch.arange(T, device=self.device).view(T, 1, 1)
            idx = (b * T + (t - s).clamp(min=0)).to(torch.int32).reshape(T, B * T).contiguous()
            row2img = (b + 0 * t).to(torch.int32).reshape(B * T).contiguous()
            rowid = torch.arange(B * T, device=self.device, dtype=torch.int32)
            self._idx_cache[key] = (idx, row2img, rowid)
        return self._idx_cache[key]

    def relevance(self, enc, tr, head_idx, lens=None, compact=True):
        """explain_caption_wordt (:1064-1156) for every (image, word) row -> r_feat (B*T,P,C), r_words (B*T,T), row -> image.
        lens (one caption length per image; explainers/ragged.py): padded words are skipped by the lock-step kernels and, with
        compact=True (the image path: the rows feed the VGG16 chain), the (word, pixel) rules run on the valid rows only:
        r_feat is then (sum(lens), P, C) with the matching row -> image table."""
        lib = _lib.load()
        st = stream_ptr()
        B, T, P, H, E, Cc = tr["B"], tr["T"], tr["P"], self.H, self.E, self.C
        rows = B * T
        rg = ragged(lens, B, T, self.device)
        lens = rg.lens if rg is not None else None
        e = lambda *s: torch.empty(*s, device=self.device, dtype=torch.float32)
        rs = dict(r_hn=e(rows, H), r_glob=e(rows, H), A=e(rows, H), rx=e(rows, E + 2 * H), r_words=e(rows, T))
        c = AoaRelState()
        c.lens = ptr(lens)
        for k, v in rs.items():
            setattr(c, k, ptr(v))
        ctr, crs = C.byref(tr["_c"]), C.byref(c)
        idx, row2img, rowid = self._row_index(B, T)
        check(lib.lrpx_aoa_rel_init(ctr, crs, ptr(self.sd["fc.weight"]), ptr(tr["logit"]), ptr(tr["captions"]), T + 1, st))
        # decoder_aoa_linear dense rule (:1107-1110): r_ctx = ctx * (W^T (r_caoa / z~(lin)))
        r_ctx = e(rows, H)
        f16 = 1 if (self.lockstep_f16 and self._f16()) else 0
        ops.conv_mfma(rs["A"], self.p_lin_rel_h if f16 else self.p_lin_rel, rows, 0, H, H, 1, EPI_REL, pix_per_map=1, oc_split=H,
                      x=tr["ctx"], map2img=rowid, out0=r_ctx, f16x3=f16)
        W = E + 2 * H
        # the lock-steps s = 0..T-1 (:1114-1134) in one native call: phase 0, the LSTM dense rule with map2img = idx[s], phase 1
        dense = ops.conv_desc(rs["A"], self.p_wg_h if f16 else self.p_wg, rows, 0, H, W, 1, EPI_REL, pix_per_map=1, oc_split=W,
                              x=tr["xh"], map2img=idx[0], out0=rs["rx"], f16x3=f16)
        fused = self.fused_rel and E == H == 512 and (f16 or self.fused_rel_exact)     # (exact modes: the same fusion in the fp32 K-split kernel, csrc/dense_small.hip)
        if fused:      # one launch per lock-step: the step's point-wise code in the GEMM's epilogue (lrpx_aoa_rel_steps_fused)
            a_alt, wpart, coef = e(rows, H), e(rows, T, 4), e(2 * rows * H + rows)
            check(lib.lrpx_aoa_rel_steps_fused(ctr, crs, C.byref(dense), ptr(idx), idx.shape[1], ptr(a_alt), ptr(wpart), ptr(coef), st))
        else:
            check(lib.lrpx_aoa_rel_steps(ctr, crs, T, C.byref(dense), ptr(idx), idx.shape[1], st))
        # :1136-1144  r_proj = eye rule on the mean (U) + v_proj dense rule; fused division for the projector rule
        U = e(rows, H)
        check(lib.lrpx_rel_avg_u(ptr(rs["r_glob"]), ptr(enc["glob"]), ptr(U), rows, T, H, P, st))
        if not fused:
            check(lib.lrpx_rel_words_norm(ptr(rs["r_words"]), rows, T, st))
        n, rowlist = rows, None
        if rg is not None and not rg.full and compact:     # unequal lengths: the (word, pixel) rules on the valid rows only
            n, rowlist, row2img = rg.n, rg.rows, rg.row2img
            if n == 0:
                return e(0, P, Cc), rs["r_words"], row2img
            U = ops.gather_rows(U, rowlist)
        a_proj = e(n, P, H)
        r_feat = e(n, P, Cc)
        f16_rules = self.p_v_rel_h is not None and P >= 32 and self._f16()
        b6_rules = not f16_rules and self.p_v_rel_6 is not None and self.dense_bf16x6           # (any P: a row's kernel never depends on the batch)
        head_only = self.head_only and ((f16_rules and self.p_v_rel_head is not None) or (b6_rules and self.p_v_rel_head6 is not None))
        if head_only:
            dk = H // self.NH
            a_val = e(n, P, dk)
            check(lib.lrpx_aoa_rel_value_head(ctr, crs, ptr(r_ctx), ptr(enc["value"]), int(head_idx), ptr(a_val), ptr(rowlist), n, st))
        else:
            a_val = e(n, P, H)
            check(lib.lrpx_aoa_rel_value_rows(ctr, crs, ptr(r_ctx), ptr(enc["value"]), int(head_idx), ptr(a_val), ptr(rowlist), n, st))
        if f16_rules:
            amax2 = ops.zeros(n, dtype=torch.int32, device=self.device)           # max|a_proj| per row: recorded by the first GEMM
            ops.conv_mfma(a_val, self.p_v_rel_head[int(head_idx)] if head_only else self.p_v_rel_h, n, 0, dk if head_only else H, H, 1,
                          EPI_REL, pix_per_map=P, oc_split=H, x=enc["Vp"], u=U,
                          zdiv=enc["proj_pre"], stab=STAB_EPS, map2img=row2img, out1=a_proj, f16x3=1,
                          in_amax=ops.amax_maps(a_val, n), out1_amax=amax2)
            ops.conv_mfma(a_proj, self.p_proj_rel_h, n, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc,
                          x=enc["feats"], map2img=row2img, out0=r_feat, f16x3=1, in_amax=amax2)       # :1145-1148
        elif b6_rules:
            ops.conv_mfma(a_val, self.p_v_rel_head6[int(head_idx)] if head_only else self.p_v_rel_6, n, 0, dk if head_only else H, H, 1,
                          EPI_REL, pix_per_map=P, oc_split=H, x=enc["Vp"], u=U, zdiv=enc["proj_pre"], stab=STAB_EPS, map2img=row2img,
                          out1=a_proj, bf16x6=1)
            ops.conv_mfma(a_proj, self.p_proj_rel_6, n, 0, H, -(-Cc // 32) * 32, 1, EPI_REL, pix_per_map=P, oc_split=Cc,
                          x=enc["feats"], map2img=row2img, out0=r_feat, bf16x6=1)                     # :1145-1148
        else:
            ops.conv_mfma(a_val, self.p_v_rel, n, 0, H, H, 1, EPI_REL, pix_per_map=P, oc_split=H, x=enc["Vp"], u=U,
                          zdiv=enc["proj_pre"], stab=STAB_EPS, map2img=row2img, out1=a_proj)
            ops.conv_mfma(a_proj, self.p_proj_rel, n, 0, H, Cc, 1, EPI_REL, pix_per_map=P, oc_split=Cc, x=enc["feats"],
                          map2img=row2img, out0=r_feat)                                   # :1145-1148
        return r_feat, rs["r_words"], row2img

    def explain_batch(self, captions, head_idx, images=None, features=None, lens=None, accumulate=False,
                      return_features=False, predictions=False):
        """Batched `explain_caption(img, head_idx)` (:1165-1181).  With images: maps (B,T,3,224,224); with
        `features` (bottom-up): the region-feature relevance (B,T,P,C) is the result (no CNN stage).
        predictions=True keeps the (T,V) scores of the trace as the reference's explainer does (:1026)."""
        enc = self.encode(images, features)
        captions = captions.to(self.device, torch.int64).contiguous()
        B, T = captions.shape[0], captions.shape[1] - 1
        tr = self.trace(enc, captions, predictions=predictions)
        rg = ragged(lens, B, T, self.device)
        if features is not None:         # bottom-up: the region relevance IS the result; rows behind an image's last word are zero
            r_feat, r_words, _ = self.relevance(enc, tr, head_idx, rg, compact=False)
            return r_feat.view(B, T, enc["P"], self.C), r_words.view(B, T, T)
        r_feat, r_words, row2img = self.relevance(enc, tr, head_idx, rg)
        if rg is not None and not rg.full:
            # unequal caption lengths (models/aoamodel.py:1171-1176 explains `caption_length` words): the VGG16 chain runs on the
            # sum(lens) valid maps; back to the padded layout afterwards (running sums per image over ITS words)
            maps = self.vgg.relevance(r_feat, row2img) if rg.n else r_feat.new_zeros(0, 3, 224, 224)
            maps = ops.scatter_maps(maps, rg, accumulate=accumulate)
            if return_features:
                r_feat = ops.scatter_maps(r_feat, rg)
        else:
            maps = self.vgg.relevance(r_feat, row2img)
            if accumulate:
                maps = ops.cumsum_maps(maps, B, T)
        out = (maps.view(B, T, 3, 224, 224), r_words.view(B, T, T))
        if return_features:
            out = out + (r_feat.view(B, T, enc["P"], self.C), tr, enc)
        return out


    def explain_batch_graph(self, captions, head_idx, images=None, features=None, accumulate=False, predictions=False):
        """`explain_batch` replayed from a captured HIP graph (one per input shape / head), as GridTDEngine.explain_batch_graph:
        one hipGraphLaunch instead of the ~330 launches of a bottom-up step.  Worth it for a SINGLE batch in flight (the host
        issues ~4 us per launch); with batches in flight on several streams eager launches are faster (config 5, B = 32:
        442 000 maps/s eager against 343 000 - 391 000 replayed, gpurun_out/r3g) - a replayed graph does not overlap with its
        neighbours the way independent kernels do.  Inputs are copied into the graph's static buffers; the returned tensors
        are the graph's static outputs (overwritten by the next call with the same shapes)."""
        src = features if features is not None else images
        src = src.to(self.device, torch.float32)
        captions = captions.to(self.device, torch.int64)
        key = (features is not None, tuple(src.shape), tuple(captions.shape), int(head_idx), bool(accumulate), bool(predictions), self._f16())
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        g = self._graphs.get(key)
        if g is None:
            st_src, st_cap = src.clone(), captions.clone()
            kw = dict(features=st_src) if features is not None else dict(images=st_src)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up outside capture (kernel attributes, index caches)
                self.explain_batch(st_cap, head_idx, accumulate=accumulate, predictions=predictions, **kw)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.explain_batch(st_cap, head_idx, accumulate=accumulate, predictions=predictions, **kw)
            g = self._graphs[key] = (graph, st_src, st_cap, out)
        graph, st_src, st_cap, out = g
        st_src.copy_(src)
        st_cap.copy_(captions)
        graph.replay()
        return out

    def explain_batch_replay(self, captions, head_idx, images=None, features=None, accumulate=False, predictions=False):
        """`explain_batch` as a RECORDED step (lrp_amd._lib.Recording): the first call with a given input shape runs the step eagerly
        on static copies of the inputs and keeps its library calls - functions, arguments, and every buffer they point at; later calls
        copy the inputs into those static buffers and issue the same calls again: the same kernels in the same order on the same
        stream, as ordinary launches (they overlap with other streams' kernels like any launch; a HIP graph replay did not), without the
        interpreter's ~9 us per launch.  Bit-identical to `explain_batch` by construction.  Like a graph's, the returned tensors are
        the recording's static outputs: overwritten by the next call of the same shape on this engine (take `replica()`s for batches
        in flight).  Captions of equal length only (`lens` makes the launch sequence data-dependent)."""
        src = features if features is not None else images
        src = src.to(self.device, torch.float32)
        captions = captions.to(self.device, torch.int64)
        key = (tuple(src.shape), tuple(captions.shape), bool(accumulate), bool(predictions), _lib.stream_ptr().value,
               self.vgg.conv_mode if self.vgg is not None else None, int(head_idx), features is not None, self._f16())
        if not hasattr(self, "_recordings"):
            self._recordings = {}
        rec = self._recordings.get(key)
        if rec is None:
            st_src, st_cap = src.clone(), captions.clone()
            # warm-up outside the recording: one-time work (kernel attributes, index caches, workspace allocations) must not be replayed
            self.explain_batch(st_cap, head_idx, accumulate=accumulate, predictions=predictions, **({"features": st_src} if features is not None else {"images": st_src}))
            rec = _lib.Recording()
            with rec:
                rec.result = self.explain_batch(st_cap, head_idx, accumulate=accumulate, predictions=predictions, **({"features": st_src} if features is not None else {"images": st_src}))
            rec.inputs = (st_src, st_cap)
            self._recordings[key] = rec
            return rec.result
        st_src, st_cap = rec.inputs
        st_src.copy_(src)
        st_cap.copy_(captions)
        return rec.replay()

    def _f16(self):
        """the decoder GEMMs on the fp16 split products?  (ops.decoder_f16: with conv modes 2 / 3 only - the engine's own `vgg.conv_mode` or the
        process default; `force_f16` overrides per engine)"""
        if self.force_f16 is not None:
            return bool(self.force_f16)
        return ops.decoder_f16(self.vgg.conv_mode if self.vgg is not None else None)

    def replica(self):
        """A second execution context over the SAME weights: own VGG16 trace / workspace buffers, so that several
        batches can be in flight on separate HIP streams (as GridTDEngine.replica)."""
        import copy
        r = copy.copy(self)
        if self.vgg is not None:
            r.vgg = self.vgg.replica()
        r._idx_cache = {}
        for k in ("_graphs", "_replicas", "_streams", "_recordings"):
            r.__dict__.pop(k, None)
        return r

    def explain_stream(self, batches, head_idx, depth=3, accumulate=False):
        """`explain_batch` over an iterable of independent (images, captions) batches with `depth` batches in flight,
        each on its own HIP stream and buffer set (the decoder's latency-bound kernels of one batch overlap the CNN
        relevance chain of another).  Yields (maps, r_words) in input order, complete when yielded; bit-identical to
        serial `explain_batch` calls."""
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
            st.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(st):
                out = self._replicas[k].explain_batch(captions, head_idx, images=images, lens=lens, accumulate=accumulate)
                ev = torch.cuda.Event()
                ev.record(st)
            for t in out:
                t.record_stream(torch.cuda.current_stream(self.device))
            pending.append((out, ev, images, captions))      # inputs stay alive until the side stream has read them
            if len(pending) >= depth:
                o, e, _, _ = pending.pop(0)
                e.synchronize()
                yield o
        for o, e, _, _ in pending:
            e.synchronize()
            yield o


class ExplainAOAAttention(object):
    """Drop-in for the reference's `ExplainAOAAttention` (models/aoamodel.py:748-1194), called as evaluation.py:637,702,767 call
    it: `explain_caption(img_filepath, head_idx)`, `explain_caption_wordt(t, head_idx)`, `explain_cnn(R)`,
    `explain_caption_words(img_filepath)`, `teacherforce_forward(img, beam_caption_encode)`, `get_hidden_parameters(img_filepath)`,
    `preprocess_img(img_filepath)`; attributes `.model .word_map .img .img_filepath .beam_caption .beam_caption_encode .predictions
    .alphas .args`.  Wherever the reference takes a file path a (1,3,H,W) tensor is accepted too.  See
    explainers/gridtd.py:ExplainGridTDAttention for the conventions (without `caption_encode=` the image is captioned by the
    reference's own procedure, beam search with beam 3 over 20 steps, :992; nothing written to disk)."""
    EPS = 0.01
    EX_TYPE = 'lrp'
    TF_MODEL_BIAS = False      # the LRP explainer's LanguageLSTM forward adds bias_ih twice (:873); the gradient family's is correct (:1298)

    def __init__(self, args, word_map, model=None):
        self.args = args
        self.word_map = word_map
        self.vocab_size = len(word_map)
        self.num_head = getattr(args, "num_head", 8)
        from . import engine_cache
        key = engine_cache.fingerprint("aoa", args.weight if model is None else model, (self.num_head,))

        def build():
            if model is None:
                state = torch.load(args.weight, map_location="cpu")['state_dict']
            elif hasattr(model, "state_dict"):
                state = model.state_dict()
            else:
                state = model
            return AOAEngine(state, self.num_head)
        self.model = model
        # one device engine per weight set (explainers/engine_cache.py); weights shared, trace / workspace buffers this explainer's own
        self.engine = engine_cache.get(key, build, hold=engine_cache.source_tensors(model)).replica()
        self.rev_word_map = {v: k for k, v in word_map.items()}
        self.mean = list(IMAGENET_MEAN)
        self.std = list(IMAGENET_STD)

    def preprocess_img(self, img_filepath):
        """Resize -> ToTensor -> Normalize (models/aoamodel.py:864-868), host side."""
        return load_image(img_filepath, getattr(self.args, "height", 224), getattr(self.args, "width", 224), self.mean, self.std,
                          self.engine.device)

    def get_hidden_parameters(self, img, caption_encode=None):
        """Forward trace (:990-1062).  `img`: file path (the reference's argument) or a (1,3,224,224) tensor."""
        eng = self.engine
        if isinstance(img, str):
            self.img_filepath = img
            self.img = self.preprocess_img(img)
        else:
            self.img = img.to(eng.device, torch.float32)
        # (a caption that is handed over is uploaded before the encoder is enqueued: the copy of a pageable list waits for the stream)
        cap_dev = None if caption_encode is None else torch.tensor([[int(c) for c in caption_encode]], dtype=torch.int64, device=eng.device)
        self._enc = eng.encode(self.img)
        if caption_encode is None:       # the reference captions the image itself: beam 3, 20 steps (:992-995)
            from .beam import caption_from_sequence
            seq = eng.beam_search(self._enc, 3, 20, self.word_map['<start>'], self.word_map['<end>'])
            caption_encode = caption_from_sequence(seq, self.word_map)
        self.beam_caption_encode = [int(c) for c in caption_encode]
        self.caption_length = len(self.beam_caption_encode) - 1
        special = {self.word_map[k] for k in ('<start>', '<end>', '<unk>', '<pad>') if k in self.word_map}
        self.beam_caption = [' '.join(self.rev_word_map.get(c, str(c)) for c in self.beam_caption_encode[1:]
                                      if c not in special)]
        self._rel = {}
        if self.caption_length == 0:
            return
        cap = cap_dev if cap_dev is not None else torch.tensor([self.beam_caption_encode], dtype=torch.int64, device=eng.device)
        self._cap_dev = cap
        self._tr = eng.trace(self._enc, cap, predictions=True)
        self.image_features = ops.nhwc_to_nchw(self._enc["feats"].contiguous(), eng.C, 14, 14)
        self.num_pixels = self._enc["P"]
        self.predictions = self._tr["pred"][0]
        self.alphas = self._tr["alpha"][0]
        self._rel = {}

    def _relevance(self, head_idx):
        if head_idx not in self._rel:
            self._rel[head_idx] = self.engine.relevance(self._enc, self._tr, head_idx)
        return self._rel[head_idx]

    def explain_caption_wordt(self, t, head_idx):
        assert t < self.caption_length
        r_feat, r_words, _ = self._relevance(head_idx)
        return ops.nhwc_to_nchw(r_feat[t:t + 1].contiguous(), self.engine.C, 14, 14), r_words[t, :t + 1].clone()

    def explain_cnn(self, r_img_feature):
        t_nhwc = ops.nchw_to_nhwc(r_img_feature.to(torch.float32))
        r = self.engine.vgg.relevance(t_nhwc, torch.zeros(r_img_feature.shape[0], dtype=torch.int32,
                                                          device=self.engine.device))
        if getattr(self, "_img_grad", None) is None:
            self._img_grad = r
        else:
            check(_lib.load().lrpx_accumulate(ptr(self._img_grad), ptr(r), r.numel(), stream_ptr()))
        ops.check_relevance(self._img_grad, finite=True, nonzero=True)
        return self._img_grad.clone()

    def teacherforce_forward(self, img, beam_caption_encode):
        """(:952-988; gradient family :1377-1413) -> predictions (len(beam_caption_encode), V) under teacher forcing: step t reads
        token t (evaluation.py:702,767 hand the caption WITH <start>), with this explainer's own LanguageLSTM forward."""
        eng = self.engine
        if isinstance(img, str):
            img = self.preprocess_img(img)
        enc = eng.encode(img.to(eng.device, torch.float32))
        cap = torch.tensor([[int(c) for c in beam_caption_encode] + [0]], dtype=torch.int64, device=eng.device)
        n = cap.shape[1] - 1
        tr = eng.trace(enc, cap, model_bias=self.TF_MODEL_BIAS, predictions=False)
        return eng.logits(tr["hc"].view(n, eng.H))       # the fp32 kernel of the decoding loops, at any caption length

    def explain_caption(self, img_filepath, head_idx, t_list=None, caption_encode=None):
        """(:1165-1181); the returned maps are the reference's running sums (lrp_wrapper.py:64-82)."""
        img = img_filepath
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img, caption_encode)
        if self.caption_length == 0:
            return [], []
        self._img_grad = None
        r_feat, r_words, row2img = self._relevance(head_idx)
        maps = ops.cumsum_maps(self.engine.vgg.relevance(r_feat, row2img), 1, self.caption_length)
        ops.check_relevance(maps, finite=True, nonzero=True)
        return ([maps[t:t + 1] for t in range(self.caption_length)],
                [r_words[t, :t + 1] for t in range(self.caption_length)])

    def explain_caption_words(self, img_filepath, caption_encode=None):
        """(:1183-1194) linguistic relevance only, head 0."""
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:
            return []
        _, r_words, _ = self._relevance(0)
        return [r_words[t, :t + 1] for t in range(self.caption_length)]


class ExplainAOAGradient(ExplainAOAAttention):
    """Drop-in for the reference's `ExplainAOAGradient` (models/aoamodel.py:1257-1592): hand-written BPTT through the
    language LSTM for one attention head (`explain_caption_wordt`, :1435-1499) and the autograd gradient through the
    encoder (`explain_cnn`, :1501-1515).  Same surface: `explain_caption(img, head_idx) -> (maps, word scores)`."""
    EX_TYPE = 'gradient'
    TF_MODEL_BIAS = True

    def get_hidden_parameters(self, img, caption_encode=None):
        super().get_hidden_parameters(img, caption_encode)
        if self.caption_length == 0:
            return
        self._tr = self.engine.trace(self._enc, self._cap_dev, predictions=True, grad=True)      # :1309-1376 (correct LSTM bias)
        self.predictions = self._tr["pred"][0]
        self.alphas = self._tr["alpha"][0]

    def _relevance(self, head_idx):
        if head_idx not in self._rel:
            self._rel[head_idx] = self.engine.gradient(self._enc, self._tr, head_idx)
        return self._rel[head_idx]

    def _cnn(self, d_feat_nhwc, row2img):
        return self.engine.vgg.gradient(d_feat_nhwc, row2img)

    def explain_cnn(self, d_img_feature):
        t_nhwc = ops.nchw_to_nhwc(d_img_feature.to(torch.float32))
        return self._cnn(t_nhwc, torch.zeros(d_img_feature.shape[0], dtype=torch.int32, device=self.engine.device))

    def explain_caption(self, img_filepath, head_idx, t_list=None, caption_encode=None):
        """(:1517-1534) no running sums here: the image gradient is a fresh tensor per word."""
        self.img_filepath = img_filepath
        self.get_hidden_parameters(img_filepath, caption_encode)
        if self.caption_length == 0:
            return [], []
        d_feat, r_words, row2img = self._relevance(head_idx)
        maps = self._cnn(d_feat, row2img)
        return ([maps[t:t + 1] for t in range(self.caption_length)],
                [r_words[t, :t + 1] for t in range(self.caption_length)])


class ExplainAOAGuidedGradient(ExplainAOAGradient):
    """Drop-in for `ExplainAOAGuidedGradient` (models/aoamodel.py:1594-1667): same decoder gradient, guided backprop
    through the encoder (:1621-1640)."""
    EX_TYPE = 'GuidedBackpropagate'

    def _cnn(self, d_feat_nhwc, row2img):
        return self.engine.vgg.guided_backprop(d_feat_nhwc, row2img)


class ExplainAOAGuidedGradCam(ExplainAOAGuidedGradient):
    """Drop-in for `ExplainAOAGuidedGradCam` (models/aoamodel.py:1714-1751): the guided-backprop map of every word times
    the Grad-CAM heat map of the same decoder gradient, expanded 16x by `skimage.transform.pyramid_expand` (:1741; here
    one matrix product per axis, ops.pyramid_expand_matrix)."""
    EX_TYPE = 'GuidedGradCam'

    def _cnn(self, d_feat_nhwc, row2img):
        rows, P = d_feat_nhwc.shape[0], d_feat_nhwc.shape[1]
        guided = self.engine.vgg.guided_backprop(d_feat_nhwc, row2img)
        cam = torch.empty(rows, P, device=self.engine.device, dtype=torch.float32)
        check(_lib.load().lrpx_gradcam(ptr(self._enc["feats"]), ptr(d_feat_nhwc.contiguous()), ptr(row2img), ptr(cam), rows,
                                       P, self.engine.C, stream_ptr()))
        return ops.guided_gradcam(guided, cam, int(round(P ** 0.5)))


class ExplainAOAGradCam(ExplainAOAGradient):
    """Drop-in for `ExplainAOAGradCam` (models/aoamodel.py:1669-1711): per word the (1, h*w) Grad-CAM heat map."""
    EX_TYPE = 'GradCam'

    def _cnn(self, d_feat_nhwc, row2img):
        rows, P = d_feat_nhwc.shape[0], d_feat_nhwc.shape[1]
        cam = torch.empty(rows, P, device=self.engine.device, dtype=torch.float32)
        check(_lib.load().lrpx_gradcam(ptr(self._enc["feats"]), ptr(d_feat_nhwc.contiguous()), ptr(row2img), ptr(cam), rows,
                                       P, self.engine.C, stream_ptr()))
        return cam

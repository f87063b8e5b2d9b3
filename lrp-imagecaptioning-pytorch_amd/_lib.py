"""ctypes binding of csrc/liblrpx.so (the C ABI declared in include/lrpx.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Error codes map to the exceptions the reference raises at the same places
(AssertionError for NaN/Inf/zero relevance, ValueError for unsupported layers/shapes)."""
import ctypes as C
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LRPX_LIB_PATH: another build of the same ABI (A/B of kernel variants on one box, tools/ab_chain.sh); default = in-tree
LIB_PATH = os.environ.get("LRPX_LIB_PATH") or os.path.join(_HERE, "csrc", "liblrpx.so")

OK, EINVAL, EARCH, ELAUNCH, ENONFINITE, EZERO = range(6)
PACK_FWD_DUAL, PACK_BWD_POS, PACK_BWD_FIRST, PACK_BWD_PLAIN, PACK_DENSE_T, PACK_DENSE, PACK_FWD, PACK_FWD_DUAL_FIRST = range(8)
EPI_FWD_DUAL, EPI_REL, EPI_FIRST, EPI_PLAIN, EPI_GUIDED, EPI_REL_MUL = range(6)
STAB_NONE, STAB_SAFE, STAB_EPS = range(3)

_f = C.c_void_p      # device pointers travel as void*
_i = C.c_int
_l = C.c_long
_sz = C.c_size_t


class GridTrace(C.Structure):
    _fields_ = [("B", _i), ("T", _i), ("H", _i), ("E", _i), ("P", _i)] + [(k, _f) for k in (
        "xh1", "xh2", "h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "hc",
        "alpha", "beta", "o1", "o2", "sgate")]


class GridGradState(C.Structure):
    _fields_ = [(k, _f) for k in ("lens", "d_h2n", "d_c2", "d_c1", "d_ch0", "d_h2p", "d_glob", "gates", "dx", "wacc",
                                  "r_words")]


class GridRelState(C.Structure):
    _fields_ = [(k, _f) for k in ("lens", "r_h2n", "r_c2", "r_c1", "r_ch0", "r_h2p", "r_glob", "A", "rx", "wacc",
                                  "r_words")]


class AoaTrace(C.Structure):
    _fields_ = [("B", _i), ("T", _i), ("H", _i), ("E", _i), ("P", _i), ("NH", _i)] + [(k, _f) for k in (
        "xh", "h", "c", "g", "i", "f", "ctx", "lin", "c_aoa", "hc", "alpha", "o", "sg")]


class AoaStepArgs(C.Structure):
    """lrpx_aoa_step_args"""
    _fields_ = [("glob", _f), ("emb", _f), ("tok", _f), ("tok_ld", _i)] + [(k, _f) for k in (
        "w_cat", "b_cat", "w_cat_il", "b_cat_il", "w_qg", "b_qg", "w_lin", "b_lin", "key", "value", "zz", "qg", "lin")]


class GridStepArgs(C.Structure):
    """lrpx_gridtd_step_args"""
    _fields_ = [("glob", _f), ("emb", _f), ("tok", _f), ("tok_ld", _i)] + [(k, _f) for k in (
        "w_cat1", "b_cat1", "w_cat2", "b_cat2", "Vp", "att_img", "Wg", "Ws", "bs", "wh", "zz1", "zz2", "att_scratch",
        "w_il1", "b_il1", "w_il2", "b_il2")]


class AoaGradState(C.Structure):
    _fields_ = [(k, _f) for k in ("lens", "d_h", "d_c", "dA", "dB", "gates", "dx", "d_glob", "r_words")]


class AoaRelState(C.Structure):
    _fields_ = [(k, _f) for k in ("lens", "r_hn", "r_glob", "A", "rx", "r_words")]


class ConvDesc(C.Structure):
    _fields_ = [("in_", _f), ("wpacked", _f),
                ("n_maps", _i), ("hw", _i), ("cin", _i), ("n_oc", _i), ("taps", _i), ("pix_per_map", _i),
                ("epi", _i), ("stab", _i), ("oc_split", _i), ("relu", _i), ("in_chunked", _i), ("bf16x6", _i),
                ("bias", _f), ("x", _f), ("u", _f), ("zdiv", _f), ("map2img", _f),
                ("out0", _f), ("out1", _f),
                ("f16x3", _i), ("out_chunk", _i), ("in_amax", _f), ("out1_amax", _f), ("out0_amax", _f), ("pool_am", _f),
                ("tile_group", _i), ("blocked", _i)]


class VggOpts(C.Structure):
    """lrpx_vgg16_opts: the per-call context of the VGG16 chains (conv mode, forward switch, per-layer timing)"""
    _fields_ = [("conv_mode", _i), ("forward_f16", _i), ("layer_ms", C.POINTER(C.c_float))]


# name -> (restype, argtypes); must list every symbol of include/lrpx.h (tests/test_abi.py checks it)
SIGNATURES = {
    "lrpx_version": (_i, []),
    "lrpx_last_error_string": (C.c_char_p, []),
    "lrpx_build_flags": (C.c_char_p, []),
    "lrpx_blocked_floats": (_sz, [_l, _i]),
    "lrpx_nhwc_to_blocked": (_i, [_f, _f, _l, _i, _i, _f]),
    "lrpx_blocked_to_nhwc": (_i, [_f, _f, _l, _i, _i, _f]),
    "lrpx_packed_floats": (_sz, [_i, _i, _i, _i]),
    "lrpx_pack_weights": (_i, [_f, _i, _i, _i, _i, _i, _f, _f]),
    "lrpx_packed_bf16x3_bytes": (_sz, [_i, _i, _i]),
    "lrpx_pack_weights_bf16x3": (_i, [_f, _i, _i, _i, _i, _f, _f]),
    "lrpx_packed_f16x2_bytes": (_sz, [_i, _i, _i]),
    "lrpx_pack_weights_f16x2": (_i, [_f, _i, _i, _i, _i, _f, _f]),
    "lrpx_packed_f16f8_bytes": (_sz, [_i, _i]),
    "lrpx_pack_weights_f16f8": (_i, [_f, _i, _i, _i, _f, _f]),
    "lrpx_conv_kc": (_i, [_i, _i, _i]),
    "lrpx_conv_mfma": (_i, [C.POINTER(ConvDesc), _f]),
    "lrpx_nchw_to_nhwc": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "lrpx_nhwc_to_nchw": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "lrpx_nchw_to_nhwc_posneg": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "lrpx_maxpool2x2_fwd": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "lrpx_maxpool2x2_relevance": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "lrpx_divide_stab": (_i, [_f, _f, _f, _f, _i, _l, _i, _f]),
    "lrpx_pool_winner": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "lrpx_unpool_winner": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "lrpx_aoa_grad_init": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaGradState), _f, _f, _i, _f]),
    "lrpx_aoa_grad_step": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaGradState), _i, _i, _f]),
    "lrpx_aoa_grad_pix": (_i, [C.POINTER(AoaTrace), _i, _f, _f, _f, _i, _f]),
    "lrpx_keep_cols": (_i, [_f, _l, _i, _i, _i, _f]),
    "lrpx_spatial_reduce": (_i, [_f, _i, _i, _l, _i, _f, _f]),
    "lrpx_project_maxabs": (_i, [_f, _i, _l, _f]),
    "lrpx_patch_mask": (_i, [_f, _i, _i, _i, _i, _i, _f, _f]),
    "lrpx_bbox_ratio": (_i, [_f, _i, _i, _i, _f, _f, _i, _f, _f]),
    "lrpx_map_stats": (_i, [_f, _i, _l, _f, _f]),
    "lrpx_map_quantiles_workspace": (C.c_size_t, [_i, _l]),
    "lrpx_map_quantiles": (_i, [_f, _i, _l, _f, _i, _f, _f, C.c_size_t, _f]),
    "lrpx_heatmap": (_i, [_f, _i, _i, _l, C.c_float, _f, _i, _f, _f, _f]),
    "lrpx_amax_maps": (_i, [_f, _i, _l, _f, _f]),
    "lrpx_cumsum_maps": (_i, [_f, _f, _i, _i, _l, _f]),
    "lrpx_zero": (_i, [_f, _sz, _f]),
    "lrpx_scatter_maps": (_i, [_f, _f, _i, _i, _f, _f, _l, _i, _f]),
    "lrpx_gather_rows": (_i, [_f, _f, _f, _i, _i, _f]),
    "lrpx_gridtd_rel_pix_rows": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _f, _f, _f, _f, _i, _f]),
    "lrpx_aoa_rel_value_rows": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _f, _f, _i, _f, _f, _i, _f]),
    "lrpx_aoa_rel_value_head": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _f, _f, _i, _f, _f, _i, _f]),
    "lrpx_aoa_grad_pix_rows": (_i, [C.POINTER(AoaTrace), _i, _f, _f, _f, _i, _f, _i, _f]),
    "lrpx_spread_pixels_rows": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f, _i, _f]),
    "lrpx_accumulate": (_i, [_f, _f, _l, _f]),
    "lrpx_fold_halves": (_i, [_f, _f, _l, _i, _f]),
    "lrpx_check": (_i, [_f, _l, _i, _f]),
    "lrpx_linear_small": (_i, [_f, _l, _f, _f, _f, _l, _i, _i, _i, _i, _f]),
    "lrpx_mean_pixels": (_i, [_f, _f, _i, _i, _i, _f]),
    "lrpx_relu": (_i, [_f, _f, _l, _f]),
    "lrpx_argmax_rows": (_i, [_f, _l, _i, _i, _f, _f]),
    "lrpx_target_logit": (_i, [_f, _f, _f, _f, _i, _f, _i, _i, _i, _f]),
    "lrpx_gridtd_fwd_pre": (_i, [C.POINTER(GridTrace), _i, _f, _f, _f, _i, _f]),
    "lrpx_gridtd_fwd_lstm": (_i, [C.POINTER(GridTrace), _i, _f, _i, _i, _f]),
    "lrpx_gridtd_fwd_gate_input": (_i, [C.POINTER(GridTrace), _i, _f, _f]),
    "lrpx_gridtd_fwd_sentinel": (_i, [C.POINTER(GridTrace), _i, _f, _i, _f]),
    "lrpx_gridtd_lrp_reweight": (_i, [C.POINTER(GridTrace), _i, _f, _l, _i, _f, _f, _f, _f]),
    "lrpx_lrp_reweight_rows": (_i, [_f, _l, _i, _f, _l, _f, _l, _f, _f, _f, _i, _i, _i, _f]),
    "lrpx_argmax_logprob_rows": (_i, [_f, _l, _i, _i, _f, _f, _f]),
    "lrpx_gridtd_fwd_attention": (_i, [C.POINTER(GridTrace), _i, _f, _f, _f, _f, _f, _f, _f, _f]),
    "lrpx_gridtd_fwd_steps": (_i, [C.POINTER(GridTrace), _i, _i, C.POINTER(GridStepArgs), _f]),
    "lrpx_gridtd_rel_steps": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _i, C.POINTER(ConvDesc), C.POINTER(ConvDesc), _f, _i, _f]),
    "lrpx_gridtd_rel_init": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _f, _f, _f, _i, _f]),
    "lrpx_gridtd_rel_step": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _i, _i, _f]),
    "lrpx_gridtd_rel_glob": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _f, _f, _f]),
    "lrpx_rel_avg_u": (_i, [_f, _f, _f, _i, _i, _i, _i, _f]),
    "lrpx_gridtd_rel_pix": (_i, [C.POINTER(GridTrace), C.POINTER(GridRelState), _f, _f, _f, _f]),
    "lrpx_rel_words_norm": (_i, [_f, _i, _i, _f]),
    "lrpx_aoa_fwd_pre": (_i, [C.POINTER(AoaTrace), _i, _f, _f, _f, _i, _f]),
    "lrpx_aoa_fwd_lstm": (_i, [C.POINTER(AoaTrace), _i, _f, _i, _f]),
    "lrpx_aoa_fwd_attention": (_i, [C.POINTER(AoaTrace), _i, _f, _i, _f, _f, _f]),
    "lrpx_aoa_fwd_post": (_i, [C.POINTER(AoaTrace), _i, _f, _i, _f, _f]),
    "lrpx_aoa_fwd_steps": (_i, [C.POINTER(AoaTrace), _i, _i, C.POINTER(AoaStepArgs), _f]),
    "lrpx_aoa_fwd_inputs": (_i, [C.POINTER(AoaTrace), _f, _f, _f, _i, _f, _f]),
    "lrpx_aoa_fwd_recurrence": (_i, [C.POINTER(AoaTrace), _f, _f, _f]),
    "lrpx_aoa_fwd_recurrence_tab": (_i, [C.POINTER(AoaTrace), _f, _f, _f, _f, _i, _f]),
    "lrpx_aoa_fwd_gather_h": (_i, [C.POINTER(AoaTrace), _f, _f, _f]),
    "lrpx_aoa_fwd_attention_all": (_i, [C.POINTER(AoaTrace), _f, _i, _f, _f, _f, _f]),
    "lrpx_aoa_fwd_post_all": (_i, [C.POINTER(AoaTrace), _f, _i, _f, _f, _f]),
    "lrpx_aoa_rel_init": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _f, _f, _f, _i, _f]),
    "lrpx_aoa_rel_value": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _f, _f, _i, _f, _f]),
    "lrpx_aoa_rel_step": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _i, _i, _f]),
    "lrpx_aoa_rel_steps": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), _i, C.POINTER(ConvDesc), _f, _i, _f]),
    "lrpx_aoa_rel_steps_fused": (_i, [C.POINTER(AoaTrace), C.POINTER(AoaRelState), C.POINTER(ConvDesc), _f, _i, _f, _f, _f, _f]),
    "lrpx_gridtd_grad_init": (_i, [C.POINTER(GridTrace), C.POINTER(GridGradState), _f, _f, _i, _f]),
    "lrpx_gridtd_grad_step": (_i, [C.POINTER(GridTrace), C.POINTER(GridGradState), _i, _i, _f]),
    "lrpx_spread_pixels": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "lrpx_scale": (_i, [_f, _f, _l, C.c_float, _f]),
    "lrpx_positive_mask": (_i, [_f, _f, _l, _f]),
    "lrpx_vgg16_guided_backprop": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, _f]),
    "lrpx_vgg16_gradient": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, _f]),
    "lrpx_gradcam": (_i, [_f, _f, _f, _f, _i, _i, _i, _f]),
    "lrpx_set_bf16x6": (_i, [_i]),
    "lrpx_set_conv_mode": (_i, [_i]),
    "lrpx_set_forward_f16": (_i, [_i]),
    "lrpx_vgg16_packed_bytes": (_sz, []),
    "lrpx_vgg16_trace_bytes": (_sz, [_i]),
    "lrpx_vgg16_workspace_bytes": (_sz, [_i]),
    "lrpx_vgg16_pack": (_i, [C.POINTER(_f), C.POINTER(_f), _f, _f]),
    "lrpx_vgg16_forward": (_i, [_f, _f, _i, _f, _f, _f]),
    "lrpx_vgg16_relevance": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, _f]),
    "lrpx_vgg16_layer_timing": (_i, [_i, _f]),
    "lrpx_vgg16_trace_derive": (_i, [_f, _i, _f]),
    "lrpx_vgg16_trace_layout": (_i, [_i, C.POINTER(_sz), C.POINTER(_sz)]),
    "lrpx_vgg16_trace_features": (_f, [_f, _i]),
    "lrpx_vgg16_channel_scales": (_f, [_f, _i, C.POINTER(_i)]),
    "lrpx_vgg16_row_spread": (_f, [_f]),
    "lrpx_vgg16_resolve_opts": (_i, [C.POINTER(VggOpts), C.POINTER(_i), C.POINTER(_i)]),
    "lrpx_vgg16_forward_ex": (_i, [_f, _f, _i, _f, _f, C.POINTER(VggOpts), _f]),
    "lrpx_vgg16_relevance_ex": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, C.POINTER(VggOpts), _f]),
    "lrpx_vgg16_guided_backprop_ex": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, C.POINTER(VggOpts), _f]),
    "lrpx_vgg16_gradient_ex": (_i, [_f, _f, _i, _f, _f, _i, _f, _f, C.POINTER(VggOpts), _f]),
    "lrpx_beam_topk": (_i, [_f, _l, _i, _i, _f, _i, _f, _f, _f]),
    "lrpx_guided_gradcam": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f]),
    "lrpx_linear_eps_rule": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _f]),
    "lrpx_batchnorm_rule": (_i, [_f, _f, _f, _f, _f, _f, C.c_float, _f, _l, _i, _l, _i, _f]),
    "lrpx_add_rule": (_i, [_f, _f, _f, _f, _f, _l, _f]),
    "lrpx_avgpool_rule": (_i, [_f, _f, _f, _f, _l] + [_i] * 12 + [_f]),
    "lrpx_max_abs_diff": (_i, [_f, _f, _l, _f, _f]),
}

_lib = None


class LrpxError(RuntimeError):
    pass


def load():
    """Load liblrpx.so once.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LrpxError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                        "(or __graft_entry__.build()); there is no CPU fallback for the LRP hot path")
    # torch first: it ships its own libamdhip64; liblrpx.so must bind to THAT runtime (same device context, same
    # streams).  Loaded the other way round the process holds two HIP runtimes and ours sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    ns = _Namespace()
    ns._cdll = lib
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name) and os.environ.get("LRPX_LIB_PATH"):
            continue          # (A/B against an OLDER build of the ABI through LRPX_LIB_PATH: entry points it predates stay unbound)
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
        setattr(ns, name, _recordable(fn))
    _lib = ns
    return ns


class _Namespace(object):
    """the bound entry points of liblrpx.so (attributes by name; `_cdll`: the ctypes library itself)"""


# ---- recorded steps ------------------------------------------------------------------------------------------------------------
# A step of the decoders is 50 - 330 launches of 5 - 20 us; issuing one from Python costs ~9 us (pointer objects, argument
# conversion, the status check), so the bottom-up step (config 5) and the one-image drop-in are bound by the INTERPRETER, not by
# the GPU.  A `Recording` keeps the (function, arguments) pairs of one eager run of a step together with every
# tensor / descriptor the arguments point at; `replay()` issues the same calls again - the same kernels in the same order on the
# same buffers, ~1.5 us of host time each - as ordinary launches on the current stream (no HIP graph: graphs were measured to
# overlap worse with the other batches in flight).  Like a graph, a recording works on static buffers: the caller copies new
# inputs into the recorded input tensors first (explainers: explain_batch_replay).
_TLS = threading.local()


class Recording(object):
    def __init__(self):
        self.calls, self.keep, self.stream, self.result = [], [], None, None

    def __enter__(self):
        if getattr(_TLS, "rec", None) is not None:
            raise LrpxError("a recording is already active on this thread")
        self.stream = stream_ptr().value
        _TLS.rec = self
        return self

    def __exit__(self, *exc):
        _TLS.rec = None
        return False

    def replay(self):
        """issue the recorded calls again on the current stream (which must be the stream of the recording)"""
        if stream_ptr().value != self.stream:
            raise LrpxError("a recorded step must be replayed on the stream it was recorded on")
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:
                check(rc)
        return self.result


# int-valued entry points that are QUERIES / setters, not launches with a status: never part of a recorded step
_QUERIES = {"lrpx_version", "lrpx_conv_kc", "lrpx_set_bf16x6", "lrpx_set_conv_mode", "lrpx_set_forward_f16"}


def _recordable(fn):
    status = fn.restype is _i and getattr(fn, "__name__", "") not in _QUERIES

    def call(*args):
        rec = getattr(_TLS, "rec", None)
        if rec is not None and status:
            rec.calls.append((fn, args))
        return fn(*args)
    call.raw = fn
    call.__name__ = getattr(fn, "__name__", "lrpx")
    return call


def check(rc):
    """Translate a status code into the exception the reference raises at the same place."""
    if rc == OK:
        return
    msg = load().lrpx_last_error_string().decode()
    if rc in (ENONFINITE, EZERO):
        raise AssertionError(msg)          # lrp_modules.py:154-155 / lrp_wrapper.py:81
    if rc == EINVAL:
        raise ValueError(msg)              # lrp_modules.py:338 style
    raise LrpxError(f"lrpx error {rc}: {msg}")


_get_dev = None


def _same_device(t):
    """a tensor of another GPU than the thread's current one would meet a stream (stream_ptr: the CURRENT device's) and kernels of the
    wrong device - the same class of wrong-address fault as a host pointer; the C ABI checks it again (check_dev_ptrs)"""
    global _get_dev
    if _get_dev is None:
        import torch
        # one visible GPU (one process per GPU behind HIP_VISIBLE_DEVICES): nothing to confuse, and the check costs ~0.25 us on each of
        # the ~10 pointers of a 5-us decoder launch
        _get_dev = (getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device) if torch.cuda.device_count() > 1 else False
    if _get_dev is False:
        return
    cur = _get_dev()
    if t.device.index is not None and t.device.index != cur:
        raise ValueError("lrpx: tensor on cuda:%d, the current device is cuda:%d (torch.cuda.set_device / a device context first)" % (t.device.index, cur))


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    # a host tensor's address handed to a kernel is a GPU page fault that comes and goes with what the runtime happens to have
    # mapped (round 5: a test that passed CPU images aborted the process in two of three sessions): refuse it here, loudly
    if not t.is_cuda:
        raise TypeError("lrpx kernels take CUDA tensors (got a %s tensor): there is no CPU path" % t.device.type)
    _same_device(t)
    assert t.is_contiguous(), "lrpx needs contiguous tensors"
    rec = getattr(_TLS, "rec", None)
    if rec is not None:
        rec.keep.append(t)          # the recorded arguments hold this address: the tensor lives as long as the recording
    return C.c_void_p(t.data_ptr())


def ptr_at(t, offset_elems=0):
    """Device pointer of t's storage start + offset (for strided row views)."""
    if not t.is_cuda:
        raise TypeError("lrpx kernels take CUDA tensors (got a %s tensor): there is no CPU path" % t.device.type)
    _same_device(t)
    if t.element_size() != 4:
        raise TypeError("ptr_at counts 4-byte elements (got %s)" % t.dtype)
    rec = getattr(_TLS, "rec", None)
    if rec is not None:
        rec.keep.append(t)
    return C.c_void_p(t.data_ptr() + 4 * offset_elems)


_raw_stream = None


def stream_ptr():
    """the current HIP stream of the current device as a void* (torch owns the streams).  `torch.cuda.current_stream()` builds a
    Stream object through three Python layers (~3 us, ~20 times per decoder step); the raw getter of the same value is ~0.3 us."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get_raw, get_dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None:
            _raw_stream = lambda: get_raw(get_dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return C.c_void_p(_raw_stream())

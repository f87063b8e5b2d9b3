"""One device engine per set of weights.

The reference's evaluation script builds a NEW explainer for every image (evaluation.py:806-838: `ExplainGridTDAttention(args,
word_map)` inside the loop over the test set), and every constructor loads the same checkpoint (models/gridTDmodel.py:717-718).
Here an explainer's construction is the weight upload plus ~170 weight packs (13 VGG16 layers x 12 operand formats, the decoder's
GEMM operands): far more than one 20-word explanation.  So engines are kept per weight set:

  * a checkpoint path: keyed by (absolute path, size, mtime) - the evaluation loop's case;
  * an nn.Module / a state dict of torch tensors: keyed by every tensor's (data pointer, shape, in-place version counter): a
    module trained or patched between two constructions gets a fresh engine;
  * numpy arrays (the test generators) carry no version counter: never cached.

Engines hold only read-only weight state plus per-call scratch; an explainer that wants buffers of its own (two explanations in
flight on two streams) takes `engine.replica()`.  `clear()` drops everything (tests; memory)."""
import os
import threading

import numpy as np
import torch

_LOCK = threading.Lock()
_CACHE = {}
MAX_ENGINES = 4


def fingerprint(kind, source, extra=()):
    """cache key of a weight source, or None when it cannot be fingerprinted cheaply and safely"""
    if isinstance(source, (str, os.PathLike)):
        try:
            st = os.stat(source)
        except OSError:
            return None
        return (kind, "path", os.path.abspath(source), st.st_size, st.st_mtime_ns) + tuple(extra)
    state = source.state_dict() if hasattr(source, "state_dict") else source
    if not isinstance(state, dict) or not state:
        return None
    items = []
    for k, v in state.items():
        if not torch.is_tensor(v):
            return None
        items.append((k, v.data_ptr(), tuple(v.shape), str(v.dtype), str(v.device), v._version))
    return (kind, "tensors", tuple(items)) + tuple(extra)


def get(key, build):
    """the cached engine of `key` (built by `build()` on a miss); key None: always build"""
    if key is None:
        return build()
    with _LOCK:
        eng = _CACHE.get(key)
        if eng is not None:
            _CACHE[key] = _CACHE.pop(key)          # most recently used last
            return eng
    eng = build()
    with _LOCK:
        _CACHE[key] = eng
        while len(_CACHE) > MAX_ENGINES:
            _CACHE.pop(next(iter(_CACHE)))
    return eng


def clear():
    with _LOCK:
        _CACHE.clear()

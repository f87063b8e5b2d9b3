"""One device engine per set of weights.

The reference's evaluation script builds a NEW explainer for every image (evaluation.py:806-838: `ExplainGridTDAttention(args,
word_map)` inside the loop over the test set), and every constructor loads the same checkpoint (models/gridTDmodel.py:717-718).
Here an explainer's construction is the weight upload plus ~170 weight packs (13 VGG16 layers x 12 operand formats, the decoder's
GEMM operands): far more than one 20-word explanation.  So engines are kept per weight set:

  * a checkpoint path: keyed by (absolute path, size, mtime) - the evaluation loop's case;
  * an nn.Module / a state dict of torch tensors: keyed by every tensor's (data pointer, shape, in-place version counter, a sampled
    content digest): a module trained or patched between two constructions gets a fresh engine.  The cache entry HOLDS the source
    tensors (`get(..., hold=)`): an address in the key can therefore not be handed to another checkpoint's tensors while the entry
    lives (two `torch.load`ed state dicts, the first freed before the second was read, had identical fingerprints -
    same addresses, `_version` 0 - and the second explainer silently got the first one's engine).  Cost: up to MAX_ENGINES source
    state dicts stay alive on the host (~130 MB each for VGG16 + decoder), next to each engine's device state - for the AoA engines
    that includes the V x 4H token table of the decoupled trace (~80 MB at V = 9.6 k); `clear()` releases both.  The digest (four
    samples per tensor) also catches edits made through `.data`, which bump no version counter;
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
        items.append((k, v.data_ptr(), tuple(v.shape), str(v.dtype), str(v.device), v._version, _digest(v)))
    return (kind, "tensors", tuple(items)) + tuple(extra)


def _digest(v):
    """four samples of the tensor (first, last, two thirds): a few microseconds, content-dependent"""
    n = v.numel()
    if n == 0:
        return ()
    flat = v.detach().reshape(-1)
    return tuple(float(flat[i]) for i in sorted({0, n // 3, (2 * n) // 3, n - 1}))


def source_tensors(source):
    """what a cache entry must keep alive for its key to stay unique: the source's tensors (None for paths)"""
    if isinstance(source, (str, os.PathLike)) or source is None:
        return None
    state = source.state_dict() if hasattr(source, "state_dict") else source
    return [v for v in state.values() if torch.is_tensor(v)] if isinstance(state, dict) else None


def get(key, build, hold=None):
    """the cached engine of `key` (built by `build()` on a miss); key None: always build.  `hold`: objects the entry keeps alive
    (source_tensors(source) for tensor-keyed entries: the key holds their addresses)"""
    if key is None:
        return build()
    with _LOCK:
        ent = _CACHE.get(key)
        if ent is not None:
            _CACHE[key] = _CACHE.pop(key)          # most recently used last
            return ent[0]
    eng = build()
    with _LOCK:
        _CACHE[key] = (eng, hold)
        while len(_CACHE) > MAX_ENGINES:
            _CACHE.pop(next(iter(_CACHE)))
    return eng


def clear():
    with _LOCK:
        _CACHE.clear()

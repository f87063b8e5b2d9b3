"""Captions of unequal length inside one batch.

The reference explains one image at a time, whatever caption length its beam search returned
(models/gridTDmodel.py:935-937, loop :1147-1153; models/aoamodel.py:992-995, :1171-1176).  A batch pads the captions
to a common width T; `lens[b]` is the number of words of image b.  The lock-step decoder kernels skip the padded
words themselves (`lens` in the relevance-state structs of include/lrpx.h); everything that works per (word, pixel)
row - the projector rules and the VGG16 chains, > 95 % of a step - runs on the VALID rows only, compacted:

    rows[r]     = b * T + t   of the r-th valid (image, word) pair, image-major
    row2img[r]  = b
    offs[b]     = first compact row of image b
"""
import numpy as np
import torch


class RaggedRows:
    def __init__(self, lens, B, T, device):
        if torch.is_tensor(lens):
            lens = lens.detach().cpu().tolist()
        lens = [int(x) for x in lens]
        if len(lens) != B:
            raise ValueError(f"one caption length per image: got {len(lens)} lengths for {B} images")
        if any(x < 0 or x > T for x in lens):
            raise ValueError(f"caption lengths must lie in [0, {T}] (the padded width of the batch): {lens}")
        self.B, self.T = B, T
        self.lens_host = lens
        self.full = all(x == T for x in lens)
        self.n = int(sum(lens))
        rows = np.fromiter((b * T + t for b in range(B) for t in range(lens[b])), dtype=np.int32, count=self.n)
        offs = np.zeros(B, dtype=np.int32)
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.lens = to(np.asarray(lens, dtype=np.int32))
        self.rows = to(rows)
        self.row2img = to((rows // T).astype(np.int32)) if self.n else to(np.zeros(0, np.int32))
        self.offs = to(offs)


_CACHE = {}


def ragged(lens, B, T, device):
    """None for lens=None, else the (cached) RaggedRows of this length pattern"""
    if lens is None:
        return None
    if isinstance(lens, RaggedRows):
        return lens
    if torch.is_tensor(lens):
        lens = lens.detach().cpu().tolist()
    key = (tuple(int(x) for x in lens), B, T, str(device))
    r = _CACHE.get(key)
    if r is None:
        if len(_CACHE) > 64:
            _CACHE.clear()
        r = _CACHE[key] = RaggedRows(key[0], B, T, device)
    return r

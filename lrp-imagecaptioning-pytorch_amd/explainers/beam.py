"""Host bookkeeping of the reference's beam search (`GridTDModel.beam_search`, models/gridTDmodel.py:400-478;
`AOAModel.beam_search`, models/aoamodel.py, the same algorithm) around device steps: the explainers caption the image
with it (`get_hidden_parameters`, models/gridTDmodel.py:935: beam 2, 50 steps; models/aoamodel.py:992: beam 3, 20 steps)
before they explain, so a drop-in without `caption_encode=` has to decode the same caption.

The model step, the logits and the top-k (`lrpx_beam_topk`: log-softmax + cumulative score + k best over the live beams)
run on the device; what happens here is what the reference does in Python lists too (sequence bookkeeping, `<end>`
detection - one small device->host read per step, as the reference's `enumerate(next_word_idx)`)."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def run_beam_search(step, logits, reorder, vocab, beam_size, max_cap_length, start_id, end_id, device):
    """step(t, prev_words): model step t for all `beam_size` rows (prev_words: int64 device tensor, live rows first);
    logits(t) -> (beam_size, vocab) scores of step t; reorder(t, src): state row j of time t+1 <- row src[j].
    Returns the chosen sequence incl. <start> (and <end> if it was produced), as the reference's `seq`."""
    lib = _lib.load()
    assert 1 <= beam_size <= 4, "lrpx_beam_topk keeps at most 4 candidates"
    seqs = [[start_id] for _ in range(beam_size)]
    cum = torch.zeros(beam_size, dtype=torch.float32, device=device)
    prev = torch.full((beam_size,), start_id, dtype=torch.int64, device=device)
    # indices (4 x int64) and scores (4 x float32) of a step in ONE 48-byte buffer: one device -> host read per step instead of two
    buf = torch.zeros(6, dtype=torch.int64, device=device)
    idx, val = buf[:4], buf[4:].view(torch.float32)
    complete, complete_scores = [], []
    n_live = beam_size
    for t in range(max_cap_length):
        step(t, prev)
        lg = logits(t)
        rows, k = (1, beam_size) if t == 0 else (n_live, n_live)          # :440-443
        check(lib.lrpx_beam_topk(ptr(lg), lg.shape[1], rows, vocab, ptr(cum), k, ptr(idx), ptr(val), stream_ptr()))
        host = buf.cpu()
        top, sc = host[:k].tolist(), host[4:].view(torch.float32)[:k].tolist()
        beam_idx = [w // vocab for w in top]                               # :444 (floor division: PyTorch 1.4 `/` on int64)
        nxt = [w % vocab for w in top]
        seqs = [seqs[b] + [w] for b, w in zip(beam_idx, nxt)]              # :447
        inc = [i for i, w in enumerate(nxt) if w != end_id]                # :448
        for i in sorted(set(range(len(nxt))) - set(inc)):                  # :449-453
            complete.append(seqs[i])
            complete_scores.append(sc[i])
        n_live = n_live - (len(nxt) - len(inc))
        if n_live == 0:
            break
        seqs = [seqs[i] for i in inc]                                      # :458
        reorder(t, [beam_idx[i] for i in inc])                             # :460-465
        cum[:n_live] = torch.tensor([sc[i] for i in inc], dtype=torch.float32, device=device)
        prev[:n_live] = torch.tensor([nxt[i] for i in inc], dtype=torch.int64, device=device)
    if complete:
        return complete[complete_scores.index(max(complete_scores))]       # :469-470
    return seqs[0][:20]                                                    # :472 (the cut at 20 tokens is the reference's)


def caption_from_sequence(seq, word_map):
    """`sen_idx` (:474): the sequence without <start>, <end>, <unk>, <pad>; the explainers prepend <start> (:937)."""
    drop = {word_map[k] for k in ('<start>', '<end>', '<unk>', '<pad>') if k in word_map}
    return [word_map['<start>']] + [int(w) for w in seq if w not in drop]

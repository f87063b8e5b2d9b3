"""Multi-GPU sharding of the explanation batch.

Every image is independent, so a batch is cut into contiguous per-rank blocks (one process per GPU,
weights replicated) and explained with NO collective on the data path.  The only communication is the
optional terminal gather of the relevance maps to one rank (`torch.distributed` — backend "nccl" is RCCL
over xGMI on MI355X; "gloo" in the CPU tests).  The reference has no distributed code at all."""
import torch


def shard_bounds(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_bounds(lens, world):
    """Contiguous blocks balanced by cost instead of count for captions of unequal length: the decoder pass
    of a T-word caption costs ~T(T+1)/2 lock-step rows and the CNN pass ~c*T (c ~ 40: one VGG16 relevance
    pass per word dominates).  Returns [(lo, hi)] per rank."""
    cost = [t * (t + 1) / 2 + 40.0 * t for t in lens]
    total = sum(cost)
    bounds, lo, acc = [], 0, 0.0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        # take the next image while that brings the running cost closer to this rank's share of the total
        while hi < len(cost) and (acc + cost[hi] <= target + 1e-9 or target - acc > acc + cost[hi] - target or hi == lo) \
                and (len(cost) - hi) > (world - 1 - r):
            acc += cost[hi]
            hi += 1
        if r == world - 1:
            hi = len(cost)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def gather_to_rank0(local, group=None):
    """Gather per-rank result tensors (first dim = items of the shard, may differ by rank) to rank 0.
    Returns the concatenation on rank 0 and None elsewhere."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad.contiguous(), bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def explain_sharded(explain_fn, images, captions=None, gather=True, group=None, lens=None, n_items=None):
    """Run `explain_fn(images_shard, captions_shard) -> (maps, r_words)` on this rank's block of the global
    batch; with gather=True rank 0 gets the whole batch's results in input order.
    `images` is the global batch tensor, or a LOADER `images(lo, hi) -> (images_shard, captions_shard)` together with
    `n_items` (the global batch size; `lens` gives it when present): then a rank only ever materialises its own block
    [lo, hi) - at BASELINE config 4 (B = 256 over 8 GPUs) 32 images per rank instead of 256 on every one.
    lens (optional, one caption length per image; captions are then padded to a common width): the blocks are cut by
    COST (`balanced_bounds`: ~T(T+1)/2 decoder rows + one CNN pass per word) instead of by count, so a rank that holds the
    long captions holds fewer images (SURVEY §8(e) load balance), and `explain_fn(images, captions, lens_shard)` receives
    its block's lengths."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    loader = callable(images)
    if loader:
        assert captions is None, "a loader returns (images, captions) of its block"
        assert n_items is not None or lens is not None, "a loader needs n_items (or lens) to know the global batch size"
        n_all = len(lens) if n_items is None else int(n_items)
    else:
        n_all = images.shape[0]
    if lens is None:
        lo, hi = shard_bounds(n_all, world, rank)
    else:
        lens = [int(t) for t in lens]
        assert len(lens) == n_all, "one caption length per image"
        lo, hi = balanced_bounds(lens, world)[rank]
    im, cp = images(lo, hi) if loader else (images[lo:hi], captions[lo:hi])
    assert im.shape[0] == hi - lo and cp.shape[0] == hi - lo, "the loader must return exactly its block"
    maps, r_words = explain_fn(im, cp) if lens is None else explain_fn(im, cp, lens[lo:hi])
    if not gather:
        return maps, r_words
    return gather_to_rank0(maps, group), gather_to_rank0(r_words, group)

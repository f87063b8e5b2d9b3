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


def gather_to_rank0(local, group=None, sizes=None, out=None):
    """Gather per-rank result tensors (first dim = items of the shard, may differ by rank) to rank 0, in rank order.
    Returns the concatenation on rank 0 and None elsewhere.
    sizes: the first dimension of every rank's tensor when the caller knows it (the block bounds are a function of the
    batch alone: `explain_sharded` passes them) - nothing is exchanged or synchronised then.  Equal shards land directly in
    their slice of ONE result tensor (`out`, or a new one): no padding, no concatenation."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = local.contiguous()
    if sizes is None:          # one exchange of the sizes, read in one go (a single synchronisation)
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_n = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(all_n, n, group=group)
        sizes = all_n.tolist()
    sizes = [int(x) for x in sizes]
    assert len(sizes) == world and sizes[rank] == local.shape[0], "sizes must list every rank's first dimension"
    tail = tuple(local.shape[1:])
    if len(set(sizes)) == 1:
        bufs = None
        if rank == 0:
            if out is None:
                out = torch.empty((world * sizes[0],) + tail, dtype=local.dtype, device=local.device)
            assert tuple(out.shape) == (world * sizes[0],) + tail and out.is_contiguous()
            bufs = list(out.view((world, sizes[0]) + tail).unbind(0))       # views: the collective writes the result in place
        dist.gather(local, bufs, dst=0, group=group)
        return out if rank == 0 else None
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = local.new_zeros((mx,) + tail)
        pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    if out is None:
        out = torch.empty((sum(sizes),) + tail, dtype=local.dtype, device=local.device)
    o = 0
    for b, n_r in zip(bufs, sizes):
        out[o:o + n_r] = b[:n_r]
        o += n_r
    return out


def reduce_for_gather(maps, how):
    """What leaves the GPU (the terminal gather must never become the bottleneck - a config-4 step is ~80 ms per
    GPU, 8 x 385 MB of fp32 maps into one rank would be a tenth of it):
      "maps"    the (N,3,H,W) maps themselves;
      "heatmap" the channel mean (N,H,W) the evaluation experiments look at (evaluation.py:134; `lrpx_spatial_reduce` on
                the device: a third of the bytes);
      "stats"   the tpfp statistics (N,4) of that heat map (evaluation.py:506-513; `lrpx_map_stats`: ~1e-4 of the bytes)."""
    if callable(how):          # a reducer of the caller's (maps -> tensor with the same leading dimension)
        return how(maps)
    if how == "maps":
        return maps
    if how not in ("heatmap", "stats"):
        raise ValueError(f"unknown gather mode {how!r}: maps | heatmap | stats")
    from . import evaluation as ev
    lead = tuple(maps.shape[:-3])
    heat = ev.spatial_relevance(maps.reshape(-1, *maps.shape[-3:]), "mean")
    if how == "heatmap":
        return heat.reshape(*lead, *heat.shape[-2:])
    return ev.map_statistics(heat).reshape(*lead, 4)


class OverlappedGather:
    """The terminal collective of a step issued on a SIDE stream from a double-buffered copy, so that the gather of step i
    runs under the compute of step i + 1 (the data path itself has no collective: SURVEY §8(e)).

        og = OverlappedGather(shape_of_the_local_result, group=...)
        for batch in batches:
            res = explain(batch)                # on the compute stream
            k = og.submit(res)                  # copy + collective queued on the side stream; returns at once
            ...
        og.finish()                             # all collectives done; rank 0 reads og.result(k)

    mode "gather": rank 0 receives (world, *shape); "all_gather": every rank does (`all_gather_into_tensor` on a preallocated
    tensor - for the small reduced results).  CPU tensors (gloo in the tests) take the same calls without streams."""

    def __init__(self, shape, dtype=torch.float32, device="cuda", group=None, depth=2, mode="gather", host_staged=False):
        import torch.distributed as dist
        self.dist, self.group, self.mode, self.depth = dist, group, mode, max(1, int(depth))
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = torch.device(device).type == "cuda"
        shape = tuple(shape)
        # host slots (the gloo rehearsal of a GPU run): PINNED here, so that the non_blocking device -> host copy really is asynchronous
        # (into pageable memory it blocks the host: no overlap)
        pin = (not self.cuda) and torch.cuda.is_available()
        self.stage = [torch.empty(shape, dtype=dtype, device=device, pin_memory=pin) for _ in range(self.depth)]
        # host-staged collectives are ISSUED by one worker thread in submit order (see _HostStaged); host_staged=True forces that path
        # for CPU inputs too (tests)
        self.force_host_staged = bool(host_staged) and not self.cuda
        self._q, self._worker = None, None
        need_out = mode == "all_gather" or self.rank == 0
        self.out = [torch.empty((self.world,) + shape, dtype=dtype, device=device) if need_out else None for _ in range(self.depth)]
        self.work = [None] * self.depth
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.k = 0

    def _collective(self, k):
        if self.mode == "all_gather":
            return self.dist.all_gather_into_tensor(self.out[k].view(-1), self.stage[k].view(-1), group=self.group, async_op=True)
        bufs = list(self.out[k].unbind(0)) if self.rank == 0 else None
        return self.dist.gather(self.stage[k], bufs, dst=0, group=self.group, async_op=True)

    class _HostStaged:
        """handle of a collective whose device -> pinned-host copy is still in flight (the rehearsal backend, gloo, moves host tensors
        only).  ONE worker thread per OverlappedGather takes the handles from a FIFO queue, waits for each copy and starts its
        collective: gloo pairs collectives across ranks by ISSUE ORDER and all the buffers have one shape, so with a thread per
        submit (round 5) and `depth` >= 2 two ranks could issue steps i and i + 1 in different orders and gather one step's data
        against another's without any error.  wait() blocks until the collective has been issued, then until it is done."""

        def __init__(self, og, k, ev):
            import threading
            self.work, self.error = None, None
            self.issued = threading.Event()
            self.og, self.k, self.ev = og, k, ev

        def issue(self):                    # (worker thread)
            try:
                if self.ev is not None:
                    self.ev.synchronize()
                self.work = self.og._collective(self.k)
            except BaseException as e:      # surfaced by wait() on the caller's thread
                self.error = e
            finally:
                self.issued.set()

        def wait(self):
            self.issued.wait()
            if self.error is not None:
                raise self.error
            if self.work is not None:
                self.work.wait()

    def _enqueue(self, handle):
        import queue
        import threading
        if self._worker is None:
            self._q = queue.Queue()

            def loop():
                while True:
                    h = self._q.get()
                    if h is None:
                        return
                    h.issue()
            self._worker = threading.Thread(target=loop, daemon=True)
            self._worker.start()
        self._q.put(handle)
        return handle

    def submit(self, local):
        k = self.k
        self.k = (k + 1) % self.depth
        if self.work[k] is not None:
            self.work[k].wait()                 # the slot's previous collective (two steps ago) has its data out
        if not self.cuda and local.is_cuda:     # host-staged (gloo rehearsal on a GPU box): async copy into the pinned slot
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=local.device)
            ev0 = torch.cuda.Event()
            ev0.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev0)
                self.stage[k].copy_(local, non_blocking=True)
                local.record_stream(self.stream)
                ev = torch.cuda.Event()
                ev.record()
            self.work[k] = self._enqueue(OverlappedGather._HostStaged(self, k, ev))
            return k
        if self.force_host_staged:
            self.stage[k].copy_(local)
            self.work[k] = self._enqueue(OverlappedGather._HostStaged(self, k, None))
            return k
        if not self.cuda:
            self.stage[k].copy_(local)
            self.work[k] = self._collective(k)
            return k
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self.stage[k].copy_(local, non_blocking=True)
            local.record_stream(self.stream)
            self.work[k] = self._collective(k)
        return k

    def result(self, k):
        if self.work[k] is not None:
            self.work[k].wait()
        return self.out[k]

    def finish(self):
        for w in self.work:
            if w is not None:
                w.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        elif self.stream is not None:
            self.stream.synchronize()

    def close(self):
        """stop the worker thread of the host-staged path (idempotent; the thread is a daemon, so forgetting this leaks nothing)"""
        self.finish()
        if self._worker is not None:
            self._q.put(None)
            self._worker.join()
            self._worker = None


def explain_sharded(explain_fn, images, captions=None, gather=True, group=None, lens=None, n_items=None, reduce="maps"):
    """Run `explain_fn(images_shard, captions_shard) -> (maps, r_words)` on this rank's block of the global
    batch; with gather=True rank 0 gets the whole batch's results in input order.
    `images` is the global batch tensor, or a LOADER `images(lo, hi) -> (images_shard, captions_shard)` together with
    `n_items` (the global batch size; `lens` gives it when present): then a rank only ever materialises its own block
    [lo, hi) - at BASELINE config 4 (B = 256 over 8 GPUs) 32 images per rank instead of 256 on every one.
    lens (optional, one caption length per image; captions are then padded to a common width): the blocks are cut by
    COST (`balanced_bounds`: ~T(T+1)/2 decoder rows + one CNN pass per word) instead of by count, so a rank that holds the
    long captions holds fewer images (SURVEY §8(e) load balance), and `explain_fn(images, captions, lens_shard)` receives
    its block's lengths.
    reduce: what is gathered - "maps" (default), "heatmap", "stats" or a callable (`reduce_for_gather`: reduced on the device first)."""
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
        bounds = [shard_bounds(n_all, world, r) for r in range(world)]
    else:
        lens = [int(t) for t in lens]
        assert len(lens) == n_all, "one caption length per image"
        bounds = balanced_bounds(lens, world)
    lo, hi = bounds[rank]
    sizes = [h - l for l, h in bounds]         # every rank knows every block: the gather exchanges no sizes and never synchronises
    im, cp = images(lo, hi) if loader else (images[lo:hi], captions[lo:hi])
    assert im.shape[0] == hi - lo and cp.shape[0] == hi - lo, "the loader must return exactly its block"
    maps, r_words = explain_fn(im, cp) if lens is None else explain_fn(im, cp, lens[lo:hi])
    if not gather:
        return maps, r_words
    maps = reduce_for_gather(maps, reduce)
    return gather_to_rank0(maps, group, sizes), gather_to_rank0(r_words, group, sizes)

"""World-size-2 test of the sharded driver on CPU with the gloo backend (the N>1 path of bench.py / shard.py):
block partition, no collective on the data path, terminal gather in input order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lrp_amd  # noqa: F401
from lrp_amd import shard


def test_bounds_cover_everything_once():
    for n in (0, 1, 7, 16, 33):
        for world in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_balanced_bounds_by_caption_length():
    lens = [20, 3, 3, 3, 3, 3, 18, 2]
    b = shard.balanced_bounds(lens, 2)
    assert b[0][0] == 0 and b[-1][1] == len(lens) and b[0][1] == b[1][0]
    cost = lambda lo, hi: sum(t * (t + 1) / 2 + 40 * t for t in lens[lo:hi])
    assert abs(cost(*b[0]) - cost(*b[1])) < 0.5 * (cost(0, len(lens)))
    assert all(hi > lo for lo, hi in shard.balanced_bounds([5] * 8, 8))


def _fake_explain(images, captions):
    """deterministic stand-in for the GPU engine: a 'map' and 'r_words' that depend on the inputs only"""
    maps = images.mean(dim=(2, 3), keepdim=True) * captions.float().sum(1).view(-1, 1, 1, 1)
    return maps.expand(-1, -1, 4, 4).contiguous(), captions[:, 1:].float()


def _fake_explain_lens(images, captions, lens):
    """as above with per-image caption lengths: words past an image's length contribute nothing (as the engines treat `lens`)"""
    keep = (torch.arange(captions.shape[1] - 1).view(1, -1) < torch.tensor(lens).view(-1, 1)).float()
    rw = captions[:, 1:].float() * keep
    maps = images.mean(dim=(2, 3), keepdim=True) * rw.sum(1).view(-1, 1, 1, 1)
    return maps.expand(-1, -1, 4, 4).contiguous(), rw


def _worker(rank, world, port, n_img, out, lens=None, loader=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(n_img, 3, 8, 8, generator=g)
    caps = torch.randint(1, 50, (n_img, 5 if lens is None else max(lens) + 1), generator=g)
    if loader:
        # per-rank loader: the rank asks for its block only, once, and never sees the global batch
        asked = []

        def load(lo, hi):
            asked.append((lo, hi))
            return images[lo:hi].clone(), caps[lo:hi].clone()
        if lens is None:
            maps, rw = shard.explain_sharded(_fake_explain, load, gather=True, n_items=n_img)
            assert asked == [shard.shard_bounds(n_img, world, rank)]
        else:
            maps, rw = shard.explain_sharded(_fake_explain_lens, load, gather=True, lens=lens)
            assert asked == [shard.balanced_bounds(lens, world)[rank]]
    elif lens is None:
        maps, rw = shard.explain_sharded(_fake_explain, images, caps, gather=True)
    else:
        seen = []

        def fn(im, cp, ln):
            seen.append(len(ln))
            return _fake_explain_lens(im, cp, ln)
        maps, rw = shard.explain_sharded(fn, images, caps, gather=True, lens=lens)
        lo, hi = shard.balanced_bounds(lens, world)[rank]
        assert seen == [hi - lo]                          # this rank explained exactly its cost-balanced block
    if rank == 0:
        want_m, want_w = _fake_explain(images, caps) if lens is None else _fake_explain_lens(images, caps, lens)
        out.put((torch.equal(maps, want_m), torch.equal(rw, want_w), tuple(maps.shape)))
    else:
        assert maps is None and rw is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_img", [4, 5])
def test_two_ranks_gloo(n_img):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_img, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_m, ok_w, shape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_m and ok_w and shape[0] == n_img


def test_two_ranks_gloo_unequal_caption_lengths():
    """SURVEY §8(e) load balance: with caption lengths the blocks are cut by cost - rank 0 gets two of the three 20-word
    captions (cost 2020), rank 1 the third and the five short ones (1598; by count it would be 3232 / 386) - and rank 0
    still receives everything in input order"""
    lens = [20, 20, 20, 3, 2, 4, 3, 2]
    b = shard.balanced_bounds(lens, 2)
    assert b == [(0, 2), (2, 8)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, len(lens), q, lens)) for r in range(2)]
    for p in procs:
        p.start()
    ok_m, ok_w, shape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_m and ok_w and shape[0] == len(lens)


@pytest.mark.parametrize("lens", [None, [20, 20, 20, 3, 2, 4, 3, 2]])
def test_two_ranks_gloo_per_rank_loader(lens):
    """`explain_sharded(fn, loader, n_items=...)` - every rank asks its loader for its own block [lo, hi)
    only (by count, or by cost when caption lengths are given) and rank 0 still gets the whole batch in input order"""
    n_img = 5 if lens is None else len(lens)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_img, q, lens, True)) for r in range(2)]
    for p in procs:
        p.start()
    ok_m, ok_w, shape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_m and ok_w and shape[0] == n_img


def _gather_worker(rank, world, port, out):
    """gather_to_rank0 with known / unknown sizes, equal / unequal shards, a preallocated result; the reduced gather of
    explain_sharded; OverlappedGather in both modes (depth 2, five steps: every slot reused)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # equal shards, sizes known: the result tensor is written in place (no padding, no concatenation)
    local = torch.arange(3 * 4, dtype=torch.float32).view(3, 4) + 100 * rank
    pre = torch.full((world * 3, 4), -1.0) if rank == 0 else None
    got = shard.gather_to_rank0(local, sizes=[3] * world, out=pre)
    want = torch.cat([torch.arange(12, dtype=torch.float32).view(3, 4) + 100 * r for r in range(world)])
    if rank == 0:
        ok &= got is pre and torch.equal(got, want)
    else:
        ok &= got is None
    # sizes unknown: one exchange; unequal shards keep the rank order
    loc2 = torch.full((2 + rank, 5), float(rank))
    got2 = shard.gather_to_rank0(loc2)
    if rank == 0:
        ok &= tuple(got2.shape) == (sum(2 + r for r in range(world)), 5)
        ok &= torch.equal(got2, torch.cat([torch.full((2 + r, 5), float(r)) for r in range(world)]))
    got3 = shard.gather_to_rank0(loc2, sizes=[2 + r for r in range(world)])
    if rank == 0:
        ok &= torch.equal(got3, got2)
    # explain_sharded with a reducer: the channel mean is what travels (a third of the bytes), in input order
    g = torch.Generator().manual_seed(0)
    images = torch.randn(5, 3, 8, 8, generator=g)
    caps = torch.randint(1, 50, (5, 5), generator=g)
    heat = lambda m: m.mean(dim=1)
    hm, rw = shard.explain_sharded(_fake_explain, images, caps, gather=True, reduce=heat)
    if rank == 0:
        ok &= torch.equal(hm, heat(_fake_explain(images, caps)[0])) and tuple(hm.shape) == (5, 4, 4)
    try:
        shard.reduce_for_gather(images, "nonsense")
        ok = False
    except ValueError:
        pass
    # the overlapped gather: results of step i are complete when asked for, slots are reused safely
    for mode in ("gather", "all_gather"):
        og = shard.OverlappedGather((2, 3), device="cpu", depth=2, mode=mode)
        slots = []
        for step in range(5):
            k = og.submit(torch.full((2, 3), float(10 * step + rank)))
            slots.append(k)
            res = og.result(k)
            if mode == "all_gather" or rank == 0:
                ok &= torch.equal(res, torch.stack([torch.full((2, 3), float(10 * step + r)) for r in range(world)]))
            else:
                ok &= res is None
        og.finish()
        ok &= slots == [0, 1, 0, 1, 0]
    # the host-staged path (gloo rehearsal of a GPU run) with SEVERAL submits outstanding before any result is asked for:
    # one worker issues the collectives in submit order on every rank, however the ranks' timing differs; rank 1 is slowed down
    # between its submits so that a per-submit thread of rank 0 would run ahead
    import time
    for mode in ("gather", "all_gather"):
        og = shard.OverlappedGather((4, 3), device="cpu", depth=4, mode=mode, host_staged=True)
        ks = []
        for step in range(3):
            ks.append(og.submit(torch.full((4, 3), float(10 * step + rank))))
            if rank == 1:
                time.sleep(0.05)
        for step, k in enumerate(ks):
            res = og.result(k)
            if mode == "all_gather" or rank == 0:
                ok &= torch.equal(res, torch.stack([torch.full((4, 3), float(10 * step + r)) for r in range(world)]))
        # a second round re-uses the slots in order
        k = og.submit(torch.full((4, 3), float(77 + rank)))
        res = og.result(k)
        if mode == "all_gather" or rank == 0:
            ok &= torch.equal(res, torch.stack([torch.full((4, 3), float(77 + r)) for r in range(world)]))
        og.close()
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_gather_paths():
    """the terminal gather without size exchange / padding / concatenation when the blocks are known, the
    reduced gather (channel mean) of `explain_sharded`, and `OverlappedGather` (double-buffered, gather and all_gather_into_tensor)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def _config4_worker(rank, world, port, out):
    """BASELINE config 4's block map at world size 8: B = 256 images, LRP + Guided-Backprop side by side; every rank explains its 32
    images only, rank 0 receives both families' (reduced) maps in input order through the known-size gather (no size exchange)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T = 256, 3
    g = torch.Generator().manual_seed(1)
    images = torch.randn(B, 3, 4, 4, generator=g)
    caps = torch.randint(1, 50, (B, T + 1), generator=g)
    asked = []

    def load(lo, hi):
        asked.append((lo, hi))
        return images[lo:hi].clone(), caps[lo:hi].clone()

    def explain_both(im, cp):          # (LRP maps | guided maps) side by side on the map axis, as bench.py --config 4 produces them
        m, rw = _fake_explain(im, cp)
        return torch.cat([m, -2.0 * m], dim=1), rw
    maps, rw = shard.explain_sharded(explain_both, load, gather=True, n_items=B)
    ok = asked == [shard.shard_bounds(B, world, rank)] and asked[0] == (32 * rank, 32 * rank + 32)
    if rank == 0:
        m, w = _fake_explain(images, caps)
        ok &= torch.equal(maps, torch.cat([m, -2.0 * m], dim=1)) and torch.equal(rw, w) and maps.shape[0] == B
    else:
        ok &= maps is None and rw is None
    # the reduced gather of the same batch (heat map = channel mean: a third of the bytes)
    hm, _ = shard.explain_sharded(_fake_explain, images, caps, gather=True, reduce=lambda m: m.mean(dim=1))
    if rank == 0:
        ok &= tuple(hm.shape) == (B, 4, 4) and torch.allclose(hm, _fake_explain(images, caps)[0].mean(dim=1))
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_gloo_config4_block_map():
    """the first real 8-GPU run must not fail on bookkeeping - world size 8 on gloo (CPU), config 4's B = 256:
    bounds, order, per-rank loader, full and reduced gather"""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_config4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {r: True for r in range(world)}

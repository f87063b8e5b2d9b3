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


def _worker(rank, world, port, n_img, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(n_img, 3, 8, 8, generator=g)
    caps = torch.randint(1, 50, (n_img, 5), generator=g)
    maps, rw = shard.explain_sharded(_fake_explain, images, caps, gather=True)
    if rank == 0:
        want_m, want_w = _fake_explain(images, caps)
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

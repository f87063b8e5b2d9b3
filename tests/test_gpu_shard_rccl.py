"""The sharded driver and the terminal gather on the RCCL backend itself (`torch.distributed` backend "nccl" is RCCL on ROCm), as far
as ONE GPU allows: a process group of world size 1 on cuda:0.  The multi-rank logic (block bounds, order, unequal shards) is covered
on CPU with gloo (tests/test_shard_gloo.py); what only a device backend can show is that the collectives take the engines' DEVICE
tensors as they are - `gather` into views of one preallocated result, `all_gather_into_tensor`, both issued asynchronously on a side
stream from the double-buffered slots of `OverlappedGather` - and that the reduced gathers (`heatmap`, `stats`) equal the device
reductions of `lrp_amd.evaluation` on the full maps.  The real engine runs underneath, with captions of unequal length."""
import os
import socket

import pytest
import torch

import lrp_amd  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_explain_sharded_and_gathers_on_rccl(group):
    from lrp_amd import evaluation as ev, shard, weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    V, T, lens = 331, 4, [4, 1, 3]
    eng = GridTDEngine(weights.make_gridtd_state(seed=51, vocab_size=V))
    images = torch.from_numpy(weights.make_images(52, 3)).cuda()
    caps = torch.from_numpy(weights.make_captions(53, 3, T, V)).cuda()
    want_maps, want_words = eng.explain_batch(images, caps, lens=lens)
    want_maps, want_words = want_maps.clone(), want_words.clone()

    def fn(im, cp, ln):
        return eng.explain_batch(im, cp, lens=ln)
    maps, words = shard.explain_sharded(fn, images, caps, gather=True, lens=lens)
    assert torch.equal(maps, want_maps) and torch.equal(words, want_words) and maps.is_cuda
    heat, _ = shard.explain_sharded(fn, images, caps, gather=True, lens=lens, reduce="heatmap")
    want_heat = ev.spatial_relevance(want_maps.view(-1, 3, 224, 224), "mean").view(3, T, 224, 224)
    assert torch.equal(heat, want_heat)
    stats, _ = shard.explain_sharded(fn, images, caps, gather=True, lens=lens, reduce="stats")
    assert torch.equal(stats, ev.map_statistics(want_heat.view(-1, 224, 224)).view(3, T, 4))
    # a loader instead of the global batch, no gather: the rank's own block
    m2, w2 = shard.explain_sharded(lambda im, cp: eng.explain_batch(im, cp), lambda lo, hi: (images[lo:hi], caps[lo:hi]), gather=False, n_items=3)
    assert m2.shape[0] == 3 and torch.equal(w2, eng.explain_batch(images, caps)[1])
    # the overlapped gather on device tensors: five steps through two slots, both collectives, issued from a side stream
    for mode in ("gather", "all_gather"):
        og = shard.OverlappedGather((3 * T, 224, 224), device="cuda", depth=2, mode=mode)
        seen = []
        for step in range(5):
            local = want_heat.view(3 * T, 224, 224) * float(step + 1)
            k = og.submit(local)
            seen.append((k, og.result(k).clone()))
        og.finish()
        torch.cuda.synchronize()
        for step, (k, res) in enumerate(seen):
            assert k == step % 2 and tuple(res.shape) == (1, 3 * T, 224, 224)
            assert torch.equal(res[0], want_heat.view(3 * T, 224, 224) * float(step + 1)), (mode, step)

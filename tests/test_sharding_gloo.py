"""N > 1 path on CPU: two gloo ranks.  Checks the sharding arithmetic, that per-rank shards of the oracle's output
concatenate to the full-batch output (what the GPU ranks rely on), and that the flat all-reduce of the aggregator's
1x1-conv gradients equals the full-batch gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from multiviewhmr_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_the_batch():
    for B in (1, 2, 7, 32, 33):
        for R in (1, 2, 3, 4, 8):
            spans = [sharding.shard_bounds(B, R, r) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cport
        d = load_golden("unproj", "tiny_b2v2c4")
        feats, proj, coords = (torch.from_numpy(d[k]) for k in ("features", "proj", "coords"))
        f, p, c = sharding.shard_batch(feats, proj, coords)
        assert f.shape[0] == 1 and f.data_ptr() == feats[rank:rank + 1].data_ptr()          # a view, not a copy
        # the oracle stands in for the kernel here (no GPU): shard in, shard out
        part = torch.from_numpy(cport.forward(f.numpy(), p.numpy(), c.numpy(), "softmax"))
        gathered = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(gathered, part)
        full = torch.cat(gathered)
        np.testing.assert_array_equal(full.numpy(), cport.forward(d["features"], d["proj"], d["coords"], "softmax"))

        # aggregator 1x1 conv: per-rank gradient on its shard, one flat all-reduce == full-batch gradient / world
        torch.manual_seed(0)
        conv = torch.nn.Sequential(torch.nn.Conv2d(4, 3, 1))
        x = feats.view(-1, 4, 12, 12)
        full_loss = conv(x).square().sum() / world
        gw, gb = torch.autograd.grad(full_loss, list(conv.parameters()))
        lo, hi = sharding.shard_bounds(feats.shape[0], world, rank)
        conv.zero_grad()
        conv(feats[lo:hi].reshape(-1, 4, 12, 12)).square().sum().backward()
        n = sharding.allreduce_aggregator_grads(conv)
        assert n == 4 * 3 + 3
        np.testing.assert_allclose(conv[0].weight.grad.numpy(), gw.numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(conv[0].bias.grad.numpy(), gb.numpy(), rtol=1e-5, atol=1e-5)

        # the reference's batch dict shards the same way
        batch = dict(images=np.zeros((2, 2, 8, 8, 3)), cameras=[["a0", "a1"], ["b0", "b1"]], keypoints_3d=[np.zeros((17, 3))] * 2)
        sb = sharding.shard_batch_dict(batch)
        assert sb["images"].shape[0] == 1 and sb["cameras"] == [["a%d" % rank], ["b%d" % rank]] and len(sb["keypoints_3d"]) == 1
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_equivalence_and_grad_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]

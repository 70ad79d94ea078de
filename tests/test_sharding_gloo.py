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


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --train-step's control flow on two gloo ranks: a real VolumeGenerator per rank on its (uneven) shard of the batch, backward,
# ONE flat all-reduce of the aggregator's gradients -- with the HIP op replaced by an oracle-backed stand-in (no GPU here; the
# oracle is test infrastructure).  What must hold on the GPU ranks: the payload layout is identical on every rank even when a
# parameter received no gradient, and the reduced gradients are the full batch's.
def _stub_unprojection_cuboid():
    from oracle import cport

    def coords_of(rots, centers, position, sides, vol):
        g = np.stack(np.meshgrid(*[np.arange(n) for n in vol], indexing="ij"), -1).astype(np.float32)
        pos, step = np.asarray(position, np.float32), (np.asarray(sides, np.float64) / (np.asarray(vol) - 1)).astype(np.float32)
        grid = (pos + step * g).astype(np.float32)                                  # aggregation.py:157-159
        out = np.empty((len(rots),) + grid.shape, np.float32)
        for b in range(len(rots)):
            c = centers[b].numpy().astype(np.float32)
            out[b] = (grid - c) @ rots[b].numpy().astype(np.float32).T + c          # :184-186
        return out

    class Stub(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, proj, coords):
            ctx.save_for_backward(features, proj, coords)
            return torch.from_numpy(cport.forward(features.detach().numpy(), proj.numpy(), coords.numpy(), "softmax"))

        @staticmethod
        def backward(ctx, go):
            f, p, c = ctx.saved_tensors
            return torch.from_numpy(cport.backward(go.contiguous().numpy(), f.detach().numpy(), p.numpy(), c.numpy(), "softmax")), None, None

    def unprojection_cuboid(features, proj, rots, centers, position, sides, volume_shape, aggregation_method='softmax', *, out_dtype=None, variant='auto'):
        assert aggregation_method == "softmax"
        coords = torch.from_numpy(coords_of(rots, centers, position, sides, tuple(volume_shape)))
        return Stub.apply(features.contiguous(), proj.float().contiguous(), coords)
    return unprojection_cuboid


def _train_step_batch(B, V, img):
    from multiviewhmr_amd import multiview
    cams = [[None] * B for _ in range(V)]
    for v in range(V):
        for b in range(B):
            az = 2 * np.pi * v / V + 0.3 + 0.05 * b
            eye = np.array([5000 * np.cos(az), 5000 * np.sin(az), 1500.0])
            fwd = -eye / np.linalg.norm(eye)
            right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
            R = np.stack([right, np.cross(fwd, right), fwd])
            cam = multiview.Camera(R, -R @ eye, [[1145.0, 0, 512], [0, 1145.0, 512], [0, 0, 1]])
            cam.update_after_crop((150, 150, 850, 850))
            cam.update_after_resize((700, 700), (img, img))
            cams[v][b] = cam
    rng = np.random.default_rng(5)
    return dict(images=np.zeros((B, V, img, img, 3), np.uint8), cameras=cams, keypoints_3d=[rng.normal(0, 100, (17, 3)).astype(np.float32) for _ in range(B)])


def _train_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multiviewhmr_amd import aggregation
        aggregation.unprojection_cuboid = _stub_unprojection_cuboid()              # the kernel hook: VolumeGenerator.forward calls this name
        B, V, Cin, Cout, S, hw, IMG = 3, 2, 6, 4, 8, 12, 48                        # 3 samples over 2 ranks: shards of 2 and 1
        batch = _train_step_batch(B, V, IMG)
        torch.manual_seed(0)                                                       # same initial weights on every rank (bench.py does the same)
        gen = aggregation.VolumeGenerator(volume_size=S, input_channels=Cin, output_channels=Cout, device="cpu").eval()
        gen.extra_head = torch.nn.Parameter(torch.ones(5))                         # a parameter this step never touches: no .grad on any rank
        feats = torch.randn(B, V, Cin, hw, hw, generator=torch.Generator().manual_seed(1))
        go = torch.randn(B, Cout, S, S, S, generator=torch.Generator().manual_seed(2))
        proj_org = torch.from_numpy(np.stack([[batch["cameras"][v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32))
        # full batch on this process: the gradients the ranks must end up with
        gen.zero_grad(set_to_none=True)
        (gen(feats, proj_org, batch) * go).sum().backward()
        want = {n: p.grad.clone() for n, p in gen.named_parameters() if p.grad is not None}
        assert set(want) == {"process_feature.0.weight", "process_feature.0.bias"}
        # this rank's shard
        lo, hi = sharding.shard_bounds(B, world, rank)
        assert (lo, hi) == ((0, 2), (2, 3))[rank]
        gen.zero_grad(set_to_none=True)
        sb = sharding.shard_batch_dict(batch)
        (gen(feats[lo:hi], proj_org[lo:hi], sb) * go[lo:hi]).sum().backward()
        assert gen.extra_head.grad is None
        n = sharding.allreduce_aggregator_grads(gen, average=False)
        assert n == Cout * Cin + Cout + 5                                           # identical flat layout on both ranks, unused parameter included
        for name, p in gen.named_parameters():
            if name in want:
                np.testing.assert_allclose(p.grad.numpy(), want[name].numpy(), rtol=2e-5, atol=2e-5)
            else:
                assert p.grad is not None and float(p.grad.abs().max()) == 0.0     # zeros went into the reduce, zeros came back
        # average=True (what bench.py --train-step calls): the sum divided by the world size
        gen.zero_grad(set_to_none=True)
        (gen(feats[lo:hi], proj_org[lo:hi], sb) * go[lo:hi]).sum().backward()
        sharding.allreduce_aggregator_grads(gen)
        np.testing.assert_allclose(gen.process_feature[0].weight.grad.numpy(), want["process_feature.0.weight"].numpy() / world, rtol=2e-5, atol=2e-5)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_control_flow_with_a_stub_kernel(tmp_path):
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]

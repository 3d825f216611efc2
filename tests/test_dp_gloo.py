"""CPU, world_size 2 over gloo: the data-parallel harness (closerlook3d_amd/dp.py) -- shard ranges and
gradient averaging (bucketed/overlapped and one-shot) equal the single-process mean of per-shard
gradients.  The model is a small stand-in: dp.py is independent of what produces the gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from closerlook3d_amd.dp import FlatGradients, GradientSynchronizer, allreduce_gradients, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 16, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv1d(6, 16, 1, bias=False), torch.nn.BatchNorm1d(16), torch.nn.ReLU(),
                               torch.nn.Conv1d(16, 4, 1))


def _batch():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 6, 32, generator=g)


def _worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        lo, hi = shard_range(8, rank, world)
        x = _batch()[lo:hi]
        params = list(model.parameters())
        if mode == "bucketed":
            sync = GradientSynchronizer(params, world, bucket_bytes=256)  # several tiny buckets
            model(x).square().sum().backward()
            sync.finish()
            model.zero_grad()
            model(x).square().sum().backward()  # second step reuses the buckets
            sync.finish()
        elif mode == "flat":
            flat = FlatGradients(params)
            for _ in range(2):  # the second step must land in the same views again
                flat.zero_()
                model(x).square().sum().backward()
                assert all(p.grad.data_ptr() >= flat.buffer.data_ptr() and
                           p.grad.data_ptr() < flat.buffer.data_ptr() + 4 * flat.buffer.numel() for p in params)
                flat.allreduce_mean(world)
        else:
            model(x).square().sum().backward()
            allreduce_gradients(params, world)
        if rank == 0:
            torch.save([p.grad.clone() for p in params], out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bucketed", "oneshot", "flat"])
def test_gradient_mean_world2(tmp_path, mode):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = torch.load(out)
    want = None
    for r in range(2):
        model = _model()
        lo, hi = shard_range(8, r, 2)
        model(_batch()[lo:hi]).square().sum().backward()
        g = [p.grad for p in model.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    for a, b in zip(got, want):
        assert torch.allclose(a, b / 2, atol=1e-6, rtol=1e-5)


class _TwoHeads(torch.nn.Module):
    """Shared trunk + per-shape heads: a rank whose shard holds only one shape gives the other head no gradient."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.trunk = torch.nn.Conv1d(6, 12, 1)
        self.heads = torch.nn.ModuleList([torch.nn.Conv1d(12, 3, 1), torch.nn.Conv1d(12, 5, 1)])

    def forward(self, x, shape):
        return self.heads[shape](torch.relu(self.trunk(x)))


def _heads_worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _TwoHeads()
        params = list(model.parameters())
        x = _batch()[4 * rank:4 * rank + 4]
        if mode == "bucketed":
            sync = GradientSynchronizer(params, world, bucket_bytes=128)  # heads and trunk in different buckets
            for _ in range(2):
                model.zero_grad(set_to_none=True)
                model(x, rank).square().sum().backward()  # rank r only touches head r
                sync.finish()
        else:
            model(x, rank).square().sum().backward()
            allreduce_gradients(params, world)
        assert all(p.grad is not None for p in params)  # every replica steps every parameter
        torch.save([p.grad.clone() for p in params], f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bucketed", "oneshot"])
def test_gradient_mean_when_a_rank_skips_a_head(tmp_path, mode):
    """parameters without a gradient on one rank -- collectives stay in one order on all ranks, the
    absent gradient counts as zero, and both replicas end up with the same .grad for every parameter."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grads.pt")
    mp.spawn(_heads_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(2)]
    want = None
    for r in range(2):
        model = _TwoHeads()
        model(_batch()[4 * r:4 * r + 4], r).square().sum().backward()
        g = [p.grad if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    for a, b, w in zip(got[0], got[1], want):
        assert torch.equal(a, b)
        assert torch.allclose(a, w / 2, atol=1e-6, rtol=1e-5)

"""CPU tests of the N>1 path (gloo, world_size 2): row/frame sharding, the single flat
all-reduce of the parameter gradients, replica consistency of the data-parallel step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_amd import parallel


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(fn, world, *args):
    port = free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def test_row_shards_partition_the_frame():
    for h in (1, 5, 100, 800, 801):
        for world in (1, 2, 3, 4, 8):
            blocks = [parallel.shard_rows(h, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == h
            for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_rows(800, 3, 8) == (300, 400)       # BASELINE config 4: 100 rows per GPU


def make_net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.LayerNorm(32), torch.nn.ReLU(),
                               torch.nn.Linear(32, 3))


def batch(step, n=64):
    g = torch.Generator().manual_seed(100 + step)
    return torch.randn(n, 6, generator=g), torch.rand(n, 3, generator=g)


def mse(model, b):
    x, y = b
    return ((model(x) - y) ** 2).mean()


def _reduce_worker(rank, world, out_dir):
    net = make_net()
    x, y = batch(0)
    lo, hi = parallel.shard_items(x.shape[0], rank, world)
    mse(net, (x[lo:hi], y[lo:hi])).backward()
    flat = parallel.FlatGradientAllReduce(net.parameters())()
    assert flat.numel() == sum(p.numel() for p in net.parameters())
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f"g{rank}.pt"))


def test_flat_all_reduce_equals_full_batch_gradient(tmp_path):
    run_ranks(_reduce_worker, 2, str(tmp_path))
    net = make_net()
    mse(net, batch(0)).backward()
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"g{r}.pt"))
        for a, p in zip(got, net.parameters()):
            assert (a - p.grad).abs().max() <= 1e-6


def _trainer_worker(rank, world, out_dir):
    net = make_net()
    if rank == 1:                                    # replicas start different: broadcast must fix it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    parallel.broadcast_parameters(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)

    def loss_fn(model, b):
        lo, hi = parallel.shard_items(b[0].shape[0], rank, world)
        return mse(model, (b[0][lo:hi], b[1][lo:hi]))

    trainer = parallel.DataParallelTrainer(net, opt, loss_fn)
    for step in range(4):
        trainer.step(batch(step))
    torch.save([p.detach().clone() for p in net.parameters()], os.path.join(out_dir, f"p{rank}.pt"))


def test_data_parallel_trainer_matches_single_process(tmp_path):
    run_ranks(_trainer_worker, 2, str(tmp_path))
    net = make_net()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    single = parallel.DataParallelTrainer(net, opt, mse)
    for step in range(4):
        single.step(batch(step))
    p0 = torch.load(os.path.join(tmp_path, "p0.pt"))
    p1 = torch.load(os.path.join(tmp_path, "p1.pt"))
    for a, b, ref in zip(p0, p1, net.parameters()):
        assert torch.equal(a, b)                     # replicas stay bit-identical
        assert (a - ref).abs().max() <= 1e-5


def _gather_worker(rank, world, out_dir):
    h, w = 5, 3
    full = torch.arange(2 * h * w * 4, dtype=torch.float32).reshape(2, h, w, 4)
    b, e = parallel.shard_rows(h, rank, world)
    got = parallel._gather_rows(full[:, b:e].contiguous(), h, world)
    assert torch.equal(got, full)


def test_row_gather_with_uneven_blocks(tmp_path):
    run_ranks(_gather_worker, 2, str(tmp_path))


class FlatNet(torch.nn.Module):
    """A module whose backward hands out gradients as views of ONE flat vector, like the HIP
    backward does (nerf_amd/backward.py): lets the in-place path of FlatGradientAllReduce run on CPU."""

    def __init__(self):
        super().__init__()
        self.net = make_net()
        self.last_flat_grad = None

    def forward(self, x):
        return self.net(x)

    def flatten_grads(self):
        flat = torch.cat([p.grad.reshape(-1) for p in self.parameters()])
        off = 0
        for p in self.parameters():
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.last_flat_grad = flat


def _uneven_worker(rank, world, out_dir, split):
    """Uneven (and empty) shards: rank 0 takes `split` of the 64 examples, rank 1 the rest; each
    weights its mean-loss gradient by local / global before the SUM all-reduce."""
    net = FlatNet()
    x, y = batch(0)
    lo, hi = (0, split) if rank == 0 else (split, x.shape[0])
    n = hi - lo
    loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).sum() / max(3 * n, 1)        # mean, 0 for an empty shard
    loss.backward()
    net.flatten_grads()
    reduce = parallel.FlatGradientAllReduce(net.parameters())
    reduce(net.last_flat_grad, n / x.shape[0])
    assert reduce.in_place_calls == 1                                      # no pack / unpack copies
    assert all(p.grad.data_ptr() >= net.last_flat_grad.data_ptr() for p in net.parameters())
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f"u{rank}.pt"))


@pytest.mark.parametrize("split", [40, 63, 64])
def test_weighted_all_reduce_handles_uneven_and_empty_shards(tmp_path, split):
    run_ranks(_uneven_worker, 2, str(tmp_path), split)
    net = make_net()
    mse(net, batch(0)).backward()
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, f"u{r}.pt"))
        for a, p in zip(got, net.parameters()):
            assert (a - p.grad).abs().max() <= 1e-6


def test_all_reduce_falls_back_when_gradients_are_not_views():
    net = make_net()
    mse(net, batch(0)).backward()
    reduce = parallel.FlatGradientAllReduce(net.parameters())
    assert not reduce._aliases(torch.zeros(reduce.numel))
    assert not reduce._aliases(None)

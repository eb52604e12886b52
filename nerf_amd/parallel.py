"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

Inference shards with NO collective: rays/pixels are independent, so a rank renders its block of
image rows (or its frames of a batch) with replicated parameters.
Training is data-parallel over rays: every rank renders its own sub-batch, then ONE all-reduce of
the flat parameter-gradient vector (304,438 fp32 = 1.2 MB) per step; at that size the collective
is latency-bound (a few tens of microseconds), so it is a single bucket with no overlap, and Adam
runs redundantly on every rank, which keeps the replicas bit-identical.

Everything here is device-agnostic torch code (it is also exercised on CPU with the gloo backend
in tests/test_parallel_cpu.py); the renderer itself only runs on the GPU.
"""
import torch
import torch.distributed as dist


def shard_rows(image_h, rank, world_size):
    """Contiguous block of image rows of rank ``rank``: [begin, end).  Blocks differ by at most one
    row and cover [0, image_h) exactly."""
    base, extra = divmod(image_h, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_items(n_items, rank, world_size):
    """Contiguous block of items (frames of a batch, rays of a batch) of rank ``rank``."""
    return shard_rows(n_items, rank, world_size)


def render_image_sharded(model, camera_o, camera_r, image_h, image_w, focal_length, num_samples,
                         rank=None, world_size=None, gather=False, **kwargs):
    """Row-block sharded ``NeRF.render_image``: each rank renders rows ``shard_rows(...)`` of every
    frame.  No collective unless ``gather`` (then every rank gets the assembled frame; row blocks
    are padded to the largest block for ``all_gather``)."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    begin, end = shard_rows(image_h, rank, world_size)
    image, seg = model.render_image(camera_o, camera_r, image_h, image_w, focal_length, num_samples,
                                    row_begin=begin, row_end=end, **kwargs)
    if not gather:
        return image, seg, (begin, end)
    return (_gather_rows(image, image_h, world_size), _gather_rows(seg, image_h, world_size),
            (begin, end))


def _gather_rows(block, image_h, world_size):
    rows_max = -(-image_h // world_size)
    pad = torch.zeros(block.shape[0], rows_max, *block.shape[2:], dtype=block.dtype, device=block.device)
    pad[:, :block.shape[1]] = block
    parts = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(parts, pad)
    rows = [shard_rows(image_h, r, world_size) for r in range(world_size)]
    return torch.cat([p[:, :e - b] for p, (b, e) in zip(parts, rows)], dim=1)


class FlatGradientAllReduce:
    """Weighted sum of the gradients of ``params`` over the process group with ONE collective.

    ``weight`` is this rank's share of the global batch (local rays / global rays; 1 / world for
    equal shards): with a loss that is the MEAN over the rank's rays, sum_r weight_r * grad_r is
    the gradient of the mean over the global batch, also when the shards are uneven or empty.

    Fast path: the HIP backward already emits the flat 304,438-element vector in ``params`` order
    with every ``p.grad`` a view of it (``model.last_flat_grad``; true whenever the gradients were
    ``None`` before ``backward()``, i.e. ``zero_grad(set_to_none=True)``).  Pass it as ``flat`` and
    the collective runs in place on it: one scale + one all-reduce, no copies.  Otherwise the
    gradients are packed into a flat buffer and unpacked again (any module, CPU tests)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None
        self.in_place_calls = 0

    def _aliases(self, flat):
        """Every gradient is the view of ``flat`` at its offset (host-side pointer checks only)."""
        if flat is None or flat.numel() != self.numel or not flat.is_contiguous():
            return False
        off, base, item = 0, flat.data_ptr(), flat.element_size()
        for p in self.params:
            g = p.grad
            if g is None or g.dtype != flat.dtype or not g.is_contiguous() or g.data_ptr() != base + off * item:
                return False
            off += p.numel()
        return True

    def __call__(self, flat=None, weight=None):
        world = dist.get_world_size(self.group)
        weight = 1.0 / world if weight is None else float(weight)
        if self._aliases(flat):
            self.in_place_calls += 1
            # (also with ONE rank: the callers only come here when a process group exists, and a single-rank RCCL
            #  group is how the captured-collective path is exercised on a one-GPU box)
            if weight != 1.0:
                flat.mul_(weight)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return flat
        ref = self.params[0]
        if self._flat is None or self._flat.device != ref.device:
            self._flat = torch.empty(self.numel, dtype=torch.float32, device=ref.device)
        flat, off = self._flat, 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if weight != 1.0:
            flat.mul_(weight)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        return flat


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class DataParallelTrainer:
    """Data-parallel training step: local forward/backward on this rank's rays, one flat
    all-reduce, redundant optimiser step.  ``loss_fn(model, batch) -> scalar`` must be the MEAN
    over the rank's rays; ``weight`` (local rays / global rays) makes the reduced gradient the one
    of the mean over the global batch for uneven shards too (default: equal shards)."""

    def __init__(self, model, optimizer, loss_fn, group=None):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.reduce = FlatGradientAllReduce(model.parameters(), group)
        self.distributed = dist.is_available() and dist.is_initialized()

    def step(self, batch, weight=None):
        self.optimizer.zero_grad(set_to_none=True)       # p.grad become views of the flat gradient
        loss = self.loss_fn(self.model, batch)
        loss.backward()
        if self.distributed:
            self.reduce(getattr(self.model, "last_flat_grad", None), weight)
        self.optimizer.step()
        return loss.detach()

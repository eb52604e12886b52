"""Mirrors of ``nerf.dataset.PixelRayDataset`` (nerf/dataset.py:175-316), with an on-device batched
sampler, and of ``nerf.dataset.ImageRayDataset`` (nerf/dataset.py:6-172).

``dataset[idx]`` keeps the reference's per-example dictionary (same keys, shapes and the
``pose_d`` quirk) as plain torch indexing, so it still plugs into a ``DataLoader``.  The fast path
is ``dataset.gather(indices)`` / ``dataset.batches(batch_size)``: one HIP launch
(``nerf_hip_gather_pixel_rays``) decodes a whole batch of example ids, gathers pixels/labels and
builds world-space rays on the device — the reference's Python loop + ``default_collate`` tops out
at ~1.7e4 rays/s, the renderer consumes >4e5 rays/s in training.
"""
import ctypes

import torch
import torch.utils.data as data

from . import _lib
from .model import NeRF


class PixelRayDataset(data.Dataset):
    """One example per pixel of ``images`` [B,H,W,3] with ``poses`` [B,4,4] (camera-to-world) and a
    pinhole camera of ``focal_length`` pixels (nerf/dataset.py:194-230)."""

    def __init__(self, images, segmentation, poses, focal_length, states_x=None, states_d=None):
        self.images = images
        self.segmentation = segmentation
        self.poses = poses
        self.states_x = states_x
        self.states_d = states_d
        self.focal_length = focal_length
        self.rays = NeRF.generate_rays(images.shape[1], images.shape[2], focal_length,
                                       dtype=images.dtype, device=images.device)

    def __len__(self):
        return self.images.shape[0] * self.images.shape[1] * self.images.shape[2]

    def decode(self, idx):
        """example id -> (column, row, image)  (nerf/dataset.py:283-291)."""
        wi = idx % self.images.shape[2]
        idx = idx // self.images.shape[2]
        hi = idx % self.images.shape[1]
        idx = idx // self.images.shape[1]
        return wi, hi, idx % self.images.shape[0]

    def __getitem__(self, idx):
        """The reference's example dictionary (nerf/dataset.py:246-316); note ``pose_d`` is the
        world-space ray direction, as in the reference (:315)."""
        wi, hi, bi = self.decode(idx)
        pixel = self.images[bi, hi, wi]
        ray = self.rays[hi, wi]
        pose = self.poses[bi]
        dev = pixel.device
        rays_o, rays_d = NeRF.rays_to_world_coordinates(ray, pose[:3, 3], pose[:3, :3])
        return dict(image_wi=torch.tensor([wi], dtype=torch.int64, device=dev),
                    image_hi=torch.tensor([hi], dtype=torch.int64, device=dev),
                    image_bi=torch.tensor([bi], dtype=torch.int64, device=dev),
                    states_x=(self.states_x[bi] if self.states_x is not None else torch.zeros(0, device=dev)),
                    states_d=(self.states_d[bi] if self.states_d is not None else torch.zeros(0, device=dev)),
                    pixels=pixel, label=self.segmentation[bi, hi, wi], rays=ray,
                    pose_o=pose[:3, 3], pose_d=rays_d, rays_o=rays_o, rays_d=rays_d)

    # ---- on-device batched path -------------------------------------------------------------------

    def gather(self, index):
        """Collated batch for example ids ``index`` [n] (int64, on the images' ROCm device): the
        dictionary ``default_collate`` would build from ``[self[i] for i in index]``."""
        images = self.images
        if not images.is_cuda:
            raise RuntimeError("PixelRayDataset.gather runs on the GPU: move images/poses to the device")
        dev, n = images.device, int(index.shape[0])
        index = index.to(device=dev, dtype=torch.int64).contiguous()
        images = images.contiguous()
        poses = self.poses.to(device=dev, dtype=torch.float32).contiguous()
        seg = self.segmentation
        seg = None if seg is None else seg.to(device=dev, dtype=torch.int64).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        out = dict(pixels=torch.empty(n, 3, **f32), rays=torch.empty(n, 3, **f32),
                   rays_o=torch.empty(n, 3, **f32), rays_d=torch.empty(n, 3, **f32),
                   image_wi=torch.empty(n, 1, **i64), image_hi=torch.empty(n, 1, **i64),
                   image_bi=torch.empty(n, 1, **i64))
        if seg is not None:
            out["label"] = torch.empty(n, **i64)
        args = _lib.GatherArgs()
        args.index, args.n = _lib.ptr(index), n
        args.images, args.segmentation, args.poses = _lib.ptr(images), _lib.ptr(seg), _lib.ptr(poses)
        args.batch, args.image_h, args.image_w = images.shape[0], images.shape[1], images.shape[2]
        args.focal_length = float(self.focal_length)
        for key in ("pixels", "rays", "rays_o", "rays_d", "image_wi", "image_hi", "image_bi"):
            setattr(args, key, _lib.ptr(out[key]))
        args.label = _lib.ptr(out.get("label"))
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib().nerf_hip_gather_pixel_rays(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_gather_pixel_rays")
        out["pose_o"], out["pose_d"] = out["rays_o"], out["rays_d"]       # dataset.py:315 quirk
        bi = out["image_bi"][:, 0]
        out["states_x"] = (self.states_x[bi] if self.states_x is not None else torch.zeros(n, 0, device=dev))
        out["states_d"] = (self.states_d[bi] if self.states_d is not None else torch.zeros(n, 0, device=dev))
        return out

    def batches(self, batch_size, shuffle=True, generator=None, drop_last=False, rank=0, world_size=1):
        """One epoch of collated batches: a uniform permutation without replacement (what
        ``DataLoader(shuffle=True)`` draws), cut into batches; with ``world_size`` > 1 every rank
        takes its contiguous share of each global batch (same permutation on all ranks when the
        generators are seeded alike).  Shares differ by at most one example (``parallel.shard_items``;
        the last batch of an epoch may leave a rank with fewer, even zero, examples) and every batch
        carries ``global_n``, the size of the global batch, so that a data-parallel step can weight
        its gradient by ``local / global`` (``parallel.FlatGradientAllReduce``)."""
        from .parallel import shard_items
        dev, total = self.images.device, len(self)
        order = (torch.randperm(total, device=dev, generator=generator) if shuffle
                 else torch.arange(total, device=dev))
        for lo in range(0, total, batch_size):
            idx = order[lo:lo + batch_size]
            if drop_last and idx.shape[0] < batch_size:
                return
            global_n = int(idx.shape[0])
            if world_size > 1:
                begin, end = shard_items(global_n, rank, world_size)
                idx = idx[begin:end]
            batch = self.gather(idx)
            batch["global_n"] = global_n
            yield batch


class ImageRayDataset(data.Dataset):
    """Block-stratified ray batches (nerf/dataset.py:6-172): every example is ``num_samples_per_block`` rays from
    EACH cell of a ``num_vertical_blocks`` x ``num_horizontal_blocks`` grid over one image — [samples, blocks]
    rays with their pixels, poses and states.  No script or notebook of the reference uses it; it is mirrored so
    that a caller of the reference's data module finds both classes.  Same constructor, length, keys, shapes and
    dtypes; the in-cell picks are ONE ``torch.multinomial`` over uniform weights on the CPU generator, exactly
    where the reference draws them (:134-137), so a seeded run picks the same rays (fixture G10).  Kept quirk: the
    image is ``idx // cell_area`` while the length is ``images * cell_area // samples`` (:96-97, :156-158), so only
    the first ``images / samples`` images are ever visited."""

    def __init__(self, images, poses, states, focal_length, num_vertical_blocks=8, num_horizontal_blocks=8,
                 num_samples_per_block=2):
        self.images, self.poses, self.states = images, poses, states
        self.num_vertical_blocks = num_vertical_blocks
        self.num_horizontal_blocks = num_horizontal_blocks
        self.num_samples_per_block = num_samples_per_block
        height, width = images.shape[1], images.shape[2]
        self.rays = NeRF.generate_rays(height, width, focal_length, dtype=images.dtype, device=images.device)
        self.vertical_block_size = height // num_vertical_blocks
        self.horizontal_block_size = width // num_horizontal_blocks
        self.total_block_area = self.vertical_block_size * self.horizontal_block_size
        # top-left pixel of every grid cell, cells in row-major order: [1, cells]
        cell = torch.arange(num_vertical_blocks * num_horizontal_blocks, device=images.device).unsqueeze(0)
        self._cell_row0 = (cell // num_horizontal_blocks) * self.vertical_block_size
        self._cell_col0 = (cell % num_horizontal_blocks) * self.horizontal_block_size

    def __len__(self):
        return (self.images.shape[0] * self.total_block_area) // self.num_samples_per_block

    def __getitem__(self, idx):
        dev = self.images.device
        cells = self.num_vertical_blocks * self.num_horizontal_blocks
        # position inside the cell, one per (sample, cell): uniform with replacement (the reference's draw)
        pick = torch.multinomial(torch.ones(self.num_samples_per_block, self.total_block_area), cells,
                                 replacement=True).to(dev)
        image_hi = pick // self.horizontal_block_size + self._cell_row0
        image_wi = pick % self.horizontal_block_size + self._cell_col0
        image_bi = torch.full(image_hi.shape, idx // self.total_block_area, dtype=torch.int64, device=dev)
        pose = self.poses[image_bi]
        rays = self.rays[image_hi, image_wi]
        pose_o, pose_d = pose[..., :3, 3], pose[..., :3, :3]
        rays_o, rays_d = NeRF.rays_to_world_coordinates(rays, pose_o, pose_d)
        return dict(image_bi=image_bi, image_hi=image_hi, image_wi=image_wi,
                    pixels=self.images[image_bi, image_hi, image_wi], states=self.states[image_bi], rays=rays,
                    pose_o=pose_o, pose_d=pose_d, rays_o=rays_o, rays_d=rays_d)

"""Training / evaluation loop of the reference script (train_conditional_nerf.py:106-174) on the
generation-C call surface, single GPU or data-parallel (one process per GPU, RCCL all-reduce of
the flat gradient).  Same recipe: Adam, MSE of ``render_rays`` pixels against the sampled pixels
(``.unsqueeze(1)`` target, :132), last view held out (:89-95), evaluation by ``render_image``
every ``log_interval`` iterations with PSNR = -10 ln(mse) / 2.30258509299 (:152-153), and the
same files: ``params.json``, ``model.pth``, ``psnrs.npy``, ``iternums.npy``,
``rendered_images.npy``, ``ground_truth_images.npy`` (:53-69, :160-174).
"""
import json
import os

import numpy as np
import torch
import torch.distributed as dist

from .dataset import PixelRayDataset
from .loss import mse_and_grad
from .model import NeRF
from . import parallel


def psnr(rendered, target):
    """train_conditional_nerf.py:152-153 (natural log over a hard-coded ln 10)."""
    return -10.0 * torch.log(((rendered - target) ** 2).mean()) / 2.30258509299


def synthetic_scene(num_views=12, size=32, focal=None, num_samples=48, device="cuda", seed=0,
                    weight_scale=3.0, radius=4.0):
    """A self-consistent scene when no dataset is on disk (tiny_nerf_data.npz is not shipped):
    views of a 'teacher' field (default init, Linear weights x ``weight_scale``) rendered by the
    renderer itself from poses on a sphere.  Returns images [V,H,W,3], poses [V,4,4], focal."""
    focal = float(focal if focal is not None else size * 1.12)
    gen = torch.Generator().manual_seed(seed)
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    teacher = NeRF(focal_length=focal)
    torch.set_rng_state(state)
    with torch.no_grad():
        for slot in (0, 3, 6, 9, 12, 15):
            teacher.prediction_heads[slot].weight.mul_(weight_scale)
    teacher = teacher.to(device)
    yaw = torch.rand(num_views, generator=gen) * 2 * np.pi
    elev = 0.3 + 0.5 * torch.rand(num_views, generator=gen)
    pos = NeRF.spherical_to_cartesian(yaw, elev) * radius
    eye = -pos / pos.norm(dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]]).expand_as(eye)
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / up.norm(dim=-1, keepdim=True)
    rot = torch.stack([torch.linalg.cross(eye, up, dim=-1), up, -eye], dim=-1)
    poses = torch.eye(4).repeat(num_views, 1, 1)
    poses[:, :3, :3], poses[:, :3, 3] = rot, pos
    poses = poses.to(device)
    with torch.no_grad():
        images, _ = teacher.render_image(poses[:, :3, 3].contiguous(), poses[:, :3, :3].contiguous(),
                                         size, size, focal, num_samples)
    return images, poses, focal


def load_scene(path, device):
    """``.npz`` in the tiny_nerf layout (images [V,H,W,3], poses [V,4,4], focal)."""
    with np.load(path) as z:
        images = torch.from_numpy(z["images"].astype(np.float32)).to(device)
        poses = torch.from_numpy(z["poses"].astype(np.float32)).to(device)
        focal = float(z["focal"])
    return images, poses, focal


def load_pickled_scene(path, device, camera_focal_length, camera_ccd_width):
    """The reference script's data file (train_conditional_nerf.py:70-87): a pickled dict with
    ``images`` [V,H,W,3], ``poses`` [V,6] (camera position | viewing direction) and ``states``.
    Focal length in pixels = W * focal_mm / ccd_mm (:78-80); a pose becomes [R | t] with R from the
    viewing direction (:86 — ``direction_to_rotation_matrix`` no longer exists in nerf/model.py; the
    stated choice of nerf_amd.compat is used).  ``states`` are read and dropped: the generation-C
    network ignores them (nerf/model.py:596-598).  Only unpickle files you trust."""
    import pickle
    from .compat import LegacyNeRF
    with open(path, "rb") as f:
        blob = pickle.load(f)
    images = torch.as_tensor(np.asarray(blob["images"]), dtype=torch.float32)
    raw = torch.as_tensor(np.asarray(blob["poses"]), dtype=torch.float32)
    focal = float(images.shape[2]) * (float(camera_focal_length) / float(camera_ccd_width))
    poses = torch.eye(4).repeat(raw.shape[0], 1, 1)
    poses[:, :3, :3] = LegacyNeRF.direction_to_rotation_matrix(raw[:, 3:])
    poses[:, :3, 3] = raw[:, :3]
    return images.to(device), poses.to(device), focal


class Trainer:
    """``model``: a generation-C ``NeRF`` (default) or a ``LegacyNeRF8x256`` — the network of the notebook
    and of examples/nerf.pth, trained the notebook's way (examples/example.ipynb cell 8): explicit
    ``near`` / ``far`` planes, ``render_rays`` returning [N, 3], the same MSE / Adam / PSNR recipe."""

    def __init__(self, images, poses, focal_length, logging_dir=None, batch_size=1024,
                 learning_rate=1e-4, num_samples_per_ray=64, density_noise_std=1.0, log_interval=1000,
                 segmentation=None, model=None, seed=0, rng="torch", graph=False, near=2.0, far=6.0):
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        device = images.device
        self.image_h, self.image_w = images.shape[1], images.shape[2]
        self.focal_length = focal_length
        # hold out the last view (train_conditional_nerf.py:89-95)
        self.test_image, self.test_pose = images[-1:], poses[-1:]
        if segmentation is None:
            segmentation = torch.zeros(images.shape[:3], dtype=torch.int64, device=device)
        self.dataset = PixelRayDataset(images[:-1], segmentation[:-1], poses[:-1], focal_length)
        if model is None:
            torch.manual_seed(seed)
            model = NeRF(focal_length=focal_length).to(device)
        self.model = model
        from .legacy import LegacyNeRF8x256
        self.legacy = isinstance(model, LegacyNeRF8x256)
        self.near, self.far = float(near), float(far)
        if self.legacy and rng != "torch":
            raise ValueError("the legacy network draws with torch's generator only (rng='torch')")
        self.model.rng = rng
        # draws: with rng="torch" the stratified / noise draws come from torch's generator, so give
        # every rank its own stream (seed + rank); the in-kernel Philox path folds the rank into its
        # key by itself (NeRF._next_philox_state).  The example order (self.sampler) stays common.
        self.draws = torch.Generator(device=device).manual_seed(seed + 7919 * (self.rank + 1))
        if self.distributed:
            parallel.broadcast_parameters(self.model)
        # graph: replay forward + loss + backward (+ the optimiser step when single-process) as ONE HIP
        # graph per iteration.  At 512 rays per GPU (BASELINE config 5 on 8 GPUs) a step is launch-bound:
        # ~27 kernel launches and the autograd hop cost more than the 0.33 ms the kernels run
        # (0.64 -> 0.46 ms per step, scripts/graph_step.py).
        on_gpu = next(self.model.parameters()).is_cuda
        self.use_graph = bool(graph) and on_gpu
        self._graph = None
        self._graph_rays = -1
        self._eager_steps = 0
        self._stale_grads = False
        begin, end = parallel.shard_items(batch_size, self.rank, self.world)
        self._full_share = end - begin                    # rays of a full global batch that land on this rank
        if self.use_graph and rng == "torch":
            torch.cuda.manual_seed(seed + 7919 * (self.rank + 1))      # draws come from the default generator
        # (rng="philox" under graph replay: the launch's Philox offset has a device-resident part that a captured
        #  one-thread launch advances, include/nerf_hip.h: rng_counter — every replay draws new samples)
        # With RCCL (backend "nccl") the collective is a stream-ordered kernel like any other and is captured
        # with the step: scale, all-reduce, Adam all replay from the one graph.  Over gloo (CPU rendezvous: tests,
        # rehearsals) it is a host-side operation and stays outside the captured region, as does the optimiser.
        self.collective_in_graph = bool(self.distributed and self.use_graph and "nccl" in str(dist.get_backend()))
        # the reference's optimiser (train_conditional_nerf.py:106-107: Adam, default betas / eps) as one launch
        # over all parameter tensors (nerf_amd/optim.py; graph-capturable by construction: its step count lives on
        # the device); torch's own on the CPU
        if on_gpu:
            from .optim import Adam
            self.optimizer = Adam(self.model.parameters(), lr=learning_rate)
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=learning_rate)
        self.reduce = parallel.FlatGradientAllReduce(self.model.parameters())
        self.batch_size = batch_size
        self.num_samples = num_samples_per_ray
        self.density_noise_std = density_noise_std
        self.log_interval = log_interval
        self.logging_dir = logging_dir
        self.sampler = torch.Generator(device=device).manual_seed(seed)     # same order on all ranks
        self.psnrs, self.iternums, self.rendered, self.truth = [], [], [], []
        self.iteration = -1
        self._side = None
        if logging_dir is not None and self.rank == 0:
            os.makedirs(logging_dir, exist_ok=True)

    def write_params(self, params):
        if self.logging_dir is not None and self.rank == 0:
            with open(os.path.join(self.logging_dir, "params.json"), "w") as f:
                json.dump(params, f, indent=4)

    def _draw(self, n, device, generator):
        """The reference's draws in the reference's order (rand, then randn: nerf/model.py:432, :652); the
        legacy network evaluates S points per ray, generation C S - 1 intervals."""
        u = torch.rand(n, self.num_samples, dtype=torch.float32, device=device, generator=generator)
        shape = (n, self.num_samples) if self.legacy else (n, self.num_samples - 1, 1)
        return u, torch.randn(*shape, dtype=torch.float32, device=device, generator=generator)

    def _render(self, rays_o, rays_d, u, noise):
        """Pixels [n, 3] of a training batch (``u`` / ``noise`` None: drawn by the renderer itself)."""
        if noise is not None and self.density_noise_std == 0.0:
            noise = None
        if self.legacy:
            return self.model.render_rays(rays_o, rays_d, self.near, self.far, self.num_samples,
                                          randomly_sample=True, density_noise_std=self.density_noise_std,
                                          u=u, noise=noise)
        pixels, _ = self.model.render_rays(rays_o, rays_d, self.num_samples, randomly_sample=True,
                                           density_noise_std=self.density_noise_std, u=u, noise=noise)
        return pixels.view(pixels.shape[0], -1)           # the single stage (train_conditional_nerf.py:132); a view:
                                                          # its backward launches nothing (a select would: zeros + copy)

    @staticmethod
    def _loss_backward(pixels, target):
        """The loop's ``loss = MSE; loss.backward()`` (train_conditional_nerf.py:132-133); returns the loss.
        On the GPU the loss and its gradient come from one launch (nerf_amd.loss) and the backward starts at
        ``pixels`` with that gradient; an empty shard (tail of an epoch) gives loss 0, not NaN, either way."""
        if pixels.is_cuda:
            loss, grad = mse_and_grad(pixels, target)
            pixels.backward(grad)
            return loss
        loss = ((pixels - target) ** 2).sum() / max(pixels.numel(), 1)
        loss.backward()
        return loss.detach()

    def _global_n(self, batch, n):
        """Rays of the GLOBAL batch this step belongs to: ``batch["global_n"]`` when the producer says (dataset.batches
        always does); else — an external producer — the configured batch size when this rank holds its full share of
        one (uneven shards: ``n * world`` would never equal it and every step would silently run eagerly),
        and ``n * world`` only for a short batch, where nothing better is known."""
        if "global_n" in batch:
            return int(batch["global_n"])
        if not self.distributed:
            return n
        return self.batch_size if n == self._full_share else n * self.world

    # ---- HIP-graph path ---------------------------------------------------------------------------
    def _graph_body(self, o, d, pix, u=None, noise=None):
        n, dev = o.shape[0], o.device
        if u is None and self.model.rng == "torch":
            u, noise = self._draw(n, dev, None)           # graph-safe default generator
        self.last_draws = (u, noise)                      # (static tensors of the graph once captured; None: Philox)
        pixels = self._render(o, d, u, noise)
        loss = self._loss_backward(pixels, pix)
        if self.collective_in_graph:                      # full-size batches only: the share is a constant
            self.reduce(self.model.last_flat_grad, n / float(self.batch_size))
        if not self.distributed or self.collective_in_graph:
            self.optimizer.step()
        return loss

    def _graph_step(self, batch):
        """Steps 0-2 run eagerly (lazy initialisation, optimiser state), steps 3-4 eagerly on the side
        stream the capture will use (autograd's accumulation nodes must live there), step 5 is captured
        and from then on every full-size batch is a copy into the static inputs + one replay."""
        n = batch["rays_o"].shape[0]
        # Only this rank's FULL-batch share is warmed up, captured and replayed: a short batch (the
        # tail of an epoch, an uneven shard) runs eagerly and consumes no warm-up step, so the capture
        # can never freeze a tail size and then refuse every full batch for the rest of the run.
        # The GLOBAL batch must be full as well: with the collective inside the graph the all-reduce weight is
        # baked in as share / batch_size; a short global batch can still hand SOME ranks their full share, and
        # those would replay with that weight while the others go eager with n / global_n — weights that no
        # longer sum to 1.  `global_n` is the same number on every rank, so all ranks take the same path.
        global_n = self._global_n(batch, n)
        # a batch may bring its own draws (batch["u"] [n, S], batch["noise"]: a parity study replaying captured draws,
        # tests/psnr_parity.py); they become static inputs of the graph, so every batch of the run must bring them
        keys = ("rays_o", "rays_d", "pixels") + (("u", "noise") if "u" in batch else ())
        if n != self._full_share or (self.distributed and global_n != self.batch_size) or \
                (self._graph is not None and keys != tuple(self._static)):
            self._stale_grads = True                      # the eager step re-points p.grad
            return None
        if self._graph is None and self._eager_steps < 5:
            self._eager_steps += 1
            if self._eager_steps <= 3:
                return None
            if self._side is None:
                self._side = torch.cuda.Stream(device=batch["rays_o"].device)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self.optimizer.zero_grad(set_to_none=True)
                loss = self._graph_body(batch["rays_o"], batch["rays_d"], batch["pixels"], batch.get("u"), batch.get("noise"))
                if self.distributed and not self.collective_in_graph:
                    self.reduce(self.model.last_flat_grad, n / max(self._global_n(batch, n), 1))
                    self.optimizer.step()
            torch.cuda.current_stream().wait_stream(self._side)
            return loss
        if self._graph is None:
            self._static = {k: batch[k].clone() for k in keys}
            self.optimizer.zero_grad(set_to_none=True)    # the gradients are created inside the graph's pool
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=self._side):
                self._static_loss = self._graph_body(self._static["rays_o"], self._static["rays_d"],
                                                     self._static["pixels"], self._static.get("u"),
                                                     self._static.get("noise"))
            self._graph_rays = n
            self._static_grads = [p.grad for p in self.model.parameters()]
            self._static_flat = self.model.last_flat_grad
        for k, t in self._static.items():
            t.copy_(batch[k])
        self._graph.replay()
        if self._stale_grads:                             # an eager tail step replaced them: p.grad must
            for p, g in zip(self.model.parameters(), self._static_grads):     # show what the replay wrote
                p.grad = g
            self.model.last_flat_grad = self._static_flat
            self._stale_grads = False
        if self.distributed and not self.collective_in_graph:
            self.reduce(self._static_flat, n / max(self._global_n(batch, n), 1))
            self.optimizer.step()
        if getattr(self.model, "train_precision", "fp32") == "f16x3" and self.iteration % 64 == 0:
            self.model.check_split_precision_range()      # the replay runs no host code
        return self._static_loss

    def train_step(self, batch):
        if self.use_graph:
            loss = self._graph_step(batch)
            if loss is not None:
                return loss
        n = batch["rays_o"].shape[0]
        u, noise = batch.get("u"), batch.get("noise")     # the caller's own draws, if it brings any
        if u is None and self.model.rng == "torch":
            # the reference's draws in the reference's order (rand, then randn: nerf/model.py:432, :652):
            # from this rank's own generator when data-parallel, else from torch's default one
            u, noise = self._draw(n, batch["rays_o"].device, self.draws if self.distributed else None)
        self.last_draws = (u, noise)                      # what this step rendered with (None: in-kernel Philox)
        pixels = self._render(batch["rays_o"], batch["rays_d"], u, noise)
        self.optimizer.zero_grad(set_to_none=True)       # p.grad become views of the flat gradient
        loss = self._loss_backward(pixels, batch["pixels"])
        if self.distributed:
            self.reduce(self.model.last_flat_grad, n / max(self._global_n(batch, n), 1))
        self.optimizer.step()
        return loss.detach()

    def evaluate(self):
        cam_o, cam_r = self.test_pose[..., :3, 3].contiguous(), self.test_pose[..., :3, :3].contiguous()
        # graph replays update the parameters without running host code (no version bump): the evaluation
        # render must not trust a range check cached before them
        self.model._forget_range_check()
        with torch.no_grad():
            if self.legacy:                               # the notebook's call (cell 8)
                render = self.model.render_image(cam_o, cam_r, self.image_h, self.image_w, self.focal_length,
                                                 self.near, self.far, self.num_samples)
            else:
                render, _ = self.model.render_image(cam_o, cam_r, self.image_h, self.image_w,
                                                    self.focal_length, self.num_samples)
        value = psnr(render, self.test_image)
        self.psnrs.append(value.cpu().numpy())
        self.iternums.append(self.iteration)
        self.rendered.append(render.cpu().numpy())
        self.truth.append(self.test_image.cpu().numpy())
        if self.logging_dir is not None and self.rank == 0:
            d = self.logging_dir
            torch.save(self.model.state_dict(), os.path.join(d, "model.pth"))
            np.save(os.path.join(d, "psnrs.npy"), np.asarray(self.psnrs))
            np.save(os.path.join(d, "iternums.npy"), np.asarray(self.iternums))
            np.save(os.path.join(d, "rendered_images.npy"), np.asarray(self.rendered))
            np.save(os.path.join(d, "ground_truth_images.npy"), np.asarray(self.truth))
        return float(value)

    def fit(self, epochs=1, max_iterations=None):
        last = None
        for _ in range(epochs):
            for batch in self.dataset.batches(self.batch_size, shuffle=True, generator=self.sampler,
                                              rank=self.rank, world_size=self.world):
                self.iteration += 1
                last = self.train_step(batch)
                if self.iteration % self.log_interval == 0:
                    self.evaluate()
                if max_iterations is not None and self.iteration + 1 >= max_iterations:
                    return last
        return last

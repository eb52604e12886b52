"""The LEGACY (generation-A) network of ``examples/nerf.pth`` on the MI355X renderer.

The reference ships trained Lego weights for a vanilla-NeRF-style network (sin/cos positional
encoding, 8 x 256 trunk with a skip-concatenation, density head, 2 x 256 view branch, color head)
whose SOURCE is no longer in the repository (SURVEY.md sections 0.4, 2.3).  This module gives that
checkpoint a home: the module tree is the one its 44 tensor names describe, so
``load_state_dict(torch.load("examples/nerf.pth"))`` works, and the call surface is the notebook's
(examples/example.ipynb cells 6, 8):

    NeRF(normalize_position=6.0)
    render_rays(rays_o, rays_d, near, far, num_samples, randomly_sample, density_noise_std) -> [N, 3]
    render_image(camera_o, camera_r, H, W, focal, near, far, num_samples)                  -> [B, H, W, 3]

PARITY UNPINNED.  What the checkpoint cannot settle — activation (ReLU), frequency multiplier (pi),
concatenation order ([hidden, encoding]), normalised view directions — are stated choices that
``oracle/legacy_oracle.py`` restates on the CPU and SURVEY.md's probe found to render the Lego
scene; they are constructor keywords.  Everything render_rays computes is ONE fused HIP launch
(``nerf_hip_legacy_render_forward``, nerf_amd/csrc/nerf_legacy.hip).

Training (the notebook's loop, cell 8: ``((pixels - batch['pixels']) ** 2).mean().backward()``, Adam):
with gradients enabled ``render_rays`` goes through ``LegacyRenderRaysFunction``, whose backward is
``nerf_hip_legacy_render_backward`` (nerf_amd/csrc/nerf_legacy_backward.hip): the 44 parameter gradients as
views of ONE flat vector in ``parameters()`` order (``model.last_flat_grad``, what the data-parallel
all-reduce runs on in place).  ``train_precision`` selects the training arithmetic like
``nerf_amd.model.NeRF.train_precision``: "fp32" = fp32 MFMA (forward, data gradient) and bf16 triples (weight
gradient), exact-fp32 products throughout; "f16x3" = f16 pairs in all three (three f16 MFMAs per product, fp32
accumulation, power-of-two scales of dY per sample / per batch), held to the same gradient tests.
``precision`` selects the arithmetic of inference launches.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from .model import NeRF as _GenerationC, _require_device

__all__ = ["LegacyNeRF8x256"]

FLOP_PER_SAMPLE = 1261568          # 2 x (60 + 3*256 + 316 + 3*256 + 1 + 292 + 256 + 3) x 256: SURVEY.md section 2.3


def _block(first_inputs, layers):
    mods = []
    for i in range(layers):
        mods += [nn.Linear(first_inputs if i == 0 else 256, 256), nn.ReLU(), nn.LayerNorm(256)]
    return nn.Sequential(*mods)


class LegacyRenderRaysFunction(torch.autograd.Function):
    """rgb [N,3] = f(parameters): forward = the training instantiation of the fused kernel (it also saves
    the encodings, the LayerNorm statistics and a_hat of every wide layer, the head outputs), backward = four
    HIP launches that leave the flat gradient (nerf_hip_legacy_render_backward).  Inputs after
    ``density_noise_std`` are the 44 parameters in the kernels' order.  Rays and draws take no gradient."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, near, far, num_samples, u, noise, density_noise_std, *params):
        lib = _lib.lib()
        n_rays, device = rays_o.shape[0], rays_o.device
        ws_bytes = lib.nerf_hip_legacy_train_workspace_bytes(n_rays, num_samples)
        workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=device)
        rgb, _, _ = model._launch(n_rays, num_samples, device, near, far, rays_o=rays_o, rays_d=rays_d, u=u,
                                  noise=noise, density_noise_std=density_noise_std, train_workspace=workspace)
        if getattr(model, "keep_workspace", False):      # debugging / stage-parity tests only
            model.last_workspace = workspace
        ctx.model = model
        ctx.call = (rays_o, rays_d, near, far, num_samples, u, noise, density_noise_std)
        ctx.workspace = workspace
        ctx.packed = model._last_packed               # the image this forward used (its own buffer)
        ctx.precision = model.train_precision         # the backward runs in the forward's arithmetic
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(rgb)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        lib = _lib.lib()
        model = ctx.model
        (rgb,) = ctx.saved_tensors
        rays_o, rays_d, near, far, num_samples, u, noise, std = ctx.call
        n_rays, device = rays_o.shape[0], rays_o.device
        d_rgb = d_rgb.contiguous()
        args = _lib.LegacyBackwardArgs()
        model._fill_args(args.fwd, n_rays, num_samples, device, near, far, rays_o=rays_o, rays_d=rays_d, u=u,
                         noise=noise, density_noise_std=std, packed=ctx.packed, rgb=rgb,
                         train_workspace=ctx.workspace, precision=ctx.precision)
        grad = torch.empty(lib.nerf_hip_legacy_grad_elements(), dtype=torch.float32, device=device)
        scratch = model._scratch(lib.nerf_hip_legacy_backward_scratch_bytes(n_rays, num_samples), device)
        args.d_rgb, args.grad, args.scratch = _lib.ptr(d_rgb), _lib.ptr(grad), _lib.ptr(scratch)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nerf_hip_legacy_render_backward(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_legacy_render_backward")
        ctx.workspace = None
        grads, off = [], 0
        for shape in ctx.shapes:                  # views of the flat vector, parameters() order
            n = 1
            for d in shape:
                n *= d
            grads.append(grad[off:off + n].view(shape))
            off += n
        model.last_flat_grad = grad
        return (None,) * 9 + tuple(grads)


class LegacyNeRF8x256(nn.Module):
    def __init__(self, normalize_position=6.0, multiplier=math.pi, normalize_directions=True):
        super().__init__()
        self.normalize_position = float(normalize_position)
        self.multiplier = float(multiplier)
        self.normalize_directions = bool(normalize_directions)
        # the checkpoint's tree: block_0.{0,3,6,9} Linear, .{2,5,8,11} LayerNorm, slots 1,4,7,10 parameter-less.
        # Registered in the kernels' tensor order (nerf_legacy_layout.h) so that parameters() IS that order:
        # the flat gradient the backward writes then aliases every p.grad in optimiser / all-reduce order.
        # (Creation order below keeps the seeded default initialisation of earlier rounds: density, color first.)
        density, color = nn.Linear(256, 1), nn.Linear(256, 3)
        block_0, block_1, block_2 = _block(60, 4), _block(256 + 60, 4), _block(256 + 36, 2)
        self.block_0, self.block_1, self.density, self.block_2, self.color = block_0, block_1, density, block_2, color
        self._packed = None
        self._packed_key = None
        self._tables = {}
        # arithmetic of the twelve matrix products: "fp32" = exact-fp32 MFMA, "f16x3" = every operand
        # as an f16 pair, three f16 MFMAs per product, fp32 accumulation (as nerf_amd.model.NeRF.precision)
        self.precision = "fp32"
        # arithmetic of launches that record a backward (forward, data gradient, weight gradient)
        self.train_precision = "fp32"

    # pose helpers of the reference's class, unchanged (nerf/model.py:243-367)
    generate_rays = staticmethod(_GenerationC.generate_rays)
    rays_to_world_coordinates = staticmethod(_GenerationC.rays_to_world_coordinates)
    get_rotation_matrix = staticmethod(_GenerationC.get_rotation_matrix)
    spherical_to_cartesian = staticmethod(_GenerationC.spherical_to_cartesian)

    def _param_list(self):
        """The 44 tensors in the order nerf_hip_legacy_pack_weights takes them."""
        order = []
        for block in (self.block_0, self.block_1):
            for slot in (0, 3, 6, 9):
                order += [block[slot].weight, block[slot].bias, block[slot + 2].weight, block[slot + 2].bias]
        order += [self.density.weight, self.density.bias]
        for slot in (0, 3):
            order += [self.block_2[slot].weight, self.block_2[slot].bias,
                      self.block_2[slot + 2].weight, self.block_2[slot + 2].bias]
        return order + [self.color.weight, self.color.bias]

    def packed_parameters(self, fresh=False):
        """Re-packed on every call (one small launch), like nerf_amd.model.NeRF.packed_parameters:
        version counters miss fused-optimiser and ``p.data`` updates.  ``fresh``: into a buffer of its
        own (a training forward: its backward reads the image later)."""
        params = self._param_list()
        dev = params[0].device
        for p in params:
            _require_device(p, "parameter")
        lib = _lib.lib()
        keep = [p.detach().contiguous() for p in params]
        ptrs = (ctypes.c_void_p * _lib.NUM_LEGACY_PARAM_TENSORS)(*[p.data_ptr() for p in keep])
        packed = self._packed
        if fresh or packed is None or packed.device != dev:
            packed = torch.empty(lib.nerf_hip_legacy_packed_bytes() // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.nerf_hip_legacy_pack_weights(ptrs, _lib.ptr(packed), ctypes.c_void_p(stream)),
                       "nerf_hip_legacy_pack_weights")
        if not fresh:
            self._packed = packed
        self._last_packed = packed
        return packed

    def _check_f16x3_range(self, training=False):
        """The split-precision kernel holds 2^8 * w and 2^4 * (gamma * x_hat + beta) as f16 pairs
        (|x_hat| < 16 for 256 features): refuse parameters outside that range instead of saturating.
        The check is a device -> host copy, so — the rule of ``nerf_amd.model.NeRF`` — it runs when a parameter
        tensor's version or storage changed and otherwise on every 64th split-precision launch; a training
        forward, whose versions change with every optimiser step, only uses the every-64th rule; never
        inside a HIP-graph capture (the trainer re-checks between replays)."""
        if torch.cuda.is_current_stream_capturing():
            return
        key = tuple((p._version, p.data_ptr()) for p in self.parameters())
        self._f16x3_calls = getattr(self, "_f16x3_calls", -1) + 1
        periodic = self._f16x3_calls % 64 == 0
        if not periodic and (training or getattr(self, "_f16x3_checked", None) == key):
            return
        with torch.no_grad():
            linears = [m for m in self.modules() if isinstance(m, nn.Linear)]
            norms = [m for m in self.modules() if isinstance(m, nn.LayerNorm)]
            w_dev = torch.stack([m.weight.abs().max() for m in linears]).max()
            a_dev = torch.stack([16.0 * m.weight.abs().max() + m.bias.abs().max() for m in norms]).max()
            w_max, act_max = (float(v) for v in torch.stack([w_dev, a_dev]).cpu())
        if w_max * 256.0 >= 65504.0 or act_max * 16.0 >= 65504.0:
            raise ValueError(f"nerf_amd: parameters out of range for precision='f16x3' (max |w| {w_max:.3g}, "
                             f"max 16|gamma|+|beta| {act_max:.3g}); use precision='fp32'")
        self._f16x3_checked = key

    def _forget_range_check(self):
        self._f16x3_checked = None
        self._f16x3_calls = -1

    def load_state_dict(self, *args, **kwargs):
        """As nn.Module.load_state_dict; the next split-precision launch re-checks the f16 range."""
        out = super().load_state_dict(*args, **kwargs)
        self._forget_range_check()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._forget_range_check()
        return out

    def check_split_precision_range(self):
        """Force the f16 range check of the parameters now (raises ValueError when they left it)."""
        self._forget_range_check()
        self._check_f16x3_range(training=False)

    def _table(self, near, far, num_samples, device):
        """Linear sample positions in [near, far] (torch.linspace on the CPU, cached on the device)."""
        key = (float(near), float(far), int(num_samples), str(device))
        if key not in self._tables:
            self._tables[key] = torch.linspace(float(near), float(far), num_samples,
                                               dtype=torch.float32).to(device)
        return self._tables[key]

    def _fill_args(self, args, n_rays, num_samples, device, near, far, *, rays_o=None, rays_d=None, cameras=None,
                   ray_begin=0, u=None, noise=None, density_noise_std=0.0, packed=None, rgb=None, raw=None,
                   weights=None, train_workspace=None, precision=None):
        """Fill a NerfHipLegacyArgs block (include/nerf_hip.h) from tensors."""
        table = self._table(near, far, num_samples, device)
        r = args.render
        r.rays_o, r.rays_d = _lib.ptr(rays_o), _lib.ptr(rays_d)
        if cameras is not None:
            cam_o, cam_r, image_h, image_w, focal = cameras
            r.camera_o, r.camera_r = _lib.ptr(cam_o), _lib.ptr(cam_r)
            r.image_h, r.image_w, r.focal_length = image_h, image_w, float(focal)
        r.ray_begin, r.n_rays, r.num_samples = int(ray_begin), int(n_rays), int(num_samples)
        r.t_table, r.t_scale = _lib.ptr(table), 1.0
        r.u, r.noise = _lib.ptr(u), _lib.ptr(noise)
        r.density_noise_std = float(density_noise_std)
        r.packed, r.rgb = _lib.ptr(packed), _lib.ptr(rgb)
        r.out_raw, r.out_weights = _lib.ptr(raw), _lib.ptr(weights)
        r.train_workspace = _lib.ptr(train_workspace)
        r.precision = _lib.PRECISIONS[precision]
        args.normalize_position = self.normalize_position
        args.multiplier = self.multiplier
        args.normalize_directions = 1 if self.normalize_directions else 0

    def _scratch(self, nbytes, device):
        """Cached scratch buffer for the backward's partial slabs."""
        cur = getattr(self, "_scratch_buf", None)
        if cur is None or cur.numel() * 4 < nbytes or cur.device != device:
            self._scratch_buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        return self._scratch_buf

    def _launch(self, n_rays, num_samples, device, near, far, *, rays_o=None, rays_d=None, cameras=None,
                ray_begin=0, u=None, noise=None, density_noise_std=0.0, rgb=None, per_sample=False,
                train_workspace=None):
        lib = _lib.lib()
        packed = self.packed_parameters(fresh=train_workspace is not None)
        if rgb is None:
            rgb = torch.empty(n_rays, 3, dtype=torch.float32, device=device)
        raw = weights = None
        if per_sample:
            raw = torch.empty(n_rays, num_samples, 4, dtype=torch.float32, device=device)
            weights = torch.empty(n_rays, num_samples, dtype=torch.float32, device=device)
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"nerf_amd: precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")
        precision = self.train_precision if train_workspace is not None else self.precision
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"nerf_amd: precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        if precision == "f16x3":
            self._check_f16x3_range(training=train_workspace is not None)
        args = _lib.LegacyArgs()
        self._fill_args(args, n_rays, num_samples, device, near, far, rays_o=rays_o, rays_d=rays_d, cameras=cameras,
                        ray_begin=ray_begin, u=u, noise=noise, density_noise_std=density_noise_std, packed=packed,
                        rgb=rgb, raw=raw, weights=weights, train_workspace=train_workspace, precision=precision)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nerf_hip_legacy_render_forward(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_legacy_render_forward")
        return rgb, raw, weights

    def render_rays(self, rays_o, rays_d, near, far, num_samples, randomly_sample=False,
                    density_noise_std=0.0, u=None, noise=None, per_sample=False):
        """Pixels [N, 3] of a batch of rays (the notebook's signature, cell 8).  ``u`` [N, S] /
        ``noise`` [N, S] replace the torch draws; ``per_sample`` also returns (density | color
        logits [N, S, 4], compositing weights [N, S])."""
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        lead = rays_o.shape[:-1]
        flat_o = rays_o.detach().reshape(-1, 3).contiguous()
        flat_d = rays_d.detach().reshape(-1, 3).contiguous()
        n_rays, dev = flat_o.shape[0], flat_o.device
        if u is None and randomly_sample:
            u = torch.rand(n_rays, num_samples, dtype=torch.float32, device=dev)
        if noise is None and density_noise_std != 0.0:
            noise = torch.randn(n_rays, num_samples, dtype=torch.float32, device=dev)
        u = None if u is None else u.detach().reshape(n_rays, num_samples).contiguous()
        noise = None if noise is None else noise.detach().reshape(n_rays, num_samples).contiguous()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad and not per_sample:
            rgb = LegacyRenderRaysFunction.apply(self, flat_o, flat_d, float(near), float(far), num_samples, u, noise,
                                                 float(density_noise_std), *self._param_list())
            return rgb.reshape(*lead, 3)
        rgb, raw, weights = self._launch(n_rays, num_samples, dev, near, far, rays_o=flat_o, rays_d=flat_d,
                                         u=u, noise=noise, density_noise_std=density_noise_std,
                                         per_sample=per_sample)
        rgb = rgb.reshape(*lead, 3)
        return (rgb, raw, weights) if per_sample else rgb

    def render_image(self, camera_o, camera_r, image_h, image_w, focal_length, near, far, num_samples,
                     row_begin=0, row_end=None):
        """Frames [B, rows, W, 3]: one launch per frame, rays generated inside the kernel from the
        poses (the notebook's signature, cell 8; ``row_begin``/``row_end`` as in the generation-C class)."""
        _require_device(camera_o, "camera_o"), _require_device(camera_r, "camera_r")
        device, batch = camera_o.device, camera_o.shape[0]
        row_end = image_h if row_end is None else row_end
        rows = row_end - row_begin
        cameras = (camera_o.detach().contiguous(), camera_r.detach().contiguous(), image_h, image_w, focal_length)
        out = torch.empty(batch, rows, image_w, 3, dtype=torch.float32, device=device)
        for b in range(batch):
            n_rays = rows * image_w
            self._launch(n_rays, num_samples, device, near, far, cameras=cameras,
                         ray_begin=(b * image_h + row_begin) * image_w, rgb=out[b].reshape(n_rays, 3))
        return out

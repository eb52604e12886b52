"""Host-side mirror of ``nerf.model`` (generation C of brandontrabucco/nerf) for MI355X.

Same importable names, constructor keywords, state-dict keys and method signatures as the
reference's ``nerf/model.py`` so a caller switches with ``from nerf_amd.model import NeRF``.
Everything ``NeRF.render_rays`` / ``render_image`` / ``forward`` compute is done by one fused HIP
kernel behind the C ABI of ``include/nerf_hip.h``; torch only owns device memory and streams here.
There is no CPU or eager fallback: tensors must live on a ROCm device and the library must be built.

The small pose helpers (``generate_rays`` ... ``get_rotation_matrix``) and the module-level
mip-NeRF functions stay thin torch expressions, as the reference has them, for debuggability;
they are not on the hot path (``render_image`` generates its rays inside the kernel).
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as functional

from . import _lib

__all__ = ["NeRF", "expected_sin", "lift_gaussian", "conical_frustum_to_gaussian", "cast_rays",
           "integrated_pos_enc"]

_LOG2_NEAR = -9.43633744014          # nerf/model.py:414


# ----------------------------------------------------------------------------------------------
# module-level helpers (reference: nerf/model.py:24-163) — debugging aids, plain torch
# ----------------------------------------------------------------------------------------------

def expected_sin(x, x_var):
    """Mean and variance of sin(z), z ~ N(x, x_var)  (nerf/model.py:24-30)."""
    damp = torch.exp(-0.5 * x_var)
    y = damp * torch.sin(x)
    y_var = (0.5 * (1 - torch.exp(-2 * x_var) * torch.cos(2 * x)) - y ** 2).clamp(min=0.0)
    return y, y_var


def lift_gaussian(d, t_mean, t_var, r_var, diag=True):
    """Gaussian along a ray -> 3-D diagonal Gaussian  (nerf/model.py:33-45; diag only)."""
    if not diag:
        raise NotImplementedError("full covariance is unreachable from NeRF (nerf/model.py:46-53)")
    mean = d[..., None, :] * t_mean[..., None]
    d_sq = d ** 2
    mag = torch.sum(d_sq, dim=-1, keepdim=True).clamp(min=1e-10)
    cov = t_var[..., None] * d_sq[..., None, :] + r_var[..., None] * (1 - d_sq / mag)[..., None, :]
    return mean, cov


def conical_frustum_to_gaussian(d, t0, t1, base_radius, diag=True, stable=True):
    """Conical frustum [t0, t1] -> Gaussian, stable form  (nerf/model.py:56-87)."""
    if not stable:
        raise NotImplementedError("only the stable formulation is on the render path")
    mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
    denom = 3 * mu ** 2 + hw ** 2
    t_mean = mu + (2 * mu * hw ** 2) / denom
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / denom ** 2)
    r_var = base_radius ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 * (hw ** 4) / denom)
    return lift_gaussian(d, t_mean, t_var, r_var, diag)


def cast_rays(t_vals, origins, directions, radii, ray_shape="cone", diag=True):
    """Fenceposts -> per-interval Gaussians  (nerf/model.py:112-136; cone only)."""
    if ray_shape != "cone":
        raise NotImplementedError("NeRF only casts cones (nerf/model.py:548)")
    means, covs = conical_frustum_to_gaussian(directions, t_vals[..., :-1], t_vals[..., 1:],
                                              radii, diag)
    return means + origins[..., None, :], covs


def integrated_pos_enc(x_coord, min_deg, max_deg):
    """Integrated positional encoding of (mean, diag cov)  (nerf/model.py:139-163)."""
    x, x_cov = x_coord
    scales = torch.as_tensor([2 ** i for i in range(min_deg, max_deg)]).to(x)
    shape = list(x.shape[:-1]) + [-1]
    y = (x[..., None, :] * scales[:, None]).reshape(*shape)
    y_var = (x_cov[..., None, :] * scales[:, None] ** 2).reshape(*shape)
    return expected_sin(torch.cat([y, y + 0.5 * np.pi], dim=-1), torch.cat([y_var] * 2, dim=-1))[0]


# ----------------------------------------------------------------------------------------------
# the renderer
# ----------------------------------------------------------------------------------------------

def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"nerf_amd: `{name}` is on {t.device}; the renderer runs on an MI355X "
                           "(ROCm device) only and has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"nerf_amd: `{name}` must be float32, got {t.dtype}")


class NeRF(nn.Module):
    """mip-NeRF-style radiance field with the call surface of the reference's ``NeRF``
    (nerf/model.py:166-770): LayerNorm MLP 96-256x5-54 on integrated positional encodings of
    conical frusta, one output head split into density | color | segmentation."""

    def __init__(self, color_outputs=3, segmentation_outputs=50, hidden_size=256, encoding_size=32,
                 focal_length=112.0, min_x=-20.0, max_x=20.0, min_y=-20.0, max_y=20.0,
                 min_z=-20.0, max_z=20.0):
        super().__init__()
        self.focal_length = focal_length
        self.color_outputs = color_outputs
        self.segmentation_outputs = segmentation_outputs
        self.hidden_size = hidden_size
        self.encoding_size = encoding_size
        self.register_buffer("rays_min", torch.as_tensor([[[min_x, min_y, min_z]]],
                                                         dtype=torch.float32))
        self.register_buffer("rays_max", torch.as_tensor([[[max_x, max_y, max_z]]],
                                                         dtype=torch.float32))
        # Parameter container with the reference's module tree (nerf/model.py:525-542) so that
        # state-dict keys, default initialisation and RNG consumption are identical.  The
        # modules are never called: the fused kernel reads their tensors.
        layers = [nn.Linear(3 * encoding_size, hidden_size), nn.LayerNorm(hidden_size), nn.ReLU()]
        for _ in range(4):
            layers += [nn.Linear(hidden_size, hidden_size), nn.LayerNorm(hidden_size), nn.ReLU()]
        layers.append(nn.Linear(hidden_size, 1 + color_outputs + segmentation_outputs))
        self.prediction_heads = nn.Sequential(*layers)
        # rng: "torch" draws u / noise with torch's generator exactly where the reference does
        # (model.py:432, :652); "philox" lets the kernel draw them (no HBM round trip).
        self.rng = "torch"
        # precision of the MLP in inference launches (anything that does not record a backward):
        # "fp32" = exact-fp32 MFMA, "f16x3" = split-precision f16 MFMA (include/nerf_hip.h).
        self.precision = "fp32"
        # arithmetic of the training FORWARD's MLP (launches that record a backward); the data and
        # weight gradients keep their own arithmetic (fp32 MFMA / bf16 triples).
        self.train_precision = "fp32"
        self._packed = None
        self._packed_key = None
        self._tables = {}
        self._dead_draw_offsets = {}

    # ---- statics (reference: nerf/model.py:243-367, :438-469) --------------------------------

    @staticmethod
    def generate_rays(image_h, image_w, focal_length, dtype=torch.float32, device="cpu"):
        """Camera-frame pinhole ray per pixel, [H, W, 3] = (x, -y, -1)  (nerf/model.py:243-278)."""
        rows = torch.arange(image_h, dtype=dtype, device=device)
        cols = torch.arange(image_w, dtype=dtype, device=device)
        yy, xx = torch.meshgrid(rows, cols, indexing="ij")
        xx = (xx - 0.5 * float(image_w - 1)) / focal_length
        yy = (yy - 0.5 * float(image_h - 1)) / focal_length
        return torch.stack([xx, -yy, -torch.ones_like(xx)], dim=-1)

    @staticmethod
    def spherical_to_cartesian(yaw, elevation):
        """Unit vector of (yaw, elevation), z up  (nerf/model.py:281-306)."""
        ce = torch.cos(elevation)
        return torch.stack([torch.cos(yaw) * ce, torch.sin(yaw) * ce, torch.sin(elevation)], dim=-1)

    @staticmethod
    def get_rotation_matrix(eye_vector, up_vector):
        """Camera-to-world rotation with columns [eye x up, up, -eye]  (nerf/model.py:309-334).
        The reference calls ``torch.cross`` without ``dim``, which crosses along the FIRST axis of
        size 3 (so a batch of exactly three poses is crossed along the batch axis); kept."""
        dim = next(i for i, n in enumerate(eye_vector.shape) if n == 3)
        side = torch.linalg.cross(eye_vector, up_vector, dim=dim)
        return torch.stack([side, up_vector, -eye_vector], dim=-1)

    @staticmethod
    def rays_to_world_coordinates(rays, camera_o, camera_r):
        """(origin, R . ray)  (nerf/model.py:337-367)."""
        return camera_o, (camera_r * rays.unsqueeze(-2)).sum(dim=-1)

    @staticmethod
    def alpha_compositing_coefficients(points, density_outputs):
        """Compositing weights from points [.., P, 3] and densities [.., P, 1]
        (nerf/model.py:438-469); torch helper, the renderer computes these in-kernel."""
        gaps = points[..., 1:, :] - points[..., :-1, :]
        dists = functional.pad(torch.linalg.norm(gaps, dim=-1, keepdim=True), (0, 0, 0, 1),
                               value=1e10)
        trans = torch.exp(-functional.relu(density_outputs) * dists)
        return (1.0 - trans) * functional.pad(
            torch.cumprod(trans[..., :-1, :] + 1e-10, dim=-2), (0, 0, 1, 0), value=1.0)

    # ---- sampling ------------------------------------------------------------------------------

    def _fencepost_table(self, num_samples, device):
        """Unscaled 2^linspace table (nerf/model.py:414-415), built with the same torch CPU ops as
        the reference and cached on the device."""
        key = (int(num_samples), str(device))
        if key not in self._tables:
            table = torch.pow(2.0, torch.linspace(_LOG2_NEAR, 0.0, num_samples,
                                                  dtype=torch.float32))
            self._tables[key] = table.to(device)
        return self._tables[key]

    def _t_scale(self):
        """|rays_max - rays_min| (nerf/model.py:435) as a host float.  The two buffers only change
        at construction, ``load_state_dict`` or ``.to()``, so the value is cached on their
        (storage, version) and the device -> host copy happens once, not on every launch."""
        key = (self.rays_min.data_ptr(), self.rays_min._version, self.rays_max.data_ptr(),
               self.rays_max._version)
        if getattr(self, "_t_scale_key", None) != key:
            self._t_scale_value = float(torch.linalg.norm(self.rays_max.detach().cpu()
                                                          - self.rays_min.detach().cpu()))
            self._t_scale_key = key
        return self._t_scale_value

    def _next_philox_state(self):
        """(seed, offset) of this module's in-kernel draws.  ONE word carries the launch sequence: the
        device-resident counter of ``_philox_device_counter``, which the kernel adds to this offset and a
        one-thread launch advances behind every drawing launch — eager launches and HIP-graph replays alike, so
        an eager step between two replays cannot land on a replay's key (a host-side count frozen into a captured
        argument block could: eager launches would move host + device, replays only the device word).  The
        host part is therefore only the data-parallel rank in the high bits: neither successive launches nor the
        ranks of a job share a Philox key (include/nerf_hip.h: key = seed ^ (offset + *rng_counter))."""
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        return (int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, (rank & 0xFFFFFF) << 40)

    def _philox_device_counter(self, device):
        """The launch counter of this module's in-kernel draws, a DEVICE word (include/nerf_hip.h: rng_counter):
        every launch that draws in-kernel adds it to its Philox offset and is followed by a one-thread launch
        that increments it.  Device-resident because a HIP-graph replay repeats its argument block: this is what
        gives every replayed step new draws (``Trainer(graph=True, rng="philox")``), and being the ONLY counter it
        keeps eager and replayed launches on one sequence."""
        cur = getattr(self, "_philox_counter", None)
        if cur is None or cur.device != device:
            self._philox_counter = cur = torch.zeros(1, dtype=torch.int64, device=device)
        return cur

    def sample_along_rays(self, rays_o, rays_d, num_samples, states_x=None, states_d=None,
                          randomly_sample=True):
        """Fenceposts [..., S] along each ray, log-spaced, optionally stratified
        (nerf/model.py:369-435).  Only the shape/device of ``rays_o`` matters."""
        lead = list(rays_o.shape[:-1])
        table = self._fencepost_table(num_samples, rays_o.device).to(rays_o.dtype)
        samples = torch.broadcast_to(table.reshape([1] * len(lead) + [num_samples]),
                                     lead + [num_samples])
        if randomly_sample:
            mid = 0.5 * (samples[..., 1:] + samples[..., :-1])
            lower = torch.cat([samples[..., :1], mid], dim=-1)
            upper = torch.cat([mid, samples[..., -1:]], dim=-1)
            draw = torch.rand(*samples.shape, dtype=rays_o.dtype, device=rays_o.device)
            samples = lower + (upper - lower) * draw
        return samples * torch.linalg.norm(self.rays_max - self.rays_min)

    # ---- plumbing to the C ABI -------------------------------------------------------------------

    @property
    def num_outputs(self):
        """Rows of the last Linear: density | color | segmentation (nerf/model.py:541-542)."""
        return 1 + self.color_outputs + self.segmentation_outputs

    @property
    def enc_inputs(self):
        """Inputs of the first Linear: 3 coordinates x encoding_size (nerf/model.py:526)."""
        return 3 * self.encoding_size

    def _check_shape(self):
        """hidden_size (<= 256), encoding_size (even, <= 32), color_outputs (1 .. 12) and the number of segmentation
        classes (1 + colors + classes <= 64: the padded output tile) are run-time arguments of the kernels, which exist
        at 16 / 8 / 4 register tiles per sample: a launch runs at the smallest width that holds hidden_size (a narrow
        network at its own cost) and zero-padded inside it, which is exact (nerf_amd/csrc/nerf_layout.h: Shape, Narrow;
        color channels: row_of_slot)."""
        if not 1 <= self.color_outputs <= 12 or not 1 <= self.hidden_size <= 256 or self.encoding_size % 2 != 0 or \
                not 2 <= self.encoding_size <= 32 or self.segmentation_outputs < 0 or self.num_outputs > 64:
            raise NotImplementedError(
                "libnerf_hip takes 1 <= hidden_size <= 256, an even encoding_size in 2 .. 32, 1 <= color_outputs <= 12 "
                "and 1 + color_outputs + segmentation_outputs <= 64 (the reference's defaults: 256 / 32 / 3 / 50)")

    def _param_list(self):
        heads = self.prediction_heads
        order = []
        for slot in (0, 1, 3, 4, 6, 7, 9, 10, 12, 13, 15):
            order += [heads[slot].weight, heads[slot].bias]
        return order

    def packed_parameters(self, fresh=False):
        """The parameters re-laid for the kernels (nerf_hip_pack_weights, one 8 us launch).  Re-packed
        on EVERY call: tensor version counters cannot be trusted to say "unchanged" — fused optimisers
        (torch.optim.Adam(fused=True)) and ``p.data`` edits update parameters without bumping them,
        and a stale image is a silently wrong render.  ``fresh``: write into a new buffer (a training
        forward: its backward reads the image later, after other launches may have re-packed)."""
        self._check_shape()
        params = self._param_list()
        dev = params[0].device
        for p in params:
            _require_device(p, "parameter")
        lib = _lib.lib()
        keep = [p.detach().contiguous() for p in params]
        ptrs = (ctypes.c_void_p * _lib.NUM_PARAM_TENSORS)(*[p.data_ptr() for p in keep])
        packed = self._packed
        if fresh or packed is None or packed.device != dev:
            packed = torch.empty(lib.nerf_hip_packed_bytes() // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.nerf_hip_pack_weights(ptrs, self.hidden_size, self.enc_inputs, self.num_outputs,
                                                 self.color_outputs, _lib.ptr(packed), ctypes.c_void_p(stream)),
                       "nerf_hip_pack_weights")
        if not fresh:
            self._packed = packed
        self._packed_key = tuple((p.data_ptr(), p._version) for p in params)
        self._last_packed = packed
        return packed

    def _check_f16x3_range(self, training=False):
        """The split-precision kernel holds 2^8 * w and 2^4 * relu(gamma * x_hat + beta) as f16
        pairs (|x_hat| < 16 for 256 features); refuse parameters that would leave the f16 range
        instead of saturating silently.  The check is a device -> host copy, so it runs when a
        parameter tensor's version or storage changed and otherwise on every 64th split-precision
        launch (updates that bypass the version counters — fused optimisers, ``p.data`` — are caught
        there; a weight does not grow from O(0.1) to 256 within 64 steps).  A training forward, whose
        versions change with every ordinary optimiser step, only uses the every-64th rule."""
        if torch.cuda.is_current_stream_capturing():
            return                                   # no host sync inside a HIP-graph capture (the trainer
                                                     # re-checks between replays, Trainer.train_step)
        key = self._packed_key
        self._f16x3_calls = getattr(self, "_f16x3_calls", -1) + 1
        periodic = self._f16x3_calls % 64 == 0
        if not periodic and (training or getattr(self, "_f16x3_checked", None) == key):
            return
        heads = self.prediction_heads
        with torch.no_grad():                       # one device -> host copy per parameter version
            w_dev = torch.stack([heads[i].weight.abs().max() for i in (0, 3, 6, 9, 12, 15)]).max()
            a_dev = torch.stack([16.0 * heads[i].weight.abs().max() + heads[i].bias.abs().max()
                                 for i in (1, 4, 7, 10, 13)]).max()
            w_max, act_max = (float(v) for v in torch.stack([w_dev, a_dev]).cpu())
        if w_max * 256.0 >= 65504.0 or act_max * 16.0 >= 65504.0:
            raise ValueError(f"nerf_amd: parameters out of range for precision='f16x3' (max |w| {w_max:.3g}, "
                             f"max 16|gamma|+|beta| {act_max:.3g}); use precision='fp32'")
        self._f16x3_checked = key

    def _forget_range_check(self):
        self._f16x3_checked = None
        self._f16x3_calls = -1

    def load_state_dict(self, *args, **kwargs):
        """As nn.Module.load_state_dict; the next split-precision launch (inference or training)
        re-checks the f16 range of the new parameters instead of waiting for its 64-launch period."""
        out = super().load_state_dict(*args, **kwargs)
        self._forget_range_check()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)         # .to() / .cuda() / .float(): new storages
        self._forget_range_check()
        return out

    def check_split_precision_range(self):
        """Force the f16 range check of the parameters now (raises ValueError when they left it)."""
        self._forget_range_check()
        self._check_f16x3_range(training=False)

    def _fill_args(self, args, n_rays, num_samples, device, *, rays_o=None, rays_d=None,
                   cameras=None, ray_begin=0, t_values=None, u=None, noise=None,
                   density_noise_std=0.0, rng_mode=0, rng_state=None, rng_counter=None, packed=None, rgb=None,
                   seg=None, mean=None, cov=None, raw=None, weights=None, train_workspace=None, out_t=None,
                   precision=None):
        """Fill a NerfHipRenderArgs block (include/nerf_hip.h) from tensors.  ``precision``: the
        backward passes the arithmetic its forward ran with (already range-checked)."""
        args.rays_o, args.rays_d = _lib.ptr(rays_o), _lib.ptr(rays_d)
        if cameras is not None:
            cam_o, cam_r, image_h, image_w, focal = cameras
            args.camera_o, args.camera_r = _lib.ptr(cam_o), _lib.ptr(cam_r)
            args.image_h, args.image_w, args.focal_length = image_h, image_w, float(focal)
        args.ray_begin, args.n_rays, args.num_samples = int(ray_begin), int(n_rays), int(num_samples)
        table = self._fencepost_table(num_samples, device)
        args.t_table, args.t_scale = _lib.ptr(table), self._t_scale()
        args.t_values, args.u, args.noise = _lib.ptr(t_values), _lib.ptr(u), _lib.ptr(noise)
        args.density_noise_std = float(density_noise_std)
        args.rng_mode = int(rng_mode)
        if rng_mode:
            args.rng_seed, args.rng_offset = rng_state
            args.rng_counter = _lib.ptr(rng_counter)
        r_dot = 1.0 / (math.sqrt(3.0) * self.focal_length)          # nerf/model.py:546
        args.base_radius_sq = r_dot ** 2
        args.packed = _lib.ptr(packed)
        args.rgb, args.seg = _lib.ptr(rgb), _lib.ptr(seg)
        args.out_mean, args.out_cov, args.out_t = _lib.ptr(mean), _lib.ptr(cov), _lib.ptr(out_t)
        args.out_raw, args.out_weights = _lib.ptr(raw), _lib.ptr(weights)
        args.train_workspace = _lib.ptr(train_workspace)
        args.num_outputs, args.color_outputs = self.num_outputs, self.color_outputs
        args.hidden, args.enc_inputs = self.hidden_size, self.enc_inputs
        if precision is not None:
            args.precision = precision
            return
        which = self.precision if train_workspace is None else self.train_precision
        if which not in _lib.PRECISIONS:
            raise ValueError(f"nerf_amd: precision must be one of {sorted(_lib.PRECISIONS)}, got {which!r}")
        args.precision = _lib.PRECISIONS[which]
        if args.precision == _lib.PRECISIONS["f16x3"]:
            self._check_f16x3_range(training=train_workspace is not None)

    def _scratch(self, nbytes, device):
        """Cached scratch buffer for the backward's partial slabs."""
        cur = getattr(self, "_scratch_buf", None)
        if cur is None or cur.numel() * 4 < nbytes or cur.device != device:
            self._scratch_buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        return self._scratch_buf

    def _launch(self, n_rays, num_samples, device, *, rays_o=None, rays_d=None, cameras=None,
                ray_begin=0, t_values=None, u=None, noise=None, density_noise_std=0.0, rng_mode=0,
                want_seg=True, per_sample=False, rgb=None, seg=None, rng_state=None, rng_counter=None,
                train_workspace=None, want_weights=False, cov=None, out_t=None, composite=True):
        """``rng_state`` None with ``rng_mode``: the module's own launch sequence (rank bits + the device counter,
        advanced behind the launch); an explicit state is used as it is (reproducible draws), plus
        ``rng_counter`` if the caller passes one (the training forward: its own sequence state, kept for the
        backward's argument block)."""
        lib = _lib.lib()
        packed = self.packed_parameters(fresh=train_workspace is not None)
        P = num_samples - 1
        if not composite:                               # a training forward for the per-sample outputs only
            want_seg = False
        elif rgb is None:
            rgb = torch.empty(n_rays, self.color_outputs, dtype=torch.float32, device=device)
        if seg is None and want_seg:
            seg = torch.empty(n_rays, self.segmentation_outputs, dtype=torch.float32, device=device)
        if self.segmentation_outputs == 0:
            seg_given, seg = seg, None                  # no classes: nothing for the kernel to write
        mean = raw = weights = None
        if per_sample:
            mean = torch.empty(n_rays, P, 3, dtype=torch.float32, device=device)
            raw = torch.empty(n_rays, P, self.num_outputs, dtype=torch.float32, device=device)
            if composite:
                weights = torch.empty(n_rays, P, dtype=torch.float32, device=device)
        elif want_weights:
            weights = torch.empty(n_rays, P, dtype=torch.float32, device=device)
        if rng_mode and rng_state is None:
            rng_state = self._next_philox_state()
            rng_counter = self._philox_device_counter(device)
        args = _lib.RenderArgs()
        self._fill_args(args, n_rays, num_samples, device, rays_o=rays_o, rays_d=rays_d,
                        cameras=cameras, ray_begin=ray_begin, t_values=t_values, u=u, noise=noise,
                        density_noise_std=density_noise_std, rng_mode=rng_mode, rng_state=rng_state,
                        rng_counter=rng_counter if rng_mode else None,
                        packed=packed, rgb=rgb, seg=seg, mean=mean, cov=cov, raw=raw, weights=weights,
                        train_workspace=train_workspace, out_t=out_t)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.nerf_hip_render_forward(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_render_forward")
            if rng_mode and rng_counter is not None:
                _lib.check(lib.nerf_hip_rng_advance(_lib.ptr(rng_counter), 1, ctypes.c_void_p(stream)),
                           "nerf_hip_rng_advance")
        if self.segmentation_outputs == 0:
            seg = seg_given                              # the caller's empty [n, 0] tensor (or None)
        return rgb, seg, mean, raw, weights

    def _draws(self, n_rays, num_samples, device, randomly_sample, density_noise_std):
        """u / noise for a batch.  "torch": consume torch's generator exactly like the reference
        (rand only when stratified, randn always — model.py:432, :652).  "philox": in-kernel."""
        if self.rng == "philox":
            mode = (1 if randomly_sample else 0) | (2 if density_noise_std != 0.0 else 0)
            return None, None, mode
        u = torch.rand(n_rays, num_samples, dtype=torch.float32, device=device) \
            if randomly_sample else None
        noise = torch.randn(n_rays, num_samples - 1, 1, dtype=torch.float32, device=device)
        if density_noise_std == 0.0:
            noise = None                       # drawn (generator advanced) but adds nothing
        return u, noise, 0

    def _skip_dead_draws(self, ray_lists, num_samples, step, device):
        """Advance torch's generator of ``device`` as the reference's chunk loop does on the deterministic
        path: one ``randn([n, S-1, 1])`` per chunk of ``step`` rays of every list (nerf/model.py:652-654,
        :750-761), without drawing.  How far one such call moves the Philox offset is a property of torch's
        kernel launch policy on this device, so it is MEASURED once per chunk shape on a private generator
        and cached, not derived."""
        if torch.cuda.is_current_stream_capturing():
            return                                   # a captured region cannot move the host-side offset
        gen = torch.cuda.default_generators[device.index if device.index is not None
                                            else torch.cuda.current_device()]
        advance = 0
        for count in ray_lists:
            full, tail = divmod(int(count), step)
            for n, times in ((step, full), (tail, 1 if tail else 0)):
                if times == 0:
                    continue
                key = (n, int(num_samples), str(device))
                if key not in self._dead_draw_offsets:
                    probe = torch.Generator(device=device)
                    probe.manual_seed(0)
                    torch.randn(n, num_samples - 1, 1, dtype=torch.float32, device=device, generator=probe)
                    self._dead_draw_offsets[key] = int(probe.get_offset())
                advance += times * self._dead_draw_offsets[key]
        if advance:
            gen.set_offset(gen.get_offset() + advance)

    # ---- the reference's methods -----------------------------------------------------------------

    def forward(self, rays_o, rays_d, samples, states_x=None, states_d=None):
        """Field values on the intervals of ``samples`` [N, S]: returns (mean [N,S-1,3],
        density [N,S-1,1], color [N,S-1,3], segmentation [N,S-1,50])  (nerf/model.py:553-594).
        ``states_*`` are accepted and ignored, as in the reference.  Like the reference's it is differentiable
        w.r.t. the parameters when gradients are enabled (a loss on density / color / segmentation trains the
        network: nerf_amd/backward.py FieldFunction); under ``torch.no_grad()`` it is the plain inference launch.
        Arithmetic: the differentiable call is a TRAINING forward and runs in ``self.train_precision`` (its backward
        must match it), the no_grad call in ``self.precision``; with the two set differently the same inputs give
        values that differ by the arithmetics' distance (~1e-6).  NOT differentiable w.r.t. ``rays_o`` /
        ``rays_d`` / ``samples`` (the reference's autograd reaches them; its scripts never ask): they are detached,
        with a warning when one of them requires grad."""
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        _require_device(samples, "samples")
        if torch.is_grad_enabled() and any(t.requires_grad for t in (rays_o, rays_d, samples)):
            import warnings
            warnings.warn("nerf_amd.NeRF.forward is differentiable w.r.t. the parameters only: rays_o / rays_d / "
                          "samples are detached (no gradient reaches them)", stacklevel=2)
        n_rays, num_samples = samples.shape[0], samples.shape[-1]
        o, d, t = rays_o.detach().contiguous(), rays_d.detach().contiguous(), samples.detach().contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .backward import FieldFunction
            mean, raw = FieldFunction.apply(self, o, d, t, *self._param_list())
        else:
            _, _, mean, raw, _ = self._launch(n_rays, num_samples, rays_o.device, rays_o=o, rays_d=d, t_values=t,
                                              want_seg=False, per_sample=True)
        density, color, seg = raw.split([1, self.color_outputs, self.segmentation_outputs], dim=2)
        return mean, density, color, seg

    def fenceposts_used(self, rays_o, rays_d, num_samples, randomly_sample=False, u=None, rng_state=None):
        """The fenceposts [N, S] a render of these rays uses, straight from the kernel's front end:
        deterministic, stratified from given ``u`` [N, S], or — ``self.rng == "philox"`` — from the
        in-kernel draws of the NEXT launch state (or of ``rng_state``).  Debugging / parity aid."""
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        n_rays = rays_o.shape[0]
        out = torch.empty(n_rays, num_samples, dtype=torch.float32, device=rays_o.device)
        mode = 1 if (randomly_sample and u is None) else 0
        self._launch(n_rays, num_samples, rays_o.device, rays_o=rays_o.detach().contiguous(),
                     rays_d=rays_d.detach().contiguous(), u=None if u is None else u.detach().contiguous(),
                     rng_mode=mode, rng_state=rng_state, want_seg=False, out_t=out)
        return out

    def integrated_pe(self, rays_o, rays_d, samples):
        """(means, covs, h) of the intervals of ``samples`` [N, S]  (nerf/model.py:544-551): the
        Gaussians come from the kernel's front end (the values the MLP is fed from), the 96
        features from the torch helper above on the same device — a debugging aid, like the
        reference's method; the renderer never materialises ``h``."""
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        _require_device(samples, "samples")
        n_rays, num_samples = samples.shape[0], samples.shape[-1]
        cov = torch.empty(n_rays, num_samples - 1, 3, dtype=torch.float32, device=rays_o.device)
        _, _, mean, _, _ = self._launch(
            n_rays, num_samples, rays_o.device, rays_o=rays_o.detach().contiguous(),
            rays_d=rays_d.detach().contiguous(), t_values=samples.detach().contiguous(),
            want_seg=False, per_sample=True, cov=cov)
        return mean, cov, integrated_pos_enc((mean, cov), -4, self.encoding_size // 2 - 4)

    def render_rays(self, rays_o, rays_d, num_samples, states_x=None, states_d=None,
                    randomly_sample=False, density_noise_std=0.0, u=None, noise=None):
        """Render a batch of rays: returns (image [N,1,3], segmentation [N,1,50]); the middle axis
        is the reference's (single) stage axis  (nerf/model.py:596-668).

        Extension: ``u`` [N,S] and ``noise`` [N,S-1,1] may be passed to replace the random draws
        (bit-reproducible stochastic path); otherwise they are drawn per ``self.rng``."""
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        lead = rays_o.shape[:-1]
        flat_o = rays_o.detach().reshape(-1, 3).contiguous()
        flat_d = rays_d.detach().reshape(-1, 3).contiguous()
        n_rays = flat_o.shape[0]
        mode = 0
        if u is None and noise is None:
            u, noise, mode = self._draws(n_rays, num_samples, flat_o.device, randomly_sample,
                                         density_noise_std)
        u = None if u is None else u.detach().reshape(n_rays, num_samples).contiguous()
        noise = None if noise is None else noise.detach().reshape(n_rays, num_samples - 1).contiguous()
        from .autograd import render_rays_function
        rgb, seg, _ = render_rays_function(self, flat_o, flat_d, num_samples, u, noise,
                                           float(density_noise_std), mode)
        return (rgb.reshape(*lead, 1, self.color_outputs),
                seg.reshape(*lead, 1, self.segmentation_outputs))

    # ---- hierarchical sampling (BASELINE config 3; parity unpinned, see include/nerf_hip.h) -----

    def resample_fenceposts(self, t_coarse, weights, num_fine, u=None, pdf_floor=1e-5):
        """Sorted union [N, S_c + num_fine] of the coarse fenceposts ``t_coarse`` [N,S_c] and
        ``num_fine`` fenceposts drawn by inverse-transform sampling from the piecewise-constant
        PDF that the coarse compositing ``weights`` [N,S_c-1] define (Mildenhall et al. 2020,
        section 5.2).  ``u`` [N,num_fine]: ascending uniforms in [0,1); default (k + 0.5)/num_fine."""
        _require_device(t_coarse, "t_coarse"), _require_device(weights, "weights")
        n_rays, num_coarse = t_coarse.shape
        out = torch.empty(n_rays, num_coarse + num_fine, dtype=torch.float32, device=t_coarse.device)
        args = _lib.ResampleArgs()
        args.n_rays, args.num_coarse, args.num_fine = n_rays, num_coarse, int(num_fine)
        t_coarse, weights = t_coarse.detach().contiguous(), weights.detach().contiguous()
        u = None if u is None else u.detach().contiguous()
        args.t_coarse, args.weights, args.u = _lib.ptr(t_coarse), _lib.ptr(weights), _lib.ptr(u)
        args.pdf_floor, args.t_union = float(pdf_floor), _lib.ptr(out)
        with torch.cuda.device(out.device):
            stream = torch.cuda.current_stream(out.device).cuda_stream
            _lib.check(_lib.lib().nerf_hip_resample_pdf(ctypes.byref(args), ctypes.c_void_p(stream)),
                       "nerf_hip_resample_pdf")
        return out

    def render_rays_hierarchical(self, rays_o, rays_d, num_coarse, num_fine, states_x=None,
                                 states_d=None, randomly_sample=False, density_noise_std=0.0):
        """Two-stage render: ``num_coarse`` log-spaced fenceposts, then the sorted union with
        ``num_fine`` fenceposts resampled from the coarse weights.  Returns
        (image [N,2,3], segmentation [N,2,50]) — coarse and fine on the reference's stage axis
        (model.py:645, :667-668); the same network serves both stages.  Gradients flow through both
        renders, not through the sampling (the paper's recipe)."""
        from .autograd import render_rays_function
        _require_device(rays_o, "rays_o"), _require_device(rays_d, "rays_d")
        flat_o = rays_o.detach().reshape(-1, 3).contiguous()
        flat_d = rays_d.detach().reshape(-1, 3).contiguous()
        n_rays, dev = flat_o.shape[0], flat_o.device
        t_c = self.sample_along_rays(flat_o, flat_d, num_coarse, randomly_sample=randomly_sample)
        t_c = t_c.contiguous()

        def noise_for(num_samples):
            draw = torch.randn(n_rays, num_samples - 1, dtype=torch.float32, device=dev)
            return draw if density_noise_std != 0.0 else None

        std = float(density_noise_std)
        rgb_c, seg_c, w_c = render_rays_function(self, flat_o, flat_d, num_coarse, None,
                                                 noise_for(num_coarse), std, 0, t_values=t_c,
                                                 want_weights=True)
        u_f = None
        if randomly_sample:
            k = torch.arange(num_fine, dtype=torch.float32, device=dev)
            u_f = (k + torch.rand(n_rays, num_fine, dtype=torch.float32, device=dev)) / num_fine
            u_f = u_f.clamp(max=1.0 - 1e-6)
        t_u = self.resample_fenceposts(t_c, w_c, num_fine, u=u_f)
        rgb_f, seg_f, _ = render_rays_function(self, flat_o, flat_d, num_coarse + num_fine, None,
                                               noise_for(num_coarse + num_fine), std, 0, t_values=t_u)
        lead = rays_o.shape[:-1]
        image = torch.stack([rgb_c, rgb_f], dim=-2).reshape(*lead, 2, self.color_outputs)
        seg = torch.stack([seg_c, seg_f], dim=-2).reshape(*lead, 2, self.segmentation_outputs)
        return image, seg

    def render_image_hierarchical(self, camera_o, camera_r, image_h, image_w, focal_length, num_coarse,
                                  num_fine, max_chunk_size=262144, randomly_sample=False,
                                  density_noise_std=0.0):
        """``render_image`` with two-stage sampling: returns the LAST stage, like the reference's
        ``x[:, -1]`` (model.py:757): (image [B,H,W,3], segmentation [B,H,W,50])."""
        _require_device(camera_o, "camera_o"), _require_device(camera_r, "camera_r")
        batch = camera_o.shape[0]
        rays = self.generate_rays(image_h, image_w, focal_length, dtype=camera_o.dtype,
                                  device=camera_o.device)
        rays = torch.broadcast_to(rays.unsqueeze(0), [batch, image_h, image_w, 3])
        cam_o = torch.broadcast_to(camera_o[:, None, None, :], [batch, image_h, image_w, 3])
        cam_r = torch.broadcast_to(camera_r[:, None, None, :, :], [batch, image_h, image_w, 3, 3])
        rays_o, rays_d = self.rays_to_world_coordinates(rays, cam_o, cam_r)
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        images, segs = [], []
        for o_i, d_i in zip(torch.split(rays_o, max_chunk_size), torch.split(rays_d, max_chunk_size)):
            img, seg = self.render_rays_hierarchical(o_i, d_i, num_coarse, num_fine,
                                                     randomly_sample=randomly_sample,
                                                     density_noise_std=density_noise_std)
            images.append(img[:, -1])
            segs.append(seg[:, -1])
        return (torch.cat(images).reshape(batch, image_h, image_w, self.color_outputs),
                torch.cat(segs).reshape(batch, image_h, image_w, self.segmentation_outputs))

    def render_image(self, camera_o, camera_r, image_h, image_w, focal_length, num_samples,
                     states_x=None, states_d=None, max_chunk_size=1024, randomly_sample=False,
                     density_noise_std=0.0, row_begin=0, row_end=None):
        """Render full frames: returns (image [B,H,W,3], segmentation [B,H,W,50])
        (nerf/model.py:670-770).  Rays are generated inside the kernel from the poses; the frame
        is one launch, so ``max_chunk_size`` (the reference's memory knob) only bounds the size
        of the random-draw tensors on the stochastic path.

        Extension for multi-GPU sharding: ``row_begin``/``row_end`` render only that block of
        image rows (of every frame in the batch) and return [B, rows, W, .].

        Like the reference's, the result is differentiable w.r.t. the parameters when gradients are enabled
        (the reference has no ``no_grad`` inside, model.py:754-770; its callers wrap the call,
        train_conditional_nerf.py:139): the frame then goes through ``render_rays`` chunk by chunk exactly as
        the reference's loop does, every chunk keeping its training workspace (11 KB per sample) until the
        backward.  Wrap inference in ``torch.no_grad()`` for the one-launch path."""
        _require_device(camera_o, "camera_o"), _require_device(camera_r, "camera_r")
        device = camera_o.device
        batch = camera_o.shape[0]
        row_end = image_h if row_end is None else row_end
        rows = row_end - row_begin
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # the reference's body (model.py:727-770) on the differentiable render_rays
            rays = self.generate_rays(image_h, image_w, focal_length, dtype=camera_o.dtype, device=device)
            rays = torch.broadcast_to(rays[row_begin:row_end].unsqueeze(0), [batch, rows, image_w, 3])
            cam_o = torch.broadcast_to(camera_o[:, None, None, :], [batch, rows, image_w, 3])
            cam_r = torch.broadcast_to(camera_r[:, None, None, :, :], [batch, rows, image_w, 3, 3])
            rays_o, rays_d = self.rays_to_world_coordinates(rays, cam_o, cam_r)
            rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
            step = max(int(max_chunk_size), 1)
            parts = [self.render_rays(o_i, d_i, num_samples, randomly_sample=randomly_sample,
                                      density_noise_std=density_noise_std)
                     for o_i, d_i in zip(torch.split(rays_o, step), torch.split(rays_d, step))]
            return (torch.cat([p[0][:, -1] for p in parts]).reshape(batch, rows, image_w, self.color_outputs),
                    torch.cat([p[1][:, -1] for p in parts]).reshape(batch, rows, image_w,
                                                                    self.segmentation_outputs))
        cam_o = camera_o.detach().contiguous()
        cam_r = camera_r.detach().contiguous()
        cameras = (cam_o, cam_r, image_h, image_w, focal_length)
        stochastic = randomly_sample or density_noise_std != 0.0
        rgb_out = torch.empty(batch, rows, image_w, self.color_outputs, dtype=torch.float32,
                              device=device)
        seg_out = torch.empty(batch, rows, image_w, self.segmentation_outputs, dtype=torch.float32,
                              device=device)
        n_rays = rows * image_w
        if not stochastic or self.rng == "philox":
            # one launch per frame, written straight into the output block.  On the deterministic path the
            # reference still draws a randn per chunk that it then multiplies by 0 (model.py:652-654, once
            # per chunk of its loop :757-761): nothing is drawn here, but with rng="torch" the device
            # generator is ADVANCED by what those draws consume, so that whatever the caller draws next is
            # what it would have drawn after the reference's render_image.
            mode = 0
            if stochastic:
                mode = (1 if randomly_sample else 0) | (2 if density_noise_std != 0.0 else 0)
            elif self.rng == "torch":
                total = batch * n_rays
                self._skip_dead_draws([total] if rows == image_h else [n_rays] * batch, num_samples,
                                      max(int(max_chunk_size), 1), device)
            for b in range(batch):
                self._launch(n_rays, num_samples, device, cameras=cameras,
                             ray_begin=(b * image_h + row_begin) * image_w,
                             density_noise_std=density_noise_std, rng_mode=mode,
                             rgb=rgb_out[b].reshape(n_rays, -1), seg=seg_out[b].reshape(n_rays, -1))
            return rgb_out, seg_out
        # Stochastic with torch draws: the reference flattens ALL B*H*W rays and splits that list into
        # chunks of max_chunk_size — across frame boundaries — drawing rand [n,S] then randn [n,S-1,1]
        # per chunk (model.py:750-761, :432, :652).  Same chunks, same draw order here; the kernel takes
        # a chunk that straddles two frames as it is (global ray id -> frame, row, column).  A row block
        # (the sharding extension, no reference counterpart) is chunked frame by frame.
        step = max(int(max_chunk_size), 1)
        if rows == image_h:
            spans = [(0, batch * n_rays, rgb_out.reshape(batch * n_rays, -1), seg_out.reshape(batch * n_rays, -1))]
        else:
            spans = [((b * image_h + row_begin) * image_w, n_rays, rgb_out[b].reshape(n_rays, -1),
                      seg_out[b].reshape(n_rays, -1)) for b in range(batch)]
        for begin, count, flat_rgb, flat_seg in spans:
            for lo in range(0, count, step):
                n = min(step, count - lo)
                u, noise, mode = self._draws(n, num_samples, device, randomly_sample, density_noise_std)
                u = None if u is None else u.contiguous()
                noise = None if noise is None else noise.reshape(n, num_samples - 1).contiguous()
                self._launch(n, num_samples, device, cameras=cameras, ray_begin=begin + lo, u=u, noise=noise,
                             density_noise_std=density_noise_std, rng_mode=mode,
                             rgb=flat_rgb[lo:lo + n], seg=flat_seg[lo:lo + n])
        return rgb_out, seg_out

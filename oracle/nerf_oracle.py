"""CPU oracle for the volume-render hot path of brandontrabucco/nerf.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: it may be imported by
``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline``
leg.  Nothing under ``nerf_amd/`` (the product) imports it, and the product has
no CPU fallback: it fails loudly when the HIP library is missing.

It is a functional restatement, in fp32 torch CPU ops, of generation C of the
reference renderer (``nerf/model.py``); every function cites the reference
``file:line`` it follows.  torch is used (rather than numpy) because the path
is floating point and because autograd then gives the gradient oracle for the
training path for free.  Parity pin: ``tests/test_oracle_golden.py`` checks
every stage against the fixtures under ``tests/golden/`` that
``tests/golden/make_golden.py`` produced by importing the reference itself in
the build container (max |delta rgb| <= 2e-6; in practice bit-exact, the same
ATen kernels are reached).

Parameters travel as a plain ``dict`` keyed by the reference's state_dict names
(``prediction_heads.{0,1,3,...,15}.{weight,bias}``, ``rays_min``, ``rays_max``),
so a reference checkpoint can be fed to the oracle directly.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# Constant of nerf/model.py:414 (log2 of the nearest fencepost, unscaled).
LOG2_NEAR = -9.43633744014

LINEAR_IDS = (0, 3, 6, 9, 12, 15)     # nn.Sequential slots of the 6 Linear   (model.py:525-542)
NORM_IDS = (1, 4, 7, 10, 13)          # nn.Sequential slots of the 5 LayerNorm (model.py:525-542)


def default_config():
    """Constructor defaults of nerf/model.py:471-475."""
    return dict(color_outputs=3, segmentation_outputs=50, hidden_size=256,
                encoding_size=32, focal_length=112.0)


def init_params(seed=0, cfg=None, box=20.0):
    """PyTorch-default initialisation with the same RNG consumption order as
    ``NeRF.__init__`` (nerf/model.py:525-542): six nn.Linear built in order
    (LayerNorm draws nothing), so ``torch.manual_seed(seed)`` followed by this
    yields the same tensors as ``torch.manual_seed(seed); NeRF()``."""
    cfg = cfg or default_config()
    torch.manual_seed(seed)
    hid = cfg["hidden_size"]
    n_out = 1 + cfg["color_outputs"] + cfg["segmentation_outputs"]
    dims = [(3 * cfg["encoding_size"], hid)] + [(hid, hid)] * 4 + [(hid, n_out)]
    params = {
        "rays_min": torch.tensor([[[-box, -box, -box]]], dtype=torch.float32),
        "rays_max": torch.tensor([[[box, box, box]]], dtype=torch.float32),
    }
    for slot, (fan_in, fan_out) in zip(LINEAR_IDS, dims):
        lin = torch.nn.Linear(fan_in, fan_out)
        params[f"prediction_heads.{slot}.weight"] = lin.weight.detach().clone()
        params[f"prediction_heads.{slot}.bias"] = lin.bias.detach().clone()
    for slot in NORM_IDS:
        params[f"prediction_heads.{slot}.weight"] = torch.ones(hid)
        params[f"prediction_heads.{slot}.bias"] = torch.zeros(hid)
    return params


# --------------------------------------------------------------------------
# a1-a3: camera helpers
# --------------------------------------------------------------------------

def pinhole_rays(image_h, image_w, focal_length, dtype=torch.float32):
    """nerf/model.py:243-278 — camera-frame ray per pixel, ij indexing,
    pixel-centre convention 0.5*(W-1), not normalised, [x, -y, -1]."""
    rows = torch.arange(image_h, dtype=dtype)
    cols = torch.arange(image_w, dtype=dtype)
    yy, xx = torch.meshgrid(rows, cols, indexing="ij")
    xx = (xx - 0.5 * float(image_w - 1)) / focal_length
    yy = (yy - 0.5 * float(image_h - 1)) / focal_length
    return torch.stack([xx, -yy, -torch.ones_like(xx)], dim=-1)


def spherical_to_cartesian(yaw, elevation):
    """nerf/model.py:281-306."""
    ce = torch.cos(elevation)
    return torch.stack([torch.cos(yaw) * ce, torch.sin(yaw) * ce,
                        torch.sin(elevation)], dim=-1)


def legacy_cross_dim(shape):
    """``torch.cross`` without ``dim`` (as called at model.py:333) crosses along the FIRST
    dimension of size 3, not the last: a batch of exactly three poses is crossed along the
    batch axis.  Restated here so that quirk is part of the contract."""
    for i, n in enumerate(shape):
        if n == 3:
            return i
    raise RuntimeError("no dimension of size 3")


def rotation_from_eye_up(eye, up):
    """nerf/model.py:309-334 — columns [eye x up, up, -eye]."""
    side = torch.linalg.cross(eye, up, dim=legacy_cross_dim(eye.shape))
    return torch.stack([side, up, -eye], dim=-1)


def rays_to_world(rays, camera_o, camera_r):
    """nerf/model.py:337-367 — origin unchanged, direction = R . ray (row dot)."""
    return camera_o, (camera_r * rays.unsqueeze(-2)).sum(dim=-1)


# --------------------------------------------------------------------------
# a5: fenceposts
# --------------------------------------------------------------------------

def fencepost_table(num_samples, dtype=torch.float32):
    """Unscaled log-spaced fenceposts 2^linspace(LOG2_NEAR, 0, S)  (model.py:414-415)."""
    return torch.pow(2.0, torch.linspace(LOG2_NEAR, 0.0, num_samples, dtype=dtype))


def box_diagonal(params):
    """|rays_max - rays_min|_2, the scale applied at model.py:435."""
    return torch.linalg.norm(params["rays_max"] - params["rays_min"])


def sample_t(params, n_rays, num_samples, u=None):
    """nerf/model.py:369-435.  ``u`` ([n_rays, S] uniform draws) replaces the
    ``torch.rand`` of :432 so the stochastic path is reproducible."""
    table = fencepost_table(num_samples).reshape(1, num_samples)
    t = torch.broadcast_to(table, (n_rays, num_samples))
    if u is not None:
        mid = 0.5 * (t[..., 1:] + t[..., :-1])
        lower = torch.cat([t[..., :1], mid], dim=-1)
        upper = torch.cat([mid, t[..., -1:]], dim=-1)
        t = lower + (upper - lower) * u
    return t * box_diagonal(params)


# --------------------------------------------------------------------------
# a6: conical frustum -> Gaussian
# --------------------------------------------------------------------------

def frustum_gaussians(rays_o, rays_d, t, base_radius):
    """cast_rays('cone') -> conical_frustum_to_gaussian(stable) -> lift_gaussian(diag)
    (nerf/model.py:112-136, :56-87, :33-45).  Returns means, covs [N, S-1, 3]."""
    t0, t1 = t[..., :-1], t[..., 1:]
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    denom = 3 * mu ** 2 + hw ** 2
    t_mean = mu + (2 * mu * hw ** 2) / denom
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / denom ** 2)
    r_var = base_radius ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2
                                - 4 / 15 * (hw ** 4) / denom)
    mean = rays_d[..., None, :] * t_mean[..., None]
    d_sq = rays_d ** 2
    d_mag_sq = torch.sum(d_sq, dim=-1, keepdim=True).clamp(min=1e-10)
    null_diag = 1 - d_sq / d_mag_sq
    cov = t_var[..., None] * d_sq[..., None, :] + r_var[..., None] * null_diag[..., None, :]
    return mean + rays_o[..., None, :], cov


# --------------------------------------------------------------------------
# a7: integrated positional encoding
# --------------------------------------------------------------------------

def ipe_features(means, covs, min_deg, max_deg):
    """integrated_pos_enc + expected_sin()[0]  (nerf/model.py:139-163, :24-30).
    Layout [sin: scale-major x coord-minor | sin(.+pi/2): same]."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)], dtype=means.dtype)
    shape = list(means.shape[:-1]) + [-1]
    y = (means[..., None, :] * scales[:, None]).reshape(*shape)
    y_var = (covs[..., None, :] * scales[:, None] ** 2).reshape(*shape)
    arg = torch.cat([y, y + 0.5 * np.pi], dim=-1)
    var = torch.cat([y_var, y_var], dim=-1)
    return torch.exp(-0.5 * var) * torch.sin(arg)


# --------------------------------------------------------------------------
# a9: the MLP
# --------------------------------------------------------------------------

def mlp(params, h, gates=None, record=None):
    """prediction_heads: Linear -> (LayerNorm -> ReLU -> Linear) x 5  (model.py:525-542).

    Test aids (not in the reference): ``record`` — a list that receives the five ReLU gates (LayerNorm output > 0)
    this evaluation ran with; ``gates`` — five boolean tensors to run with INSTEAD (x * gate in place of relu(x)).  A
    gradient is discontinuous in every gate, and a gate within rounding of zero falls on either side depending on the
    summation order of the LayerNorm that feeds it: forcing the gates another implementation reports makes both sides
    differentiate the same piecewise-linear function, so that their gradients can be compared at arithmetic accuracy
    and the flipped gates be counted separately."""
    x = F.linear(h, params["prediction_heads.0.weight"], params["prediction_heads.0.bias"])
    for i, (norm_slot, lin_slot) in enumerate(zip(NORM_IDS, LINEAR_IDS[1:])):
        x = F.layer_norm(x, (x.shape[-1],), params[f"prediction_heads.{norm_slot}.weight"],
                         params[f"prediction_heads.{norm_slot}.bias"], 1e-5)
        if record is not None:
            record.append((x > 0).detach())
        x = F.relu(x) if gates is None else x * gates[i].to(x.dtype)
        x = F.linear(x, params[f"prediction_heads.{lin_slot}.weight"],
                     params[f"prediction_heads.{lin_slot}.bias"])
    return x


def mlp_stages(params, h):
    """The same network as ``mlp`` with the LayerNorm internals exposed: per hidden layer the
    normalised pre-affine activations x_hat = (x - mean) / sqrt(var + 1e-5) and 1 / sqrt(var + 1e-5)
    (biased variance, torch.nn.LayerNorm, nerf/model.py:525-542).  For stage-by-stage parity of the
    training forward's saved tensors."""
    x = F.linear(h, params["prediction_heads.0.weight"], params["prediction_heads.0.bias"])
    x_hats, rstds = [], []
    for norm_slot, lin_slot in zip(NORM_IDS, LINEAR_IDS[1:]):
        mean = x.mean(dim=-1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        x_hat = (x - mean) * rstd
        x_hats.append(x_hat)
        rstds.append(rstd[..., 0])
        x = F.relu(x_hat * params[f"prediction_heads.{norm_slot}.weight"]
                   + params[f"prediction_heads.{norm_slot}.bias"])
        x = F.linear(x, params[f"prediction_heads.{lin_slot}.weight"],
                     params[f"prediction_heads.{lin_slot}.bias"])
    return x, x_hats, rstds


def field(params, cfg, rays_o, rays_d, t, gates=None, record=None):
    """NeRF.forward (model.py:553-594): returns means, covs, h, density, color, seg.  (gates / record: ``mlp``.)"""
    base_radius = 1 / (np.sqrt(3) * cfg["focal_length"])                       # :546
    means, covs = frustum_gaussians(rays_o, rays_d, t, base_radius)
    h = ipe_features(means, covs, -4, cfg["encoding_size"] // 2 - 4)           # :550-551
    out = mlp(params, h, gates=gates, record=record)
    density, color, seg = out.split(
        [1, cfg["color_outputs"], cfg["segmentation_outputs"]], dim=-1)         # :591-592
    return means, covs, h, density, color, seg


# --------------------------------------------------------------------------
# a11-a13: compositing
# --------------------------------------------------------------------------

def composite_weights(points, density, gate=None):
    """alpha_compositing_coefficients (model.py:438-469).  ``gate`` (test aid, see ``mlp``): a boolean tensor to use
    INSTEAD of the density's own ReLU gate (density * gate in place of relu(density))."""
    gaps = points[..., 1:, :] - points[..., :-1, :]
    dists = F.pad(torch.linalg.norm(gaps, dim=-1, keepdim=True), (0, 0, 0, 1), value=1e10)
    trans = torch.exp(-(F.relu(density) if gate is None else density * gate.to(density.dtype)) * dists)
    return (1.0 - trans) * F.pad(torch.cumprod(trans[..., :-1, :] + 1e-10, dim=-2),
                                 (0, 0, 1, 0), value=1.0)


def render_rays(params, cfg, rays_o, rays_d, num_samples, u=None, noise=None,
                density_noise_std=0.0, return_stages=False, gates=None, record=None):
    """NeRF.render_rays (model.py:596-668) without the stage axis.

    ``u`` [N,S] / ``noise`` [N,S-1,1] stand for the torch.rand (:432) and
    torch.randn (:652) draws.  Returns rgb [N,3], seg [N,50] (log-probs)."""
    n_rays = rays_o.shape[0]
    t = sample_t(params, n_rays, num_samples, u)
    means, covs, h, density, color, seg = field(params, cfg, rays_o, rays_d, t, gates=gates, record=record)
    if noise is not None:
        density = density + noise * density_noise_std                           # :652-654
    # (test aid: a sixth entry of ``gates`` / ``record`` is the ReLU gate of the (noisy) density in the compositing)
    if record is not None:
        record.append((density > 0).detach())
    weights = composite_weights(means, density, gates[5] if gates is not None and len(gates) > 5 else None)   # :658
    rgb = (weights * torch.sigmoid(color)).sum(dim=-2)                          # :660
    seg_out = (torch.log(weights + 1e-10)
               + torch.log_softmax(seg, dim=-1)).logsumexp(dim=-2)              # :661-663
    if return_stages:
        return rgb, seg_out, dict(t=t, means=means, covs=covs, h=h, density=density,
                                  color=color, seg=seg, weights=weights)
    return rgb, seg_out


def image_rays(camera_o, camera_r, image_h, image_w, focal_length):
    """Ray set of render_image (model.py:727-751): flattened [B*H*W, 3] o and d."""
    batch = camera_o.shape[0]
    rays = pinhole_rays(image_h, image_w, focal_length, dtype=camera_o.dtype)
    rays = torch.broadcast_to(rays.unsqueeze(0), [batch, image_h, image_w, 3])
    cam_o = torch.broadcast_to(camera_o[:, None, None, :], [batch, image_h, image_w, 3])
    cam_r = torch.broadcast_to(camera_r[:, None, None, :, :], [batch, image_h, image_w, 3, 3])
    rays_o, rays_d = rays_to_world(rays, cam_o, cam_r)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def render_image(params, cfg, camera_o, camera_r, image_h, image_w, focal_length,
                 num_samples, max_chunk_size=1024):
    """NeRF.render_image (model.py:670-770), deterministic sampling."""
    rays_o, rays_d = image_rays(camera_o, camera_r, image_h, image_w, focal_length)
    rgb, seg = [], []
    for o_i, d_i in zip(torch.split(rays_o, max_chunk_size), torch.split(rays_d, max_chunk_size)):
        a, b = render_rays(params, cfg, o_i, d_i, num_samples)
        rgb.append(a)
        seg.append(b)
    batch = camera_o.shape[0]
    return (torch.cat(rgb).reshape(batch, image_h, image_w, cfg["color_outputs"]),
            torch.cat(seg).reshape(batch, image_h, image_w, cfg["segmentation_outputs"]))


# --------------------------------------------------------------------------
# metric + synthetic camera shared by tests / bench
# --------------------------------------------------------------------------

def psnr(a, b):
    """train_conditional_nerf.py:152-153 (natural log / hard-coded ln 10)."""
    return -10.0 * torch.log(((a - b) ** 2).mean()) / 2.30258509299


def look_at_pose(camera_o):
    """Camera at ``camera_o`` looking at the origin, z-up; returns R [1,3,3] built
    with the reference's get_rotation_matrix convention (model.py:333-334)."""
    cam = torch.as_tensor(camera_o, dtype=torch.float32).reshape(1, 3)
    eye = -cam / torch.linalg.norm(cam, dim=-1, keepdim=True)
    z = torch.tensor([[0.0, 0.0, 1.0]])
    up = z - (z * eye).sum(-1, keepdim=True) * eye
    up = up / torch.linalg.norm(up, dim=-1, keepdim=True)
    return rotation_from_eye_up(eye, up)


def training_loss(params, cfg, rays_o, rays_d, num_samples, target, u, noise, noise_std, gates=None, record=None):
    """MSE of train_conditional_nerf.py:132 on the generation-C output shape.  (gates / record: ``mlp``.)"""
    rgb, _ = render_rays(params, cfg, rays_o, rays_d, num_samples, u=u, noise=noise,
                         density_noise_std=noise_std, gates=gates, record=record)
    return ((rgb.unsqueeze(1) - target.unsqueeze(1)) ** 2).mean()


def flops_per_sample(cfg=None):
    """GEMM FLOPs per evaluated sample: 601,088 for the defaults."""
    cfg = cfg or default_config()
    hid = cfg["hidden_size"]
    n_out = 1 + cfg["color_outputs"] + cfg["segmentation_outputs"]
    return 2 * (3 * cfg["encoding_size"] * hid + 4 * hid * hid + hid * n_out)


assert math.isclose(flops_per_sample(), 601088)


# --------------------------------------------------------------------------
# hierarchical sampling — PARITY UNPINNED: the reference has no code for it (only docstrings,
# model.py:191-193).  This restates Mildenhall et al. 2020 section 5.2 and is the spec the HIP
# resampler is tested against.
# --------------------------------------------------------------------------

def resample_fenceposts(t_coarse, weights, num_fine, u=None, pdf_floor=1e-5):
    """Sorted union of the coarse fenceposts [N,S_c] with num_fine fenceposts drawn by inverse
    transform sampling from the piecewise-constant PDF of weights [N,S_c-1]."""
    w = weights + pdf_floor
    cdf = torch.cumsum(w, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf / cdf[..., -1:]], dim=-1)
    cdf[..., -1] = 1.0
    if u is None:
        u = ((torch.arange(num_fine, dtype=t_coarse.dtype) + 0.5) / num_fine).expand(
            t_coarse.shape[0], num_fine)
    idx = (torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True) - 1).clamp(
        0, weights.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, idx), torch.gather(cdf, -1, idx + 1)
    t0, t1 = torch.gather(t_coarse, -1, idx), torch.gather(t_coarse, -1, idx + 1)
    den = c1 - c0
    frac = torch.where(den > 0, (u - c0) / den, torch.zeros_like(den))
    t_fine = t0 + frac * (t1 - t0)
    return torch.sort(torch.cat([t_coarse, t_fine], dim=-1), dim=-1).values


def render_rays_t(params, cfg, rays_o, rays_d, t):
    """render_rays on explicit fenceposts t [N,S] (deterministic): rgb, seg, weights."""
    means, covs, h, density, color, seg = field(params, cfg, rays_o, rays_d, t)
    weights = composite_weights(means, density)
    rgb = (weights * torch.sigmoid(color)).sum(dim=-2)
    seg_out = (torch.log(weights + 1e-10) + torch.log_softmax(seg, dim=-1)).logsumexp(dim=-2)
    return rgb, seg_out, weights[..., 0]


def render_rays_hierarchical(params, cfg, rays_o, rays_d, num_coarse, num_fine):
    """Coarse + fine (sorted union) deterministic render: rgb [N,2,3], seg [N,2,50], t_union."""
    t_c = sample_t(params, rays_o.shape[0], num_coarse)
    rgb_c, seg_c, w_c = render_rays_t(params, cfg, rays_o, rays_d, t_c)
    t_u = resample_fenceposts(t_c, w_c, num_fine)
    rgb_f, seg_f, _ = render_rays_t(params, cfg, rays_o, rays_d, t_u)
    return torch.stack([rgb_c, rgb_f], dim=-2), torch.stack([seg_c, seg_f], dim=-2), t_u

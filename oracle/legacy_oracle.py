"""CPU restatement of the LEGACY (generation-A) network whose trained weights ship as
``examples/nerf.pth`` — TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.

The source of that network is NOT in the reference repository (SURVEY.md sections 0.4, 2.3): only
the checkpoint's tensor names / shapes and two dead fragments of ``nerf/model.py`` survive.  So this
file restates a STRUCTURE RECOVERED FROM THE CHECKPOINT, and every constant that the checkpoint cannot
settle is a named parameter of ``default_config()`` — a stated choice, not recovered behaviour:

  recovered from ``examples/nerf.pth`` (44 tensors):
    block_0  Linear 60->256, then 3 x Linear 256->256; after each: parameter-less slot, LayerNorm(256)
             -> [Linear, activation, LayerNorm] x 4               (keys .0/.3/.6/.9 and .2/.5/.8/.11)
    block_1  the same with a first Linear of 316 = 256 + 60 inputs  (skip-concatenation of PE(x))
    density  Linear 256->1 on block_1's output
    block_2  Linear 292 = 256 + 36 -> 256, Linear 256->256, each followed by activation, LayerNorm
    color    Linear 256->3 on block_2's output
  from the dead ``NeRF.positional_encoding`` (nerf/model.py:197-240): per coordinate
    [sin(x f_0) .. sin(x f_{F-1}), cos(x f_0) .. cos(x f_{F-1})], f_k = 2^k * multiplier, flattened
    coordinate-major (60 = 3 x 2 x 10 for positions, 36 = 3 x 2 x 6 for directions)
  from the notebook's call sites (examples/example.ipynb cells 6, 8): NeRF(normalize_position=6.0),
    render_rays(o, d, near=2, far=6, S, randomly_sample, density_noise_std) -> [N, 3]
  compositing: nerf/model.py:438-469, :660 (the only compositing code in the repository)

  NOT recoverable, chosen here (SURVEY.md section 2.3's probe renders a recognisable Lego bulldozer with
  them): activation = ReLU; frequency multiplier = pi; positions divided by normalize_position;
  skip / view concatenation order = [hidden, encoding]; view directions normalised before encoding.
"""
import math

import torch
import torch.nn.functional as F

BLOCK0 = ("block_0.0", "block_0.3", "block_0.6", "block_0.9")
BLOCK1 = ("block_1.0", "block_1.3", "block_1.6", "block_1.9")
BLOCK2 = ("block_2.0", "block_2.3")


def norm_key(linear_key):
    block, slot = linear_key.split(".")
    return f"{block}.{int(slot) + 2}"


def default_config():
    return dict(normalize_position=6.0, position_freqs=10, direction_freqs=6, multiplier=math.pi,
                activation="relu", concat_order="hidden_first", normalize_directions=True)


def state_dict_keys():
    """The checkpoint's 44 keys in the order the kernels' pack routine takes them."""
    keys = []
    for lin in BLOCK0 + BLOCK1:
        keys += [lin + ".weight", lin + ".bias", norm_key(lin) + ".weight", norm_key(lin) + ".bias"]
    keys += ["density.weight", "density.bias"]
    for lin in BLOCK2:
        keys += [lin + ".weight", lin + ".bias", norm_key(lin) + ".weight", norm_key(lin) + ".bias"]
    keys += ["color.weight", "color.bias"]
    return keys


def init_params(seed=0):
    """PyTorch default initialisation of the recovered module tree (for tests without the checkpoint)."""
    gen = torch.Generator().manual_seed(seed)
    shapes = {"block_0.0": (256, 60), "block_1.0": (256, 316), "block_2.0": (256, 292),
              "density": (1, 256), "color": (3, 256)}
    params = {}
    for lin in BLOCK0 + BLOCK1 + BLOCK2 + ("density", "color"):
        out_f, in_f = shapes.get(lin, (256, 256))
        bound = 1.0 / math.sqrt(in_f)
        params[lin + ".weight"] = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
        params[lin + ".bias"] = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
        if lin not in ("density", "color"):
            params[norm_key(lin) + ".weight"] = 1.0 + 0.1 * (torch.rand(256, generator=gen) - 0.5)
            params[norm_key(lin) + ".bias"] = 0.1 * (torch.rand(256, generator=gen) - 0.5)
    return params


def positional_encoding(x, num_freqs, multiplier):
    """[..., 3] -> [..., 3 * 2 * num_freqs], layout of nerf/model.py:221-240."""
    freqs = multiplier * torch.pow(2.0, torch.arange(num_freqs, dtype=x.dtype))
    arg = x.unsqueeze(-1) * freqs
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1).flatten(start_dim=-2)


def _act(x, cfg):
    return F.relu(x) if cfg["activation"] == "relu" else F.gelu(x)


def _block(params, keys, h, cfg, gates=None, record=None):
    """``gates``: an iterator of boolean masks, one per layer, that REPLACE the activation's own decision
    (relu(y) becomes y * gate): the gradient tests take them from the kernel under test, so that both sides
    differentiate the same piecewise-linear function (a ReLU gate within rounding of zero is a discontinuity
    of the gradient, not an error of either side).  ``record``: list that receives each layer's own gates."""
    for lin in keys:
        h = F.linear(h, params[lin + ".weight"], params[lin + ".bias"])
        if record is not None:
            record.append((h > 0).detach())
        a = _act(h, cfg) if gates is None else h * next(gates).to(h.dtype)
        h = F.layer_norm(a, (256,), params[norm_key(lin) + ".weight"], params[norm_key(lin) + ".bias"], 1e-5)
    return h


def _concat(h, enc, cfg):
    return torch.cat([h, enc] if cfg["concat_order"] == "hidden_first" else [enc, h], dim=-1)


def field(params, cfg, points, directions, gates=None, record=None):
    """density [..., 1], color logits [..., 3] at `points` seen along `directions`.  ``gates`` / ``record``:
    the ten wide layers' ReLU gates in the order block_0, block_1, block_2 (see ``_block``)."""
    gates = None if gates is None else iter(gates)
    pe_x = positional_encoding(points / cfg["normalize_position"], cfg["position_freqs"], cfg["multiplier"])
    d = directions
    if cfg["normalize_directions"]:
        d = d / d.norm(dim=-1, keepdim=True)
    pe_d = positional_encoding(d, cfg["direction_freqs"], cfg["multiplier"])
    h = _block(params, BLOCK0, pe_x, cfg, gates, record)
    h = _block(params, BLOCK1, _concat(h, pe_x, cfg), cfg, gates, record)
    density = F.linear(h, params["density.weight"], params["density.bias"])
    hv = _block(params, BLOCK2, _concat(h, pe_d, cfg), cfg, gates, record)
    color = F.linear(hv, params["color.weight"], params["color.bias"])
    return density, color


def sample_t(n_rays, near, far, num_samples, u=None):
    """Linear sample positions in [near, far] (Mildenhall et al. 2020), stratified by `u` [N, S]."""
    t = torch.linspace(float(near), float(far), num_samples).expand(n_rays, num_samples)
    if u is not None:
        mid = 0.5 * (t[..., 1:] + t[..., :-1])
        lower = torch.cat([t[..., :1], mid], dim=-1)
        upper = torch.cat([mid, t[..., -1:]], dim=-1)
        t = lower + (upper - lower) * u
    return t


def render_rays(params, cfg, rays_o, rays_d, near, far, num_samples, u=None, noise=None,
                density_noise_std=0.0, return_stages=False, gates=None, record=None):
    """-> rgb [N, 3].  S samples = S network evaluations (points, not intervals); compositing as
    nerf/model.py:438-469 / :660 on the sample points."""
    n = rays_o.shape[0]
    t = sample_t(n, near, far, num_samples, u)
    points = rays_o[:, None, :] + rays_d[:, None, :] * t[..., None]
    density, color = field(params, cfg, points, rays_d[:, None, :].expand_as(points),
                           None if gates is None else gates[:10], record)
    if noise is not None:
        density = density + noise * density_noise_std
    gaps = points[..., 1:, :] - points[..., :-1, :]
    dists = F.pad(torch.linalg.norm(gaps, dim=-1, keepdim=True), (0, 0, 0, 1), value=1e10)
    # an ELEVENTH gate: the ReLU of the (noisy) density the compositing runs with — with a noise draw a density can sit
    # within rounding of zero like any pre-activation (gates[10] forces it, record receives the oracle's own)
    if record is not None:
        record.append((density > 0).detach())
    dens_pos = F.relu(density) if gates is None or len(gates) <= 10 else density * gates[10].to(density.dtype)
    trans = torch.exp(-dens_pos * dists)
    weights = (1.0 - trans) * F.pad(torch.cumprod(trans[..., :-1, :] + 1e-10, dim=-2), (0, 0, 1, 0), value=1.0)
    rgb = (weights * torch.sigmoid(color)).sum(dim=-2)
    if return_stages:
        return rgb, dict(t=t, points=points, density=density, color=color, weights=weights)
    return rgb


def flops_per_sample():
    macs = 60 * 256 + 3 * 256 * 256 + 316 * 256 + 3 * 256 * 256 + 256 + 292 * 256 + 256 * 256 + 3 * 256
    return 2 * macs

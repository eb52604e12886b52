"""Gate-aware gradient comparison for the main network (tests/test_legacy_backward.py does the same for the legacy one).

A parameter gradient is discontinuous in every ReLU gate, and a gate whose pre-activation sits within rounding of zero
falls on either side depending on the summation order of the LayerNorm that feeds it — one such gate moves a gradient
tensor by 1e-4 of its largest element (round 5: the first G11 fixture).  So the comparison is split in two:

* the ARITHMETIC: the oracle differentiates with the gates the KERNEL ran with (read from the workspace its training
  forward saved: fma(x_hat, gamma, beta) > 0 of the five hidden layers and density + noise > 0 of the compositing,
  tests/workspace_mirror.py: saved_gates / saved_density_gate), i.e. both sides differentiate
  the same piecewise-linear function — every gradient tensor must agree to ``grad_bound`` of its largest element;
* the GATES: the kernel's gates against the oracle's own — they may differ only in a bounded share of elements
  (``flip_bound``), the ones within rounding of zero.

No fixture has to be picked for having no borderline gate."""
import torch

import workspace_mirror as W

GRAD_BOUND = 5e-6            # on top of the fp32 oracle's own distance from the fp64 oracle (same forced gates: arithmetic only)
FLIP_BOUND = 1e-5            # share of gates that may differ from the oracle's own


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def rays_with_weight(params, cfg, o, d, num_samples, u, noise, noise_std, least=0.02):
    """Mask [n] of the rays whose total compositing weight (fp64 oracle) is at least ``least``.  The segmentation
    output is log(w + 1e-10) + ..., whose gradient carries 1 / w: on a ray of total weight 5e-3 whose one contributing
    sample has a density that all but cancels its noise draw (sigma = 1.2 - 1.195), the 1e-5 with which ANY fp32
    forward rounds that density is 4e-4 of sigma, of w and of every gradient the ray's segmentation term feeds — the
    fp32 oracle itself moves by that much from one summation order to the next (tests/diag_seg_grad.py; profiles/
    r06_d_seg_gradient_diag.log).  Gradient tests put their segmentation loss on the rays above the threshold."""
    from oracle import nerf_oracle as O
    with torch.no_grad():
        p64 = {k: v.double() for k, v in params.items()}
        st = O.render_rays(p64, cfg, o.double(), d.double(), num_samples, u=None if u is None else u.double(),
                           noise=None if noise is None else noise.double(), density_noise_std=noise_std,
                           return_stages=True)[2]
    return st["weights"].sum(dim=(1, 2)) >= least


def caster(p):
    """t -> t in the dtype of the parameter dict ``p`` (the oracle runs in fp32 and in fp64)."""
    dtype = next(v for v in p.values()).dtype
    return lambda t: t.to(dtype)


def kernel_gates(model, params, n_rays, num_samples):
    """Gates of the training forward ``model`` just ran (``model.keep_workspace`` must have been set before it)."""
    return W.saved_gates(model.last_workspace, params, n_rays, num_samples) + \
        [W.saved_density_gate(model.last_workspace, params, n_rays, num_samples)]


def oracle_gradients(params, loss_fn, gates=None, record=None, dtype=torch.float32):
    p = {k: v.to(dtype).clone().requires_grad_(k.startswith("prediction")) for k, v in params.items()}
    loss = loss_fn(p, gates, record)
    loss.backward()
    return float(loss.detach()), {k: v.grad.float() for k, v in p.items() if v.grad is not None}


def check(model, params, n_rays, num_samples, loss_fn, grad_bound=GRAD_BOUND, flip_bound=FLIP_BOUND, tag="", gates=None):
    """``loss_fn(p, gates, record)``: the oracle's loss on parameter dict ``p`` (in ITS dtype: the function casts its
    other inputs to ``next(iter(p.values())).dtype``), passing ``gates`` / ``record`` through to
    oracle.nerf_oracle.mlp.  ``model`` holds the kernel's gradients (p.grad) of the same loss.  Every tensor must lie
    within ``grad_bound`` + 4 x (the fp32 oracle's own distance from the fp64 oracle, both on the kernel's gates: the
    largest of four fp32 evaluations) of the fp64 or the fp32 oracle on the kernel's gates, relative to its largest
    element.  ``gates``: the kernel's six gates when the caller read them itself (a loss without compositing has no
    density gate in the workspace).
    Returns (flips, total, worst error, the fp32 oracle's gradients on its OWN gates, and on the KERNEL's gates)."""
    if gates is None:
        gates = kernel_gates(model, params, n_rays, num_samples)
    own = []
    _, plain = oracle_gradients(params, loss_fn, record=own)
    _, ref32 = oracle_gradients(params, loss_fn, gates=gates)
    _, ref64 = oracle_gradients(params, loss_fn, gates=gates, dtype=torch.float64)
    # the floor: what fp32 ROUNDING does to these gradients.  One fp32 evaluation is one draw from a heavy-tailed
    # distribution when a ray of little weight amplifies the forward's rounding by 1 / w (rays_with_weight above), so
    # three more draws are taken — the fp32 oracle on parameters moved by one ulp at random (a change of 6e-8 in the
    # true gradient) — and the largest distance from the fp64 oracle counts
    instances = [ref32]
    gen = torch.Generator().manual_seed(12345)
    for _ in range(3):
        moved = {k: (v * (1.0 + (torch.randint(0, 2, v.shape, generator=gen).to(v.dtype) * 2 - 1) * 2.0 ** -23)
                     if k.startswith("prediction") else v) for k, v in params.items()}
        instances.append(oracle_gradients(moved, loss_fn, gates=gates)[1])
    assert len(own) == len(gates) == 6 and all(a.shape == b.shape for a, b in zip(own, gates))      # five LayerNorm-ReLU gates + the density's
    flips = sum(int((a != b).sum()) for a, b in zip(own, gates))
    total = sum(a.numel() for a in gates)
    assert flips <= max(flip_bound * total, 2), (flips, total)
    worst, table, bad = 0.0, [], []
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == ref64[k].shape, k
        floor = max(rel_err(inst[k], ref64[k]) for inst in instances)
        # (against the nearer of the two oracles: what fp32 does to the encoding's large arguments moves layer 0's
        #  gradient by 4e-5 for kernel and fp32 oracle alike, while an accumulation over thousands of samples lands
        #  nearer the fp64 one)
        e64, e32 = rel_err(p.grad.cpu(), ref64[k]), rel_err(p.grad.cpu(), ref32[k])
        e = min(e64, e32)
        worst = max(worst, e)
        table.append(f"{k:32s} vs fp64 {e64:.2e}  vs fp32 {e32:.2e}  fp32-vs-fp64 floor {floor:.2e}")
        if e > grad_bound + 4.0 * floor:
            bad.append(k)
    assert not bad, f"[{tag}] {bad}\n" + "\n".join(table)
    print(f"[{tag}] worst relative gradient error on the kernel's gates {worst:.2e} (fp32 oracle, same gates: "
          f"{max(rel_err(inst[k], ref64[k]) for inst in instances for k in ref64):.2e}); {flips} of {total} gates differ from the oracle's own "
          f"(against the oracle on ITS gates: {max(rel_err(p.grad.cpu(), plain[k]) for k, p in model.named_parameters()):.2e})")
    return flips, total, worst, plain, ref32

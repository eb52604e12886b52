"""Python mirror of the training workspace layout (nerf_amd/csrc/nerf_device.h: TrainLayout) — what
the training forward saves for the backward.  Debugging and stage-parity aid: tests read the encoded
inputs ``h``, the normalised activations ``x_hat`` and the LayerNorm statistics of every layer
straight from the buffer the kernels wrote (``model.keep_workspace = True`` keeps it after a
forward).  The product never reads the workspace from Python.
"""
import torch

HIDDEN, ENC_IN, OUT_PAD = 256, 96, 64
SAMPLES_PER_WAVE = 16


def chunks_of(num_samples):
    return (num_samples - 1 + SAMPLES_PER_WAVE - 1) // SAMPLES_PER_WAVE


def train_width(hidden_size):
    """Features per saved x_hat / dY row: 128 when the network trains at 8 register tiles (hidden_size <= 128,
    nerf_device.h: train_tiles), else 256."""
    return 128 if hidden_size <= 128 else HIDDEN


def train_layout(n_rays, num_samples, width=HIDDEN):
    """Offsets (in floats) of the saved tensors, like ``make_train_layout``: ``mp`` padded samples
    = ceil4(n_rays) * chunks * 16; row tensors are [mp, features] (``width`` for x_hat / dY)."""
    chunks = chunks_of(num_samples)
    mp = (n_rays + 3) // 4 * 4 * chunks * 16
    lay, off = {"mp": mp, "chunks": chunks, "width": width}, 0
    lay["h"] = off
    off += mp * ENC_IN
    lay["dy"] = []
    for _ in range(5):
        lay["dy"].append(off)
        off += mp * width
    lay["dy5"] = off
    off += mp * OUT_PAD
    lay["xhat"] = []
    for _ in range(5):
        lay["xhat"].append(off)
        off += mp * width
    lay["rstd"] = []
    for _ in range(5):
        lay["rstd"].append(off)
        off += mp
    lay["out"] = off
    off += mp * OUT_PAD
    lay["comp"] = off
    off += mp * 4
    lay["total"] = off
    return lay


def layer0_feature_order():
    """Column c = 16 t + 4 g + r of the saved ``h`` rows holds the reference's feature
    ``layer0_source_feature(t, g, r)`` (nerf_layout.h): [sin: scale-major x coord-minor | shifted]."""
    order = []
    for c in range(ENC_IN):
        t, g, r = c // 16, (c % 16) // 4, c % 4
        q = 4 * t + r
        part, p = q // 12, q % 12
        order.append(part * 48 + 3 * (4 * g + p // 3) + p % 3)
    return torch.tensor(order, dtype=torch.int64)


def _row_major(workspace, lay, offset, width):
    """The [mp, width] rows of one saved tensor.  Narrow tensors are stored like that; the 256-wide ones (x_hat, dY)
    are TILE-MAJOR (nerf_device.h: tile_lane_base): a wave's 16 samples as [16 register tiles T][g][sample][4 features],
    i.e. element (sample s, feature f) of tile n at n * 4096 + (f >> 4) * 256 + ((f & 15) >> 2) * 64 + s * 4 + (f & 3)."""
    mp = lay["mp"]
    flat = workspace[offset:offset + mp * width]
    if width not in (HIDDEN, 128):
        return flat.view(mp, width)
    tiles = flat.view(mp // 16, width // 16, 4, 16, 4)       # [n][T][g][s][r]
    return tiles.permute(0, 3, 1, 2, 4).reshape(mp, width)   # -> [n][s][T][g][r]


def _rows(workspace, lay, offset, width, n_rays, num_samples):
    """[mp, width] rows -> [n_rays, S-1, width] (padded ray slots and samples dropped)."""
    chunks, mp = lay["chunks"], lay["mp"]
    rows = _row_major(workspace, lay, offset, width).view(mp // (chunks * 16), chunks * 16, width)
    return rows[:n_rays, :num_samples - 1]


def saved_h(workspace, n_rays, num_samples):
    """Encoded inputs [n_rays, S-1, 96] in the REFERENCE's feature order."""
    lay = train_layout(n_rays, num_samples)
    h = _rows(workspace, lay, lay["h"], ENC_IN, n_rays, num_samples)
    out = torch.empty_like(h)
    out[..., layer0_feature_order().to(h.device)] = h
    return out


def saved_xhat(workspace, layer, n_rays, num_samples, width=HIDDEN):
    """Normalised pre-affine activations of hidden layer ``layer`` (0..4): [n_rays, S-1, width]."""
    lay = train_layout(n_rays, num_samples, width)
    return _rows(workspace, lay, lay["xhat"][layer], width, n_rays, num_samples)


def saved_rstd(workspace, layer, n_rays, num_samples, width=HIDDEN):
    """1 / sqrt(var + eps) of hidden layer ``layer``: [n_rays, S-1]."""
    lay = train_layout(n_rays, num_samples, width)
    return _rows(workspace, lay, lay["rstd"][layer], 1, n_rays, num_samples)[..., 0]


def saved_density_gate(workspace, params, n_rays, num_samples):
    """The ReLU gate of the (noisy) density the compositing ran with and its backward differentiates through:
    slot 3 of the compositing state [sp][4] = (alpha, T_exclusive, dist, density + noise) > 0; [n_rays, S-1, 1]."""
    hidden = params["prediction_heads.1.weight"].shape[0]
    lay = train_layout(n_rays, num_samples, train_width(hidden))
    comp = _rows(workspace, lay, lay["comp"], 4, n_rays, num_samples)
    return (comp[..., 3:4] > 0).cpu()


def saved_gates(workspace, params, n_rays, num_samples):
    """The five ReLU gates [n_rays, S-1, hidden_size] the training forward ran with AND its backward differentiates
    through: both evaluate fma(x_hat, gamma, beta) > 0 on the saved x_hat (nerf_fused.h: normalize_tile,
    nerf_backward_common.h: layer_norm_relu_bwd).  Evaluated here in float64, where the product of two fp32 values
    is exact and the sum's sign is the fused operation's sign."""
    hidden = params["prediction_heads.1.weight"].shape[0]
    width = train_width(hidden)
    gates = []
    for layer, slot in enumerate((1, 4, 7, 10, 13)):
        xhat = saved_xhat(workspace, layer, n_rays, num_samples, width)[..., :hidden].double().cpu()
        z = xhat * params[f"prediction_heads.{slot}.weight"].double().cpu() + params[f"prediction_heads.{slot}.bias"].double().cpu()
        gates.append(z > 0)
    return gates


# ---- the legacy 8 x 256 network's workspace (nerf_amd/csrc/nerf_legacy_layout.h: LegacyTrainLayout) ----------
LEGACY_WIDE, LEGACY_ENC_PAD = 10, 64


def legacy_chunks_of(num_samples):
    return (num_samples + SAMPLES_PER_WAVE - 1) // SAMPLES_PER_WAVE      # S samples = S evaluations here


def legacy_train_layout(n_rays, num_samples):
    """Offsets (in floats) like ``make_legacy_train_layout``: encodings, then per wide layer a_hat rows, 1/std and
    the gate threshold ``shift``, then the head outputs and the compositing state.  (dY rows are the backward's.)"""
    chunks = legacy_chunks_of(num_samples)
    mp = (n_rays + 3) // 4 * 4 * chunks * 16
    lay, off = {"mp": mp, "chunks": chunks}, 0
    for name in ("pos", "dir"):
        lay[name] = off
        off += mp * LEGACY_ENC_PAD
    for name, width in (("xhat", HIDDEN), ("rstd", 1), ("shift", 1)):
        lay[name] = []
        for _ in range(LEGACY_WIDE):
            lay[name].append(off)
            off += mp * width
    lay["out"] = off
    off += mp * OUT_PAD
    lay["comp"] = off
    off += mp * 4
    lay["total"] = off
    return lay


def _legacy_rows(workspace, lay, offset, width, n_rays, num_samples):
    chunks, mp = lay["chunks"], lay["mp"]
    rows = _row_major(workspace, lay, offset, width).view(mp // (chunks * 16), chunks * 16, width)
    return rows[:n_rays, :num_samples]


def legacy_saved_layer(workspace, layer, n_rays, num_samples):
    """(a_hat [n, S, 256], 1/std [n, S], shift [n, S]) of wide layer ``layer`` (0..3 block_0, 4..7 block_1,
    8..9 block_2): a_hat = (relu(y) - mean) / std, ``shift`` = the a_hat of a closed gate."""
    lay = legacy_train_layout(n_rays, num_samples)
    assert workspace.numel() == lay["total"], (workspace.numel(), lay["total"])
    return (_legacy_rows(workspace, lay, lay["xhat"][layer], HIDDEN, n_rays, num_samples),
            _legacy_rows(workspace, lay, lay["rstd"][layer], 1, n_rays, num_samples)[..., 0],
            _legacy_rows(workspace, lay, lay["shift"][layer], 1, n_rays, num_samples)[..., 0])


def legacy_saved_gates(workspace, n_rays, num_samples):
    """The ReLU gates the kernels differentiate through, one bool [n, S, 256] per wide layer: y > 0 <=> a_hat >
    shift (exact: the forward moves an open gate whose a_hat rounds onto ``shift`` one ulp up)."""
    gates = []
    for layer in range(LEGACY_WIDE):
        xhat, _, shift = legacy_saved_layer(workspace, layer, n_rays, num_samples)
        gates.append(xhat > shift[..., None])
    return gates


def legacy_saved_density_gate(workspace, n_rays, num_samples):
    """The ReLU gate of the (noisy) density the legacy compositing ran with: slot 3 of its state [sp][4] = (alpha,
    T_exclusive, dist, density + noise) > 0; [n, S, 1]."""
    lay = legacy_train_layout(n_rays, num_samples)
    comp = _legacy_rows(workspace, lay, lay["comp"], 4, n_rays, num_samples)
    return (comp[..., 3:4] > 0).cpu()

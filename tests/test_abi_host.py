"""CPU tests of the boundary: the C-ABI library builds/loads without a GPU and exports every
symbol include/nerf_hip.h declares; the Python mirror keeps the reference's call surface."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_hip_[a-z_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def handle():
    from nerf_amd import build as nerf_build
    nerf_build.build()
    from nerf_amd import _lib
    return _lib.lib()


def test_library_exports_every_declared_symbol(handle):
    from nerf_amd import _lib
    names = declared_symbols()
    assert len(names) >= 7
    for name in names:
        assert hasattr(handle, name), name
    assert set(_lib.EXPORTS) == set(names)
    assert handle.nerf_hip_version() == _lib.ABI_VERSION == 8
    # packed image = {74 forward stages + 3,904 small floats + 68 transposed stages of 16 KiB} x {fp32, f16 pairs}
    # + four bound constants + the narrow images: fp32 forward (21 stages at 8 register tiles, 7 at 4), transposed
    # fp32 (18 at 8), f16-pair forward (21) and transposed (18) at 8, transposed fp32 at 4 (5)
    assert handle.nerf_hip_packed_bytes() == 2 * (74 * 16384 + 3904 * 4 + 68 * 16384) + 16 + (21 + 7 + 18 + 21 + 18 + 5) * 16384
    ge = handle.nerf_hip_grad_elements
    assert ge(256, 96, 54) == 304438                          # the reference's defaults: hidden 256, 3 x 32 inputs, 1 + 3 + 50 outputs
    assert ge(256, 96, 4) == 304438 - 50 * 257 and ge(256, 96, 64) == 304438 + 10 * 257
    assert ge(256, 96, 1) == 0 and ge(256, 96, 65) == 0 and ge(256, 96, 2) == 304438 - 52 * 257      # (density + 1 color)
    # narrower networks (run zero-padded inside the kernels): sum of the 22 tensors' PyTorch sizes
    for hid, enc, n_out in ((128, 96, 54), (256, 48, 54), (64, 48, 11), (17, 6, 4)):
        want = hid * enc + hid + 4 * (hid * hid + hid) + 10 * hid + n_out * hid + n_out
        assert ge(hid, enc, n_out) == want, (hid, enc, n_out)
    assert ge(257, 96, 54) == 0 and ge(256, 97, 54) == 0 and ge(256, 102, 54) == 0 and ge(0, 96, 54) == 0
    assert handle.nerf_hip_train_workspace_bytes(4096, 64) == 4096 * 64 * 2793 * 4
    assert handle.nerf_hip_train_workspace_bytes(0, 64) == 0


def header_fields(text, name):
    body = text[text.index(f"typedef struct {name}"):text.index("} " + name + ";")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = re.sub(r"\[[^\]]*\]", "", decl).split(",")          # array bounds carry no name
        fields.append(names[0].split()[-1])
        fields += [n.strip() for n in names[1:]]
    return [f.lstrip("*") for f in fields]


def test_args_structs_match_the_header():
    """Field order/names of every ctypes mirror == its C struct, and the array bounds they share (guards silent
    ABI drift)."""
    from nerf_amd import _lib
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    for c_name, mirror in (("NerfHipRenderArgs", _lib.RenderArgs), ("NerfHipAdamArgs", _lib.AdamArgs),
                           ("NerfHipMseArgs", _lib.MseArgs), ("NerfHipLegacyBackwardArgs", _lib.LegacyBackwardArgs)):
        assert header_fields(text, c_name) == [f[0] for f in mirror._fields_], c_name
    defines = dict(re.findall(r"#define (NERF_HIP_\w+) (\d+)", text))
    assert int(defines["NERF_HIP_ADAM_MAX_TENSORS"]) == _lib.ADAM_MAX_TENSORS
    assert int(defines["NERF_HIP_ADAM_STEP_SLOTS"]) == _lib.ADAM_STEP_SLOTS
    assert int(defines["NERF_HIP_ABI_VERSION"]) == _lib.ABI_VERSION


def test_argument_errors_do_not_touch_the_gpu(handle):
    from nerf_amd import _lib
    assert handle.nerf_hip_render_forward(None, None) == -1
    assert b"null args" in handle.nerf_hip_last_error()
    args = _lib.RenderArgs()
    args.n_rays, args.num_samples = 8, 1
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -1
    args.n_rays = 0                       # empty batch is a no-op, not an error
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == 0
    # unknown precision: refused before any HIP call
    dummy = ctypes.c_void_p(16)
    args = _lib.RenderArgs()
    args.n_rays, args.num_samples = 8, 4
    args.rays_o = args.rays_d = args.packed = args.rgb = args.t_table = dummy
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -2      # num_outputs unset (0): unsupported shape
    assert b"num_outputs" in handle.nerf_hip_last_error()
    args.num_outputs = 65
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -2
    args.num_outputs, args.color_outputs = 54, 13              # more color channels than output tile 0 holds
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -2
    args.num_outputs, args.color_outputs = 5, 5                # 1 + colors > rows
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -2
    args.num_outputs, args.color_outputs = 54, 0               # 0 = the reference's 3
    args.precision = 7
    assert handle.nerf_hip_render_forward(ctypes.byref(args), None) == -1
    assert b"precision" in handle.nerf_hip_last_error()
    assert handle.nerf_hip_build_flags() == b""          # the product build carries no experiment macro
    assert handle.nerf_hip_pack_weights(None, 256, 96, 54, 3, None, None) == -1
    assert handle.nerf_hip_render_backward(None, None) == -1
    bargs = _lib.BackwardArgs()
    assert handle.nerf_hip_render_backward(ctypes.byref(bargs), None) == -1
    # the training loop's optimiser and loss launches: argument errors are caught on the host too
    assert handle.nerf_hip_adam_step(None, None) == -1 and handle.nerf_hip_mse_loss(None, None) == -1
    adam = _lib.AdamArgs()
    assert handle.nerf_hip_adam_step(ctypes.byref(adam), None) == -1           # no tensors
    adam.num_tensors, adam.total = 1, 8
    adam.offsets[1] = 8
    adam.exp_avg = adam.exp_avg_sq = adam.step = ctypes.cast(dummy, _lib._f32p)
    assert handle.nerf_hip_adam_step(ctypes.byref(adam), None) == -1           # null parameter / gradient
    assert b"null tensor" in handle.nerf_hip_last_error()
    adam.offsets[1] = 7
    adam.params[0] = adam.grads[0] = 16
    assert handle.nerf_hip_adam_step(ctypes.byref(adam), None) == -1           # offsets do not end at total
    mse = _lib.MseArgs()
    mse.n_rays, mse.stages = 4, 0
    assert handle.nerf_hip_mse_loss(ctypes.byref(mse), None) == -1
    mse.stages = 1
    mse.loss = ctypes.cast(dummy, _lib._f32p)
    assert handle.nerf_hip_mse_loss(ctypes.byref(mse), None) == -1             # rays but no tensors
    assert b"null tensor" in handle.nerf_hip_last_error()


def test_mirror_keeps_reference_call_surface():
    from nerf_amd.model import NeRF
    import nerf_amd.model as m
    for fn in ("expected_sin", "lift_gaussian", "conical_frustum_to_gaussian", "cast_rays",
               "integrated_pos_enc"):
        assert callable(getattr(m, fn))
    sig = inspect.signature(NeRF.__init__)
    assert list(sig.parameters)[1:] == ["color_outputs", "segmentation_outputs", "hidden_size",
                                        "encoding_size", "focal_length", "min_x", "max_x", "min_y",
                                        "max_y", "min_z", "max_z"]
    assert list(inspect.signature(NeRF.render_rays).parameters)[:8] == [
        "self", "rays_o", "rays_d", "num_samples", "states_x", "states_d", "randomly_sample",
        "density_noise_std"]
    assert list(inspect.signature(NeRF.render_image).parameters)[:12] == [
        "self", "camera_o", "camera_r", "image_h", "image_w", "focal_length", "num_samples",
        "states_x", "states_d", "max_chunk_size", "randomly_sample", "density_noise_std"]
    assert list(inspect.signature(NeRF.forward).parameters) == [
        "self", "rays_o", "rays_d", "samples", "states_x", "states_d"]
    torch.manual_seed(0)
    model = NeRF()
    from conftest import golden_params
    ref = golden_params()
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:                         # same init under the same seed as the reference
        assert torch.equal(sd[k], ref[k]), k


def test_host_helpers_match_reference_fixtures():
    from conftest import load_golden
    from nerf_amd.model import NeRF
    g = load_golden("g7_statics")
    assert torch.equal(NeRF.generate_rays(5, 7, 112.0), g["rays_5x7"])
    assert torch.equal(NeRF.spherical_to_cartesian(g["yaw"], g["elevation"]), g["cartesian"])
    rot = NeRF.get_rotation_matrix(g["cartesian"], g["up"])
    assert torch.equal(rot, g["rotation"])
    _, wd = NeRF.rays_to_world_coordinates(g["rays_5x7"][None], g["cartesian"][:, None, None, :] * 2.0,
                                           rot[:, None, None, :, :])
    assert torch.equal(wd, g["world_d"])
    model = NeRF()
    t = model.sample_along_rays(torch.zeros(2, 3), torch.zeros(2, 3), 64, randomly_sample=False)
    assert torch.equal(t[1], g["t64"])
    g1 = load_golden("g1_stages")
    w = NeRF.alpha_compositing_coefficients(g1["means"], g1["density"])
    # (exp / cumprod on the host: identical on the machine that made the fixture, last-bit
    #  differences between CPU models)
    assert (w - g1["weights"]).abs().max() <= 1e-7


def test_no_cpu_fallback():
    from nerf_amd.model import NeRF
    model = NeRF()
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.render_rays(torch.zeros(4, 3), torch.ones(4, 3), 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.render_image(torch.zeros(1, 3), torch.eye(3)[None], 4, 4, 4.0, 8)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: the product may CITE it in comments (which file states the spec
    of an unpinned row), but never import, load or execute anything under it."""
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|\boracle\.[A-Za-z_]+\(|import_module\([^)]*oracle|"
                     r"__import__\([^)]*oracle|sys\.path[^\n]*oracle|open\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nerf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f
    for f in os.listdir(os.path.join(ROOT, "scripts")):              # measurement helpers: same rule
        if f.endswith(".py"):
            assert not bad.search(open(os.path.join(ROOT, "scripts", f)).read()), f
    assert bad.search("from oracle import nerf_oracle as O") and bad.search("  import oracle.legacy_oracle")


def test_missing_library_fails_loudly(tmp_path):
    """No CPU/eager fallback: without the HIP library the package refuses to render."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['NERF_HIP_LIB'] = %r\n"
            "from nerf_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU' in str(e) or 'missing' in str(e))\n"
            % (ROOT, str(tmp_path / "absent.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr

"""ctypes binding of libnerf_hip.so (C ABI: include/nerf_hip.h).

There is no fallback: if the library is missing or a call fails this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_HIP_LIB selects another build of the same ABI (an experimental variant: scripts/ab_libs.py)
LIB_PATH = os.environ.get("NERF_HIP_LIB") or os.path.join(_HERE, "csrc", "libnerf_hip.so")
ABI_VERSION = 8
NUM_PARAM_TENSORS = 22
PRECISIONS = {"fp32": 0, "f16x3": 1}      # NERF_HIP_PRECISION_*

_f32p = ctypes.c_void_p


class RenderArgs(ctypes.Structure):
    """Mirror of NerfHipRenderArgs (include/nerf_hip.h)."""
    _fields_ = [
        ("rays_o", _f32p), ("rays_d", _f32p),
        ("camera_o", _f32p), ("camera_r", _f32p),
        ("image_h", ctypes.c_int32), ("image_w", ctypes.c_int32),
        ("focal_length", ctypes.c_float),
        ("ray_begin", ctypes.c_int64), ("n_rays", ctypes.c_int64),
        ("num_samples", ctypes.c_int32),
        ("t_table", _f32p), ("t_scale", ctypes.c_float),
        ("t_values", _f32p), ("u", _f32p), ("noise", _f32p),
        ("density_noise_std", ctypes.c_float),
        ("rng_mode", ctypes.c_int32),
        ("rng_seed", ctypes.c_uint64), ("rng_offset", ctypes.c_uint64), ("rng_counter", _f32p),
        ("base_radius_sq", ctypes.c_float),
        ("packed", _f32p),
        ("rgb", _f32p), ("seg", _f32p),
        ("out_mean", _f32p), ("out_cov", _f32p), ("out_t", _f32p), ("out_raw", _f32p), ("out_weights", _f32p),
        ("train_workspace", _f32p),
        ("precision", ctypes.c_int32),
        ("num_outputs", ctypes.c_int32),
        ("hidden", ctypes.c_int32), ("enc_inputs", ctypes.c_int32),
        ("color_outputs", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class BackwardArgs(ctypes.Structure):
    """Mirror of NerfHipBackwardArgs (include/nerf_hip.h)."""
    _fields_ = [("fwd", RenderArgs), ("d_rgb", _f32p), ("d_seg", _f32p), ("grad", _f32p),
                ("scratch", _f32p), ("d_raw", _f32p)]


class GatherArgs(ctypes.Structure):
    """Mirror of NerfHipGatherArgs (include/nerf_hip.h)."""
    _fields_ = [("index", _f32p), ("n", ctypes.c_int64), ("images", _f32p), ("segmentation", _f32p),
                ("poses", _f32p), ("batch", ctypes.c_int32), ("image_h", ctypes.c_int32),
                ("image_w", ctypes.c_int32), ("focal_length", ctypes.c_float),
                ("pixels", _f32p), ("label", _f32p), ("rays", _f32p), ("rays_o", _f32p),
                ("rays_d", _f32p), ("image_wi", _f32p), ("image_hi", _f32p), ("image_bi", _f32p)]


class ResampleArgs(ctypes.Structure):
    """Mirror of NerfHipResampleArgs (include/nerf_hip.h)."""
    _fields_ = [("n_rays", ctypes.c_int64), ("num_coarse", ctypes.c_int32), ("num_fine", ctypes.c_int32),
                ("t_coarse", _f32p), ("weights", _f32p), ("u", _f32p), ("pdf_floor", ctypes.c_float),
                ("t_union", _f32p)]


class LegacyArgs(ctypes.Structure):
    """Mirror of NerfHipLegacyArgs (include/nerf_hip.h)."""
    _fields_ = [("render", RenderArgs), ("normalize_position", ctypes.c_float),
                ("multiplier", ctypes.c_float), ("normalize_directions", ctypes.c_int32)]


class LegacyBackwardArgs(ctypes.Structure):
    """Mirror of NerfHipLegacyBackwardArgs (include/nerf_hip.h)."""
    _fields_ = [("fwd", LegacyArgs), ("d_rgb", _f32p), ("grad", _f32p), ("scratch", _f32p)]


ADAM_MAX_TENSORS = 64
ADAM_STEP_SLOTS = 2048


class AdamArgs(ctypes.Structure):
    """Mirror of NerfHipAdamArgs (include/nerf_hip.h)."""
    _fields_ = [("num_tensors", ctypes.c_int32), ("total", ctypes.c_int64),
                ("offsets", ctypes.c_int64 * (ADAM_MAX_TENSORS + 1)),
                ("params", ctypes.c_void_p * ADAM_MAX_TENSORS), ("grads", ctypes.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", _f32p), ("exp_avg_sq", _f32p), ("step", _f32p),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float)]


class MseArgs(ctypes.Structure):
    """Mirror of NerfHipMseArgs (include/nerf_hip.h)."""
    _fields_ = [("pred", _f32p), ("target", _f32p), ("n_rays", ctypes.c_int64), ("stages", ctypes.c_int32),
                ("loss", _f32p), ("grad", _f32p), ("channels", ctypes.c_int32)]


NUM_LEGACY_PARAM_TENSORS = 44
_lib = None


def _check_stamp():
    """A library older than the sources next to it must not load silently: the ABI number only changes
    with the header, a kernel edit does not bump it.  ``build.build()`` writes the content hash of
    everything the library's bytes depend on beside the .so; the in-tree library is refused when that
    hash is not the hash of the tree it sits in.  A library named through NERF_HIP_LIB (an experimental
    variant with its own -D list) is the caller's responsibility."""
    if os.environ.get("NERF_HIP_LIB"):
        return
    from . import build
    try:
        with open(LIB_PATH + ".stamp") as f:
            have = f.read().strip().split("-")[0]
    except OSError:
        have = None
    want = build.source_stamp()
    if have != want:
        raise RuntimeError(
            f"{LIB_PATH} is stale: its build stamp ({have and have[:12]}) is not the stamp of the sources "
            f"under {build.CSRC} ({want[:12]}). Rebuild with `python -c 'import __graft_entry__ as g; "
            "g.build()'`.")


def lib():
    """The loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). nerf_amd has no CPU or PyTorch fallback.")
    handle = ctypes.CDLL(LIB_PATH)
    handle.nerf_hip_version.restype = ctypes.c_int
    handle.nerf_hip_last_error.restype = ctypes.c_char_p
    handle.nerf_hip_packed_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_pack_weights.restype = ctypes.c_int
    handle.nerf_hip_pack_weights.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    handle.nerf_hip_render_forward.restype = ctypes.c_int
    handle.nerf_hip_render_forward.argtypes = [ctypes.POINTER(RenderArgs), ctypes.c_void_p]
    handle.nerf_hip_train_workspace_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_train_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
    handle.nerf_hip_grad_elements.restype = ctypes.c_size_t
    handle.nerf_hip_grad_elements.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    handle.nerf_hip_backward_scratch_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_backward_scratch_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
    handle.nerf_hip_render_backward.restype = ctypes.c_int
    handle.nerf_hip_render_backward.argtypes = [ctypes.POINTER(BackwardArgs), ctypes.c_void_p]
    handle.nerf_hip_gather_pixel_rays.restype = ctypes.c_int
    handle.nerf_hip_gather_pixel_rays.argtypes = [ctypes.POINTER(GatherArgs), ctypes.c_void_p]
    handle.nerf_hip_resample_pdf.restype = ctypes.c_int
    handle.nerf_hip_resample_pdf.argtypes = [ctypes.POINTER(ResampleArgs), ctypes.c_void_p]
    handle.nerf_hip_legacy_packed_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_legacy_pack_weights.restype = ctypes.c_int
    handle.nerf_hip_legacy_pack_weights.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                                    ctypes.c_void_p]
    handle.nerf_hip_legacy_render_forward.restype = ctypes.c_int
    handle.nerf_hip_legacy_render_forward.argtypes = [ctypes.POINTER(LegacyArgs), ctypes.c_void_p]
    handle.nerf_hip_legacy_train_workspace_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_legacy_train_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
    handle.nerf_hip_legacy_grad_elements.restype = ctypes.c_size_t
    handle.nerf_hip_legacy_backward_scratch_bytes.restype = ctypes.c_size_t
    handle.nerf_hip_legacy_backward_scratch_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
    handle.nerf_hip_legacy_render_backward.restype = ctypes.c_int
    handle.nerf_hip_legacy_render_backward.argtypes = [ctypes.POINTER(LegacyBackwardArgs), ctypes.c_void_p]
    handle.nerf_hip_adam_step.restype = ctypes.c_int
    handle.nerf_hip_adam_step.argtypes = [ctypes.POINTER(AdamArgs), ctypes.c_void_p]
    handle.nerf_hip_rng_advance.restype = ctypes.c_int
    handle.nerf_hip_rng_advance.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    handle.nerf_hip_mse_loss.restype = ctypes.c_int
    handle.nerf_hip_mse_loss.argtypes = [ctypes.POINTER(MseArgs), ctypes.c_void_p]
    handle.nerf_hip_timing.restype = ctypes.c_int
    handle.nerf_hip_timing.argtypes = [ctypes.c_int]
    handle.nerf_hip_timing_read.restype = ctypes.c_int
    handle.nerf_hip_timing_read.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_int64)]
    handle.nerf_hip_timing_read_tagged.restype = ctypes.c_int
    handle.nerf_hip_timing_read_tagged.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.POINTER(ctypes.c_double),
                                                   ctypes.POINTER(ctypes.c_int64)]
    handle.nerf_hip_timing_tag_name.restype = ctypes.c_char_p
    handle.nerf_hip_timing_tag_name.argtypes = [ctypes.c_int32]
    if handle.nerf_hip_version() != ABI_VERSION:
        raise RuntimeError(f"libnerf_hip.so ABI {handle.nerf_hip_version()} != expected {ABI_VERSION}")
    handle.nerf_hip_build_flags.restype = ctypes.c_char_p
    flags = handle.nerf_hip_build_flags().decode().split()
    if flags and not os.environ.get("NERF_HIP_LIB"):
        raise RuntimeError(
            f"{LIB_PATH} was compiled as an experiment {flags} (-DNERF_HIP_EXPERIMENT: not the product "
            "kernels); rebuild the product library, or select an experimental build explicitly "
            "with NERF_HIP_LIB=<path>")
    _check_stamp()
    _lib = handle
    return _lib


EXPORTS = ("nerf_hip_version", "nerf_hip_last_error", "nerf_hip_build_flags", "nerf_hip_packed_bytes",
           "nerf_hip_pack_weights", "nerf_hip_render_forward", "nerf_hip_train_workspace_bytes",
           "nerf_hip_grad_elements", "nerf_hip_backward_scratch_bytes", "nerf_hip_render_backward",
           "nerf_hip_gather_pixel_rays", "nerf_hip_resample_pdf", "nerf_hip_legacy_packed_bytes",
           "nerf_hip_legacy_pack_weights", "nerf_hip_legacy_render_forward",
           "nerf_hip_legacy_train_workspace_bytes", "nerf_hip_legacy_grad_elements",
           "nerf_hip_legacy_backward_scratch_bytes", "nerf_hip_legacy_render_backward", "nerf_hip_adam_step", "nerf_hip_mse_loss",
           "nerf_hip_rng_advance",
           "nerf_hip_timing",
           "nerf_hip_timing_read", "nerf_hip_timing_read_tagged", "nerf_hip_timing_tag_name")


def build_flags():
    """Experiment macros of the loaded library ([] for the product build)."""
    return lib().nerf_hip_build_flags().decode().split()


def check(rc, what):
    if rc != 0:
        msg = lib().nerf_hip_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def timing(enable):
    check(lib().nerf_hip_timing(1 if enable else 0), "nerf_hip_timing")


def timing_read(reset=True):
    """(average render-kernel ms, launches) since the last reset; HIP events on the launch stream."""
    avg, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib().nerf_hip_timing_read(1 if reset else 0, ctypes.byref(avg), ctypes.byref(n)),
          "nerf_hip_timing_read")
    return avg.value, n.value


TIMING_TAGS = 9         # NERF_HIP_TIMING_TAGS


def timing_read_tagged(reset=True):
    """{tag name: (average ms, launches)} of every launch kind recorded since the last reset (the launches of a
    training step: forward, composite_forward, composite_backward, data_gradient, weight_gradient, reduce, adam,
    loss, pack); tags without a launch are left out."""
    avg = (ctypes.c_double * TIMING_TAGS)()
    n = (ctypes.c_int64 * TIMING_TAGS)()
    check(lib().nerf_hip_timing_read_tagged(1 if reset else 0, TIMING_TAGS, avg, n), "nerf_hip_timing_read_tagged")
    return {lib().nerf_hip_timing_tag_name(t).decode(): (avg[t], n[t]) for t in range(TIMING_TAGS) if n[t] > 0}

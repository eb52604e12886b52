/*
 * nerf_hip.h — C ABI of libnerf_hip.so, the MI355X (gfx950) volume-render hot path.
 *
 * The reference (brandontrabucco/nerf) has no FFI of its own: its renderer is the Python
 * method surface of nerf.model.NeRF (generation C).  Each entry point below replaces the body
 * of one of those methods; the Python mirror in nerf_amd/model.py keeps the reference's
 * signatures and forwards here through ctypes (see INTEGRATION.md for the binding a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch-ROCm's
 *     caching allocator in practice); the library neither frees nor retains them;
 *   - all tensors are fp32, row-major, contiguous;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it, re-entrant,
 *     and never synchronise the device;
 *   - every call returns 0 on success or a negative NERF_HIP_E* code (never throws);
 *     nerf_hip_last_error() gives a thread-local message for the last failure.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_HIP_ABI_VERSION 8

#define NERF_HIP_OK 0
#define NERF_HIP_EINVAL (-1)   /* bad argument (null pointer, size out of range)      */
#define NERF_HIP_EUNSUPPORTED (-2) /* network shape outside what the kernels cover         */
#define NERF_HIP_EHIP (-3)     /* a HIP runtime call failed; see nerf_hip_last_error() */

/* Network shape (the constructor arguments of nerf/model.py:471-475 that size `prediction_heads`, :525-542):
 * hidden_size H in [1, 256], encoding_size in {2, 4, .. 32} (inputs of the first Linear = 3 * encoding_size in
 * {6, 12, .. 96}: nerf/model.py:526, :550-551) and the rows of the last Linear, 1 density + 3 color +
 * segmentation_outputs (nerf/model.py:541-542, :591-592) in [4, 64], are RUN-TIME arguments of pack / forward /
 * backward.  The kernels exist at three widths — 16, 8 and 4 register tiles of 16 features per sample — and a
 * launch runs at the smallest one that holds H: inference at 4 / 8 / 16 tiles in FP32 arithmetic and 8 / 16 in
 * F16X3, training (forward with saves, data gradient, weight gradient) at 8 / 16 in both; a network of H <= 128
 * therefore costs what a 128-wide one costs, not what the 256-wide one does.  Inside the chosen width (and for
 * encoding_size / num_outputs below the maxima) a network runs zero-padded, which is exact (LayerNorm divides by
 * H; nerf_amd/csrc/nerf_layout.h has the argument).  color_outputs (nerf/model.py:471, :541-542, :591-592, :660) is a
 * run-time count too, 1 .. 12: `rgb` / `d_rgb` then have that many columns.  256 / 96 / 54 / 3 for the defaults. */
#define NERF_HIP_HIDDEN 256
#define NERF_HIP_ENC_INPUTS 96
#define NERF_HIP_DEFAULT_OUTPUTS 54
#define NERF_HIP_MAX_OUTPUTS 64
#define NERF_HIP_MAX_COLORS 12
#define NERF_HIP_NUM_PARAM_TENSORS 22

#define NERF_HIP_PRECISION_FP32 0
#define NERF_HIP_PRECISION_F16X3 1

/* ABI version of the loaded library (NERF_HIP_ABI_VERSION at build time). */
int nerf_hip_version(void);

/* Message for the most recent failing call on this thread ("" if none). */
const char* nerf_hip_last_error(void);

/* Name of the experiment this library was built as (-DNERF_HIP_EXPERIMENT=name: a throw-away
 * variant for timing work), "" for the product build.  The Python loader refuses a non-empty answer
 * unless the library was selected explicitly with NERF_HIP_LIB.  No reference counterpart (build
 * hygiene). */
const char* nerf_hip_build_flags(void);

/* Size in bytes of the packed parameter image consumed by the render kernels. */
size_t nerf_hip_packed_bytes(void);

/*
 * Re-lay the 22 parameter tensors of NeRF.prediction_heads (nerf/model.py:525-542) into the
 * MFMA-fragment / LDS-image order the kernels stream.  `params` is a HOST array of 22 DEVICE
 * pointers in state_dict order (H = hidden, E = enc_inputs):
 *   prediction_heads.{0.weight[H,E], 0.bias[H], 1.weight[H], 1.bias[H], 3.weight[H,H], 3.bias,
 *   4.*, 6.*, 7.*, 9.*, 10.*, 12.*, 13.*, 15.weight[num_outputs,H], 15.bias[num_outputs]}
 * Must be called again whenever the parameters change (once per optimiser step).
 * The image ends with four floats of constants derived from the weights; [0] lets the split-precision
 * backward bound |dL/dy| of layer 0 per sample (the f16 scale of that layer's weight gradient).
 */
int nerf_hip_pack_weights(const float* const* params, int32_t hidden, int32_t enc_inputs, int32_t num_outputs,
                          int32_t color_outputs, float* packed, void* stream);

/* Where rays come from and what is written; replaces the bodies of
 * NeRF.render_rays (nerf/model.py:596-668) and NeRF.render_image (:670-770). */
typedef struct NerfHipRenderArgs {
    /* --- rays: either explicit arrays ... (render_rays, model.py:596) */
    const float* rays_o;        /* [n_rays,3] or NULL to generate from the cameras below   */
    const float* rays_d;        /* [n_rays,3] or NULL                                      */
    /* --- ... or generated on the fly from pinhole cameras (render_image, model.py:727-751):
     * global ray id r = ray_begin + i, image b = r / (H*W), row = (r / W) % H, col = r % W   */
    const float* camera_o;      /* [B,3]                                                   */
    const float* camera_r;      /* [B,3,3]                                                 */
    int32_t image_h, image_w;
    float focal_length;         /* of render_image's argument (model.py:276-277)           */
    int64_t ray_begin;          /* first global ray id of this call (row-block sharding)    */
    int64_t n_rays;             /* rays rendered by this call                              */
    /* --- sampling (sample_along_rays, model.py:369-435) */
    int32_t num_samples;        /* S fenceposts -> S-1 evaluated intervals; 2 <= S <= 4096 */
    const float* t_table;       /* [S] unscaled log-spaced fenceposts 2^linspace(...)      */
    float t_scale;              /* |rays_max - rays_min| (model.py:435)                    */
    const float* t_values;      /* [n_rays,S] explicit fenceposts (the `samples` argument of
                                   NeRF.forward, model.py:553); overrides table/u if set   */
    const float* u;             /* [n_rays,S] uniform draws of model.py:432, or NULL       */
    const float* noise;         /* [n_rays,S-1] normal draws of model.py:652, or NULL      */
    float density_noise_std;
    int32_t rng_mode;           /* bit0: draw u in-kernel (Philox) when u==NULL;
                                   bit1: draw noise in-kernel when noise==NULL            */
    uint64_t rng_seed, rng_offset;  /* Philox key = seed ^ offset: a distinct offset per launch (and
                                   per data-parallel rank) gives independent draws            */
    const uint64_t* rng_counter;    /* NULL, or a DEVICE word added to rng_offset when the kernel runs: a launch
                                   captured in a HIP graph replays its argument block, so what must differ from
                                   replay to replay lives in device memory; nerf_hip_rng_advance() moves it */
    /* --- network */
    float base_radius_sq;       /* (1/(sqrt(3)*focal))^2 with the CONSTRUCTOR focal (:546) */
    const float* packed;        /* image written by nerf_hip_pack_weights                  */
    /* --- outputs */
    float* rgb;                 /* [n_rays,color_outputs]  sum_s w_s * sigmoid(color_s)   (model.py:660); may be NULL on a
                                   training forward asked for out_raw only (then seg / out_weights NULL too):
                                   nothing is composited                                   */
    float* seg;                 /* [n_rays,classes] log-probabilities (model.py:661-663) or NULL; classes =
                                   num_outputs - 1 - color_outputs                         */
    /* optional per-sample outputs of NeRF.forward (model.py:553-594), any may be NULL      */
    float* out_mean;            /* [n_rays,S-1,3]  Gaussian means (model.py:587)           */
    float* out_cov;             /* [n_rays,S-1,3]  diagonal covariances (debug / parity)   */
    float* out_t;               /* [n_rays,S] the fenceposts used (sample_along_rays,
                                   model.py:369-435; with rng_mode bit0: the in-kernel draws) */
    float* out_raw;             /* [n_rays,S-1,num_outputs] density | color | segmentation logits */
    float* out_weights;         /* [n_rays,S-1] compositing weights (model.py:438-469)     */
    /* training: non-NULL makes the forward also save what the backward needs (activations,
     * LayerNorm statistics, compositing state); nerf_hip_train_workspace_bytes() floats.
     * A training forward may also be asked for out_raw / out_mean / out_cov (a DIFFERENTIABLE
     * NeRF.forward, model.py:553-594: its backward is nerf_hip_render_backward with d_raw);
     * out_t is not produced by it. */
    float* train_workspace;
    /* arithmetic of the MLP.  On a training forward it also selects the arithmetic of the data
     * gradient (dX = W^T dY) and of the weight gradient (dW = dY^T X; f16 pairs with one power-of-two
     * scale per layer and batch, else bf16 triples) in nerf_hip_render_backward, which reads this
     * struct back:
     * FP32  = exact-fp32 MFMA, the reference's arithmetic (torch fp32, nerf/model.py:525-542);
     * F16X3 = every fp32 operand split into an f16 pair, three f16 MFMAs per product with fp32
     *         accumulation (~2^-22 relative per product; same 1e-4 RGB parity bar)            */
    int32_t precision;
    /* rows of the last Linear = 1 density + color_outputs + segmentation classes, 2 .. 64 (54 for the
     * reference's defaults; must be what nerf_hip_pack_weights was given).  Without classes `seg`
     * must be NULL.  The legacy-network entry points ignore it. */
    int32_t num_outputs;
    /* hidden_size and 3 * encoding_size of the network (what nerf_hip_pack_weights was given); 0 means the
     * defaults 256 / 96.  The legacy-network entry points ignore them. */
    int32_t hidden, enc_inputs;
    /* color channels of the network (what nerf_hip_pack_weights was given), 1 .. 12; 0 means the default 3.
     * `rgb` is [n_rays, color_outputs], `seg` [n_rays, num_outputs - 1 - color_outputs].  The legacy-network
     * entry points ignore it (their network has three). */
    int32_t color_outputs;
    int32_t reserved;           /* set 0: the library derives a per-launch constant into its own copy of the block */
} NerfHipRenderArgs;

/* Fused forward: rays -> fenceposts -> conical-frustum Gaussians -> integrated positional
 * encoding -> 6-layer MLP on fp32 MFMA -> alpha compositing.  One persistent launch. */
int nerf_hip_render_forward(const NerfHipRenderArgs* args, void* stream);

/* Bytes of `train_workspace` for a batch of n_rays rays at num_samples fenceposts. */
size_t nerf_hip_train_workspace_bytes(int64_t n_rays, int32_t num_samples);

/* Number of fp32 elements of the flat gradient vector: the 22 parameter tensors in state_dict order,
 * each in its PyTorch layout (304,438 for 256 / 96 / 54; 0 if the shape is out of range). */
size_t nerf_hip_grad_elements(int32_t hidden, int32_t enc_inputs, int32_t num_outputs);

/* Backward of nerf_hip_render_forward w.r.t. the parameters (replaces PyTorch autograd through
 * NeRF.render_rays, driven by loss.backward() at train_conditional_nerf.py:133).  `fwd` must be
 * the argument block of the training forward call (same rays, sampling inputs, draws, packed
 * image and train_workspace, which that call filled); rays are not differentiated.
 * Two forms: the loss reached the COMPOSITED outputs (d_rgb, optionally d_seg; d_raw NULL: NeRF.render_rays,
 * model.py:596-668), or it reached the PER-SAMPLE network outputs of NeRF.forward (model.py:553-594: density |
 * color | segmentation logits; d_raw non-NULL, d_rgb / d_seg ignored) — then the compositing backward is skipped
 * and d_raw enters the data- and weight-gradient kernels directly. */
typedef struct NerfHipBackwardArgs {
    NerfHipRenderArgs fwd;
    const float* d_rgb;         /* [n_rays,color_outputs]  dL/d rgb                       */
    const float* d_seg;         /* [n_rays,classes] dL/d seg or NULL (RGB-only loss)      */
    float* grad;                /* [nerf_hip_grad_elements(hidden, enc_inputs, num_outputs)] written (not accumulated) */
    float* scratch;             /* nerf_hip_backward_scratch_bytes() bytes                 */
    const float* d_raw;         /* [n_rays,S-1,num_outputs] dL/d out_raw, or NULL          */
} NerfHipBackwardArgs;

size_t nerf_hip_backward_scratch_bytes(int64_t n_rays, int32_t num_samples);
int nerf_hip_render_backward(const NerfHipBackwardArgs* args, void* stream);

/* Batched replacement of PixelRayDataset.__getitem__ + default_collate (nerf/dataset.py:246-316,
 * train_conditional_nerf.py:100-101): decode flat example ids (w = id % W, h = (id / W) % H,
 * image = (id / (W*H)) % B, dataset.py:283-291), gather the pixel / label, build the camera-frame
 * ray of that pixel (nerf/model.py:271-278) and rotate it by the pose (model.py:367). */
typedef struct NerfHipGatherArgs {
    const int64_t* index;       /* [n] example ids                                        */
    int64_t n;
    const float* images;        /* [B,H,W,3]                                              */
    const int64_t* segmentation;/* [B,H,W] or NULL                                        */
    const float* poses;         /* [B,4,4] homogeneous camera-to-world                    */
    int32_t batch, image_h, image_w;
    float focal_length;
    float* pixels;              /* [n,3]                                                  */
    int64_t* label;             /* [n] or NULL                                            */
    float* rays;                /* [n,3] camera-frame ray                                 */
    float* rays_o;              /* [n,3] = pose[:3,3]                                     */
    float* rays_d;              /* [n,3] = pose[:3,:3] . ray                              */
    int64_t* image_wi;          /* [n] decoded column / row / image ids (any may be NULL) */
    int64_t* image_hi;
    int64_t* image_bi;
} NerfHipGatherArgs;

int nerf_hip_gather_pixel_rays(const NerfHipGatherArgs* args, void* stream);

/* Hierarchical sampling (BASELINE config 3).  The reference only carries vestigial docstrings for
 * it (nerf/model.py:191-193, :642-645), so this follows Mildenhall et al. 2020, section 5.2:
 * the coarse pass's compositing weights define a piecewise-constant PDF over the coarse intervals,
 * `num_fine` new fenceposts are drawn by inverse-transform sampling, and the fine pass evaluates
 * the sorted union of coarse and fine fenceposts.  PARITY UNPINNED: oracle/ holds the spec. */
typedef struct NerfHipResampleArgs {
    int64_t n_rays;
    int32_t num_coarse;         /* S_c coarse fenceposts per ray (2 <= S_c <= 1024)        */
    int32_t num_fine;           /* S_f fenceposts to draw (1 <= S_f <= 1024)               */
    const float* t_coarse;      /* [n_rays,S_c] coarse fenceposts, ascending               */
    const float* weights;       /* [n_rays,S_c-1] coarse compositing weights               */
    const float* u;             /* [n_rays,S_f] sorted uniforms, or NULL: (k + 0.5) / S_f  */
    float pdf_floor;            /* added to every weight (1e-5 in the paper's code)        */
    float* t_union;             /* [n_rays,S_c+S_f] sorted union, output                   */
} NerfHipResampleArgs;

int nerf_hip_resample_pdf(const NerfHipResampleArgs* args, void* stream);

/* The LEGACY (generation-A) network of examples/nerf.pth: sin/cos positional encoding, 8 x 256 trunk
 * with a skip-concatenation, density head, 2 x 256 view branch, color head.  Its
 * source is not in the reference repository; the structure is read off the checkpoint and the
 * constants the checkpoint cannot settle are fields here.  PARITY UNPINNED: oracle/legacy_oracle.py
 * holds the spec.  Call surface replaced: the notebook's NeRF(normalize_position=6.0).render_rays /
 * render_image (examples/example.ipynb cells 6, 8).
 * `params` of the pack routine: 44 DEVICE pointers, [W, b, gamma, beta] of block_0.{0,3,6,9} and
 * block_1.{0,3,6,9}, then density.{weight,bias}, then block_2.{0,3}, then color.{weight,bias}. */
#define NERF_HIP_LEGACY_PARAM_TENSORS 44
#define NERF_HIP_LEGACY_FLOP_PER_SAMPLE 1261568

typedef struct NerfHipLegacyArgs {
    /* rays (arrays or cameras), num_samples = S sample POSITIONS per ray (S network evaluations),
     * t_table [S] (+ u [n,S] for stratified sampling) or t_values [n,S], t_scale, noise [n,S],
     * density_noise_std, packed (nerf_hip_legacy_pack_weights), rgb [n,3], out_weights [n,S],
     * out_raw [n,S,4] = density | color logits, precision (FP32 or F16X3, as for the main network),
     * train_workspace (training forward, FP32 or F16X3; the backward runs in the arithmetic the training
     * forward was given: it reads `precision` back from this block; see below).
     * seg, out_mean/cov/t, rng_mode and base_radius_sq must be 0 / NULL. */
    NerfHipRenderArgs render;
    float normalize_position;   /* positions are divided by this before encoding (notebook: 6.0) */
    float multiplier;           /* frequency f_k = multiplier * 2^k (chosen: pi)                 */
    int32_t normalize_directions; /* encode d / |d| (chosen: 1)                                  */
} NerfHipLegacyArgs;

size_t nerf_hip_legacy_packed_bytes(void);
int nerf_hip_legacy_pack_weights(const float* const* params, float* packed, void* stream);
int nerf_hip_legacy_render_forward(const NerfHipLegacyArgs* args, void* stream);

/* Training of the legacy network — what the notebook's loop does through PyTorch autograd
 * (examples/example.ipynb cell 8: render_rays(...); ((pixels - target) ** 2).mean().backward(); Adam).
 * A forward with render.train_workspace set (nerf_hip_legacy_train_workspace_bytes() bytes; either precision,
 * the workspace layout is the same for both) also saves what the backward needs; nerf_hip_legacy_render_backward then writes the flat gradient:
 * nerf_hip_legacy_grad_elements() = 638,468 floats, the 44 tensors in the pack routine's order, each in its
 * PyTorch layout.  `fwd` must be the argument block of that training forward.  Rays are not differentiated. */
#define NERF_HIP_LEGACY_GRAD_ELEMENTS 638468
size_t nerf_hip_legacy_train_workspace_bytes(int64_t n_rays, int32_t num_samples);
size_t nerf_hip_legacy_grad_elements(void);

typedef struct NerfHipLegacyBackwardArgs {
    NerfHipLegacyArgs fwd;
    const float* d_rgb;         /* [n_rays,3]  dL/d rgb                                   */
    float* grad;                /* [nerf_hip_legacy_grad_elements()] written (not accumulated) */
    float* scratch;             /* nerf_hip_legacy_backward_scratch_bytes() bytes          */
} NerfHipLegacyBackwardArgs;

size_t nerf_hip_legacy_backward_scratch_bytes(int64_t n_rays, int32_t num_samples);
int nerf_hip_legacy_render_backward(const NerfHipLegacyBackwardArgs* args, void* stream);

/* The optimiser step of the reference's training loops — torch.optim.Adam(parameters, lr) with its default
 * betas / eps, no weight decay, no amsgrad (train_conditional_nerf.py:106-107, :135; examples/example.ipynb
 * cells 7, 8) — as ONE launch over all parameter tensors (22 or 44 here; torch's fused kernel spends 43 us on
 * them).  State (exp_avg, exp_avg_sq) is flat, in the order of the tensor list.  `step` holds the number of
 * updates applied so far in DEVICE memory, one copy per workgroup of the launch (NERF_HIP_ADAM_STEP_SLOTS equal
 * floats, zero-initialised): workgroup b applies update number step[b] + 1 and stores that count back into its
 * own slot, so no workgroup reads a count another one has already advanced, a captured launch replays
 * correctly, and counting costs neither a launch nor an atomic. */
#define NERF_HIP_ADAM_MAX_TENSORS 64
#define NERF_HIP_ADAM_STEP_SLOTS 2048
typedef struct NerfHipAdamArgs {
    int32_t num_tensors;
    int64_t total;                                      /* parameters in all                         */
    int64_t offsets[NERF_HIP_ADAM_MAX_TENSORS + 1];     /* prefix sums of the tensor sizes, [0] = 0  */
    float* params[NERF_HIP_ADAM_MAX_TENSORS];           /* updated in place                          */
    const float* grads[NERF_HIP_ADAM_MAX_TENSORS];      /* one per tensor (views of a flat gradient or not) */
    float* exp_avg;                                     /* [total] first moment                      */
    float* exp_avg_sq;                                  /* [total] second moment                     */
    float* step;                                        /* [NERF_HIP_ADAM_STEP_SLOTS] device: updates applied so far */
    float lr, beta1, beta2, eps;
} NerfHipAdamArgs;

int nerf_hip_adam_step(const NerfHipAdamArgs* args, void* stream);

/* *counter += delta as one tiny launch on `stream` (stream-ordered behind the launches that read the word through
 * NerfHipRenderArgs.rng_counter, in front of the next ones; capturable: a replayed training step draws new
 * samples).  No reference counterpart: the reference draws from torch's global generator (nerf/model.py:432,
 * :652), whose state a captured region advances the same way (a device-side offset). */
int nerf_hip_rng_advance(uint64_t* counter, uint64_t delta, void* stream);

/* The loss of those loops and its gradient in one launch: loss = mean((pred - target[:, None, :]) ** 2) over
 * pred [n_rays, stages, 3] and target [n_rays, 3] (train_conditional_nerf.py:132: `((pixels -
 * batch["pixels"].unsqueeze(1)) ** 2).mean()`; examples/example.ipynb cell 8), grad = d loss / d pred =
 * (1 / count) * 2 (pred - target), rounded as autograd rounds it.  n_rays == 0 gives loss 0 (an empty shard of a
 * data-parallel batch), not NaN.  One workgroup, a fixed summation order: the loss is reproducible. */
typedef struct NerfHipMseArgs {
    const float* pred;          /* [n_rays, stages, channels] */
    const float* target;        /* [n_rays, channels]         */
    int64_t n_rays;
    int32_t stages;
    float* loss;                /* [1]                        */
    float* grad;                /* [n_rays, stages, channels] */
    int32_t channels;           /* color_outputs of the network; 0 means 3 */
} NerfHipMseArgs;

int nerf_hip_mse_loss(const NerfHipMseArgs* args, void* stream);

/* Average duration in milliseconds of the render kernel over the launches issued since the
 * last call with reset != 0, measured with HIP events recorded on the launch stream.  Timing
 * is off by default (no events recorded); nerf_hip_timing(1) turns it on.  Synchronises.
 * With timing on, every launch of a training step is bracketed the same way under one of the tags below
 * (the training forward's network kernel counts as NERF_HIP_TIMING_FORWARD, like the render kernel);
 * nerf_hip_timing_read reports that tag alone, nerf_hip_timing_read_tagged all of them
 * (avg_ms / launches: arrays of n_tags entries, n_tags <= NERF_HIP_TIMING_TAGS).  No reference counterpart: the
 * reference has no timers; bench.py's roofline and its per-kernel split of a training step
 * (train_conditional_nerf.py:130-135) are measured with this. */
#define NERF_HIP_TIMING_FORWARD 0             /* render kernel / training forward (network) of either network */
#define NERF_HIP_TIMING_COMPOSITE_FORWARD 1   /* training: compositing forward (+ per-sample field outputs)    */
#define NERF_HIP_TIMING_COMPOSITE_BACKWARD 2  /* compositing backward, or the d_raw scatter                     */
#define NERF_HIP_TIMING_DATA_GRADIENT 3
#define NERF_HIP_TIMING_WEIGHT_GRADIENT 4
#define NERF_HIP_TIMING_REDUCE 5              /* split-K slabs -> the flat gradient                              */
#define NERF_HIP_TIMING_ADAM 6
#define NERF_HIP_TIMING_LOSS 7
#define NERF_HIP_TIMING_PACK 8                /* parameter re-layout (either network)                            */
#define NERF_HIP_TIMING_TAGS 9
int nerf_hip_timing(int enable);
int nerf_hip_timing_read(int reset, double* avg_ms, int64_t* launches);
int nerf_hip_timing_read_tagged(int reset, int32_t n_tags, double* avg_ms, int64_t* launches);
const char* nerf_hip_timing_tag_name(int32_t tag);      /* "forward", "data_gradient", ...; NULL out of range */

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */

// Fused render forward of the LEGACY (generation-A) network whose trained weights ship as
// examples/nerf.pth: sin/cos positional encoding -> 8 x 256 trunk with a skip-concatenation ->
// density head; view branch (2 x 256) -> color head; alpha compositing.  One persistent launch.
//
// PARITY UNPINNED: the network's source is not in the reference repository; the structure is
// recovered from the checkpoint's tensor shapes and the unrecoverable constants are arguments
// (oracle/legacy_oracle.py states every choice).  What IS the reference's:
// the encoding layout (nerf/model.py:221-240), the compositing (nerf/model.py:438-469, :660) and
// the call surface (examples/example.ipynb cells 6, 8).
//
// Same machinery as nerf_render.hip: a wave owns 16 samples x all 256 features in the fp32 MFMA
// accumulator layout (nerf_layout.h), so a layer's output is the next layer's B operand without
// leaving registers; weights stream L2 -> LDS through the 3-slot LDS-DMA ring (WeightPipe) in
// consumption order, 157 stages of 16 KiB; exact-fp32 arithmetic (v_mfma_f32_16x16x4_f32).
// Differences: [Linear, ReLU, LayerNorm] order (the LayerNorm is a VALU phase between two layers
// here, not fused into the MFMA loops), two concatenation layers (the encodings stay in registers and
// enter as extra k-groups), two one-tile heads, S samples = S evaluations at points.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerf_device.h"
#include "nerf_fused.h"
#include "nerf_legacy_layout.h"

using namespace nerf_layout;
using namespace nerf_device;
using namespace nerf_legacy;

namespace {

constexpr int kLegacySmallBytes = (kLegacySmallFloats * 4 + 127) / 128 * 128;
constexpr int kLegacyLdsBytes = kRingBytes + kLegacySmallBytes;                // 80,000 B -> 2 workgroups / CU

struct LegacyKernelArgs {
    NerfHipLegacyArgs l;
    int32_t chunks;             // ceil(S / 16)
    int64_t groups;             // inference: ceil(n_rays / 4) rays; training: (padded ray, chunk) items / 4
    LegacyTrainLayout save;     // offsets into render.train_workspace (training forward only)
};

typedef WeightPipe<kLegacyStages> LegacyPipe;
typedef WeightPipe<kLegacyHStages> LegacyHPipe;

// sin(y) or cos(y) (shift) for |y| up to a few thousand rad: the half-turn reduction of
// nerf_device.h: sin_reduced, with the cosine taken as sin(pi/2 - |r|) of the REDUCED argument
// (adding pi/2 to y itself would cost an ulp of y: 6e-5 at y = 1600).
__device__ __forceinline__ float sincos_reduced(float y, bool cosine) {
    const float n = __builtin_rintf(y * 0.318309886f);
    float r = __builtin_fmaf(-n, 3.1415927410125732f, y);
    r = __builtin_fmaf(-n, -8.742277657347586e-08f, r);
    r = __builtin_fmaf(-n, -3.4302490200117637e-15f, r);
    if (cosine) r = 1.5707963267948966f - __builtin_fabsf(r);
    const float u = r * r;
    float s = __builtin_fmaf(u, -2.3794713703943473e-08f, 2.7518855647935822e-06f);
    s = __builtin_fmaf(u, s, -0.00019840702862741812f);
    s = __builtin_fmaf(u, s, 0.008333329264456273f);
    s = __builtin_fmaf(u, s, -0.16666666541439012f);
    const float p = __builtin_fmaf(r * u, s, r);
    return ((int)n & 1) ? -p : p;
}

// This lane group's slots of an encoding [x: sin f_0..f_{F-1}, cos f_0..f_{F-1} | y: ... | z: ...]
// (nerf/model.py:233-240), f_k = multiplier 2^k: slot q = half * coord + kk is the lane group's trig function
// (g >> 1) of coordinate q / half at frequency 2^kk times the group's base frequency (nerf_legacy_layout.h:
// encoding_feature_of).  The argument v * (multiplier * 2^k) is rounded exactly as in the oracle (the extra
// factors are powers of two).
template <int kFreqs, int kSlots>
__device__ __forceinline__ void encode_slots(const float (&x)[3], float multiplier, int g, float scale,
                                             float (&out)[64]) {
    constexpr int kHalf = kFreqs / 2;
    const float base = (g & 1) ? multiplier * (float)(1 << kHalf) : multiplier;
    const bool cosine = (g >> 1) != 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (q < kSlots) {
            const float v = q / kHalf == 0 ? x[0] : (q / kHalf == 1 ? x[1] : x[2]);
            out[q] = scale * sincos_reduced(v * (base * (float)(1 << (q % kHalf))), cosine);
        } else {
            out[q] = 0.f;
        }
    }
}
// PE(position / normalize_position) at the sample t0 along the ray (15 of 60 features per lane group in 16
// slots) and PE(d / |d|) (9 of 36 in 12 slots; 16 written), times `scale`
__device__ __forceinline__ void encode_position(const Ray& ray, float t0, const NerfHipLegacyArgs& la, int g,
                                                float (&pos_act)[64], float scale = 1.0f) {
#pragma clang fp contract(off)
    const float x[3] = {(ray.d[0] * t0 + ray.o[0]) / la.normalize_position,
                        (ray.d[1] * t0 + ray.o[1]) / la.normalize_position,
                        (ray.d[2] * t0 + ray.o[2]) / la.normalize_position};
    encode_slots<kPosFreqs, kPosPerGroup>(x, la.multiplier, g, scale, pos_act);
}
__device__ __forceinline__ void encode_direction(const Ray& ray, float dlen, const NerfHipLegacyArgs& la, int g,
                                                 float (&dir_act)[64], float scale = 1.0f) {
    float dn[3] = {ray.d[0], ray.d[1], ray.d[2]};
    if (la.normalize_directions) {
        const float inv = 1.0f / dlen;
        dn[0] *= inv, dn[1] *= inv, dn[2] *= inv;
    }
    encode_slots<kDirFreqs, kDirPerGroup>(dn, la.multiplier, g, scale, dir_act);
}

// LayerNorm(256, eps 1e-5, affine) of relu(acc) -> act (the next layer's B operands).
// Lane (j, g) holds features 16 T + 4 g + r of sample j in acc[T][r].  Five VALU instructions per element (max, add,
// fma | fma, fma) — this phase is frame time in full: nothing executes beside the fp32 MFMAs of the wave's SIMD partner
// (NOTES.md section R6d; it was seven, with a two-pass variance and (a - mean) * rstd): one-pass moments, with the exact
// two-pass variance as a wave-uniform fallback when the mean carries more than 3/4 of the second moment in any sample
// (as nerf_fused.h: finish_moments_at; ReLU outputs of a zero-mean pre-activation have mean^2 = 0.32 E[a^2]).
// kTrain: also saves a_hat = (relu(y) - mean) / std (row order), 1/std and `shift` = the a_hat of a closed
// gate (fma(0, 1/std, shift) = shift exactly), so that the backward reads the ReLU gate as a_hat > shift.
// The FMA is monotonic in relu(y), so a_hat >= shift always; where an OPEN gate (y > 0, y below
// half an ulp of the mean) rounds onto `shift`, a_hat is moved one ulp up — the gate the backward sees is exact.
template <bool kTrain>
__device__ __forceinline__ void relu_layer_norm(const f32x4 (&acc)[16], const float* small_l, int g,
                                                float (&act)[64], float* xhat_row = nullptr,
                                                float* rstd_p = nullptr, float* shift_p = nullptr) {
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = relu_bits(acc[T][r]);
            act[4 * T + r] = a;
            sum += a;
            sq = __builtin_fmaf(a, a, sq);
        }
    const float mean = group_sum(sum) * (1.0f / 256.0f);
    const float ex2 = group_sum(sq) * (1.0f / 256.0f);
    float var = ex2 - mean * mean;
    if (__builtin_amdgcn_ballot_w64(mean * mean > 0.75f * ex2) != 0) {
        float sq2 = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float d = act[i] - mean;
            sq2 = __builtin_fmaf(d, d, sq2);
        }
        var = group_sum(sq2) * (1.0f / 256.0f);
    }
    const float ve = var + 1e-5f;
    float rstd = __builtin_amdgcn_rsqf(ve);
    rstd = rstd * __builtin_fmaf(-0.5f * ve * rstd, rstd, 1.5f);
    const f32x4* gam = (const f32x4*)(small_l + kHidden + g * 64);
    const f32x4* bet = (const f32x4*)(small_l + 2 * kHidden + g * 64);
    const float shift = (0.f - mean) * rstd;                  // <= 0 (mean of ReLU outputs)
    const float above = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, shift) - 1u);   // next float up (shift < 0)
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        const f32x4 ga = gam[T], be = bet[T];
        f32x4 xh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = __builtin_fmaf(act[4 * T + r], rstd, shift);
            if (kTrain) xh[r] = (act[4 * T + r] > 0.f && xh[r] <= shift) ? above : xh[r];
            act[4 * T + r] = __builtin_fmaf(xh[r], ga[r], be[r]);
        }
        if (kTrain) *(f32x4*)(xhat_row + T * kTileT) = xh;
    }
    if (kTrain && g == 0) {
        *rstd_p = rstd;
        *shift_p = shift;
    }
}

__device__ __forceinline__ void load_bias(const float* small_l, int g, f32x4 (&acc)[16]) {
    const f32x4* b = (const f32x4*)(small_l + g * 64);
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = b[T];
}

// One-tile head (256 -> <= 16 outputs): one stage = 16 quads, quad t = the A operands of k-group t.
__device__ __forceinline__ f32x4 head_layer(LegacyPipe& pipe, f32x4 acc, const float (&act)[64]) {
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const f32x4* st = pipe.open_stage();
    f32x4 a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = st[t * 64];
    pipe.prefetch_next();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        acc = mfma4(a[t].x, act[4 * t], acc);
        acc = mfma4(a[t].y, act[4 * t + 1], acc);
        acc = mfma4(a[t].z, act[4 * t + 2], acc);
        acc = mfma4(a[t].w, act[4 * t + 3], acc);
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
    return acc;
}

// kTrain: the training forward.  Compositing (the only coupling between the chunks of a ray) is then a
// kernel of its own (nerf_legacy_composite_fwd_kernel), so the unit of work is one (padded ray, chunk) item
// per wave — a 512-ray batch fills all 2,048 waves — and everything the backward needs is saved:
// the two encodings, a_hat / 1/std / shift of the ten LayerNorms, the head outputs, the sample spacing.
template <bool kTrain>
__global__ __launch_bounds__(256, 2) void nerf_legacy_fwd_kernel(const LegacyKernelArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipLegacyArgs& la = ka.l;
    const NerfHipRenderArgs& a = la.render;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int S = a.num_samples;
    float* const ws = a.train_workspace;

    float* small = (float*)(smem + kRingBytes);
    for (int i = threadIdx.x; i < kLegacySmallFloats; i += 256) small[i] = a.packed[kLegacyBlobFloats + i];
    LegacyPipe pipe;
    pipe.init(a.packed, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    for (int64_t grp = blockIdx.x; grp < ka.groups; grp += gridDim.x) {
        const int64_t unit = grp * kWavesPerWg + wave;
        const int64_t slot = kTrain ? unit / ka.chunks : unit;       // padded ray slot (workspace rows)
        int64_t local = slot;
        const bool ray_ok = local < a.n_rays;
        if (!ray_ok) local = a.n_rays - 1;
        const Ray ray = load_ray(a, local);
        // |d| (sample spacing in space) and the encoded view direction: once per ray
        const float dlen = __builtin_sqrtf((ray.d[0] * ray.d[0] + ray.d[1] * ray.d[1]) + ray.d[2] * ray.d[2]);
        float dir_act[64];
        encode_direction(ray, dlen, la, g, dir_act);
        RayAccum racc;
        racc.reset();
        const int c_begin = kTrain ? (int)(unit - slot * ka.chunks) : 0;
        const int c_end = kTrain ? c_begin + 1 : ka.chunks;
        for (int c = c_begin; c < c_end; ++c) {
            const int s = c * kSamplesPerWave + j;
            const bool ok = s < S;
            const int64_t tile = slot * ka.chunks + c;        // chunk index in the workspace
            const int64_t sp = tile * 16 + j;                 // padded sample index
            // sample positions: the fencepost routine of the main kernel on a caller-supplied table
            // (linear in [near, far] for this network, Mildenhall et al. 2020), stratified by u, or
            // explicit per-ray positions
            const int sc = s < S - 1 ? s : S - 1;
            float posts[2];
            fencepost_run<2>(a, local, sc, posts);
            const float t0 = posts[0], t1 = posts[1];
            const float dist = s >= S - 1 ? 1e10f : dlen * (t1 - t0);
            float pos_act[64];
            encode_position(ray, t0, la, g, pos_act);
            // training: lane-relative bases of this sample's saved rows (+ the tensor's offset)
            float* const xrow = kTrain ? ws + tile_lane_base(sp, g) : nullptr;      // (tile-major rows)
            float* const stat = kTrain ? ws + sp : nullptr;
            if (kTrain) {
                if (g == 0) *(f32x4*)(ws + ka.save.comp + sp * 4) = f32x4{0.f, 0.f, dist, 0.f};   // (not live across the MLP)
                float* prow = ws + ka.save.pos + sp * kEncPad + 4 * g;
                float* drow = ws + ka.save.dir + sp * kEncPad + 4 * g;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    *(f32x4*)(prow + 16 * t) = f32x4{pos_act[4 * t], pos_act[4 * t + 1], pos_act[4 * t + 2], pos_act[4 * t + 3]};
                    *(f32x4*)(drow + 16 * t) = t < kDirTiles ? f32x4{dir_act[4 * t], dir_act[4 * t + 1], dir_act[4 * t + 2], dir_act[4 * t + 3]}
                                                              : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            f32x4 acc[16];
            float act[64];
            // ---- block_0 ----
            load_bias(small, g, acc);
            layer_wide<kPosTiles>(pipe, acc, pos_act);
            relu_layer_norm<kTrain>(acc, small, g, act, xrow + ka.save.xhat[0], stat + ka.save.rstd[0], stat + ka.save.shift[0]);
#pragma unroll 1
            for (int L = 1; L <= 3; ++L) {
                const float* sl = small + L * kLegacySmallPerLayer;
                load_bias(sl, g, acc);
                layer_wide<16>(pipe, acc, act);
                relu_layer_norm<kTrain>(acc, sl, g, act, xrow + ka.save.xhat[L], stat + ka.save.rstd[L], stat + ka.save.shift[L]);
            }
            // ---- block_1: [hidden | encoded position] -> 256 ----
            {
                const float* sl = small + 4 * kLegacySmallPerLayer;
                load_bias(sl, g, acc);
                layer_wide<16>(pipe, acc, act);
                if (kTrain) {
                    // recomputed (same operations, same bits) rather than kept live across four layers: the
                    // training kernel also carries the save addresses, and 16 more live registers spill
                    float again[64];
                    encode_position(ray, t0, la, g, again);
                    layer_wide<kPosTiles>(pipe, acc, again);
                } else {
                    layer_wide<kPosTiles>(pipe, acc, pos_act);
                }
                relu_layer_norm<kTrain>(acc, sl, g, act, xrow + ka.save.xhat[4], stat + ka.save.rstd[4], stat + ka.save.shift[4]);
            }
#pragma unroll 1
            for (int L = 5; L <= 7; ++L) {
                const float* sl = small + L * kLegacySmallPerLayer;
                load_bias(sl, g, acc);
                layer_wide<16>(pipe, acc, act);
                relu_layer_norm<kTrain>(acc, sl, g, act, xrow + ka.save.xhat[L], stat + ka.save.rstd[L], stat + ka.save.shift[L]);
            }
            // ---- density head ----
            const f32x4* hb = (const f32x4*)(small + kWide * kLegacySmallPerLayer);
            const f32x4 dens = head_layer(pipe, hb[g], act);
            // ---- block_2: [hidden | encoded direction] -> 256 -> 256 ----
            {
                const float* sl = small + 8 * kLegacySmallPerLayer;
                load_bias(sl, g, acc);
                layer_wide<16>(pipe, acc, act);
                if (kTrain) {
                    float again[64];
                    encode_direction(ray, dlen, la, g, again);
                    layer_wide<kDirTiles>(pipe, acc, again);
                } else {
                    layer_wide<kDirTiles>(pipe, acc, dir_act);
                }
                relu_layer_norm<kTrain>(acc, sl, g, act, xrow + ka.save.xhat[8], stat + ka.save.rstd[8], stat + ka.save.shift[8]);
                const float* sl9 = small + 9 * kLegacySmallPerLayer;
                load_bias(sl9, g, acc);
                layer_wide<16>(pipe, acc, act);
                relu_layer_norm<kTrain>(acc, sl9, g, act, xrow + ka.save.xhat[9], stat + ka.save.rstd[9], stat + ka.save.shift[9]);
            }
            const f32x4 col = head_layer(pipe, hb[4 + g], act);
            // ---- compositing (nerf/model.py:438-469, :660): out[0] = (density, r, g, b) on lane group 0
            f32x4 out[4];
            out[0] = f32x4{dens.x, col.x, col.y, col.z};
            out[1] = out[2] = out[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kTrain) {
                float* otile = ws + ka.save.out + tile * 1024 + lane * 4;
#pragma unroll
                for (int T = 0; T < 4; ++T) *(f32x4*)(otile + T * 256) = out[T];
                continue;
            }
            const float w = composite_chunk<false, false>(a, S, local, s, ok, lane, out, dist, racc, nullptr);
            if (ray_ok && ok && g == 0) {
                const int64_t smp = local * S + s;
                if (a.out_weights != nullptr) a.out_weights[smp] = w;
                if (a.out_raw != nullptr) {
                    a.out_raw[smp * 4 + 0] = dens.x;
                    a.out_raw[smp * 4 + 1] = col.x;
                    a.out_raw[smp * 4 + 2] = col.y;
                    a.out_raw[smp * 4 + 3] = col.z;
                }
            }
        }
        if (!kTrain) store_ray<false>(a, local, ray_ok, lane, racc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Compositing of the training forward: one wave per ray over the saved head outputs.
__global__ __launch_bounds__(256) void nerf_legacy_composite_fwd_kernel(const LegacyKernelArgs ka) {
    composite_fwd_body(ka.l.render, ka.l.render.num_samples, ka.chunks, ka.save.out, ka.save.comp);
}

// ---------------------------------------------------------------------------------------------
// The same launch in split-precision arithmetic (LegacyNeRF8x256.precision = "f16x3"): every fp32
// operand of the twelve matrix products as an f16 pair, three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulation.  The ten wide layers run on the main kernel's fused layer (nerf_fused.h: layer_fused_h in
// the Linear -> ReLU -> LayerNorm order): the LayerNorm is NOT a VALU phase between two loops — the moments of
// relu(y) are gathered while the second half of the layer runs, and gamma a_hat + beta of the next layer's
// input is applied lazily and split into f16 pairs one stage ahead of the stage that consumes it.
// Accumulators hold 2^12 (W x + b) (eps * 2^24, bit-identical a_hat); gamma / beta come pre-scaled by 2^4.
//   * X and Y swap roles layer by layer (no copy-back): L0: X (encoding) -> Y, L1: Y -> X, ... L9: Y -> X.
//   * L4 / L8 = the fused layer over the hidden columns + a short loop (layer_wide_h<2>) over the two k blocks
//     of the concatenated encoding, which is RE-ENCODED right there (15 / 9 sines per lane) rather than held in
//     16 registers across four layers; their moments are gathered after that loop.
//   * the density head runs after L8's loops on x'_7, which L8's fused loop has normalised in place; the
//     color head normalises y_9 on the fly.  Both split their operands block by block between the MFMAs.
//   * the per-ray compositing state (4 sums, the sample spacing) and the ray wait in 1.4 KiB of LDS while the
//     MLP runs, so that nothing is spilled (the state is the same in all four lane groups: 16 slots per wave).
// ---------------------------------------------------------------------------------------------
constexpr int kLegacyStashBytes = kWavesPerWg * 16 * (4 + 1 + 1) * 4;  // per sample slot of a wave: 4 sums, the spacing, t0
constexpr int kLegacyRayStashBytes = kWavesPerWg * 8 * 4;
constexpr int kLegacyLdsBytesHalf = kLegacyLdsBytes + kLegacyStashBytes + kLegacyRayStashBytes;   // 81,664 B: still 2 / CU
static_assert(2 * kLegacyLdsBytesHalf <= 160 * 1024, "two workgroups per CU");

using nerf_fused::HMoments;
using nerf_fused::LazyNorm;
using nerf_fused::kOrderReluNorm;

// k block m of a 64-register activation tile set = tiles 2m, 2m + 1 -> an f16-pair B operand
__device__ __forceinline__ void split_block(const float (&act)[64], int m, h8& hi, h8& lo) {
    split8(f32x4{act[8 * m], act[8 * m + 1], act[8 * m + 2], act[8 * m + 3]},
           f32x4{act[8 * m + 4], act[8 * m + 5], act[8 * m + 6], act[8 * m + 7]}, hi, lo);
}

// One-tile head (256 -> <= 16 outputs) on f16 pairs straight from the fp32 activation tiles: one stage = pair
// i = (out tile 0, k block i); block i = tiles 2i, 2i + 1 is normalised (kNorm: the tiles still hold raw
// accumulators) and split right in front of its three MFMAs, its weights read one block ahead.
template <bool kNorm, bool kTrain>
__device__ __forceinline__ f32x4 head_layer_h(LegacyHPipe& pipe, f32x4 acc, f32x4 (&x)[16], const LazyNorm& norm) {
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = (const h8*)pipe.open_stage();
    h8 ah[2], al[2];
    ah[0] = st[0];
    al[0] = st[64];
    pipe.prefetch_next();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m + 1 < 8) {
            ah[(m + 1) & 1] = st[(2 * m + 2) * 64];
            al[(m + 1) & 1] = st[(2 * m + 3) * 64];
        }
        if (kNorm) {
            nerf_fused::normalize_tile<kTrain, false, kOrderReluNorm>(x[2 * m], norm, 2 * m);
            nerf_fused::normalize_tile<kTrain, false, kOrderReluNorm>(x[2 * m + 1], norm, 2 * m + 1);
        }
        h8 bh, bl;
        split8(x[2 * m], x[2 * m + 1], bh, bl);
        acc = mfma_h(ah[m & 1], bh, acc);
        acc = mfma_h(ah[m & 1], bl, acc);
        acc = mfma_h(al[m & 1], bh, acc);
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
    return acc;
}

// The density head on k blocks that are ALREADY f16 pairs (the B operands L8's fused loop built from x'_7,
// nerf_fused.h: layer_fused_hb): the same eight stages, splits and MFMA order as head_layer_h<false> on the fp32
// tiles — bit-identical — without holding those 16 tiles across L8 (64 registers in every layer of the shared X -> Y
// code instance: what the 256-register training kernel spilled inside its loops).
__device__ __forceinline__ f32x4 head_layer_hb(LegacyHPipe& pipe, f32x4 acc, const h8 (&bhi)[8], const h8 (&blo)[8]) {
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = (const h8*)pipe.open_stage();
    h8 ah[2], al[2];
    ah[0] = st[0];
    al[0] = st[64];
    pipe.prefetch_next();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m + 1 < 8) {
            ah[(m + 1) & 1] = st[(2 * m + 2) * 64];
            al[(m + 1) & 1] = st[(2 * m + 3) * 64];
        }
        acc = mfma_h(ah[m & 1], bhi[m], acc);
        acc = mfma_h(ah[m & 1], blo[m], acc);
        acc = mfma_h(al[m & 1], bhi[m], acc);
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
    return acc;
}

// kTrain: the training forward in this arithmetic (LegacyNeRF8x256.train_precision = "f16x3"): one (padded ray,
// chunk) item per wave, compositing in its own kernel, the same saves as the fp32 training forward
// (nerf_legacy_fwd_kernel<true>) into the same workspace — a_hat is scale-free, 1/std is un-scaled by 2^12 —
// so that either backward arithmetic reads it.
template <bool kTrain>
__global__ __launch_bounds__(256, 2) void nerf_legacy_fwd_h_kernel(const LegacyKernelArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipLegacyArgs& la = ka.l;
    const NerfHipRenderArgs& a = la.render;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int S = a.num_samples;
    float* const ws = a.train_workspace;
    constexpr float kX = (float)(1 << kXScaleLog2), kRs = (float)(1 << (kWScaleLog2 + kXScaleLog2)), kUn = 1.0f / kRs;
    constexpr float kEps = 1e-5f * kRs * kRs;

    float* small = (float*)(smem + kRingBytes);
    for (int i = threadIdx.x; i < kLegacySmallFloats; i += 256) small[i] = a.packed[kLegacyHSmallOffset + i];
    f32x4* const stash = (f32x4*)(smem + kLegacyLdsBytes) + (wave * 16 + j);          // this sample slot: 4 sums ...
    float* const stash_dist = (float*)(smem + kLegacyLdsBytes + kWavesPerWg * 16 * 16) + (wave * 16 + j);   // ... spacing
    // the sample's near fencepost, parked for the re-encoding at L4: evaluating fencepost() again there means global
    // loads (and, under stratified sampling, their vmcnt(0) waits, which also drain the weight DMA) inside the layer loop
    float* const stash_t0 = (float*)(smem + kLegacyLdsBytes + kWavesPerWg * 16 * 20) + (wave * 16 + j);
    float* const ray_stash = (float*)(smem + kLegacyLdsBytes + kLegacyStashBytes) + wave * 8;
    LegacyHPipe pipe;
    pipe.init(a.packed + kLegacyHOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    auto gamma_of = [&](const float* sl) { return (const f32x4*)(sl + kHidden + g * 64); };
    auto beta_of = [&](const float* sl) { return (const f32x4*)(sl + 2 * kHidden + g * 64); };

    for (int64_t grp = blockIdx.x; grp < ka.groups; grp += gridDim.x) {
        const int64_t unit = grp * kWavesPerWg + wave;
        const int64_t slot = kTrain ? unit / ka.chunks : unit;       // padded ray slot (workspace rows)
        int64_t local = slot;
        const bool ray_ok = local < a.n_rays;
        if (!ray_ok) local = a.n_rays - 1;
        {
            const Ray ray = load_ray(a, local);
            if (lane == 0) {
                *(f32x4*)ray_stash = f32x4{ray.o[0], ray.o[1], ray.o[2], ray.d[0]};
                ray_stash[4] = ray.d[1];
                ray_stash[5] = ray.d[2];
            }
        }
        asm volatile("" ::: "memory");            // the reads of the stash below stay below
        RayAccum racc;
        racc.reset();
        const int c_begin = kTrain ? (int)(unit - slot * ka.chunks) : 0;
        const int c_end = kTrain ? c_begin + 1 : ka.chunks;
        for (int c = c_begin; c < c_end; ++c) {
            const int s = c * kSamplesPerWave + j;
            const bool ok = s < S;
            const int sc = s < S - 1 ? s : S - 1;
            const int64_t tile = slot * ka.chunks + c;        // chunk index in the workspace
            const int64_t sp = tile * 16 + j;                 // padded sample index
            auto the_ray = [&]() {                // the wave's ray, back from LDS (broadcast reads)
                // (a compiler barrier: otherwise the reads, and what depends only on them, are hoisted out of
                //  the chunk loop and then spilled across every layer)
                asm volatile("" ::: "memory");
                Ray r;
                const f32x4 r0 = *(const f32x4*)ray_stash;
                r.o[0] = r0.x, r.o[1] = r0.y, r.o[2] = r0.z, r.d[0] = r0.w;
                r.d[1] = ray_stash[4], r.d[2] = ray_stash[5];
                return r;
            };
            // training: the wave's UNIFORM tile bases of the saved rows (+ the tensor's offset), and this lane's 32-bit
            // offsets inside a tile are taken where they are used (nerf_device.h: row_lane_offset — no per-lane 64-bit
            // pointer lives across the layers)
            float* const xrow = kTrain ? ws + tile * kTileFloats : nullptr;         // (tile-major rows)
            float* const stat = kTrain ? ws + tile * 16 : nullptr;                  // (+ j)
            f32x4 X[16], Y[16];                   // a layer's input tiles (B operands) / its accumulators, in turn
            {
                const Ray ray = the_ray();
                const float dlen = __builtin_sqrtf((ray.d[0] * ray.d[0] + ray.d[1] * ray.d[1]) + ray.d[2] * ray.d[2]);
                float posts[2];
                fencepost_run<2>(a, local, sc, posts);
                const float t0 = posts[0], t1 = posts[1];
                const float dist = s >= S - 1 ? 1e10f : dlen * (t1 - t0);
                if (g == 0) *stash_t0 = t0;
                if (kTrain) {
                    if (g == 0) *(f32x4*)(ws + ka.save.comp + sp * 4) = f32x4{0.f, 0.f, dist, 0.f};
                } else if (g == 0) {
                    *stash = f32x4{racc.carry, racc.rgb0, racc.rgb1, racc.rgb2};
                    *stash_dist = dist;
                }
                float pos_act[64];
                // (the encoders' per-lane constants — frequencies of lane group g — behind the optimisation barrier of
                //  nerf_device.h: lane_offset: hoisted out of the ray loop they are ten registers parked across every layer)
                const int ge = kTrain ? (int)lane_offset((uint32_t)g) : g;
                encode_position(ray, t0, la, ge, pos_act);
                if (kTrain) {                     // both encodings as rows (un-scaled, as the fp32 forward saves them)
                    float dir_act[64];
                    encode_direction(ray, dlen, la, ge, dir_act);
                    float* prow = ws + ka.save.pos + sp * kEncPad + 4 * g;
                    float* drow = ws + ka.save.dir + sp * kEncPad + 4 * g;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        *(f32x4*)(prow + 16 * t) = f32x4{pos_act[4 * t], pos_act[4 * t + 1], pos_act[4 * t + 2], pos_act[4 * t + 3]};
                        *(f32x4*)(drow + 16 * t) = t < kDirTiles ? f32x4{dir_act[4 * t], dir_act[4 * t + 1], dir_act[4 * t + 2], dir_act[4 * t + 3]}
                                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int t = 0; t < kPosTiles; ++t)
                    X[t] = f32x4{pos_act[4 * t], pos_act[4 * t + 1], pos_act[4 * t + 2], pos_act[4 * t + 3]} * kX;
            }
            LazyNorm norm;
            HMoments mom;
            // ---- L0: X (encoded position, 2 k blocks) -> Y ----
            load_bias(small, g, Y);
            nerf_fused::layer_fused_h<2, false, kTrain, kOrderReluNorm>(pipe, X, Y, norm, mom);
            norm = nerf_fused::finish_moments_at<kTrain, HMoments, kOrderReluNorm>(
                mom, Y, gamma_of(small), beta_of(small), g, xrow + ka.save.xhat[0], stat + ka.save.rstd[0], kEps, kRs,
                stat + ka.save.shift[0]);
            float dens = 0.f;
            // ---- L1 .. L9 in pairs (Y -> X, X -> Y): one code instance per direction ----
#pragma unroll 1
            for (int p = 0; p < 5; ++p) {
                const int la_ = 2 * p + 1;                                            // L1, L3, L5, L7, L9
                const float* sa = small + la_ * kLegacySmallPerLayer;
                load_bias(sa, g, X);
                nerf_fused::layer_fused_h<8, true, kTrain, kOrderReluNorm>(pipe, Y, X, norm, mom);
                norm = nerf_fused::finish_moments_at<kTrain, HMoments, kOrderReluNorm>(
                    mom, X, gamma_of(sa), beta_of(sa), g, xrow + ka.save.xhat[la_ < kWide ? la_ : 0],
                    stat + ka.save.rstd[la_ < kWide ? la_ : 0], kEps, kRs, stat + ka.save.shift[la_ < kWide ? la_ : 0]);
                if (p == 4) break;
                const float* sb = sa + kLegacySmallPerLayer;                          // L2, L4, L6, L8
                load_bias(sb, g, Y);
                // training: the layer's B operands (x' of the layer below as f16 pairs) stay with the caller for the
                // density head; the inference kernel, with fewer live addresses, is better off re-splitting the tiles
                h8 xb_hi[8], xb_lo[8];
                if constexpr (kTrain) nerf_fused::layer_fused_hb<8, true, kTrain, kOrderReluNorm>(pipe, X, Y, norm, mom, xb_hi, xb_lo);
                else nerf_fused::layer_fused_h<8, true, kTrain, kOrderReluNorm>(pipe, X, Y, norm, mom);
                if (p & 1) {
                    // L4: + [encoded position], L8: + [encoded direction]: two more k blocks, encoded here
                    h8 eh[2], el[2];
                    {
                        const Ray ray = the_ray();
                        float enc[64];
                        const int ge = kTrain ? (int)lane_offset((uint32_t)g) : g;
                        if (p == 1) encode_position(ray, *stash_t0, la, ge, enc, kX);
                        else encode_direction(ray, __builtin_sqrtf((ray.d[0] * ray.d[0] + ray.d[1] * ray.d[1]) + ray.d[2] * ray.d[2]),
                                              la, ge, enc, kX);
                        split_block(enc, 0, eh[0], el[0]);
                        split_block(enc, 1, eh[1], el[1]);
                    }
                    layer_wide_h<2, 0>(pipe, Y, eh, el);
                    mom.reset();
#pragma unroll
                    for (int T = 0; T < 16; ++T) mom.template add<kOrderReluNorm>(Y[T]);
                    if (p == 3) {                 // density head on x'_7: the pairs L8's fused loop made of it
                        const f32x4* hb = (const f32x4*)(small + kWide * kLegacySmallPerLayer);
                        // (the lane group behind an optimisation barrier: the address hb + g is otherwise formed once per
                        //  kernel and parked in scratch across the layers)
                        if constexpr (kTrain) dens = head_layer_hb(pipe, hb[g], xb_hi, xb_lo).x * kUn;
                        else dens = head_layer_h<false, kTrain>(pipe, hb[lane_offset((uint32_t)g)], X, norm).x * kUn;
                    }
                }
                norm = nerf_fused::finish_moments_at<kTrain, HMoments, kOrderReluNorm>(
                    mom, Y, gamma_of(sb), beta_of(sb), g, xrow + ka.save.xhat[la_ + 1], stat + ka.save.rstd[la_ + 1], kEps,
                    kRs, stat + ka.save.shift[la_ + 1]);
            }
            const f32x4* hb = (const f32x4*)(small + kWide * kLegacySmallPerLayer);
            const f32x4 col = head_layer_h<true, kTrain>(pipe, hb[4 + (kTrain ? (uint32_t)g : lane_offset((uint32_t)g))], X, norm) * kUn;
            f32x4 out[4];
            out[0] = f32x4{dens, col.x, col.y, col.z};
            out[1] = out[2] = out[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kTrain) {
                float* otile = ws + ka.save.out + tile * 1024 + lane * 4;
#pragma unroll
                for (int T = 0; T < 4; ++T) *(f32x4*)(otile + T * 256) = out[T];
                continue;
            }
            float dist;
            {
                const f32x4 s0 = *stash;
                racc.carry = s0.x, racc.rgb0 = s0.y, racc.rgb1 = s0.z, racc.rgb2 = s0.w;
                dist = *stash_dist;
            }
            const float w = composite_chunk<false, false>(a, S, local, s, ok, lane, out, dist, racc, nullptr);
            if (ray_ok && ok && g == 0) {
                const int64_t smp = local * S + s;
                if (a.out_weights != nullptr) a.out_weights[smp] = w;
                if (a.out_raw != nullptr) {
                    a.out_raw[smp * 4 + 0] = dens;
                    a.out_raw[smp * 4 + 1] = col.x;
                    a.out_raw[smp * 4 + 2] = col.y;
                    a.out_raw[smp * 4 + 3] = col.z;
                }
            }
        }
        if (!kTrain) store_ray<false>(a, local, ray_ok, lane, racc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// parameter re-layout: the checkpoint's 44 tensors (oracle/legacy_oracle.py: state_dict_keys order)
//   wide layer L = 0..9 (block_0 x4, block_1 x4, block_2 x2): params 4L .. 4L+3 = W, b, gamma, beta
//   with the two heads spliced in: density W, b at 32, 33; block_2 at 34..41; color W, b at 42, 43
// ---------------------------------------------------------------------------------------------
struct LegacyPackArgs {
    const float* p[NERF_HIP_LEGACY_PARAM_TENSORS];
    float* packed;
};


__global__ void nerf_legacy_pack_kernel(const LegacyPackArgs pa) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kLegacyPackedFloats) return;
    float v = 0.f;
    if (e >= kLegacyBwdHOffset) {
        // transposed f16-pair image of the data gradient (nerf_legacy_layout.h): two halfs of one slab per slot
        const int eb = e - kLegacyBwdHOffset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int slab = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, word = in_stage & 3;
        const int row = lane & 15, kg = lane >> 4;
        const int pair = slab >> 1;
        const bool is_lo = (slab & 1) != 0;
        int sl;
        const int L = bwd_h_layer_of_stage(stage, sl);
        _Float16 h[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int jj = 2 * word + k;
            float w = 0.f;
            if (L < 0) {
                // a head: one k block (m = 0), half = sl; its outputs sit in k slots (kg 0, jj 0..3) as in the
                // fp32 image: density at slot 0, color rows at slots 1..3
                const int in = 16 * (8 * sl + pair) + row;
                if (kg == 0 && jj < 4) {
                    if (L == -2 && jj == 0) w = pa.p[kDensityW][in];
                    if (L == -1 && jj >= 1) w = pa.p[kColorW][(jj - 1) * kHidden + in];
                }
            } else {
                const int half = sl / 8, m = sl % 8;
                const int out = 32 * m + 16 * (jj >> 2) + 4 * kg + (jj & 3);
                w = pa.p[wide_param(L)][out * wide_inputs(L) + 16 * (8 * half + pair) + row];
            }
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e >= kLegacyBwdOffset) {
        // transposed fp32 image of the data gradient (nerf_legacy_layout.h):
        // [lane (i, g)][r] = W[16 tout + 4 g + r][16 Tin + i], hidden columns only
        const int eb = e - kLegacyBwdOffset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int tin = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int i = lane & 15, g = lane >> 4;
        int tout;
        const int L = bwd_layer_of_stage(stage, tout);
        if (L < 0) {
            // a head: its outputs sit where the compositing backward leaves dL/d(density, r, g, b):
            // k slots (g 0, r 0) = density, (g 0, r 1..3) = color
            if (g == 0) {
                if (stage == kBwdDensityStage && r == 0) v = pa.p[kDensityW][16 * tin + i];
                if (stage == kBwdColorStage && r >= 1) v = pa.p[kColorW][(r - 1) * kHidden + 16 * tin + i];
            }
        } else {
            v = pa.p[wide_param(L)][(16 * tout + 4 * g + r) * wide_inputs(L) + 16 * tin + i];
        }
    } else if (e >= kLegacyHSmallOffset) {
        // small image of the split-precision path: bias * 2^12, gamma * 2^4, beta * 2^4, head biases * 2^12
        const int i = e - kLegacyHSmallOffset;
        const float sb = (float)(1 << (kWScaleLog2 + kXScaleLog2)), sx = (float)(1 << kXScaleLog2);
        if (i < kWide * kLegacySmallPerLayer) {
            const int L = i / kLegacySmallPerLayer, rem = i % kLegacySmallPerLayer;
            const int which = rem / kHidden, q = rem % kHidden;
            const int g = q / 64, T = (q % 64) / 4, reg = q & 3;
            v = pa.p[wide_param(L) + 1 + which][16 * T + 4 * g + reg] * (which == 0 ? sb : sx);
        } else {
            const int q = i - kWide * kLegacySmallPerLayer;
            const int head = q / 16, n = q % 16;
            if (head == 0 && n < 1) v = pa.p[33][n] * sb;
            if (head == 1 && n < 3) v = pa.p[43][n] * sb;
        }
    } else if (e >= kLegacyHOffset) {
        // f16-pair image: this float slot carries two halfs of one slab ([lane][8 halfs], 1 KiB)
        const int eb = e - kLegacyHOffset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int slab = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, word = in_stage & 3;
        const int row = lane & 15, kg = lane >> 4;
        const int pair = slab >> 1;
        const bool is_lo = (slab & 1) != 0;
        _Float16 h[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int jj = 2 * word + k, r = jj & 3;
            float w = 0.f;
            if (stage == kHDensityStage || stage == kHColorStage) {
                // head: pair = k block, out tile 0: A[row = output][k = 16 (2 pair + (jj >> 2)) + 4 kg + r]
                const bool color = stage == kHColorStage;
                const int nout = color ? 3 : 1;
                if (row < nout) w = pa.p[color ? 42 : 32][row * kHidden + 16 * (2 * pair + (jj >> 2)) + 4 * kg + r];
            } else {
                int L = 0, first = 0;
                for (L = 0; L < kWide; ++L) {
                    first = h_stage_of_layer(L);
                    if (stage >= first && stage < first + 2 * wide_blocks(L)) break;
                }
                // 8-block layers: stage (half, m) = half * 8 + m.  L4 / L8 (10 blocks): the 16 stages of the hidden
                // blocks first, then 4 stages (half, m - 8) of the two encoding blocks (their own short loop)
                const int KB = wide_blocks(L), sl = stage - first;
                int half, m;
                if (KB == 10) {
                    half = sl < 16 ? sl / 8 : (sl - 16) / 2;
                    m = sl < 16 ? sl % 8 : 8 + (sl - 16) % 2;
                } else {
                    half = sl / KB, m = sl % KB;
                }
                const int out = 16 * (8 * half + pair) + row;
                const int tt = 2 * m + (jj >> 2);             // register tile of the B operand
                const int K = wide_inputs(L);
                int col = -1;
                if (L == 0) {
                    const int q = 4 * tt + r;
                    if (q < kPosPerGroup) col = encoding_feature_of(kg, q, kPosFreqs);
                } else if (tt < 16) {
                    col = 16 * tt + 4 * kg + r;
                } else {
                    const int q = 4 * (tt - 16) + r;
                    if (L == 4 && q < kPosPerGroup) col = kHidden + encoding_feature_of(kg, q, kPosFreqs);
                    if (L == 8 && q < kDirPerGroup) col = kHidden + encoding_feature_of(kg, q, kDirFreqs);
                }
                if (col >= 0) w = pa.p[wide_param(L)][out * K + col];
            }
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e < kLegacyBlobFloats) {
        const int stage = e / kStageFloats;
        const int in_stage = e - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int row = lane & 15, g = lane >> 4;
        if (stage == kDensityStage || stage == kColorStage) {
            // head: quad = k-group t, A[row = output][k = 16 t + 4 g + r]
            const bool color = stage == kColorStage;
            const int nout = color ? 3 : 1;
            if (row < nout) v = pa.p[color ? 42 : 32][row * kHidden + 16 * quad + 4 * g + r];
        } else {
            int L = 0, first = 0;
            for (L = 0; L < kWide; ++L) {
                first = stage_of_layer(L);
                if (stage >= first && stage < first + wide_stages(L)) break;
            }
            const int t = stage - first;                    // k-group within the layer
            const int out = 16 * quad + row;
            const int K = wide_inputs(L);
            const float* W = pa.p[wide_param(L)];
            int col = -1;
            if (L == 0) {
                const int q = 4 * t + r;
                if (q < kPosPerGroup) col = encoding_feature_of(g, q, kPosFreqs);
            } else if (t < 16) {
                col = 16 * t + 4 * g + r;                   // hidden part: [hidden | encoding] order
            } else {
                const int q = 4 * (t - 16) + r;
                if (L == 4 && q < kPosPerGroup) col = kHidden + encoding_feature_of(g, q, kPosFreqs);
                if (L == 8 && q < kDirPerGroup) col = kHidden + encoding_feature_of(g, q, kDirFreqs);
            }
            if (col >= 0) v = W[out * K + col];
        }
    } else {
        const int i = e - kLegacyBlobFloats;
        if (i < kWide * kLegacySmallPerLayer) {
            const int L = i / kLegacySmallPerLayer, rem = i % kLegacySmallPerLayer;
            const int which = rem / kHidden, q = rem % kHidden;     // bias, gamma, beta in [g][T][reg] order
            const int g = q / 64, T = (q % 64) / 4, reg = q & 3;
            v = pa.p[wide_param(L) + 1 + which][16 * T + 4 * g + reg];
        } else {
            const int q = i - kWide * kLegacySmallPerLayer;          // head biases [head][g][reg]: row 4 g + reg
            const int head = q / 16, n = q % 16;
            if (head == 0 && n < 1) v = pa.p[33][n];
            if (head == 1 && n < 3) v = pa.p[43][n];
        }
    }
    pa.packed[e] = v;
}

}  // namespace

extern "C" {

size_t nerf_hip_legacy_packed_bytes(void) { return (size_t)kLegacyPackedFloats * sizeof(float); }

int nerf_hip_legacy_pack_weights(const float* const* params, float* packed, void* stream) {
    if (params == nullptr || packed == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_pack_weights: null pointer");
    LegacyPackArgs pa;
    for (int i = 0; i < NERF_HIP_LEGACY_PARAM_TENSORS; ++i) {
        if (params[i] == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "legacy_pack_weights: null tensor");
        pa.p[i] = params[i];
    }
    pa.packed = packed;
    const int threads = 256, blocks = (kLegacyPackedFloats + threads - 1) / threads;
    nerf_common::TimedLaunch timed((hipStream_t)stream, NERF_HIP_TIMING_PACK);
    hipLaunchKernelGGL(nerf_legacy_pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, pa);
    return nerf_common::check_hip(hipGetLastError(), "legacy_pack_weights launch");
}

size_t nerf_hip_legacy_train_workspace_bytes(int64_t n_rays, int32_t num_samples) {
    if (n_rays <= 0 || num_samples < 2) return 0;
    const int chunks = (num_samples + kSamplesPerWave - 1) / kSamplesPerWave;
    return (size_t)make_legacy_train_layout(n_rays, chunks).total * sizeof(float);
}

size_t nerf_hip_legacy_grad_elements(void) { return (size_t)kLegacyGradElements; }

int nerf_hip_legacy_render_forward(const NerfHipLegacyArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: null args");
    const NerfHipRenderArgs& a = args->render;
    if (a.n_rays == 0) return NERF_HIP_OK;
    if (a.n_rays < 0 || a.num_samples < 2 || a.num_samples > 4096)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: n_rays / num_samples out of range");
    if (a.packed == nullptr || a.rgb == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: packed / rgb is null");
    const bool arrays = a.rays_o != nullptr && a.rays_d != nullptr;
    const bool cameras = a.camera_o != nullptr && a.camera_r != nullptr && a.image_h > 0 && a.image_w > 0 &&
                         a.focal_length != 0.f;
    if (!arrays && !cameras)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: neither ray arrays nor cameras given");
    if ((a.rays_o == nullptr) != (a.rays_d == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: rays_o and rays_d must come together");
    if (a.seg != nullptr || a.out_mean != nullptr || a.out_cov != nullptr || a.out_t != nullptr || a.rng_mode != 0)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: seg / Gaussian outputs / in-kernel draws do not exist for this network");
    const bool train = a.train_workspace != nullptr;
    if (train && a.out_raw != nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: out_raw is not produced by the training forward");

    if (a.t_values == nullptr && a.t_table == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: t_table / t_values is null");
    if (!(args->normalize_position > 0.f))
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: normalize_position must be positive");

    LegacyKernelArgs ka;
    ka.l = *args;
    ka.l.render.color_outputs = 3;        // the shared compositing code stores / reads this many channels per ray
    ka.l.render.reserved = 0;
    ka.chunks = (a.num_samples + kSamplesPerWave - 1) / kSamplesPerWave;
    ka.save = make_legacy_train_layout(a.n_rays, ka.chunks);
    // inference: one ray per wave; training: one (padded ray, chunk) item per wave
    ka.groups = train ? ka.save.mp / 16 / kWavesPerWg : (a.n_rays + kWavesPerWg - 1) / kWavesPerWg;
    int device = 0, cus = 0;
    int rc = nerf_common::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (rc) return rc;
    rc = nerf_common::check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device),
                                "hipDeviceGetAttribute");
    if (rc) return rc;
    if (a.precision != NERF_HIP_PRECISION_FP32 && a.precision != NERF_HIP_PRECISION_F16X3)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_forward: unknown precision");
    const bool half = a.precision == NERF_HIP_PRECISION_F16X3;
    typedef void (*Kernel)(const LegacyKernelArgs);
    static const Kernel kernels[2][2] = {{nerf_legacy_fwd_kernel<false>, nerf_legacy_fwd_kernel<true>},
                                         {nerf_legacy_fwd_h_kernel<false>, nerf_legacy_fwd_h_kernel<true>}};
    static unsigned done[2][2] = {};
    const Kernel kernel = kernels[half][train];
    const int lds_bytes = half ? kLegacyLdsBytesHalf : kLegacyLdsBytes;
    rc = nerf_common::ensure_dynamic_lds((const void*)kernel, lds_bytes, device, &done[half][train]);
    if (rc) return rc;
    int64_t grid = (int64_t)cus * 2;
    if (grid > ka.groups) grid = ka.groups;
    hipStream_t st = (hipStream_t)stream;
    nerf_common::Timing::before(st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, ka);
    nerf_common::Timing::after(st, NERF_HIP_TIMING_FORWARD);
    if (train) {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_COMPOSITE_FORWARD);
        const int64_t blocks = (a.n_rays + kWavesPerWg - 1) / kWavesPerWg;
        hipLaunchKernelGGL(nerf_legacy_composite_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ka);
    }
    return nerf_common::check_hip(hipGetLastError(), "legacy_render_forward launch");
}

}  // extern "C"

// Backward building blocks shared by the two networks' backward files (gfx950 only): the wave-ordered
// gamma/beta partial sums, LayerNorm/ReLU backward on the register tile, the compositing backward, the
// f16-pair row scaling of the data gradient, and the weight-gradient GEMM (dW = dY^T X as one continuous
// stream of 16-sample k-steps over a 4-slot LDS ring, bf16-triple or f16-pair operands).
#ifndef NERF_BACKWARD_COMMON_H
#define NERF_BACKWARD_COMMON_H

#include <type_traits>

#include "nerf_device.h"

namespace nerf_bwd {

using namespace nerf_layout;
using namespace nerf_device;

constexpr int kMaxSplits = 128;
constexpr int kMaxDataGrid = 1024;

// gamma/beta gradient partials of a workgroup live in LDS ([layer][gamma|beta][256]).  Each wave
// row-reduces its 16 samples, then the four waves add their values in wave order, one wave per
// stage barrier of the MFMA loop that follows (no atomics: bitwise reproducible).
struct GammaBetaTurn {
    float* dst;                 // this lane's 4 gamma slots; beta slots at +256
    f32x4 kg, kb;
    int wave;
    __device__ __forceinline__ void operator()(int t) const {
        if (t < kWavesPerWg && wave == t) {
            f32x4* pg = (f32x4*)dst;
            f32x4* pb = (f32x4*)(dst + kHidden);
            *pg = *pg + kg;
            *pb = *pb + kb;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // landed before the next barrier
        }
    }
};

// Stage hook of the data-gradient MFMA loops: the wave-ordered gamma/beta adds, and at stage 1
// the loads of the NEXT LayerNorm backward's saved x_hat tile and 1/std, so that their HBM latency
// runs under this layer's MFMAs instead of in front of the LayerNorm arithmetic.  (Stage 1, not 0:
// the loads then sit behind one stage's DMA in the vmcnt queue and the next stage's counted wait
// retires them only after a whole stage of MFMAs.)
// kUniform: xhat_row / rstd_ptr are the wave's UNIFORM tile bases (scalar registers) and the lane's 32-bit offsets are
// taken here (nerf_device.h: row_lane_offset) — no per-lane 64-bit pointer per saved tensor lives across the layers.
template <int NT = 16, bool kUniform = false>
struct BwdHookN {
    GammaBetaTurn& turn;
    const float* xhat_row;      // this lane's 4 features of register tile 0 (tile-major rows: nerf_device.h)
    const float* rstd_ptr;
    f32x4 (&xh)[16];
    float& rstd;
    __device__ __forceinline__ void operator()(int t) const {
        turn(t);
        if (t == 1) {
            const float* row = kUniform ? xhat_row + row_lane_offset() : xhat_row;
#pragma unroll
            for (int T = 0; T < NT; ++T) xh[T] = *(const f32x4*)(row + T * kTileT);
            rstd = kUniform ? rstd_ptr[stat_lane_offset()] : *rstd_ptr;
        }
    }
};
typedef BwdHookN<16> BwdHook;
typedef BwdHookN<16, true> BwdHookU;

// LayerNorm + ReLU backward for hidden layer L on the register tile.
//   in : acc = dL/dx (post-ReLU activations), saved x_hat tile and 1/std
//   out: act = dL/dy (pre-LayerNorm output of the layer) = next B operands; also stored row-major
//   kScaled (split-precision chain): acc holds dL/dx times the per-sample power of two `unscale`
//   undoes (the B operands were scaled into the f16 range, the weights carry 2^kWScaleLog2)
template <bool kScaled = false, int NT = 16, bool kUniformRow = false>
__device__ __forceinline__ void layer_norm_relu_bwd(const float* small_l, int g, int j,
                                                    f32x4 (&acc)[16], float (&act)[64],
                                                    const f32x4 (&xh)[16], float rstd,
                                                    float* dy_row, float* gb_l, GammaBetaTurn& turn,
                                                    float inv_n, float unscale = 1.0f) {
    // inv_n = 1 / hidden_size: a narrower network's padded features (nerf_layout.h: Shape) have gamma = beta = 0,
    // so their gate is closed, their gamma dz is 0 and the two means below run over the real features only
    const f32x4* gam = (const f32x4*)(small_l + kSmallArrayLds + g * kSmallGStride);
    const f32x4* bet = (const f32x4*)(small_l + 2 * kSmallArrayLds + g * kSmallGStride);
    float s1 = 0.f, s2 = 0.f;
    // One tile: ReLU gate, gamma d z into the accumulator, the two LayerNorm moments; v[0..3] = d z (beta
    // gradient terms), v[4..7] = d z x_hat (gamma gradient terms) of the tile's four registers.
    auto tile = [&](int T, float (&v)[8]) {
        const f32x4 ga = gam[T], be = bet[T];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = __builtin_fmaf(xh[T][r], ga[r], be[r]);
            float dz;               // (if constexpr: a ?: on kScaled costs the fp32 kernel 60 spilled registers)
            if constexpr (kScaled) dz = z > 0.f ? acc[T][r] * unscale : 0.f;
            else dz = z > 0.f ? acc[T][r] : 0.f;
            v[r] = dz;
            v[4 + r] = dz * xh[T][r];
            const float gdz = ga[r] * dz;
            s1 += gdz;
            s2 = __builtin_fmaf(gdz, xh[T][r], s2);
            acc[T][r] = gdz;
        }
    };
    // beta / gamma gradients: sums over the 16 samples of a row, lane j keeps tile j.
    float kept[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (kScaled && NT == 16) {
        // split-precision chain (VALU-paced): a reduce-scatter butterfly (nerf_device.h: scatter_level8 / 4 /
        // take) applied as the tiles come: t, t + 8, t + 4, t + 12 — 2 DPP adds per value
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float va[8], vb[8], w0[8], w1[8], x[8];
            tile(t, va);
            tile(t + 8, vb);
            scatter_level8(va, vb, w0);
            tile(t + 4, va);
            tile(t + 12, vb);
            scatter_level8(va, vb, w1);
            scatter_level4(w0, w1, x);
            scatter_take(x, t, j, kept);
        }
    } else if constexpr (!kScaled && NT == 16) {
        // fp32 chain: HALF the butterfly — tiles T and T + 8 through scatter_level8 (lanes 0-7 then hold the pair sums of
        // tile T's eight values, lanes 8-15 of tile T + 8's), then an all-reduce inside each half row: 16 + 24 DPP adds
        // and 8 selects per tile pair where two tiles of row sums take 64 + 16, with 8 more values live (the full
        // butterfly, 32 more, costs this kernel 120 - 148 B of spills and was measured slower twice).  Data gradient
        // 1.278 -> 1.248 ms, step 3.38 -> 3.35 ms (same-box A/B).  (The legacy network's fp32 chain keeps the full
        // butterfly: with this form its allocation goes from 272 to 524 B of scratch.)
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            float va[8], vb[8], w[8];
            tile(T, va);
            tile(T + 8, vb);
            scatter_level8(va, vb, w);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = w[i];
                x += dpp<kQuadXor1>(0.f, x);
                x += dpp<kQuadXor2>(0.f, x);
                x += dpp<kRowHalfMirror>(0.f, x);
                if ((j & 7) == T) kept[i] = x;
            }
        }
    } else {
        // the narrow chains (NT = 8 / 4: the butterflies are built for sixteen tiles): one row_sum per value
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            float v[8];
            tile(T, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sum = row_sum(v[i]);
                if (j == T) kept[i] = sum;
            }
        }
    }
    const f32x4 keep_b = {kept[0], kept[1], kept[2], kept[3]}, keep_g = {kept[4], kept[5], kept[6], kept[7]};
    turn.dst = gb_l + 16 * j + 4 * g;             // features 16 j + 4 g + r, added in wave order later
    turn.kg = keep_g;
    turn.kb = keep_b;
    const float m1 = group_sum(s1) * inv_n;
    const float m2 = group_sum(s2) * inv_n;
    // dy = rstd ((gamma dz - m1) - x_hat m2) as two FMAs per element on the per-sample products (three instructions
    // as written; every VALU instruction of the fp32 chain is kernel time: NOTES.md section R6d)
    const float c1 = -(rstd * m1), c2 = rstd * m2;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        f32x4 dy;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dy[r] = __builtin_fmaf(-xh[T][r], c2, __builtin_fmaf(acc[T][r], rstd, c1));
            act[4 * T + r] = dy[r];
        }
        // (kUniformRow: dy_row is the wave's uniform tile base, the lane's offset is taken per store)
        *(f32x4*)(dy_row + (kUniformRow ? row_lane_offset() : 0u) + T * kTileT) = dy;
    }
}


// Compositing backward: one wave per padded ray slot, chunks walked last to first (the
// transmittance gradient is a suffix sum along the ray); writes dL/d(out) of every sample row-major
// ([sp][64]: the B operand of the data-gradient chain and the dY of layer 5's weight gradient).
struct CompositeBwd {
    const float* d_rgb;         // [n_rays,3]
    const float* d_seg;         // [n_rays,50] or NULL
    int32_t intervals, chunks;  // samples per ray that are composited, ceil(intervals / 16)
    int64_t mp;                 // padded samples of the workspace
    int64_t out, comp;          // workspace offsets: padded outputs (tile), compositing state
    float* dy5_rows;            // [mp][64] dL/d(out) rows, written here (in the workspace or in the backward's scratch)
};
__device__ __forceinline__ void composite_bwd_body(const NerfHipRenderArgs& a, const CompositeBwd& ba) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int P = ba.intervals;
    const int chunks = ba.chunks;
    float* const ws = a.train_workspace;
    const int64_t slot = (int64_t)blockIdx.x * kWavesPerWg + wave;
    if (slot * chunks * 16 >= ba.mp) return;
    int64_t local = slot;
    const bool ray_ok = local < a.n_rays;
    if (!ray_ok) local = a.n_rays - 1;

        // upstream gradients of this ray: this lane group's color channels 3 g .. 3 g + 2 (nerf_layout.h: color_slot),
        // zero for channels the network does not have
        const Shape sh = shape_of(a);
        const int C = sh.colors, c0 = 3 * g;
        const float g0 = ray_ok && c0 < C ? ba.d_rgb[local * C + c0] : 0.f;
        const float g1 = ray_ok && c0 + 1 < C ? ba.d_rgb[local * C + c0 + 1] : 0.f;
        const float g2 = ray_ok && c0 + 2 < C ? ba.d_rgb[local * C + c0 + 2] : 0.f;
        const bool with_seg = ba.d_seg != nullptr;

        float suffix = 0.f;                       // sum_{m in later chunks} q_m w_m
        // the chunks are a dependent chain (the suffix sum), their loads are not: chunk c - 1 is fetched while
        // chunk c is worked on
        f32x4 next_cmp, next_out[4];
        auto fetch = [&](int c) {
            const int64_t tile = slot * chunks + c;
            next_cmp = *(const f32x4*)(ws + ba.comp + (tile * 16 + j) * 4);   // alpha, T, dist, density
            const float* otile = ws + ba.out + tile * 1024 + lane * 4;
#pragma unroll
            for (int T = 0; T < 4; ++T) next_out[T] = *(const f32x4*)(otile + T * 256);
        };
        fetch(chunks - 1);
        for (int c = chunks - 1; c >= 0; --c) {
            const int s = c * kSamplesPerWave + j;
            const bool ok = ray_ok && s < P;
            const int64_t tile = slot * chunks + c;
            const int64_t sp = tile * 16 + j;

            // ---- compositing backward -> dL/d(out) in accumulator layout ----
            f32x4 dout[4];
            {
                const f32x4 cmp = next_cmp;
                const float alpha = cmp.x, t_excl = cmp.y, dist = cmp.z, dens = cmp.w;
                f32x4 out[4];
#pragma unroll
                for (int T = 0; T < 4; ++T) out[T] = next_out[T];
                if (c > 0) fetch(c - 1);
                const float w = ok ? (1.0f - alpha) * t_excl : 0.f;
                // colour: the logits of this lane group's channels in registers y, z, w of tile 0; dL/dw sums over all
                // channels, i.e. over the four lane groups of the sample (for 3 channels: lane group 0's value + zeros)
                const float sr = 1.0f / (1.0f + expf(-out[0].y));
                const float sg = 1.0f / (1.0f + expf(-out[0].z));
                const float sb = 1.0f / (1.0f + expf(-out[0].w));
                float q = group_sum((g0 * sr + g1 * sg) + g2 * sb);           // dL/dw, colour part
                // segmentation: v_sc = log(w + 1e-10) + log_softmax(x_s)[c]; seg_c = logsumexp_s v_sc
                float m = 0.f, logz = 0.f, lw = 0.f, srho = 0.f;
                float gseg[16], oseg[16];         // dL/dseg and forward seg of this lane's slots
#pragma unroll
                for (int i = 0; i < 16; ++i) gseg[i] = oseg[i] = 0.f;
                if (with_seg) {
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row_of_slot(16 * T + 4 * g + r, sh);
                            if (ray_ok && row > C) {
                                gseg[4 * T + r] = ba.d_seg[local * sh.classes() + (row - 1 - C)];
                                oseg[4 * T + r] = a.seg[local * sh.classes() + (row - 1 - C)];
                            }
                        }
                    m = -__builtin_inff();
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (is_seg_slot(T, g, r, a)) m = __builtin_fmaxf(m, out[T][r]);
                    m = group_max(m);
                    float z = 0.f;
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (is_seg_slot(T, g, r, a)) z += expf(out[T][r] - m);
                    logz = logf(group_sum(z));
                    lw = logf(w + 1e-10f);
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (is_seg_slot(T, g, r, a)) {
                                const float rho = expf(lw + ((out[T][r] - m) - logz) - oseg[4 * T + r]);
                                srho = __builtin_fmaf(gseg[4 * T + r], rho, srho);
                            }
                    srho = ok ? group_sum(srho) : 0.f;
                    q += srho / (w + 1e-10f);
                }
                const float qw = ok ? q * w : 0.f;
                const float suf_incl = row_suffix_sum(qw);
                const float suf_excl = suffix + row_shift_down(0.f, suf_incl);
                suffix += __shfl(suf_incl, lane & 48);
                const float dalpha = -q * t_excl + suf_excl / (alpha + 1e-10f);
                const float dsigma = (ok && dens > 0.f) ? dalpha * (-dist * alpha) : 0.f;
#pragma unroll
                for (int T = 0; T < 4; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = 0.f;
                        if (with_seg && is_seg_slot(T, g, r, a)) {
                            const float lp = (out[T][r] - m) - logz;
                            const float rho = expf(lw + lp - oseg[4 * T + r]);
                            v = gseg[4 * T + r] * rho - expf(lp) * srho;
                        }
                        dout[T][r] = v;
                    }
                if (g == 0) dout[0].x = dsigma;
                if (c0 < C) dout[0].y = g0 * w * sr * (1.0f - sr);
                if (c0 + 1 < C) dout[0].z = g1 * w * sg * (1.0f - sg);
                if (c0 + 2 < C) dout[0].w = g2 * w * sb * (1.0f - sb);
                if (!ok) {
#pragma unroll
                    for (int T = 0; T < 4; ++T) dout[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                float* drow = ba.dy5_rows + sp * kOutPad + 4 * g;
#pragma unroll
                for (int T = 0; T < 4; ++T) *(f32x4*)(drow + T * 16) = dout[T];
            }

        }
}


// Split-precision data gradient: a sample's dY row becomes f16 pairs after an exact per-sample
// power-of-two scaling that puts its largest magnitude in [2^12, 2^13) (the chain is linear in dY,
// so the scale is undone on the accumulators; f16 keeps 11 bits per half down to 2^-14, i.e. an
// element 2^-16 of the row's largest still has all 22 bits).  Returns the scale; `unscale` also
// removes the weights' 2^kWScaleLog2.
__device__ __forceinline__ float row_scale(float amax_lane, float& unscale, float& amax) {
    amax = group_max(amax_lane);
    uint32_t e = __builtin_bit_cast(uint32_t, amax) >> 23;            // amax >= 0
    e = e < 32u ? 32u : e;                                            // rows below 2^-95: treated as 2^-95
    unscale = __builtin_bit_cast(float, (e - 12u - (uint32_t)kWScaleLog2) << 23);
    return __builtin_bit_cast(float, (266u - e) << 23);               // 2^(12 - (e - 127))
}
__device__ __forceinline__ float abs_max4(float m, const f32x4& v) {
    m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)));
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
}

// Stage hook of the split-precision loops: only the wave-ordered gamma/beta adds (the x_hat
// prefetch is issued before the loop: a 0.2 us stage cannot hide an HBM load behind one hand-over).
// The weight-gradient kernel's f16-pair form needs ONE scale per layer for the whole batch: every
// wave folds its samples' maxima into an LDS word of its workgroup (integer max on the bits of a
// non-negative float: order-independent, so still bitwise reproducible).
__device__ __forceinline__ void note_max(int* word, float amax, int lane) {
    const float w = row_max(amax);                // lanes 0..15 = the wave's 16 samples
    if (lane == 0) atomicMax(word, __builtin_bit_cast(int, w));
}

struct TurnHook {
    GammaBetaTurn& turn;
    __device__ __forceinline__ void operator()(int t) const { turn(t); }
};

// ---------------------------------------------------------------------------------------------
// weight gradients: dW[out][in] = sum_s dY[s][out] X[s][in], split over sample tiles
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKs = 32;                          // samples per LDS tile

// kMap: how the four waves split the [OUT_W x IN_W] product into 32 x 32 accumulator tiles
//   kMapGrid: 2 x 2 waves of TO x TI tiles each;  kMapRows: wave w takes out tiles TO w .. TO w + TO - 1 and
//   all TI in tiles;  kMapCols: all TO out tiles, in tiles TI w .. TI w + TI - 1
//   kMapPrivate: every wave the WHOLE TO x TI block of its OWN job — the caller hands each wave its own dY / X /
//   slab offsets (the four hidden layers of a network of hidden_size <= 64: one 2 x 2 block each), the wave fetches
//   its own operands into its own part of the ring slot and shares nothing with the others but the barriers
//   kMapPrivatePair: the same with TWO jobs in the workgroup — the even waves run one shape, the odd waves another (the
//   caller branches on wave & 1), waves 2 and 3 repeat waves 0 and 1 to the same places (same bytes, same results)
constexpr int kMapGrid = 0, kMapRows = 1, kMapCols = 2, kMapPrivate = 3, kMapPrivatePair = 4;
// HALF: of a TILED operand (128-wide saved rows) only register tiles 0..3 of every 16-sample block are fetched — a
// network of hidden_size <= 64 that computes its forward and data gradient at 4 tiles never writes the others; their
// places in the ring slots are zeroed once, at the top of the kernel (nerf_backward.hip: nerf_wgrad_n4_kernel).
// SLOT_KIB: bytes of a ring slot of the kernel the shape runs in (all shapes of one kernel share it): 32 KiB, one
// workgroup per CU.  (16 KiB slots — a 64 KiB ring, TWO workgroups per CU — were measured on the kernels that read
// 128-wide rows: the half-fetch 4-tile form 0.330 -> 0.270 ms, the 8-tile ones 0.366 -> 0.362 ms in fp32 and 0.321 ->
// 0.333 ms on f16 pairs; the 4-tile kernel then took the per-layer wave map below, which needs the 32 KiB.)
// REGION_KIB (kMapPrivatePair): bytes of a wave's own part of a ring slot — the larger of the two shapes' needs.
template <int OUT_W, int IN_W, int TO, int TI, int MAP, int SLAB_STRIDE = IN_W, bool HALF = false, int SLOT_KIB = 32,
          int REGION_KIB = 0>
struct WgradShape {
    static constexpr int kMap = MAP;
    static constexpr bool kHalf = HALF;
    static constexpr int kSlotBytes = SLOT_KIB * 1024;
    static constexpr int kRegionBytes = REGION_KIB * 1024;
    static constexpr int kOutW = OUT_W, kInW = IN_W, kTo = TO, kTi = TI;
    static constexpr int kSlabStride = SLAB_STRIDE;            // floats between two rows of the product in the partial slab
    static constexpr int kDyBytes = kKs * OUT_W * 4, kXBytes = kKs * IN_W * 4;
    static constexpr int kTileBytes = kDyBytes + kXBytes;
    static constexpr int kPieces = kTileBytes / 1024;          // 1 KiB LDS-DMA pieces per tile
    static constexpr int kPiecesPerWave = kPieces / 4;
    static_assert(kPieces % 4 == 0, "pieces must split over 4 waves");
};
typedef WgradShape<kHidden, kEncIn, 2, 3, kMapRows> ShapeL0;        // waves: out tiles 2w..2w+1, all 3 in tiles
typedef WgradShape<kHidden, kHidden, 4, 4, kMapGrid> ShapeHid;       // waves 2x2: 4x4 tiles each
typedef WgradShape<kOutPad, kHidden, 2, 2, kMapCols> ShapeL5;        // waves: both out tiles, in tiles 2w..2w+1
// A network that trains at 8 register tiles per sample (hidden_size <= 128, fp32 arithmetic): saved rows 128 wide, the
// products land in the same full-width slab (row stride of the full-width tensors), so the reduce kernel does not change.
typedef WgradShape<128, 128, 2, 2, kMapGrid, kHidden> ShapeHidN8;    // waves 2x2: 2x2 tiles each
typedef WgradShape<kOutPad, 128, 2, 1, kMapCols, kHidden> ShapeL5N8; // waves: both out tiles, in tile w
// layer 0 (128 x 96 = 4 x 3 tiles): waves 0 and 1 take out tiles 0..1 and 2..3 with all three in tiles; waves 2 and 3
// run the same code on out tiles 4..7, which the 128-wide dY does not have — they read whatever lies behind it in the
// ring slot and write rows 128..255 of the full-width slab, which the reduce kernel never reads for such a network
// (an odd out-tile count per wave is not an option: the A operand sets ping-pong slot by slot).
typedef WgradShape<128, kEncIn, 2, 3, kMapRows> ShapeL0N8;
// ... and at 4 (hidden_size <= 64, fp32 arithmetic): the 8-tile shapes and wave maps on HALF the bytes
typedef WgradShape<128, 128, 2, 2, kMapPrivate, kHidden, true> ShapeHidN4;       // wave w: hidden layer w + 1
// layers 0 and 5 as ONE job: the even waves take layer 0 (64 x 96: 2 x 3 tiles), the odd ones layer 5 (64 x 64: 2 x 2)
typedef WgradShape<kOutPad, 128, 2, 2, kMapPrivatePair, kHidden, true, 32, 10> ShapeL5N4;
typedef WgradShape<128, kEncIn, 2, 3, kMapPrivatePair, kEncIn, true, 32, 10> ShapeL0N4;

// ---------------------------------------------------------------------------------------------
// The GEMM with every fp32 operand as a bf16 TRIPLE (hi + mid + lo = all 24 significand
// bits, by truncation, and bf16 has fp32's exponent range, so gradients of any magnitude are
// represented exactly — an f16 pair would need a data-dependent scale for dY) and six
// v_mfma_f32_32x32x16_bf16 per product: hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi, fp32
// accumulate; the dropped terms are <= 2^-24 relative.  16 samples per MFMA instead of 2:
// 96 x 32 cycles per 16 samples against 128 x 64.
//   A operand: lane l holds dY[sample 8 (l >> 5) + jj][out row l & 31], jj = 0..7
//   B operand: lane l holds  X[sample 8 (l >> 5) + jj][in  col l & 31]
//   C/D: as v_mfma_f32_32x32x2_f32 (the epilogue and the reduce kernel do not change)
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Bf3 {
    bf8 h, m, l;
};

// eight fp32 values -> their (hi, mid, lo) bf16 truncations, element jj = value jj
__device__ __forceinline__ Bf3 split_bf3(const float (&x)[8]) {
    u32x4 ph, pm, pl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hb[2], mb[2], lb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = x[2 * p + e];
            hb[e] = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
            const float r1 = v - __builtin_bit_cast(float, hb[e]);          // exact
            mb[e] = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, mb[e]);         // exact
            lb[e] = __builtin_bit_cast(unsigned, r2);
        }
        // upper halves of (value 2p, value 2p + 1) -> one dword, value 2p in the low half
        ph[p] = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
        pm[p] = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
        pl[p] = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
    }
    Bf3 r;
    r.h = __builtin_bit_cast(bf8, ph);
    r.m = __builtin_bit_cast(bf8, pm);
    r.l = __builtin_bit_cast(bf8, pl);
    return r;
}

__device__ __forceinline__ f32x16 mfma_bf(const bf8& a, const bf8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// The bf16-triple GEMM as ONE continuous stream of 16-sample k-steps over a 4-slot LDS ring
// (round 2; round 1's form — 32-sample tiles in two buffers — is in the history: commit 39d705c).
// Why: with one wave per SIMD nothing hides a tile hand-over.  The two-buffer form exposed, per
// 32-sample tile, the conversion of the tile's first operands (5 operands x 8 values x ~7 VALU)
// and a vmcnt(0) that waited for a tile whose DMA had been issued only half a tile earlier; the
// matrix pipe was ~50 % busy.  Here
//   * a ring slot holds ONE k-step (16 samples: [16][OutW] dY then [16][InW] X);
//   * during step t the VALU converts A(t, a + 1) from slot t and, for step t + 1, A(t + 1, 0) and
//     all B operands from slot t + 1, so no conversion is ever exposed after the prologue;
//   * the DMA of step t + 3 is issued during step t into the slot step t - 1 just vacated, i.e. it
//     has two whole steps to land; the hand-over at the end of step t waits only for this wave's
//     pieces of step t + 2 (counted vmcnt: the pieces of step t + 3 stay in flight), then one
//     barrier makes every wave's pieces visible and proves every wave has left step t
//     (write-after-read safety of the slot that step t + 4's DMA takes next).
// The DMA is issued from inline asm (as in WeightPipe::issue): with the builtin the compiler's
// wait-count pass would put a vmcnt(0) in front of every LDS read.
// ---------------------------------------------------------------------------------------------
constexpr int kRingStep = 16;                    // samples per ring slot = one MFMA k-step
constexpr int kRingSlots = 4;
// A 256-wide operand arrives TILE-MAJOR (nerf_device.h): a k-step's 16 KiB are 16 register tiles T of 1 KiB, each
// 64 chunks of 16 bytes (4 features of one sample) in the order [g][sample], and a DMA piece is one tile.  Which chunk
// a DMA lane fetches is a per-lane global offset, so the order of the chunks INSIDE the LDS copy of a piece is free;
// it is [s & 3][s >> 2][g] — word (s & 3) * 64 + (s >> 2) * 16 + (f & 15) — for two reasons:
//  * a lane reads samples 8 kk + jj, jj = 0..7, of feature tiles X = 0..3 (lane = feature 32 X + i): with samples s,
//    s + 1, s + 2, s + 3 exactly 64 words apart all 32 reads are (one lane base per half jj >> 2) + a multiple of 64
//    words, ds_read2st64_b32's immediate offsets.  With the samples 16 words apart the reads needed a base per X:
//    28 more v_add in a loop of 836 instructions that is issue-bound at one wave per SIMD — the kernel 4 % slower.
//  * lanes i < 16 read tile 2 X and lanes i >= 16 tile 2 X + 1 at the same offset inside the tile, which are the same
//    ds_read_b32 banks ((word address) % 32, per 32-lane half): a 2-way conflict on every read (measured: 6 %).
//    ODD tiles are therefore SAMPLE-SWIZZLED on their way in — the chunk of sample s holds sample s ^ 4 — which
//    moves them 16 banks from their even neighbours.  (Padding the odd tiles by 64 bytes instead does the same for
//    the banks, but DMA pieces that are no longer 1 KiB-aligned in LDS made the kernel 9 % slower.)
// A lane's read of (sample 8 kk + jj, feature 32 X + i) is word
//   X * 512 + (i >> 4) * 256 + (jj & 3) * 64 + ((2 kk + (jj >> 2)) ^ (i >> 4)) * 16 + (i & 15).
// Narrow operands (the 96 encoded inputs, the 64 padded outputs) are row-major [16][W] and copied as they are.
template <int W>
struct SlotLayout {
    static constexpr bool kTiled = W == kHidden || W == 128;      // (96 and 64 are the row-major operands)
    static constexpr int kPieces = kRingStep * W * 4 / 1024;
    static constexpr int kBytes = kPieces * 1024;
    // word offset of (sample 8 kk + jj, feature 32 X + i) = lane(kk, i, jj >> 2) + step(jj, X)
    __device__ static constexpr int lane(int kk, int i, int hi_jj) {
        return kTiled ? (i >> 4) * 256 + ((2 * kk + hi_jj) ^ (i >> 4)) * 16 + (i & 15) : (8 * kk + 4 * hi_jj) * W + i;
    }
    __device__ static constexpr int step(int jj, int X) { return kTiled ? (jj & 3) * 64 + X * 512 : (jj & 3) * W + 32 * X; }
};
struct OperandRows {          // a lane's view of one operand of one slot: rows[jj >> 2] + step(jj, X)
    const float* rows[2];
};
template <class L>
__device__ __forceinline__ OperandRows operand_rows(const char* region, int kk, int i, int X0) {
    const float* r = (const float*)region + L::step(0, X0);
    return OperandRows{{r + L::lane(kk, i, 0), r + L::lane(kk, i, 1)}};
}
constexpr int kRingSlotBytes = kRingStep * (kHidden + kHidden) * 4;       // 32 KiB (hidden shape)

// N (<= 4) consecutive 1 KiB pieces, the first an EVEN one: global (uniform base + k KiB + this lane's 16 bytes)
// -> LDS (base + k KiB + lane * 16).  kTiled: LDS chunk `lane` = (s & 3, s >> 2, g) takes the tile's chunk (g, s) —
// of sample s ^ 4 in odd pieces.
// (first_odd: the first piece is an odd one — wave-uniform; only the half-fetch shapes issue single pieces)
template <int N, bool kTiled>
__device__ __forceinline__ void ring_dma(const char* src, char* dst, int lane, int first_odd = 0) {
    static_assert(N >= 1 && N <= 4, "immediate offsets reach 3 KiB");
    const uint64_t base_u = (uint64_t)(uintptr_t)src;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
    const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);
    const int sample = 4 * ((lane >> 2) & 3) + (lane >> 4);               // of LDS chunk `lane`
    const int chunk_first = kTiled ? (lane & 3) * 16 + sample : lane;
    const int chunk_even = kTiled ? chunk_first ^ (first_odd << 2) : lane;      // (of this call's pieces 0, 2)
    const int chunk_odd = kTiled ? chunk_even ^ 4 : lane;                        // (pieces 1, 3)
    uint32_t m0_saved;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 2\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        ".if %c4 > 1\n\tglobal_load_lds_dwordx4 %5, %3 offset:1024\n\t.endif\n\t"
        ".if %c4 > 2\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\t.endif\n\t"
        ".if %c4 > 3\n\tglobal_load_lds_dwordx4 %5, %3 offset:3072\n\t.endif\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_saved)
        : "v"(chunk_even * 16), "s"(d), "s"(sbase), "n"(N), "v"(chunk_odd * 16)
        : "memory");
}

// DMA instructions every wave issues per k-step (the same count on every wave: the hand-over's
// vmcnt is an immediate).  A k-step of dY is 16 kOutW contiguous floats in memory, of X 16 kInW (row-major or
// tile-major: the same 1 KiB pieces either way).
template <class Sh>
struct RingPlan {
    typedef SlotLayout<Sh::kOutW> Dy;
    typedef SlotLayout<Sh::kInW> X;
    static constexpr int kDyPieces = Dy::kPieces;                         // 16 / 16 / 4
    static constexpr int kXPieces = X::kPieces;                           // 6 / 16 / 16
    // pieces FETCHED per k-step (WgradShape: HALF): the first half of a tiled operand's
    static constexpr int kDyFetch = Dy::kTiled && Sh::kHalf ? kDyPieces / 2 : kDyPieces;
    static constexpr int kXFetch = X::kTiled && Sh::kHalf ? kXPieces / 2 : kXPieces;
    static constexpr bool kPair = Sh::kMap == kMapPrivatePair;
    static constexpr bool kPrivate = Sh::kMap == kMapPrivate || kPair;
    static constexpr int kDyPerWave = kPrivate ? kDyFetch : (kDyFetch + 3) / 4, kXPerWave = kPrivate ? kXFetch : (kXFetch + 3) / 4;
    // a wave's own part of a slot
    static constexpr int kWaveBytes = !kPrivate ? 0 : (Sh::kRegionBytes ? Sh::kRegionBytes : (kDyFetch + kXFetch) * 1024);
    static constexpr int kPerWave = kDyPerWave + kXPerWave;               // 6 / 8 / 5
    static constexpr int kXOffset = kPrivate ? kDyFetch * 1024 : Dy::kBytes;       // X behind dY in the slot (or the wave's part)
    static_assert(kPrivate ? (kPair ? 2 : 4) * kWaveBytes <= Sh::kSlotBytes && (kDyFetch + kXFetch) * 1024 <= kWaveBytes
                           : Dy::kBytes + X::kBytes <= Sh::kSlotBytes, "a k-step must fit its slot");
    static_assert((!Dy::kTiled || kDyFetch % 4 == 0) && (!X::kTiled || kXFetch % 4 == 0),
                  "the waves' shares of a tiled operand are whole and equal");
    static_assert(!kPrivate || (kDyPerWave <= 8 && kXPerWave <= 8), "at most two DMA calls per operand");
};

// part 0: this wave's dY pieces of the step, part 1: its X pieces.  A wave whose share would run
// past the region re-fetches the region's last pieces instead (same bytes to the same place).
template <class Sh, int kPart>
__device__ __forceinline__ void ring_issue_part(const float* dy, const float* x, int64_t sample0, char* slot,
                                                int wave, int lane) {
    typedef RingPlan<Sh> P;
    constexpr int total = kPart == 0 ? P::kDyFetch : P::kXFetch;
    constexpr int per = kPart == 0 ? P::kDyPerWave : P::kXPerWave;
    int first = P::kPrivate ? 0 : wave * per;
    if (first + per > total) first = total - per;
    const char* src = kPart == 0 ? (const char*)(dy + sample0 * Sh::kOutW) : (const char*)(x + sample0 * Sh::kInW);
    char* dst = slot + (kPart == 0 ? 0 : P::kXOffset);
    constexpr bool tiled = kPart == 0 ? P::Dy::kTiled : P::X::kTiled;
    static_assert(!tiled || per % 2 == 0 || per == 1, "first piece of a wave even, or single pieces");
    if constexpr (P::kPrivate && per > 4) {               // (a wave's whole operand: 4 pieces, then the rest)
        ring_dma<4, tiled>(src, dst, lane);
        ring_dma<per - 4, tiled>(src + 4096, dst + 4096, lane);
    } else {
        ring_dma<per, tiled>(src + first * 1024, dst + first * 1024, lane, tiled && per == 1 ? first & 1 : 0);
    }
}

struct H2 {                   // an operand as f16 pairs: value = (h + l) / scale
    h8 h, l;
};
__device__ __forceinline__ f32x16 mfma_hw(const h8& a, const h8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// kF16 (the split-precision training mode): operands as f16 PAIRS instead of bf16 triples — three
// v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones, and 4 instead of ~7 VALU per value.
// X = relu(gamma x_hat + beta) (or the encoded inputs) is O(1) and enters times 2^4 as in the
// forward; dY enters times ONE power of two per layer and launch, chosen from the largest |dY| of
// the whole batch (`dymax`, written by the data-gradient kernel) so that it lands in [2^12, 2^13):
// a sample's row cannot have its own scale here because the product sums over samples.  Elements
// more than 2^16 below the batch maximum lose relative precision, but never more than 2^-38 of that
// maximum absolutely — below fp32 rounding of any sum the large elements take part in.  (The bf16
// form needs no scale: bf16 has fp32's exponent range; it stays the fp32 training mode's arithmetic.)
// One split of one job: which sample tiles it sums, where its partial slab goes.
struct WgradJob {
    int64_t tiles_per_split, n_tiles;             // 32-sample tiles per split, in all
    int split;
    float* slab;                                  // this split's partial slab
    const float* dymax;                           // [data_grid][dymax_stride] batch maxima of |dY| (f16 form), else unused
    int data_grid;
    int dymax_stride;
};
// "1 MFMA, the slot's raw LDS reads (at most), then 1 MFMA + 6 VALU" as one scheduling pipeline with its own sync id
template <int kSync, int kMfmas, int kReads>
__device__ __forceinline__ void mfma_valu_pattern() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, kSync);
    __builtin_amdgcn_sched_group_barrier(0x100, kReads, kSync);
#pragma unroll
    for (int m = 1; m < kMfmas; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, kSync);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, kSync);
    }
}

// kInput: what an X row holds — kInputRaw: the layer's input itself (encoded features);
// kInputAffineRelu: the previous layer's saved x_hat, input = relu(gamma x_hat + beta) (the main network:
// Linear -> LayerNorm -> ReLU); kInputAffine: input = gamma x_hat + beta (the legacy network: Linear -> ReLU ->
// LayerNorm, the LayerNorm output feeds the next Linear directly)
constexpr int kInputRaw = 0, kInputAffineRelu = 1, kInputAffine = 2;
template <class Sh, int kInput, bool kF16 = false>
__device__ __forceinline__ void wgrad_body_ring(const WgradJob& ba, char* smem, const float* dy, const float* x,
                                                const float* small_prev, int w_off, int b_off,
                                                int max_index = 0) {
    constexpr bool kAffine = kInput != kInputRaw;
    typedef RingPlan<Sh> P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int out0, in0;                                // first 32-wide tile of this wave
    if (Sh::kMap == kMapGrid) { out0 = Sh::kTo * (wave >> 1); in0 = Sh::kTi * (wave & 1); }
    else if (Sh::kMap == kMapRows) { out0 = Sh::kTo * wave; in0 = 0; }
    else if (Sh::kMap == kMapPrivate || Sh::kMap == kMapPrivatePair) { out0 = 0; in0 = 0; }
    else { out0 = 0; in0 = Sh::kTi * wave; }

    float ga[Sh::kTi], be[Sh::kTi];
#pragma unroll
    for (int b = 0; b < Sh::kTi; ++b) {
        const int f = 32 * (in0 + b) + (lane & 31);
        const int idx = (((f & 15) >> 2) * 16 + (f >> 4)) * 4 + (f & 3);
        ga[b] = kAffine ? small_prev[kHidden + idx] : 1.0f;
        be[b] = kAffine ? small_prev[2 * kHidden + idx] : 0.f;
        if (kF16) {                               // X enters the MFMAs times 2^kXScaleLog2
            ga[b] *= (float)(1 << kXScaleLog2);
            be[b] *= (float)(1 << kXScaleLog2);
        }
    }
    float a_scale = 1.0f, un_scale = 1.0f;        // dY scale and what divides it (and X's) out again
    if (kF16) {
        float m = 0.f;
        for (int q = threadIdx.x; q < ba.data_grid; q += 256) m = __builtin_fmaxf(m, ba.dymax[(int64_t)q * ba.dymax_stride + max_index]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
        float* red = (float*)smem;
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
        __syncthreads();                          // the ring's DMA may overwrite `red` from here on
        uint32_t e = __builtin_bit_cast(uint32_t, m) >> 23;
        e = e < 32u ? 32u : e;
        a_scale = __builtin_bit_cast(float, (266u - e) << 23);                      // 2^(12 - (e - 127))
        un_scale = __builtin_bit_cast(float, (e - 12u - (uint32_t)kXScaleLog2) << 23);
    }

    f32x16 acc[Sh::kTo][Sh::kTi];
#pragma unroll
    for (int a = 0; a < Sh::kTo; ++a)
#pragma unroll
        for (int b = 0; b < Sh::kTi; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[Sh::kTo];
#pragma unroll
    for (int a = 0; a < Sh::kTo; ++a) bsum[a] = 0.f;

    const int64_t tile_begin = (int64_t)ba.split * ba.tiles_per_split;
    int64_t tile_end = tile_begin + ba.tiles_per_split;
    if (tile_end > ba.n_tiles) tile_end = ba.n_tiles;
    const int64_t n_steps = tile_end > tile_begin ? 2 * (tile_end - tile_begin) : 0;   // even
    const int64_t sample_begin = tile_begin * kKs;
    const int i = lane & 31, kk = lane >> 5;

    auto slot_of = [&](int64_t t) {
        return smem + (int)(t & (kRingSlots - 1)) * Sh::kSlotBytes + (P::kPair ? wave & 1 : wave) * P::kWaveBytes;
    };
    auto issue_step = [&](int64_t t) {            // all pieces of step t (prologue)
        ring_issue_part<Sh, 0>(dy, x, sample_begin + t * kRingStep, slot_of(t), wave, lane);
        ring_issue_part<Sh, 1>(dy, x, sample_begin + t * kRingStep, slot_of(t), wave, lane);
    };
    auto b_value = [&](float raw, int b) {
        if (kInput == kInputAffineRelu) return __builtin_fmaxf(__builtin_fmaf(raw, ga[b], be[b]), 0.f);
        if (kInput == kInputAffine) return __builtin_fmaf(raw, ga[b], be[b]);
        return kF16 ? raw * ga[b] : raw;
    };

    constexpr int kBPerSlot = (Sh::kTi + Sh::kTo - 1) / Sh::kTo;
    constexpr int kPerProduct = kF16 ? 3 : 6;
    constexpr int kMfmas = kPerProduct * Sh::kTi;             // per slot
    constexpr int kLead = kMfmas / 6;
    // conversion work items per operand of the bf16 form: 8 values + 2 packing items (the f16 form has its own
    // micro-operations, below)
    constexpr int kItemsPerOp = 10;
    constexpr int kItems = kItemsPerOp * (1 + kBPerSlot);
    typedef typename std::conditional<kF16, H2, Bf3>::type Operand;
    static_assert(Sh::kTo % 2 == 0, "the A operand sets ping-pong slot by slot");

    // One k-step: MFMAs on (at[0] of slot 0, bcur) while the VALU builds the next A operands and bnext.
    // at[2]: A operand sets, slot a uses at[a & 1] and converts into at[(a & 1) ^ 1].
    Operand at[2];
    // last_tag: the job's last step converts nothing for a step behind it (its own code instance, so the
    // steady-state steps carry no selects: a v_cndmask costs 18 cycles here, a plain VALU op 5)
    auto k_step = [&](auto last_tag, int64_t t, Operand (&bcur)[Sh::kTi], Operand (&bnext)[Sh::kTi]) {
        constexpr bool kLast = decltype(last_tag)::value;
        constexpr bool has_next = !kLast;
        // one lane base per source; everything else in an address is a compile-time constant, so the
        // reads take immediate offsets (and pair up as ds_read2st64_b32) instead of one v_add each
        const OperandRows dyt = operand_rows<typename P::Dy>(slot_of(t), kk, i, out0);
        const OperandRows dyn = operand_rows<typename P::Dy>(slot_of(t + 1), kk, i, out0);
        const OperandRows xn = operand_rows<typename P::X>(slot_of(t + 1) + P::kXOffset, kk, i, in0);
        // The DMA of step t + 3 goes out UNCONDITIONALLY: behind the last step it re-fetches the last step into a
        // slot nobody reads any more.  (A run-time branch around the issue ends the scheduling region of the slot it
        // sits in: the first two slots of every step came out as 12 MFMAs | branch | DMA | all conversions — 850-920
        // cycles against 520-540 for the two slots without a DMA; scripts/experiments/stamps_wgrad_h.py.)
        const int64_t t_fill = t + 3 < n_steps ? t + 3 : n_steps - 1;
        char* fill = slot_of(t + 3);
        // the raw values slot a converts (A operand of slot a + 1, B operands of the next step): two register sets —
        // the f16 form reads slot a + 1's in the middle of slot a, when slot a's own are dead (level 0 is done)
        float raw2[2][1 + kBPerSlot][8];
        auto read_raw = [&](int a, float (&raw)[1 + kBPerSlot][8]) {
            const int na = a + 1 < Sh::kTo ? a + 1 : 0;
            const OperandRows& asrc = a + 1 < Sh::kTo ? dyt : dyn;
            if (a + 1 < Sh::kTo || has_next) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) raw[0][jj] = asrc.rows[jj >> 2][P::Dy::step(jj, na)];
            }
#pragma unroll
            for (int q = 0; q < kBPerSlot; ++q) {
                const int b = a * kBPerSlot + q;
                if (b < Sh::kTi && has_next) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) raw[1 + q][jj] = xn.rows[jj >> 2][P::X::step(jj, b)];
                }
            }
        };
#pragma unroll
        for (int a = 0; a < Sh::kTo; ++a) {
            // the DMA of step t + 3: dY pieces in front of the first slot, X pieces in front of the second (f16 form: in
            // FRONT of the slot's LDS reads and scheduling region — inside the region the volatile asm splits it, and
            // the half behind it carries the conversions without MFMAs to hide under; bf16 form: pinned behind the
            // slot's first MFMA)
            if constexpr (kF16) {
                if (a == 0) ring_issue_part<Sh, 0>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);
                if (a == 1) ring_issue_part<Sh, 1>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);
            }
            const int na = a + 1 < Sh::kTo ? a + 1 : 0;
            float (&raw)[1 + kBPerSlot][8] = raw2[a & 1];
            const bool next_a = a + 1 < Sh::kTo || has_next;       // compile-time per code instance
            if (!kF16 || a == 0) read_raw(a, raw);
            unsigned th[1 + kBPerSlot][8], tm[1 + kBPerSlot][8], tl[1 + kBPerSlot][8];
            u32x4 ph[1 + kBPerSlot], pm[1 + kBPerSlot], pl[1 + kBPerSlot];
            const Operand& ac = at[a & 1];
            Operand& an = at[(a & 1) ^ 1];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kF16) {
                // f16 form: the slot's 3 kTi MFMAs with its conversions PINNED between them, level by level: every
                // value pair goes through four dependent levels (scale / affine, hi = pkrtz, two residuals, lo =
                // pkrtz); the micro-operations are ordered level-major over all pairs of all the slot's operands, so
                // neighbours are independent, and dealt out evenly over the MFMA gaps.  (Handed to the scheduler as
                // a "1 MFMA, 6 VALU" pipeline, two of the four slots came out as 12 MFMAs | all conversions: only 38 %
                // of the matrix pipe's busy cycles had a VALU instruction beside them — SQ_VALU_MFMA_COEXEC_CYCLES,
                // profiles/r04_p_train_f16x3_wgrad_pmc.json.  Pinned pair by pair — one dependent chain per gap —
                // a chain runs at 1.7x its issue time.)
                constexpr int kOps = 1 + kBPerSlot;
                constexpr int kMicro = kOps * 4 * 4;                  // operands x pairs x levels
                float v0[kOps][4], v1[kOps][4];
                h2 nh[kOps][4], nl[kOps][4];
                auto active = [&](int op) {
                    const int bq = a * kBPerSlot + (op - 1);
                    if (op > 0) return bq < Sh::kTi && has_next;
                    return next_a;
                };
                auto micro = [&](int idx) {
                    const int level = idx / (kOps * 4), op = (idx % (kOps * 4)) / 4, pp = idx % 4;
                    if (!active(op)) return;
                    const int bq = a * kBPerSlot + (op - 1);
                    if (level == 0) {
                        float x0 = raw[op][2 * pp], x1 = raw[op][2 * pp + 1];
                        if (op > 0) {
                            x0 = b_value(x0, bq), x1 = b_value(x1, bq);
                        } else {
                            bsum[na] += x0;               // unscaled: the scaled value then dies in its split
                            asm("" : "+v"(bsum[na]));     // (no v_pk_add_f32 with an op_sel swap: isa_scan rule R5)
                            bsum[na] += x1;
                            x0 *= a_scale, x1 *= a_scale;
                        }
                        v0[op][pp] = x0, v1[op][pp] = x1;
                    } else if (level == 1) {
                        nh[op][pp] = pack_rtz(v0[op][pp], v1[op][pp]);
                    } else if (level == 2) {
                        v0[op][pp] = residual<0>(v0[op][pp], nh[op][pp]);
                        v1[op][pp] = residual<1>(v1[op][pp], nh[op][pp]);
                    } else {
                        nl[op][pp] = pack_rtz(v0[op][pp], v1[op][pp]);
                        asm volatile("" : "+v"(nl[op][pp]));
                    }
                };
#pragma unroll
                for (int m = 0; m < kMfmas; ++m) {
                    const int b = m / kPerProduct, tt = m % kPerProduct;
                    acc[a][b] = mfma_hw(tt == 2 ? ac.l : ac.h, tt == 1 ? bcur[b].l : bcur[b].h, acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int idx = m * kMicro / kMfmas; idx < (m + 1) * kMicro / kMfmas; ++idx) micro(idx);
                    if (m == kMfmas / 2 && a + 1 < Sh::kTo) read_raw(a + 1, raw2[(a + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int op = 0; op < kOps; ++op) {
                    if (!active(op)) continue;
                    Operand r;
                    r.h = join8(nh[op][0], nh[op][1], nh[op][2], nh[op][3]);
                    r.l = join8(nl[op][0], nl[op][1], nl[op][2], nl[op][3]);
                    if (op == 0) an = r;
                    else bnext[a * kBPerSlot + (op - 1)] = r;
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
            // bf16 form: every conversion instruction pinned to its MFMA gap (left alone, its longer chains were clumped
            // in front of the MFMA batch)
#pragma unroll
            for (int m = 0; m < kMfmas; ++m) {
                const int b = m / kPerProduct, tt = m % kPerProduct;
                const bf8& ta = tt == 5 ? ac.l : (tt == 2 || tt == 3 ? ac.m : ac.h);
                const bf8& tb = tt == 4 ? bcur[b].l : (tt == 1 || tt == 3 ? bcur[b].m : bcur[b].h);
                acc[a][b] = mfma_bf(ta, tb, acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
                if (m == 0 && a == 0) ring_issue_part<Sh, 0>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);
                if (m == 0 && a == 1) ring_issue_part<Sh, 1>(dy, x, sample_begin + t_fill * kRingStep, fill, wave, lane);
#pragma unroll
                for (int it = 0; it < kItems; ++it) {
                    if (kLead + it * (kMfmas - kLead) / kItems != m) continue;
                    const int op = it / kItemsPerOp, w = it % kItemsPerOp;   // op 0: next A operand, 1..: B operands of step t + 1
                    const int bq = a * kBPerSlot + (op - 1);
                    if (op > 0 && (bq >= Sh::kTi || !has_next)) continue;
                    if (op == 0 && !next_a) continue;
                    if (w < 8) {
                        float val = raw[op][w];
                        asm volatile("" : "+v"(val));
                        if (op > 0) val = b_value(val, bq);
                        // bias gradient: every dY value of the wave's out tiles is converted exactly once
                        if (op == 0) bsum[na] += val;
                        th[op][w] = __builtin_bit_cast(unsigned, val) & 0xffff0000u;
                        const float r1 = val - __builtin_bit_cast(float, th[op][w]);
                        tm[op][w] = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                        tl[op][w] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, tm[op][w]));
                        asm volatile("" : "+v"(tl[op][w]));
                    } else {
#pragma unroll
                        for (int pp = 2 * (w - 8); pp < 2 * (w - 8) + 2; ++pp) {
                            unsigned lo_h = th[op][2 * pp], lo_m = tm[op][2 * pp], lo_l = tl[op][2 * pp];
                            asm volatile("" : "+v"(lo_h), "+v"(lo_m), "+v"(lo_l));
                            ph[op][pp] = __builtin_amdgcn_perm(th[op][2 * pp + 1], lo_h, 0x07060302u);
                            pm[op][pp] = __builtin_amdgcn_perm(tm[op][2 * pp + 1], lo_m, 0x07060302u);
                            pl[op][pp] = __builtin_amdgcn_perm(tl[op][2 * pp + 1], lo_l, 0x07060302u);
                            asm volatile("" : "+v"(ph[op][pp]), "+v"(pm[op][pp]), "+v"(pl[op][pp]));
                        }
                        if (w == 9) {
                            Bf3 r;
                            r.h = __builtin_bit_cast(bf8, ph[op]);
                            r.m = __builtin_bit_cast(bf8, pm[op]);
                            r.l = __builtin_bit_cast(bf8, pl[op]);
                            if (op == 0) an = r;
                            else bnext[bq] = r;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        // hand-over: this wave's pieces of step t + 2 have landed (the kPerWave pieces of step t + 3,
        // issued above, may still fly), its LDS reads are done; behind the barrier every wave's are
        asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)" ::"n"(P::kPerWave) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (n_steps > 0) {
        // prologue: three steps in flight, the first two landed; step 0's operands converted up front
        issue_step(0);
        issue_step(1);
        if (n_steps > 2) {
            issue_step(2);
            asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(P::kPerWave) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        Operand b0[Sh::kTi], b1[Sh::kTi];
        auto convert = [&](const float (&v)[8]) {
            if constexpr (kF16) {
                Operand r;
                split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, r.h, r.l);
                return r;
            } else {
                return split_bf3(v);
            }
        };
        {
            const OperandRows dyt = operand_rows<typename P::Dy>(slot_of(0), kk, i, out0);
            const OperandRows xt = operand_rows<typename P::X>(slot_of(0) + P::kXOffset, kk, i, in0);
#pragma unroll
            for (int b = 0; b < Sh::kTi; ++b) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = b_value(xt.rows[jj >> 2][P::X::step(jj, b)], b);
                b0[b] = convert(v);
            }
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                v[jj] = dyt.rows[jj >> 2][P::Dy::step(jj, 0)];
                bsum[0] += v[jj];
                v[jj] *= a_scale;
            }
            at[0] = convert(v);
        }
        const std::false_type more_steps;
        const std::true_type last_step;
        for (int64_t t = 0; t + 2 < n_steps; t += 2) {    // two steps per trip: the B sets swap roles
            k_step(more_steps, t, b0, b1);
            k_step(more_steps, t + 1, b1, b0);
        }
        k_step(more_steps, n_steps - 2, b0, b1);
        k_step(last_step, n_steps - 1, b1, b0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetches behind the last step
    }

    float* slab = ba.slab;
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int a = 0; a < Sh::kTo; ++a)
#pragma unroll
        for (int b = 0; b < Sh::kTi; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[w_off + (32 * (out0 + a) + row) * Sh::kSlabStride + 32 * (in0 + b) + col] = acc[a][b][r] * un_scale;
            }
#pragma unroll
    for (int a = 0; a < Sh::kTo; ++a) {
        const float both = bsum[a] + __shfl_xor(bsum[a], 32);       // the two 8-sample halves of a k-step
        if (in0 == 0 && half == 0) slab[b_off + 32 * (out0 + a) + col] = both;
    }
}


// sum of p[0], p[stride], ... (n terms) in a fixed association: sixteen interleaved partial sums,
// combined pairwise — sixteen independent loads in flight per round trip (the loop is latency-bound:
// with four, a 128-slab reduction was 32 dependent trips to HBM, 74 us for 156 MB)
__device__ __forceinline__ float strided_sum(const float* p, int n, int64_t stride) {
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p[(int64_t)(i + q) * stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += v[q];
    }
    if (i < n) {                          // the tail as ONE more round trip (a scalar loop would be n - i of them)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = i + q < n ? p[(int64_t)(i + q) * stride] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += v[q];
    }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int q = 0; q < w; ++q) acc[q] += acc[q + w];
    return acc[0];
}

// One wave sums n partials p[0], p[stride], ...: lane l takes l, l + 64, ... with four loads in flight per round
// trip, then the lanes combine in a fixed butterfly (one fixed association; every lane returns the sum).
__device__ __forceinline__ float wave_strided_sum(const float* p, int n, int64_t stride, int lane) {
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = lane; q < n; q += 256) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = q + 64 * k < n ? p[(int64_t)(q + 64 * k) * stride] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k] += v[k];
    }
    float sum = (part[0] + part[1]) + (part[2] + part[3]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    return sum;
}

}  // namespace nerf_bwd
#endif

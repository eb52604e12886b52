// Backward of the fused renderer w.r.t. the 22 parameter tensors (gfx950 only).  Replaces PyTorch
// autograd through NeRF.render_rays (loss.backward(), train_conditional_nerf.py:133).  Four
// launches, no float atomics, bitwise reproducible:
//   1. nerf_composite_bwd_kernel — per ray, last chunk first: compositing backward
//      (model.py:438-469, :660-663) -> dL/d(out) of every sample.
//   2. nerf_bwd_data_kernel / nerf_bwd_data_h_kernel — per 16-sample chunk: the data-gradient chain
//      dX = W^T dY with the transposed weight image streamed through LDS exactly like the forward,
//      LayerNorm/ReLU backward in registers from the saved x_hat / 1/std.  fp32 MFMA, or (the
//      training forward's precision = F16X3) f16 pairs with an exact per-sample power-of-two scale
//      of dY.  Writes dY of every layer (row order) for the weight gradients; gamma/beta gradients
//      are row-reduced by DPP and summed per workgroup in LDS; the f16 form also records the
//      largest |dY| per layer for kernel 3.
//   3. nerf_wgrad_kernel / nerf_wgrad_h_kernel — dW_L = dY_L^T X_L as a split-K MFMA GEMM over the
//      padded samples: a continuous stream of 16-sample k-steps through a 4-slot LDS ring
//      ([sample][feature] fp32 tiles moved by LDS-DMA), operands split in registers into bf16 triples
//      (24 significand bits, fp32 exponent range; six v_mfma_f32_32x32x16_bf16 per product) or, in
//      the split-precision mode, f16 pairs under one batch-wide power-of-two scale per layer (three
//      v_mfma_f32_32x32x16_f16); fp32 accumulation (the exact-fp32 32x32x2 form of round 1 is in the history); each workgroup writes a partial slab (and the bias partial = column
//      sums of dY).
//   4. nerf_grad_reduce_kernel — sums the slabs in a fixed order into the flat gradient vector
//      (state_dict order, PyTorch layouts; undoes the layer-0 column permutation).
#include "nerf_backward_common.h"

using namespace nerf_layout;
using namespace nerf_device;
using namespace nerf_bwd;

namespace {

constexpr int kGbFloats = 5 * 2 * kHidden;                         // gamma/beta partials per workgroup
constexpr int kBwdLdsBytes = kRingBytes + kSmallLdsBytes + kGbFloats * 4;   // 74.25 KiB

// partial-slab layout (floats) for one split
constexpr int kSlabW0 = 0;                                          // [256][96] kernel column order
constexpr int kSlabWh = kSlabW0 + kHidden * kEncIn;                 // 4 x [256][256]
constexpr int kSlabW5 = kSlabWh + 4 * kHidden * kHidden;            // [64][256]
constexpr int kSlabB = kSlabW5 + kOutPad * kHidden;                 // 5 x [256] + [64]
constexpr int kSlabFloats = kSlabB + 5 * kHidden + kOutPad;

struct BwdArgs {
    NerfHipRenderArgs a;
    const float* d_rgb;
    const float* d_seg;
    const float* d_raw;         // [n_rays][P][num_outputs] or null: the loss reached NeRF.forward's outputs directly
    int32_t intervals, chunks;
    int64_t groups;
    TrainLayout L;
    float* gb_partial;          // [grid][5][2][256]
    float* dymax;               // [grid][8]: largest |dY| each data-gradient workgroup saw, per layer
                                // (0..4: dy[L], 5: dL/d(out)); split-precision path only
    float* slabs;               // [splits][kSlabFloats]
    float* grad;
    int32_t splits, data_grid;
    int64_t tiles_per_split, n_tiles;
    float inv_n;                // 1 / hidden_size (the LayerNorm backward's means: nerf_layout.h, Shape)
};

typedef WeightPipe<kBwdStages> BwdPipe;

// Compositing backward (nerf_backward_common.h): one wave per padded ray slot.
__global__ __launch_bounds__(256) void nerf_composite_bwd_kernel(const BwdArgs ba) {
    CompositeBwd cb;
    cb.d_rgb = ba.d_rgb, cb.d_seg = ba.d_seg;
    cb.intervals = ba.intervals, cb.chunks = ba.chunks;
    cb.mp = ba.L.mp, cb.out = ba.L.out, cb.comp = ba.L.comp, cb.dy5_rows = ba.a.train_workspace + ba.L.dy5;
    composite_bwd_body(ba.a, cb);
}

// dL/d(out) rows from dL/d(out_raw) (the backward of a differentiable NeRF.forward, nerf/model.py:553-594, takes
// the compositing backward's place): thread = (padded sample, four columns); padding rows / columns get zeros.
__global__ __launch_bounds__(256) void nerf_field_scatter_kernel(const BwdArgs ba) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ba.L.mp * (kOutPad / 4)) return;
    const int q = (int)(e & (kOutPad / 4 - 1));
    const int64_t sp = e >> 4;
    const int64_t tile = sp >> 4;
    const int64_t slot = tile / ba.chunks;
    const int s = (int)(tile - slot * ba.chunks) * 16 + (int)(sp & 15);
    const Shape sh = shape_of(ba.a);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (slot < ba.a.n_rays && s < ba.intervals) {
        const float* src = ba.d_raw + (slot * ba.intervals + s) * sh.n_out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = row_of_slot(4 * q + k, sh);        // slot of the padded tile -> column of d_raw
            if (row >= 0) v[k] = src[row];
        }
    }
    *(f32x4*)(ba.a.train_workspace + ba.L.dy5 + sp * kOutPad + 4 * q) = v;
}

__global__ __launch_bounds__(256, 2) void nerf_bwd_data_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gb = (float*)(smem + kRingBytes + kSmallLdsBytes);

    {
        stage_small_image(a.packed + kBlobFloats, (float*)(smem + kRingBytes));
        for (int i = threadIdx.x; i < kGbFloats; i += 256) gb[i] = 0.f;
    }
    const float* small = (const float*)(smem + kRingBytes);

    BwdPipe pipe;
    pipe.init(a.packed + kBwdBlobOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;

    // one (padded ray, chunk) item per wave: dL/d(out) of every sample was written by
    // nerf_composite_bwd_kernel (the suffix sum along the ray lives there), so the chunks of a ray
    // are independent here and a small batch still fills the chip
    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;      // = slot * chunks + c   (uniform)
        {
            // Addresses: the wave's 16-sample tile is wave-UNIFORM, so every saved row is (uniform 64-bit base of the tile,
            // in scalar registers) + (this lane's constant 32-bit offset, taken where it is used: nerf_device.h:
            // row_lane_offset) — as in the split-precision kernels below; per-lane 64-bit pointers to ten saved
            // tensors were what this kernel parked in scratch across its layers
            float* const ws_rows = ws + tile * kTileFloats;             // + L.xhat[l] / L.dy[l]: this tile's rows
            const float* const ws_stat = ws + tile * 16;                // + L.rstd[l]: this tile's 16 scalars
            f32x4 dout[4];
            {
                const float* drow = ws + ba.L.dy5 + tile * (16 * kOutPad) + lane_offset((uint32_t)(j * kOutPad + 4 * g));
#pragma unroll
                for (int T = 0; T < 4; ++T) dout[T] = *(const f32x4*)(drow + T * 16);
            }
            // ---- layer 5: dX = W5^T dOut (4 stages of the transposed image) ----
#pragma unroll
            for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 xh[16];
            float rstd;
            layer_wide_v4<kStagesL5>(pipe, acc, dout,
                                     BwdHookU{turn, ws_rows + ba.L.xhat[4], ws_stat + ba.L.rstd[4], xh, rstd});
            // ---- layers 4..1: LayerNorm/ReLU backward, then dX = W^T dY ----
#pragma unroll 1
            for (int L = 4; L >= 1; --L) {
                layer_norm_relu_bwd<false, 16, true>(small + L * kSmallPerLayerLds, g, j, acc, act, xh, rstd,
                                                     ws_rows + ba.L.dy[L], gb + L * 2 * kHidden, turn, ba.inv_n);
#pragma unroll
                for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide<kStagesHidden>(pipe, acc, act,
                                          BwdHookU{turn, ws_rows + ba.L.xhat[L - 1], ws_stat + ba.L.rstd[L - 1], xh, rstd});
            }
            layer_norm_relu_bwd<false, 16, true>(small, g, j, acc, act, xh, rstd, ws_rows + ba.L.dy[0], gb, turn, ba.inv_n);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = 0; t < kWavesPerWg; ++t) {       // the last chunk's layer-0 partials, in wave order
        __syncthreads();
        turn(t);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kGbFloats + i] = gb[i];
}

// The same chain for a network that trains at 8 register tiles per sample (hidden_size <= 128, fp32 arithmetic here;
// nerf_device.h: train_tiles): saved rows 128 wide, the transposed narrow image (nerf_layout.h: kNarrowBwd8Offset:
// 2 stages for layer 5, 4 per hidden layer).  The wave-ordered gamma / beta adds of a layer ride on the four stage
// barriers of the hidden loop that follows it; layer 0's have no such loop behind them (the next item opens with
// the two-stage layer-5 loop), so they take their turns at the end of the item, a barrier apart.
// NT = 4 (hidden_size <= 64, round 6: nerf_device.h: train_compute_tiles): the chain COMPUTES at 4 register tiles on
// the image at kNarrowBwd4Offset (one stage per layer), the rows it reads and writes stay 128 wide — it loads tiles
// 0 .. 3 of x_hat and writes tiles 0 .. 3 of dY (nerf_wgrad_n4_kernel fetches no others).  A one-stage loop has one
// barrier, so EVERY layer's gamma / beta partials take their four turns behind explicit barriers, and the next
// LayerNorm backward's x_hat is fetched at stage 0.
template <int NT>
struct NarrowBwdImage {
    static constexpr int kStages = NT == 8 ? kNarrowBwd8Stages : kNarrowBwd4Stages;
    static constexpr int kOffset = NT == 8 ? kNarrowBwd8Offset : kNarrowBwd4Offset;
};
// (the hook of the 4-tile chain: loads at stage 0 — there is no stage 1 — and no turns: they are taken explicitly)
struct BwdHookLoad4 {
    const float* xhat_row;
    const float* rstd_ptr;
    f32x4 (&xh)[16];
    float& rstd;
    __device__ __forceinline__ void operator()(int t) const {
        if (t == 0) {
#pragma unroll
            for (int T = 0; T < 4; ++T) xh[T] = *(const f32x4*)(xhat_row + T * kTileT);
            rstd = *rstd_ptr;
        }
    }
};

template <int NT>
__global__ __launch_bounds__(256, 2) void nerf_bwd_data_n_kernel(const BwdArgs ba) {
    static_assert(NT == 8 || NT == 4, "narrow data gradient: 8 or 4 register tiles");
    constexpr int kTileN = 256 * 8;               // floats of a saved 16-sample tile: the rows are 128 wide at either NT
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gb = (float*)(smem + kRingBytes + kSmallLdsBytes);

    {
        stage_small_image(a.packed + kBlobFloats, (float*)(smem + kRingBytes));
        for (int i = threadIdx.x; i < kGbFloats; i += 256) gb[i] = 0.f;
    }
    const float* small = (const float*)(smem + kRingBytes);

    WeightPipe<NarrowBwdImage<NT>::kStages> pipe;
    pipe.init(a.packed + NarrowBwdImage<NT>::kOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;
    auto take_turns = [&]() {                     // the four waves add their partials in wave order, a barrier apart
        for (int t = 0; t < kWavesPerWg; ++t) {
            __syncthreads();
            turn(t);
        }
        turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // (NT = 4: tiles 4 .. 7 of the 128-wide saved rows are never written — the weight gradient of such a network fetches
    //  tiles 0 .. 3 only, nerf_wgrad_n4_kernel)

    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;      // = slot * chunks + c
        const int64_t sp = tile * 16 + j;
        {
            const float* drow = ws + ba.L.dy5 + sp * kOutPad + 4 * g;
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const f32x4 d = *(const f32x4*)(drow + T * 16);
                act[4 * T] = d.x, act[4 * T + 1] = d.y, act[4 * T + 2] = d.z, act[4 * T + 3] = d.w;
            }
        }
        // ---- layer 5: dX = W5^T dOut (4 k-groups of padded outputs x NT in tiles) ----
#pragma unroll
        for (int T = 0; T < NT; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xh[16];
        float rstd;
        if constexpr (NT == 8)
            layer_wide_n<NT, 4>(pipe, acc, act,
                                BwdHookN<NT>{turn, ws + ba.L.xhat[4] + tile_lane_base(sp, g, kTileN), ws + ba.L.rstd[4] + sp, xh, rstd});
        else
            layer_wide_n<NT, 4>(pipe, acc, act,
                                BwdHookLoad4{ws + ba.L.xhat[4] + tile_lane_base(sp, g, kTileN), ws + ba.L.rstd[4] + sp, xh, rstd});
        // ---- layers 4..1: LayerNorm/ReLU backward, then dX = W^T dY (NT k-groups x NT in tiles) ----
#pragma unroll 1
        for (int L = 4; L >= 1; --L) {
            float* const dyrow = ws + ba.L.dy[L] + tile_lane_base(sp, g, kTileN);
            layer_norm_relu_bwd<false, NT>(small + L * kSmallPerLayerLds, g, j, acc, act, xh, rstd, dyrow, gb + L * 2 * kHidden, turn,
                                           ba.inv_n);
            if (NT == 4) take_turns();
#pragma unroll
            for (int T = 0; T < NT; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (NT == 8)
                layer_wide_n<NT, NT>(pipe, acc, act,
                                     BwdHookN<NT>{turn, ws + ba.L.xhat[L - 1] + tile_lane_base(sp, g, kTileN), ws + ba.L.rstd[L - 1] + sp,
                                                  xh, rstd});
            else
                layer_wide_n<NT, NT>(pipe, acc, act,
                                     BwdHookLoad4{ws + ba.L.xhat[L - 1] + tile_lane_base(sp, g, kTileN), ws + ba.L.rstd[L - 1] + sp,
                                                  xh, rstd});
        }
        {
            float* const dyrow = ws + ba.L.dy[0] + tile_lane_base(sp, g, kTileN);
            layer_norm_relu_bwd<false, NT>(small, g, j, acc, act, xh, rstd, dyrow, gb, turn, ba.inv_n);
        }
        take_turns();                             // layer 0's partials (NT = 8: the only ones without a loop behind them)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < kGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kGbFloats + i] = gb[i];
}

// vector-memory operations issued between the DMA of a loop's stage 1 and its first hand-overs: the 17
// x_hat / 1/std loads (layer 5's loop), or the 16 dY saves of the LayerNorm backward + those 17 loads
constexpr int kYoungerL5 = 17, kYoungerHidden = 33;
// The same chain in split-precision arithmetic (see row_scale above).
__global__ __launch_bounds__(256, 2) void nerf_bwd_data_h_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gb = (float*)(smem + kRingBytes + kSmallLdsBytes);
    int* const wmax = (int*)(smem + kBwdLdsBytes);

    {
        stage_small_image(a.packed + kBlobFloats, (float*)(smem + kRingBytes));
        for (int i = threadIdx.x; i < kGbFloats; i += 256) gb[i] = 0.f;
        if (threadIdx.x < 8) wmax[threadIdx.x] = 0;
    }
    const float* small = (const float*)(smem + kRingBytes);

    BwdPipe pipe;
    pipe.init(a.packed + kBwdHBlobOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;

    // one (padded ray, chunk) item per wave: dL/d(out) of every sample was written by
    // nerf_composite_bwd_kernel (the suffix sum along the ray lives there), so the chunks of a ray
    // are independent here and a small batch still fills the chip
    // Addresses: a wave's 16-sample tile is wave-UNIFORM, so every saved row is (uniform 64-bit base of the tile, in
    // SGPRs) + (this lane's constant 32-bit offset inside a tile) — no per-lane 64-bit pointer stays live across the
    // loops.  That is not only two or three registers per pointer: a spilled pointer that the compiler reloads BEHIND
    // a layer's burst of 16 saves + 16 loads gets a `s_waitcnt vmcnt(0)`, and vmcnt retires in order — the reload of
    // an L1-resident scratch line then waits for the whole burst's HBM round trip (round 4: one such reload, for the
    // 1/std address, cost this kernel a fifth of its time; scripts/vmcnt_drains.py finds them in the assembly).
    // (lane_word: the same value behind an optimisation barrier, taken at every use — otherwise loop-invariant code
    //  motion folds the offset into a per-lane 64-bit pointer again, and that pointer is what gets spilled)
    auto lane_word = [](uint32_t v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    const uint32_t row_off = (uint32_t)tile_lane_word(j, g);           // floats, in a tile-major [16][256] tile
    const uint32_t out_off = (uint32_t)(j * kOutPad + 4 * g);          // floats, in a [16][64] tile
    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;      // = slot * chunks + c   (uniform)
        {
            const float* const ws_rows = ws + tile * (16 * kHidden);     // + L.xhat[l] / L.dy[l]: this tile's rows
            const float* const ws_stat = ws + tile * 16;                 // + L.rstd[l]: this tile's 16 scalars
            f32x4 dout[4];
            {
                const float* drow = ws + ba.L.dy5 + tile * (16 * kOutPad);
                const uint32_t oo = lane_word(out_off);
#pragma unroll
                for (int T = 0; T < 4; ++T) dout[T] = *(const f32x4*)(drow + oo + T * 16);
            }

            f32x4 xh[16];
            float rstd;
            // x_hat / 1/std of layer 4 first: 17 loads that fly under the 4 stages of layer 5
            {
                const float* xrow = ws_rows + ba.L.xhat[4];
                rstd = (ws_stat + ba.L.rstd[4])[lane_word(j)];
                const uint32_t ro = lane_word(row_off);
#pragma unroll
                for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xrow + ro + T * kTileT);
            }
            float unscale;
            {
                float m = 0.f;
#pragma unroll
                for (int T = 0; T < 4; ++T) m = abs_max4(m, dout[T]);
                float amax;
                const float sc = row_scale(m, unscale, amax);
                note_max(wmax + 5, amax, lane);
                h8 bh[2], bl[2];
                split8(dout[0] * sc, dout[1] * sc, bh[0], bl[0]);
                split8(dout[2] * sc, dout[3] * sc, bh[1], bl[1]);
#pragma unroll
                for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide_h<2, kYoungerL5>(pipe, acc, bh, bl, TurnHook{turn});
            }
#pragma unroll 1
            for (int L = 4; L >= 0; --L) {
                layer_norm_relu_bwd<true>(small + L * kSmallPerLayerLds, g, j, acc, act, xh, rstd,
                                          const_cast<float*>(ws_rows) + ba.L.dy[L] + lane_word(row_off), gb + L * 2 * kHidden,
                                          turn, ba.inv_n, unscale);
                if (L == 0) {
                    // dy[0] feeds only the weight gradient, whose f16 pairs need ONE scale for the batch: a BOUND on
                    // this sample's largest |dy_0| from the two scalars at hand, folded into the workgroup's maximum.
                    // (The true maximum would take a pass over the 64 registers, and every place such a pass can go
                    // makes the allocator spill 270-700 B inside the loops: 0.73 -> 1.05-1.29 ms.)
                    //   |dz| <= |acc| unscale <= 2^21 C unscale   (B operands < 2^13, weights x 2^8, C = max column
                    //                                               sum of |W_1|: the accumulator cannot exceed it)
                    //   |dy| <= (|g dz| + |m1| + |x_hat| |m2|) / std <= 18 max|gamma| max|dz| / std
                    //                                              (|m1|, |m2| <= max|g dz|; |x_hat| < 16)
                    // K0 = 18 2^21 max|gamma_0| C with 1 % for rounding comes from the pack kernel.  The bound
                    // overestimates by ~2^8: the batch's largest |dy_0| enters the weight gradient near 2^4 instead
                    // of 2^12 — no overflow possible, elements down to 2^-7 of it keep all 22 bits, smaller ones
                    // lose low bits that are below 2^-29 of the largest element.
                    note_max(wmax + 0, rstd * unscale * a.packed[kBoundsOffset], lane);
                    break;
                }
                // the next LayerNorm backward's saved tile: 17 loads behind the 16 saves above (1/std first: its
                // address is the one thing here that is not a row offset), all of them younger than the two stages
                // this layer's loop opens first
                {
                    const float* xrow_n = ws_rows + ba.L.xhat[L - 1];
                    rstd = (ws_stat + ba.L.rstd[L - 1])[lane_word(j)];
                    const uint32_t ro = lane_word(row_off);
#pragma unroll
                    for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xrow_n + ro + T * kTileT);
                }
                // the sample's largest |dy|: this layer's B-operand scale, and (folded into the
                // workgroup's maximum) the weight-gradient kernel's
                float m = 0.f;
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    m = abs_max4(m, f32x4{act[4 * T], act[4 * T + 1], act[4 * T + 2], act[4 * T + 3]});
                float amax;
                const float sc = row_scale(m, unscale, amax);
                note_max(wmax + L, amax, lane);
                h8 bh[8], bl[8];
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    const int t0 = 8 * mb, t1 = 8 * mb + 4;
                    split8(f32x4{act[t0], act[t0 + 1], act[t0 + 2], act[t0 + 3]} * sc,
                           f32x4{act[t1], act[t1 + 1], act[t1 + 2], act[t1 + 3]} * sc, bh[mb], bl[mb]);
                }
#pragma unroll
                for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide_h<8, kYoungerHidden>(pipe, acc, bh, bl, TurnHook{turn});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int t = 0; t < kWavesPerWg; ++t) {       // the last chunk's layer-0 partials, in wave order
        __syncthreads();
        turn(t);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kGbFloats + i] = gb[i];
    if (threadIdx.x < 8) ba.dymax[(int64_t)blockIdx.x * 8 + threadIdx.x] = __builtin_bit_cast(float, wmax[threadIdx.x]);
}

// The split-precision chain for a network that trains at 8 register tiles per sample (hidden_size <= 128): saved rows
// 128 wide (a wave's 16-sample tile = 8 KiB), the transposed f16-pair image of nerf_layout.h: kNarrowBwdH8Offset
// (2 stages for layer 5, 4 per hidden layer), 8 x_hat loads + 1 and 8 dY saves per layer in the hand-overs' counts.
// Layer 0's gamma / beta partials take their turns at the end of the item (the two-stage layer-5 loop that opens the
// next item has only two barriers), as in nerf_bwd_data_n_kernel<8>.
constexpr int kYoungerL5N8 = 9, kYoungerHiddenN8 = 17;
__global__ __launch_bounds__(256, 2) void nerf_bwd_data_h_n8_kernel(const BwdArgs ba) {
    constexpr int NT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gb = (float*)(smem + kRingBytes + kSmallLdsBytes);
    int* const wmax = (int*)(smem + kBwdLdsBytes);

    {
        stage_small_image(a.packed + kBlobFloats, (float*)(smem + kRingBytes));
        for (int i = threadIdx.x; i < kGbFloats; i += 256) gb[i] = 0.f;
        if (threadIdx.x < 8) wmax[threadIdx.x] = 0;
    }
    const float* small = (const float*)(smem + kRingBytes);

    WeightPipe<kNarrowBwdH8Stages> pipe;
    pipe.init(a.packed + kNarrowBwdH8Offset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;

    // one (padded ray, chunk) item per wave: dL/d(out) of every sample was written by
    // nerf_composite_bwd_kernel (the suffix sum along the ray lives there), so the chunks of a ray
    // are independent here and a small batch still fills the chip
    // Addresses: a wave's 16-sample tile is wave-UNIFORM, so every saved row is (uniform 64-bit base of the tile, in
    // SGPRs) + (this lane's constant 32-bit offset inside a tile) — no per-lane 64-bit pointer stays live across the
    // loops.  That is not only two or three registers per pointer: a spilled pointer that the compiler reloads BEHIND
    // a layer's burst of 16 saves + 16 loads gets a `s_waitcnt vmcnt(0)`, and vmcnt retires in order — the reload of
    // an L1-resident scratch line then waits for the whole burst's HBM round trip (round 4: one such reload, for the
    // 1/std address, cost this kernel a fifth of its time; scripts/vmcnt_drains.py finds them in the assembly).
    // (lane_word: the same value behind an optimisation barrier, taken at every use — otherwise loop-invariant code
    //  motion folds the offset into a per-lane 64-bit pointer again, and that pointer is what gets spilled)
    auto lane_word = [](uint32_t v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    const uint32_t row_off = (uint32_t)tile_lane_word(j, g);           // floats, in a tile-major [16][256] tile
    const uint32_t out_off = (uint32_t)(j * kOutPad + 4 * g);          // floats, in a [16][64] tile
    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;      // = slot * chunks + c   (uniform)
        {
            const float* const ws_rows = ws + tile * (16 * 16 * NT);     // + L.xhat[l] / L.dy[l]: this tile's rows
            const float* const ws_stat = ws + tile * 16;                 // + L.rstd[l]: this tile's 16 scalars
            f32x4 dout[4];
            {
                const float* drow = ws + ba.L.dy5 + tile * (16 * kOutPad);
                const uint32_t oo = lane_word(out_off);
#pragma unroll
                for (int T = 0; T < 4; ++T) dout[T] = *(const f32x4*)(drow + oo + T * 16);
            }

            f32x4 xh[16];
            float rstd;
            // x_hat / 1/std of layer 4 first: NT + 1 = 9 loads that fly under the 2 stages of layer 5 (kYoungerL5N8)
            {
                const float* xrow = ws_rows + ba.L.xhat[4];
                rstd = (ws_stat + ba.L.rstd[4])[lane_word(j)];
                const uint32_t ro = lane_word(row_off);
#pragma unroll
                for (int T = 0; T < NT; ++T) xh[T] = *(const f32x4*)(xrow + ro + T * kTileT);
            }
            float unscale;
            {
                float m = 0.f;
#pragma unroll
                for (int T = 0; T < 4; ++T) m = abs_max4(m, dout[T]);
                float amax;
                const float sc = row_scale(m, unscale, amax);
                note_max(wmax + 5, amax, lane);
                h8 bh[2], bl[2];
                split8(dout[0] * sc, dout[1] * sc, bh[0], bl[0]);
                split8(dout[2] * sc, dout[3] * sc, bh[1], bl[1]);
#pragma unroll
                for (int T = 0; T < NT; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide_h<2, kYoungerL5N8, NT>(pipe, acc, bh, bl, TurnHook{turn});
            }
#pragma unroll 1
            for (int L = 4; L >= 0; --L) {
                layer_norm_relu_bwd<true, NT>(small + L * kSmallPerLayerLds, g, j, acc, act, xh, rstd,
                                          const_cast<float*>(ws_rows) + ba.L.dy[L] + lane_word(row_off), gb + L * 2 * kHidden,
                                          turn, ba.inv_n, unscale);
                if (L == 0) {
                    // dy[0] feeds only the weight gradient, whose f16 pairs need ONE scale for the batch: a BOUND on
                    // this sample's largest |dy_0| from the two scalars at hand, folded into the workgroup's maximum.
                    // (The true maximum would take a pass over the 64 registers, and every place such a pass can go
                    // makes the allocator spill 270-700 B inside the loops: 0.73 -> 1.05-1.29 ms.)
                    //   |dz| <= |acc| unscale <= 2^21 C unscale   (B operands < 2^13, weights x 2^8, C = max column
                    //                                               sum of |W_1|: the accumulator cannot exceed it)
                    //   |dy| <= (|g dz| + |m1| + |x_hat| |m2|) / std <= 18 max|gamma| max|dz| / std
                    //                                              (|m1|, |m2| <= max|g dz|; |x_hat| < 16)
                    // K0 = 18 2^21 max|gamma_0| C with 1 % for rounding comes from the pack kernel.  The bound
                    // overestimates by ~2^8: the batch's largest |dy_0| enters the weight gradient near 2^4 instead
                    // of 2^12 — no overflow possible, elements down to 2^-7 of it keep all 22 bits, smaller ones
                    // lose low bits that are below 2^-29 of the largest element.
                    note_max(wmax + 0, rstd * unscale * a.packed[kBoundsOffset], lane);
                    break;
                }
                // the next LayerNorm backward's saved tile: NT + 1 = 9 loads behind the NT = 8 saves above (kYoungerHiddenN8 = 17; 1/std first: its
                // address is the one thing here that is not a row offset), all of them younger than the two stages
                // this layer's loop opens first
                {
                    const float* xrow_n = ws_rows + ba.L.xhat[L - 1];
                    rstd = (ws_stat + ba.L.rstd[L - 1])[lane_word(j)];
                    const uint32_t ro = lane_word(row_off);
#pragma unroll
                    for (int T = 0; T < NT; ++T) xh[T] = *(const f32x4*)(xrow_n + ro + T * kTileT);
                }
                // the sample's largest |dy|: this layer's B-operand scale, and (folded into the
                // workgroup's maximum) the weight-gradient kernel's
                float m = 0.f;
#pragma unroll
                for (int T = 0; T < NT; ++T)
                    m = abs_max4(m, f32x4{act[4 * T], act[4 * T + 1], act[4 * T + 2], act[4 * T + 3]});
                float amax;
                const float sc = row_scale(m, unscale, amax);
                note_max(wmax + L, amax, lane);
                h8 bh[NT / 2], bl[NT / 2];
#pragma unroll
                for (int mb = 0; mb < NT / 2; ++mb) {
                    const int t0 = 8 * mb, t1 = 8 * mb + 4;
                    split8(f32x4{act[t0], act[t0 + 1], act[t0 + 2], act[t0 + 3]} * sc,
                           f32x4{act[t1], act[t1 + 1], act[t1 + 2], act[t1 + 3]} * sc, bh[mb], bl[mb]);
                }
#pragma unroll
                for (int T = 0; T < NT; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide_h<NT / 2, kYoungerHiddenN8, NT>(pipe, acc, bh, bl, TurnHook{turn});
            }
        }
        for (int t = 0; t < kWavesPerWg; ++t) {   // layer 0's partials, in wave order
            __syncthreads();
            turn(t);
        }
        turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < kGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kGbFloats + i] = gb[i];
    if (threadIdx.x < 8) ba.dymax[(int64_t)blockIdx.x * 8 + threadIdx.x] = __builtin_bit_cast(float, wmax[threadIdx.x]);
}

// All six layers in ONE launch: job = blockIdx.x, heavy (hidden) layers first so that the short
// layer-0 / layer-5 jobs fill the tail instead of running half-empty launches of their own.
__global__ __launch_bounds__(256, 1) void nerf_wgrad_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kSlabFloats, ba.dymax,
                      ba.data_grid, 8};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kBlobFloats;       // [layer][bias | gamma | beta][256]
    if (job < 4) {                                // layers 1..4: input = LayerNorm+ReLU of layer job
        wgrad_body_ring<ShapeHid, kInputAffineRelu>(jb, smem, ws + ba.L.dy[1] + (int64_t)job * ba.L.mp * kHidden,
                                   ws + ba.L.xhat[0] + (int64_t)job * ba.L.mp * kHidden,
                                   small + job * kSmallPerLayer, kSlabWh + job * kHidden * kHidden,
                                   kSlabB + (job + 1) * kHidden);
    } else if (job == 4) {                        // layer 0: input = encoded features
        wgrad_body_ring<ShapeL0, kInputRaw>(jb, smem, ws + ba.L.dy[0], ws + ba.L.h, nullptr, kSlabW0, kSlabB);
    } else {                                      // layer 5: input = LayerNorm+ReLU of layer 4
        wgrad_body_ring<ShapeL5, kInputAffineRelu>(jb, smem, ws + ba.L.dy5, ws + ba.L.xhat[4], small + 4 * kSmallPerLayer,
                                  kSlabW5, kSlabB + 5 * kHidden);
    }
}

// ... and for a network that trains at 8 register tiles per sample (saved rows 128 wide; shapes: nerf_backward_common.h)
__global__ __launch_bounds__(256, 1) void nerf_wgrad_n8_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kSlabFloats, ba.dymax,
                      ba.data_grid, 8};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kBlobFloats;
    if (job < 4) {
        wgrad_body_ring<ShapeHidN8, kInputAffineRelu>(jb, smem, ws + ba.L.dy[1] + (int64_t)job * ba.L.mp * 128,
                                                      ws + ba.L.xhat[0] + (int64_t)job * ba.L.mp * 128,
                                                      small + job * kSmallPerLayer, kSlabWh + job * kHidden * kHidden,
                                                      kSlabB + (job + 1) * kHidden);
    } else if (job == 4) {
        wgrad_body_ring<ShapeL0N8, kInputRaw>(jb, smem, ws + ba.L.dy[0], ws + ba.L.h, nullptr, kSlabW0, kSlabB);
    } else {
        wgrad_body_ring<ShapeL5N8, kInputAffineRelu>(jb, smem, ws + ba.L.dy5, ws + ba.L.xhat[4], small + 4 * kSmallPerLayer,
                                                     kSlabW5, kSlabB + 5 * kHidden);
    }
}

// ... and at 4 (hidden_size <= 64, fp32 arithmetic: nerf_device.h: train_compute_tiles).  The forward and the data gradient
// wrote register tiles 0..3 of the 128-wide rows only, so only those are fetched (WgradShape: HALF).  A hidden layer is
// ONE 2 x 2 block of 32 x 32 accumulator tiles — a wave's worth — so the four hidden layers are one job in which wave w
// takes layer w + 1 on the same samples (kMapPrivate: its own operands in its own quarter of the ring slot), and
// layers 0 and 5 are a second job of the same kind — even waves layer 0, odd waves layer 5 (kMapPrivatePair) —: two jobs
// of S k-steps instead of six.
constexpr int kWgradJobsN4 = 2;
__global__ __launch_bounds__(256, 1) void nerf_wgrad_n4_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kSlabFloats, ba.dymax,
                      ba.data_grid, 8};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kBlobFloats;
    const int layer = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (job == 0) {                                                                // hidden layer layer + 1: this wave's
        wgrad_body_ring<ShapeHidN4, kInputAffineRelu>(jb, smem, ws + ba.L.dy[1] + (int64_t)layer * ba.L.mp * 128,
                                                      ws + ba.L.xhat[0] + (int64_t)layer * ba.L.mp * 128,
                                                      small + layer * kSmallPerLayer, kSlabWh + layer * kHidden * kHidden,
                                                      kSlabB + (layer + 1) * kHidden);
    } else if ((layer & 1) == 0) {
        wgrad_body_ring<ShapeL0N4, kInputRaw>(jb, smem, ws + ba.L.dy[0], ws + ba.L.h, nullptr, kSlabW0, kSlabB);
    } else {
        wgrad_body_ring<ShapeL5N4, kInputAffineRelu>(jb, smem, ws + ba.L.dy5, ws + ba.L.xhat[4], small + 4 * kSmallPerLayer,
                                                     kSlabW5, kSlabB + 5 * kHidden);
    }
}

// The same launch in the split-precision training mode (f16-pair operands, see wgrad_body_ring).
__global__ __launch_bounds__(256, 1) void nerf_wgrad_h_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kSlabFloats, ba.dymax,
                      ba.data_grid, 8};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kBlobFloats;
    if (job < 4) {
        wgrad_body_ring<ShapeHid, kInputAffineRelu, true>(jb, smem, ws + ba.L.dy[1] + (int64_t)job * ba.L.mp * kHidden,
                                              ws + ba.L.xhat[0] + (int64_t)job * ba.L.mp * kHidden,
                                              small + job * kSmallPerLayer, kSlabWh + job * kHidden * kHidden,
                                              kSlabB + (job + 1) * kHidden, job + 1);
    } else if (job == 4) {
        // layer 0: the scale comes from the data gradient's BOUND on |dy_0| (see there), not from a maximum
        wgrad_body_ring<ShapeL0, kInputRaw, true>(jb, smem, ws + ba.L.dy[0], ws + ba.L.h, nullptr, kSlabW0, kSlabB, 0);
    } else {
        wgrad_body_ring<ShapeL5, kInputAffineRelu, true>(jb, smem, ws + ba.L.dy5, ws + ba.L.xhat[4], small + 4 * kSmallPerLayer,
                                             kSlabW5, kSlabB + 5 * kHidden, 5);
    }
}

// ... at 8 register tiles per sample (f16-pair operands)
__global__ __launch_bounds__(256, 1) void nerf_wgrad_h_n8_kernel(const BwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kSlabFloats, ba.dymax,
                      ba.data_grid, 8};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kBlobFloats;
    if (job < 4) {
        wgrad_body_ring<ShapeHidN8, kInputAffineRelu, true>(jb, smem, ws + ba.L.dy[1] + (int64_t)job * ba.L.mp * 128,
                                                            ws + ba.L.xhat[0] + (int64_t)job * ba.L.mp * 128,
                                                            small + job * kSmallPerLayer, kSlabWh + job * kHidden * kHidden,
                                                            kSlabB + (job + 1) * kHidden, job + 1);
    } else if (job == 4) {
        wgrad_body_ring<ShapeL0N8, kInputRaw, true>(jb, smem, ws + ba.L.dy[0], ws + ba.L.h, nullptr, kSlabW0, kSlabB, 0);
    } else {
        wgrad_body_ring<ShapeL5N8, kInputAffineRelu, true>(jb, smem, ws + ba.L.dy5, ws + ba.L.xhat[4], small + 4 * kSmallPerLayer,
                                                           kSlabW5, kSlabB + 5 * kHidden, 5);
    }
}

// ---------------------------------------------------------------------------------------------
// deterministic reduction of the partials into the flat gradient vector
// ---------------------------------------------------------------------------------------------
constexpr int kReduceThreads = 256;
constexpr int kGbElements = 5 * 2 * kHidden;                       // gamma / beta gradients
constexpr int kReduceGbBlocks = kGbElements / 4;                   // one wave per element

__device__ __forceinline__ void locate(int e, const Shape& sh, int& tensor, int& idx) {
    tensor = 0;
    int off = 0;
    for (;;) {
        const int n = tensor_elements(tensor, sh);
        if (e < off + n) break;
        off += n;
        ++tensor;
    }
    idx = e - off;
}

// Blocks [0, kReduceDirectBlocks): one thread per gradient element that sums the weight-gradient
// slabs (up to 128 terms).  Blocks behind them: the gamma / beta gradients, whose partials come one
// per data-gradient WORKGROUP (up to 1,024 terms): one wave per element, lane l sums partials
// l, l + 64, ... and the lanes combine in a fixed butterfly — still one fixed association.
__global__ void nerf_grad_reduce_kernel(const BwdArgs ba) {
    const Shape sh = shape_of(ba.a);
    const int direct_blocks = (grad_elements(sh) + kReduceThreads - 1) / kReduceThreads;
    if ((int)blockIdx.x >= direct_blocks) {
        const int lane = threadIdx.x & 63;
        const int ge = ((int)blockIdx.x - direct_blocks) * 4 + (threadIdx.x >> 6);   // [layer][gamma|beta][256]
        const int L = ge / (2 * kHidden), which = (ge / kHidden) & 1, idx = ge % kHidden;
        const float* p = ba.gb_partial + ge;
        const float sum = wave_strided_sum(p, ba.data_grid, kGbFloats, lane);
        if (lane == 0 && idx < sh.hidden) ba.grad[grad_offset(4 * L + 2 + which, sh) + idx] = sum;
        return;
    }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= grad_elements(sh)) return;
    int tensor, idx;
    locate(e, sh, tensor, idx);
    const int L = tensor / 4, which = tensor % 4;
    if (which >= 2) return;                       // gamma / beta: the blocks behind
    int so;
    if (which == 0) {          // (row, column) of the tensor -> its place in the full-width slab
        // (layer 0's columns in the order of the forward that saved the inputs: the narrow kernels spread the scales
        //  the network has over the lane groups, nerf_layout.h: scales_per_group)
        if (L == 0) so = kSlabW0 + (idx / sh.enc_in) * kEncIn +
                         layer0_kernel_column(idx % sh.enc_in, sh.scales(), train_tiles(sh.hidden) == 8 ? scales_per_group(sh.scales()) : 4);
        else if (L == 5) so = kSlabW5 + slot_of_row(idx / sh.hidden, sh) * kHidden + idx % sh.hidden;
        else so = kSlabWh + (L - 1) * kHidden * kHidden + (idx / sh.hidden) * kHidden + idx % sh.hidden;
    } else {
        so = kSlabB + L * kHidden + (L == 5 ? slot_of_row(idx, sh) : idx);
    }
    ba.grad[e] = strided_sum(ba.slabs + so, ba.splits, kSlabFloats);
}

// Split-K factor of the weight gradient: every split writes a partial slab of ALL gradients (1.2 MB; 2.7 MB for
// the legacy network) that the reduce kernel reads back, so at small batches the slab traffic, not the GEMM, sets
// the time (512 rays x 64 with 128 splits of 8 tiles: 156 MB written + read around 37 us of MFMA work).  24
// 32-sample tiles per split keep one round of workgroups on the chip at that size (6 jobs x 42 splits = 252) and
// leave the 4096-ray batch at the 128-split cap.
int choose_splits(int64_t n_tiles) {
    int64_t s = n_tiles / 24;
    if (s < 1) s = 1;
    if (s > kMaxSplits) s = kMaxSplits;
    return (int)s;
}

}  // namespace

extern "C" {

size_t nerf_hip_backward_scratch_bytes(int64_t n_rays, int32_t num_samples) {
    if (n_rays <= 0 || num_samples < 2) return 0;
    return ((size_t)kMaxSplits * kSlabFloats + (size_t)kMaxDataGrid * (kGbFloats + 8)) * sizeof(float);
}

int nerf_hip_render_backward(const NerfHipBackwardArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: null args");
    const NerfHipRenderArgs& a = args->fwd;
    if (args->grad == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: grad is null");
    if (!shape_ok(shape_of(a)))
        return nerf_common::fail(NERF_HIP_EUNSUPPORTED, "render_backward: network shape out of range (hidden 1 .. 256, enc_inputs "
                                                        "6 .. 96 in steps of 6, color_outputs 1 .. 12, num_outputs 1 + color_outputs .. 64)");
    if (a.n_rays == 0)      // empty batch (an empty data-parallel shard): the gradient is zero
        return nerf_common::check_hip(hipMemsetAsync(args->grad, 0, (size_t)grad_elements(shape_of(a)) * sizeof(float),
                                                     (hipStream_t)stream), "render_backward memset");
    if (args->scratch == nullptr || (args->d_rgb == nullptr && args->d_raw == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: scratch is null, or neither d_rgb nor d_raw is given");
    if (a.train_workspace == nullptr || a.packed == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: forward was not a training forward");
    if (args->d_seg != nullptr && a.seg == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: d_seg given but forward seg is null");
    if (a.n_rays < 0 || a.num_samples < 2 || a.num_samples > 4096)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_backward: n_rays / num_samples out of range");
    hipStream_t st = (hipStream_t)stream;

    BwdArgs ba;
    ba.a = a;
    derive_slot_constants(ba.a);
    ba.d_rgb = args->d_rgb;
    ba.d_seg = args->d_seg;
    ba.d_raw = args->d_raw;
    ba.intervals = a.num_samples - 1;
    ba.chunks = (ba.intervals + kSamplesPerWave - 1) / kSamplesPerWave;
    const int tt = train_tiles(shape_of(a).hidden);      // 8: a narrow network (hidden_size <= 128), at its own cost
    ba.L = make_train_layout(a.n_rays, ba.chunks, 16 * tt);
    ba.groups = ba.L.mp / 16 / kWavesPerWg;                 // (padded ray, chunk) items / 4 waves
    const int64_t slots = ba.L.mp / 16 / ba.chunks;         // padded rays
    ba.grad = args->grad;
    ba.n_tiles = ba.L.mp / kKs;
    ba.splits = choose_splits(ba.n_tiles);
    ba.tiles_per_split = (ba.n_tiles + ba.splits - 1) / ba.splits;
    ba.slabs = args->scratch;
    ba.inv_n = 1.0f / (float)shape_of(a).hidden;
    ba.gb_partial = args->scratch + (size_t)kMaxSplits * kSlabFloats;
    ba.dymax = ba.gb_partial + (size_t)kMaxDataGrid * kGbFloats;

    int device = 0, cus = 0;
    int rc = nerf_common::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (rc) return rc;
    rc = nerf_common::check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device),
                                "hipDeviceGetAttribute");
    if (rc) return rc;
    static unsigned done_data = 0, done_wgrad = 0;
    const bool half = a.precision == NERF_HIP_PRECISION_F16X3;   // the arithmetic of the training forward
    static unsigned done_data_h = 0;
    rc = half ? nerf_common::ensure_dynamic_lds((const void*)nerf_bwd_data_h_kernel, kBwdLdsBytes + 32, device,
                                                &done_data_h)
              : nerf_common::ensure_dynamic_lds((const void*)nerf_bwd_data_kernel, kBwdLdsBytes, device,
                                                &done_data);
    if (rc) return rc;
    rc = nerf_common::ensure_dynamic_lds((const void*)nerf_wgrad_kernel, kRingSlots * kRingSlotBytes, device,
                                         &done_wgrad);
    if (rc) return rc;
    static unsigned done_wgrad_h = 0;
    rc = nerf_common::ensure_dynamic_lds((const void*)nerf_wgrad_h_kernel, kRingSlots * kRingSlotBytes, device,
                                         &done_wgrad_h);
    if (rc) return rc;
    static unsigned done_data_n8 = 0, done_wgrad_n8 = 0, done_data_h_n8 = 0, done_wgrad_h_n8 = 0, done_data_n4 = 0, done_wgrad_n4 = 0;
    const int ct = train_compute_tiles(shape_of(a).hidden, half);     // 4: hidden_size <= 64 in fp32 arithmetic
    if (ct == 4) {
        rc = nerf_common::ensure_dynamic_lds((const void*)nerf_bwd_data_n_kernel<4>, kBwdLdsBytes, device, &done_data_n4);
        if (rc) return rc;
        rc = nerf_common::ensure_dynamic_lds((const void*)nerf_wgrad_n4_kernel, kRingSlots * kRingSlotBytes, device, &done_wgrad_n4);
        if (rc) return rc;
    }
    if (tt == 8) {
        rc = half ? nerf_common::ensure_dynamic_lds((const void*)nerf_bwd_data_h_n8_kernel, kBwdLdsBytes + 32, device, &done_data_h_n8)
                  : nerf_common::ensure_dynamic_lds((const void*)nerf_bwd_data_n_kernel<8>, kBwdLdsBytes, device, &done_data_n8);
        if (rc) return rc;
        rc = half ? nerf_common::ensure_dynamic_lds((const void*)nerf_wgrad_h_n8_kernel, kRingSlots * kRingSlotBytes, device, &done_wgrad_h_n8)
                  : nerf_common::ensure_dynamic_lds((const void*)nerf_wgrad_n8_kernel, kRingSlots * kRingSlotBytes, device, &done_wgrad_n8);
        if (rc) return rc;
    }
    int64_t grid = (int64_t)cus * 2;
    if (grid > ba.groups) grid = ba.groups;
    if (grid > kMaxDataGrid) grid = kMaxDataGrid;
    ba.data_grid = (int)grid;

    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_COMPOSITE_BACKWARD);
        if (ba.d_raw != nullptr)
            hipLaunchKernelGGL(nerf_field_scatter_kernel, dim3((unsigned)((ba.L.mp * (kOutPad / 4) + 255) / 256)), dim3(256), 0, st, ba);
        else
            hipLaunchKernelGGL(nerf_composite_bwd_kernel, dim3((unsigned)((slots + kWavesPerWg - 1) / kWavesPerWg)),
                               dim3(256), 0, st, ba);
    }
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_DATA_GRADIENT);
        if (half && tt == 8) hipLaunchKernelGGL(nerf_bwd_data_h_n8_kernel, dim3((unsigned)grid), dim3(256), kBwdLdsBytes + 32, st, ba);
        else if (half) hipLaunchKernelGGL(nerf_bwd_data_h_kernel, dim3((unsigned)grid), dim3(256), kBwdLdsBytes + 32, st, ba);
        else if (ct == 4) hipLaunchKernelGGL(nerf_bwd_data_n_kernel<4>, dim3((unsigned)grid), dim3(256), kBwdLdsBytes, st, ba);
        else if (tt == 8) hipLaunchKernelGGL(nerf_bwd_data_n_kernel<8>, dim3((unsigned)grid), dim3(256), kBwdLdsBytes, st, ba);
        else hipLaunchKernelGGL(nerf_bwd_data_kernel, dim3((unsigned)grid), dim3(256), kBwdLdsBytes, st, ba);
    }
    const int wgrad_jobs = 6;
    const bool wgrad_half = half;
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_WEIGHT_GRADIENT);
        if (tt == 8 && wgrad_half)
            hipLaunchKernelGGL(nerf_wgrad_h_n8_kernel, dim3(ba.splits * wgrad_jobs), dim3(256), kRingSlots * kRingSlotBytes, st, ba);
        else if (ct == 4)
            hipLaunchKernelGGL(nerf_wgrad_n4_kernel, dim3(ba.splits * kWgradJobsN4), dim3(256), kRingSlots * kRingSlotBytes, st, ba);
        else if (tt == 8)
            hipLaunchKernelGGL(nerf_wgrad_n8_kernel, dim3(ba.splits * wgrad_jobs), dim3(256), kRingSlots * kRingSlotBytes, st, ba);
        else if (wgrad_half)
            hipLaunchKernelGGL(nerf_wgrad_h_kernel, dim3(ba.splits * wgrad_jobs), dim3(256), kRingSlots * kRingSlotBytes, st, ba);
        else
            hipLaunchKernelGGL(nerf_wgrad_kernel, dim3(ba.splits * wgrad_jobs), dim3(256), kRingSlots * kRingSlotBytes, st, ba);
    }
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_REDUCE);
        hipLaunchKernelGGL(nerf_grad_reduce_kernel, dim3((grad_elements(shape_of(a)) + kReduceThreads - 1) / kReduceThreads + kReduceGbBlocks), dim3(kReduceThreads),
                           0, st, ba);
    }
    return nerf_common::check_hip(hipGetLastError(), "render_backward launch");
}

}  // extern "C"

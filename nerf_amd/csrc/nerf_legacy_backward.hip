// Backward of the LEGACY (generation-A) 8 x 256 network's fused renderer w.r.t. its 44 parameter tensors
// (gfx950 only): what PyTorch autograd does behind `loss.backward()` in the notebook's training loop
// (examples/example.ipynb cell 8; train_conditional_nerf.py:130-135) for the network of examples/nerf.pth.
// PARITY UNPINNED like the forward (no source of this network is in the reference repository): the spec is
// autograd through oracle/legacy_oracle.py.
//
// Same four launches as the main network's backward (nerf_backward.hip), same machinery
// (nerf_backward_common.h), no float atomics, bitwise reproducible:
//   1. nerf_legacy_composite_bwd_kernel — per ray, last chunk first: compositing backward
//      (nerf/model.py:438-469, :660) -> dL/d(density, r, g, b) of every sample.
//   2. nerf_legacy_bwd_data_kernel — per 16-sample chunk: dX = W^T dY down the chain color head -> L9 -> L8
//      (+ density head) -> L7 .. L1 with the transposed fp32 image streamed through the LDS ring, the
//      LayerNorm / ReLU backward of every wide layer in registers from the saved a_hat, 1/std and gate
//      threshold; writes dY of the ten wide layers (row order); gamma / beta gradients summed per workgroup
//      in LDS in wave order.  Layer order here is Linear -> ReLU -> LayerNorm, so the gate is applied LAST.
//   3. nerf_legacy_wgrad_kernel — the 14 weight-gradient products dW = dY^T X in ONE launch (nine 256 x 256
//      blocks, the three encoding blocks of L0 / L4 / L8, the two heads), bf16-triple operands, fp32
//      accumulation; X of a hidden block is rebuilt as gamma a_hat + beta while the operands are read.
//   4. nerf_legacy_grad_reduce_kernel — partial slabs -> the flat 638,468-element gradient (the 44 tensors
//      in nerf_legacy_layout.h's order, PyTorch layouts; undoes the encodings' column padding).
#include "nerf_backward_common.h"
#include "nerf_legacy_layout.h"

using namespace nerf_layout;
using namespace nerf_device;
using namespace nerf_bwd;
using namespace nerf_legacy;

namespace {

constexpr int kLGbFloats = kWide * 2 * kHidden;                    // gamma / beta partials per workgroup
constexpr int kGammaFloats = kWide * kHidden;                      // the ten gamma vectors, [L][g][T][r]
constexpr int kLBwdLdsBytes = kRingBytes + (kGammaFloats + kLGbFloats) * 4;      // 78 KiB -> 2 workgroups / CU

// partial-slab layout (floats) of one split
constexpr int kHiddenJobs = 9;                                      // L1, L2, L3, L4 (hidden part), L5, L6, L7, L8 (hidden part), L9
constexpr int kLSlabHid = 0;                                        // 9 x [256][256]
constexpr int kLSlabEnc = kLSlabHid + kHiddenJobs * kHidden * kHidden;   // 3 x [256][64]: L0, L4 | position, L8 | direction
constexpr int kLSlabHead = kLSlabEnc + 3 * kHidden * kEncPad;       // 2 x [64][256]: rows of (d density, d r, d g, d b, 0 ..) x x'_7 / x'_9
constexpr int kLSlabB = kLSlabHead + 2 * kOutPad * kHidden;         // 10 x [256] wide biases, 2 x [64] head rows
constexpr int kLSlabSpare = kLSlabB + kWide * kHidden + 2 * kOutPad;     // [256] bias sums nobody reads (the encoding blocks)
constexpr int kLSlabFloats = kLSlabSpare + kHidden;
constexpr int kWgradJobs = kHiddenJobs + 3 + 2;

typedef WgradShape<kHidden, kEncPad, 2, 2, kMapRows> ShapeEnc;     // waves: out tiles 2w..2w+1, both in tiles

struct LBwdArgs {
    NerfHipRenderArgs a;
    const float* d_rgb;
    int32_t samples, chunks;
    int64_t groups;
    LegacyTrainLayout L;
    float* gb_partial;          // [grid][10][2][256]
    float* dymax;               // [grid][16]: largest |dY| each data-gradient workgroup saw: 0..9 dy[l], 10 the heads'
                                // rows; split-precision path only
    float* slabs;               // [splits][kLSlabFloats]
    float* rows;                // the scratch buffer's row area: dY of the ten wide layers, dL/d(out) (L.dy[], L.dy5)
    float* grad;
    int32_t splits, data_grid;
    int64_t tiles_per_split, n_tiles;
};

typedef WeightPipe<kLegacyBwdStages> LBwdPipe;

__global__ __launch_bounds__(256) void nerf_legacy_composite_bwd_kernel(const LBwdArgs ba) {
    CompositeBwd cb;
    cb.d_rgb = ba.d_rgb, cb.d_seg = nullptr;
    cb.intervals = ba.samples, cb.chunks = ba.chunks;
    cb.mp = ba.L.mp, cb.out = ba.L.out, cb.comp = ba.L.comp, cb.dy5_rows = ba.rows + ba.L.dy5;
    composite_bwd_body(ba.a, cb);
}

// Stage hook of the data-gradient loops: the wave-ordered gamma / beta adds, and at stage 1 the loads of the
// NEXT LayerNorm backward's saved a_hat tile, 1/std and gate threshold (nerf_backward_common.h: BwdHook).
struct LegacyHook {
    GammaBetaTurn& turn;
    const float* xhat_row;
    const float* rstd_ptr;
    const float* shift_ptr;
    f32x4 (&xh)[16];
    float& rstd;
    float& shift;
    __device__ __forceinline__ void operator()(int t) const {
        turn(t);
        if (t == 1) {
#pragma unroll
            for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xhat_row + T * kTileT);
            rstd = *rstd_ptr;
            shift = *shift_ptr;
        }
    }
};

// LayerNorm + ReLU backward of one wide layer on the register tile, for the order Linear -> ReLU -> LayerNorm:
//   in : acc = dL/dx' (the LayerNorm's output = the next Linear's input), saved a_hat tile, 1/std, shift
//   out: act = dL/dy (the Linear's output) = the next B operands; also stored row-major
//   d a = (gamma d x' - mean(gamma d x') - a_hat mean(gamma d x' a_hat)) / std ;  d y = d a where y > 0
//   kScaled (split-precision chain): acc holds d x' times the per-sample power of two that `unscale` undoes
template <bool kScaled = false>
__device__ __forceinline__ void relu_layer_norm_bwd(const float* gamma_l, int g, int j, f32x4 (&acc)[16],
                                                    float (&act)[64], const f32x4 (&xh)[16], float rstd,
                                                    float shift, float* dy_row, float* gb_l, GammaBetaTurn& turn,
                                                    float unscale = 1.0f) {
    const f32x4* gam = (const f32x4*)(gamma_l + g * 64);
    float s1 = 0.f, s2 = 0.f;
    // one tile: gamma d x' into the accumulator, the two LayerNorm moments; v[0..3] = d x' (beta gradient
    // terms), v[4..7] = d x' a_hat (gamma gradient terms)
    auto tile = [&](int T, float (&v)[8]) {
        const f32x4 ga = gam[T];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // (the value the inline-asm butterfly reads must be a VALU result, not the raw accumulator of an MFMA
            //  that may still be in flight — nerf_amd/isa_scan.py rule R1: the un-scaling product, or the median
            //  of three copies)
            const float dz = kScaled ? acc[T][r] * unscale : acc[T][r];
            v[r] = kScaled ? dz : __builtin_amdgcn_fmed3f(dz, dz, dz);
            v[4 + r] = dz * xh[T][r];
            const float gdz = ga[r] * dz;
            s1 += gdz;
            s2 = __builtin_fmaf(gdz, xh[T][r], s2);
            acc[T][r] = gdz;
        }
    };
    // sums over the 16 samples of a row, lane j keeps tile j: reduce-scatter butterfly as the tiles come
    // (nerf_device.h: scatter_level8 / 4 / take)
    float kept[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    const f32x4 keep_b = {kept[0], kept[1], kept[2], kept[3]}, keep_g = {kept[4], kept[5], kept[6], kept[7]};
    turn.dst = gb_l + 16 * j + 4 * g;             // features 16 j + 4 g + r, added in wave order later
    turn.kg = keep_g;
    turn.kb = keep_b;
    const float m1 = group_sum(s1) * (1.0f / 256.0f);
    const float m2 = group_sum(s2) * (1.0f / 256.0f);
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        f32x4 dy;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float da = rstd * ((acc[T][r] - m1) - xh[T][r] * m2);
            dy[r] = xh[T][r] > shift ? da : 0.f;
            act[4 * T + r] = dy[r];
        }
        *(f32x4*)(dy_row + T * kTileT) = dy;
    }
}

__global__ __launch_bounds__(256, 2) void nerf_legacy_bwd_data_kernel(const LBwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gamma = (float*)(smem + kRingBytes);
    float* const gb = gamma + kGammaFloats;

    // gamma of the ten LayerNorms (the backward needs neither bias nor beta: the ReLU gate comes from the
    // saved a_hat), [L][g][T][r] as in the forward's small image
    for (int i = threadIdx.x; i < kGammaFloats; i += 256)
        gamma[i] = a.packed[kLegacyBlobFloats + (i / kHidden) * kLegacySmallPerLayer + kHidden + (i % kHidden)];
    for (int i = threadIdx.x; i < kLGbFloats; i += 256) gb[i] = 0.f;

    LBwdPipe pipe;
    pipe.init(a.packed + kLegacyBwdOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;

    // one (padded ray, chunk) item per wave, as in the main network's data gradient
    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;
        const int64_t sp = tile * 16 + j;
        // rows as (this tile's base: wave-uniform, SGPRs) + (the lane's 32-bit offset inside a tile): the register
        // tiles T lie 1 KiB apart, beyond the 4 KiB an instruction offset reaches — per-lane 64-bit pointers would
        // need four bases per tensor (nerf_backward.hip: nerf_bwd_data_h_kernel has the longer story)
        const float* const xbase = ws + tile * kTileFloats;           // + L.xhat[l]: this tile's a_hat
        float* const dybase = ba.rows + tile * kTileFloats;           // + L.dy[l]
        const uint32_t row_off = (uint32_t)tile_lane_word(j, g);
        auto lane_word = [](uint32_t v) {                             // (taken at every use: keeps the offset 32-bit)
            asm volatile("" : "+v"(v));
            return v;
        };
        const float* const stat = ws + sp;                            // + L.rstd[l] / L.shift[l]
        // dL/d(density, r, g, b) of this sample: k slots (g 0, r 0..3) of BOTH head stages (the transposed
        // head images carry zeros in the slots that are not theirs; lane groups 1..3 read zero columns)
        const float* const dhead = ba.rows + ba.L.dy5 + sp * kOutPad + 4 * g;
        f32x4 xh[16];
        float rstd, shift;
#pragma unroll
        for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xbase + ba.L.xhat[9] + lane_word(row_off) + T * kTileT);
        rstd = stat[ba.L.rstd[9]];
        shift = stat[ba.L.shift[9]];
        // ---- color head: dX'_9 = Wc^T d(color) ----
#pragma unroll
        for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const f32x4 dh[1] = {*(const f32x4*)dhead};
            layer_wide_v4<1>(pipe, acc, dh);
        }
        // ---- L9, L8, [density head joins: dX'_7 += Wd^T d(density)], L7 .. L1 (L4: hidden columns only, the
        // encodings take no gradient), L0's LayerNorm backward: ONE code instance of the layer body, two runs of
        // the same loop (three copies of the LayerNorm backward and two of the 16-stage loop cost this kernel
        // 356 B of scratch per lane, 85 accesses inside the loops) ----
        int l = 9;
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            const int last = phase == 0 ? 8 : 0;
#pragma unroll 1
            for (; l >= last; --l) {
                relu_layer_norm_bwd(gamma + l * kHidden, g, j, acc, act, xh, rstd, shift, dybase + ba.L.dy[l] + lane_word(row_off),
                                    gb + l * 2 * kHidden, turn);
                if (l == 0) break;
                const int ln = l - 1;
#pragma unroll
                for (int T = 0; T < 16; ++T) acc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                layer_wide<16>(pipe, acc, act,
                               LegacyHook{turn, xbase + ba.L.xhat[ln] + lane_word(row_off), stat + ba.L.rstd[ln], stat + ba.L.shift[ln], xh,
                                          rstd, shift});
            }
            if (phase == 0) {
                const f32x4 dh[1] = {*(const f32x4*)dhead};
                layer_wide_v4<1>(pipe, acc, dh);
            }
        }
        // layer 0's partials: no 4-stage loop follows inside this item (the next item opens with the
        // one-stage color head), so the four waves take their turns here, a barrier apart
        for (int t = 0; t < kWavesPerWg; ++t) {
            __syncthreads();
            turn(t);
        }
        turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < kLGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kLGbFloats + i] = gb[i];
}

// The same chain in split-precision arithmetic (the training forward's precision = F16X3), as the main
// network's nerf_bwd_data_h_kernel: a sample's dY row becomes f16 pairs after an exact per-sample power-of-two
// scaling (row_scale), the transposed f16-pair image streams through the ring (layer_wide_h), the scale is
// undone in the LayerNorm backward that consumes the accumulators.  The density head joins L8's product in the
// SAME accumulators, so its single value takes part in that row's maximum and uses that row's scale.  The
// largest |dY| of every layer is recorded per workgroup for the weight gradient's batch-wide scale.
constexpr int kLYoungerHead = 18, kLYoungerHidden = 34;       // 16 a_hat + 1/std + shift loads (+ 16 dY saves)
__global__ __launch_bounds__(256, 2) void nerf_legacy_bwd_data_h_kernel(const LBwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ba.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    float* const ws = a.train_workspace;
    float* const gamma = (float*)(smem + kRingBytes);
    float* const gb = gamma + kGammaFloats;
    int* const wmax = (int*)(smem + kLBwdLdsBytes);

    for (int i = threadIdx.x; i < kGammaFloats; i += 256)
        gamma[i] = a.packed[kLegacyBlobFloats + (i / kHidden) * kLegacySmallPerLayer + kHidden + (i % kHidden)];
    for (int i = threadIdx.x; i < kLGbFloats; i += 256) gb[i] = 0.f;
    if (threadIdx.x < 16) wmax[threadIdx.x] = 0;

    WeightPipe<kLegacyBwdHStages> pipe;
    pipe.init(a.packed + kLegacyBwdHOffset, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();

    float act[64];
    f32x4 acc[16];
    GammaBetaTurn turn;
    turn.dst = gb + 16 * j + 4 * g;
    turn.kg = turn.kb = f32x4{0.f, 0.f, 0.f, 0.f};
    turn.wave = wave;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    for (int64_t grp = blockIdx.x; grp < ba.groups; grp += gridDim.x) {
        const int64_t tile = grp * kWavesPerWg + wave;
        const int64_t sp = tile * 16 + j;
        // a_hat tiles as (this tile's base: wave-uniform) + (the lane's 32-bit offset, taken at every use); the dY rows and
        // the per-sample scalars keep their per-lane pointers.  Which of the three goes uniform was swept
        // (NOTES.md R4: all three 180-208 B of scratch and 1.98 ms; this one 68 B, none inside the loops, 1.47 against
        // 1.54 ms) — the allocator's answer to a source change in these 256-register kernels has to be measured.
        const float* const xbase = ws + tile * kTileFloats;
        const uint32_t row_off = (uint32_t)tile_lane_word(j, g);
        auto lane_word = [](uint32_t v) {
            asm volatile("" : "+v"(v));
            return v;
        };
        float* const dybase = ba.rows + tile_lane_base(sp, g);
        const float* const stat = ws + sp;
        // dL/d(density, r, g, b) on lane group 0 (zeros elsewhere), then a_hat / 1/std / shift of L9: 18 loads
        // that fly under the two stages of the color head
        const f32x4 dh = *(const f32x4*)(ba.rows + ba.L.dy5 + sp * kOutPad + 4 * g);
        f32x4 xh[16];
        float rstd, shift, unscale;
#pragma unroll
        for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xbase + ba.L.xhat[9] + lane_word(row_off) + T * kTileT);
        rstd = stat[ba.L.rstd[9]];
        shift = stat[ba.L.shift[9]];
        {
            float amax;
            const float sc = row_scale(abs_max4(0.f, dh), unscale, amax);
            note_max(wmax + 10, amax, lane);
            h8 bh[1], bl[1];
            split8(dh * sc, zero, bh[0], bl[0]);
#pragma unroll
            for (int T = 0; T < 16; ++T) acc[T] = zero;
            layer_wide_h<1, kLYoungerHead>(pipe, acc, bh, bl);           // color head: dX'_9 = Wc^T d(color)
        }
        // L9, L8, [density head], L7 .. L1: ONE code instance of the layer body (two runs of the same loop)
        int l = 9;
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            const int last = phase == 0 ? 8 : 1;
            float sc = 1.0f, ddens = 0.f;
#pragma unroll 1
            for (; l >= last; --l) {
                relu_layer_norm_bwd<true>(gamma + l * kHidden, g, j, acc, act, xh, rstd, shift, dybase + ba.L.dy[l],
                                          gb + l * 2 * kHidden, turn, unscale);
                // the next LayerNorm backward's saved tile: 18 loads behind the 16 saves above (+ the density
                // gradient, which joins L8's product in the same accumulators: same row, same scale)
#pragma unroll
                for (int T = 0; T < 16; ++T) xh[T] = *(const f32x4*)(xbase + ba.L.xhat[l - 1] + lane_word(row_off) + T * kTileT);
                rstd = stat[ba.L.rstd[l - 1]];
                shift = stat[ba.L.shift[l - 1]];
                ddens = ba.rows[ba.L.dy5 + sp * kOutPad];
                // the sample's largest |dy|: this layer's B-operand scale, and (folded into the workgroup's
                // maximum) the weight-gradient kernel's
                float m = 0.f;
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    m = abs_max4(m, f32x4{act[4 * T], act[4 * T + 1], act[4 * T + 2], act[4 * T + 3]});
                note_max(wmax + l, group_max(m), lane);
                float amax;
                sc = row_scale(l == 8 ? __builtin_fmaxf(m, __builtin_fabsf(ddens)) : m, unscale, amax);
                h8 bh[8], bl[8];
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    const int t0 = 8 * mb, t1 = 8 * mb + 4;
                    split8(f32x4{act[t0], act[t0 + 1], act[t0 + 2], act[t0 + 3]} * sc,
                           f32x4{act[t1], act[t1 + 1], act[t1 + 2], act[t1 + 3]} * sc, bh[mb], bl[mb]);
                }
#pragma unroll
                for (int T = 0; T < 16; ++T) acc[T] = zero;
                layer_wide_h<8, kLYoungerHidden + 1>(pipe, acc, bh, bl, TurnHook{turn});
            }
            if (phase == 0) {                     // density head: dX'_7 += Wd^T d(density), under L8's row scale
                h8 dbh[1], dbl[1];
                split8(f32x4{g == 0 ? ddens * sc : 0.f, 0.f, 0.f, 0.f}, zero, dbh[0], dbl[0]);
                layer_wide_h<1, 0>(pipe, acc, dbh, dbl);
            }
        }
        relu_layer_norm_bwd<true>(gamma, g, j, acc, act, xh, rstd, shift, dybase + ba.L.dy[0], gb, turn, unscale);
        {
            float m = 0.f;                        // dy[0] feeds only the weight gradient
#pragma unroll
            for (int T = 0; T < 16; ++T)
                m = abs_max4(m, f32x4{act[4 * T], act[4 * T + 1], act[4 * T + 2], act[4 * T + 3]});
            note_max(wmax, group_max(m), lane);
        }
        // layer 0's partials: the next item opens with the two-stage color head, so the four waves take their
        // turns here, a barrier apart
        for (int t = 0; t < kWavesPerWg; ++t) {
            __syncthreads();
            turn(t);
        }
        turn.kg = turn.kb = zero;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < kLGbFloats; i += 256)
        ba.gb_partial[(int64_t)blockIdx.x * kLGbFloats + i] = gb[i];
    if (threadIdx.x < 16) ba.dymax[(int64_t)blockIdx.x * 16 + threadIdx.x] = __builtin_bit_cast(float, wmax[threadIdx.x]);
}

// All 14 products in ONE launch: job = blockIdx.x / splits, heavy (256 x 256) blocks first.
__global__ __launch_bounds__(256, 1) void nerf_legacy_wgrad_kernel(const LBwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kLSlabFloats, nullptr, 0, 16};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kLegacyBlobFloats;            // [layer][bias | gamma | beta][256]
    if (job < kHiddenJobs) {
        // wide layer l = job + 1: dY_l against x'_{l-1} = gamma_{l-1} a_hat_{l-1} + beta_{l-1}
        const int l = job + 1;
        wgrad_body_ring<ShapeHid, kInputAffine>(jb, smem, ba.rows + ba.L.dy[l], ws + ba.L.xhat[l - 1],
                                                small + (l - 1) * kLegacySmallPerLayer,
                                                kLSlabHid + job * kHidden * kHidden, kLSlabB + l * kHidden);
    } else if (job < kHiddenJobs + 3) {
        // the encoding columns: L0 x position, L4 x position, L8 x direction (only L0's bias sums are read)
        const int e = job - kHiddenJobs;
        const int l = e == 0 ? 0 : (e == 1 ? 4 : 8);
        wgrad_body_ring<ShapeEnc, kInputRaw>(jb, smem, ba.rows + ba.L.dy[l], ws + (e == 2 ? ba.L.dir : ba.L.pos), nullptr,
                                             kLSlabEnc + e * kHidden * kEncPad, e == 0 ? kLSlabB : kLSlabSpare);
    } else {
        // the heads: rows (d density, d r, d g, d b, 0 ..) against x'_7 (density: row 0) or x'_9 (color: rows 1..3)
        const int h = job - kHiddenJobs - 3;
        const int l = h == 0 ? 7 : 9;
        wgrad_body_ring<ShapeL5, kInputAffine>(jb, smem, ba.rows + ba.L.dy5, ws + ba.L.xhat[l],
                                               small + l * kLegacySmallPerLayer, kLSlabHead + h * kOutPad * kHidden,
                                               kLSlabB + kWide * kHidden + h * kOutPad);
    }
}

// The same launch in the split-precision training mode: every product on f16 pairs under ONE power-of-two
// scale of its dY per batch (nerf_backward_common.h: wgrad_body_ring<.., kF16 = true>), taken from the maxima the
// data-gradient kernel recorded (dymax index: the layer, 10 for the heads' rows).
__global__ __launch_bounds__(256, 1) void nerf_legacy_wgrad_h_kernel(const LBwdArgs ba) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int job = blockIdx.x / ba.splits, split = blockIdx.x % ba.splits;
    const WgradJob jb{ba.tiles_per_split, ba.n_tiles, split, ba.slabs + (int64_t)split * kLSlabFloats, ba.dymax,
                      ba.data_grid, 16};
    const float* ws = ba.a.train_workspace;
    const float* small = ba.a.packed + kLegacyBlobFloats;
    if (job < kHiddenJobs) {
        const int l = job + 1;
        wgrad_body_ring<ShapeHid, kInputAffine, true>(jb, smem, ba.rows + ba.L.dy[l], ws + ba.L.xhat[l - 1],
                                                      small + (l - 1) * kLegacySmallPerLayer,
                                                      kLSlabHid + job * kHidden * kHidden, kLSlabB + l * kHidden, l);
    } else if (job < kHiddenJobs + 3) {
        const int e = job - kHiddenJobs;
        const int l = e == 0 ? 0 : (e == 1 ? 4 : 8);
        wgrad_body_ring<ShapeEnc, kInputRaw, true>(jb, smem, ba.rows + ba.L.dy[l], ws + (e == 2 ? ba.L.dir : ba.L.pos), nullptr,
                                                   kLSlabEnc + e * kHidden * kEncPad, e == 0 ? kLSlabB : kLSlabSpare, l);
    } else {
        const int h = job - kHiddenJobs - 3;
        const int l = h == 0 ? 7 : 9;
        wgrad_body_ring<ShapeL5, kInputAffine, true>(jb, smem, ba.rows + ba.L.dy5, ws + ba.L.xhat[l],
                                                     small + l * kLegacySmallPerLayer,
                                                     kLSlabHead + h * kOutPad * kHidden,
                                                     kLSlabB + kWide * kHidden + h * kOutPad, 10);
    }
}

// ---------------------------------------------------------------------------------------------
// deterministic reduction of the partials into the flat gradient vector
// ---------------------------------------------------------------------------------------------
constexpr int kReduceThreads = 256;
constexpr int kReduceDirectBlocks = (kLegacyGradElements + kReduceThreads - 1) / kReduceThreads;
constexpr int kReduceGbBlocks = kLGbFloats / 4;                    // one wave per gamma / beta element

__device__ __forceinline__ void legacy_locate(int e, int& tensor, int& idx) {
    tensor = 0;
    int off = 0;
    for (;;) {
        const int n = legacy_tensor_elements(tensor);
        if (e < off + n) break;
        off += n;
        ++tensor;
    }
    idx = e - off;
}

__global__ void nerf_legacy_grad_reduce_kernel(const LBwdArgs ba) {
    if ((int)blockIdx.x >= kReduceDirectBlocks) {
        // gamma / beta: partials come one per data-gradient workgroup; one wave per element
        const int lane = threadIdx.x & 63;
        const int ge = ((int)blockIdx.x - kReduceDirectBlocks) * 4 + (threadIdx.x >> 6);    // [layer][gamma|beta][256]
        const int l = ge / (2 * kHidden), which = (ge / kHidden) & 1, idx = ge % kHidden;
        const float* p = ba.gb_partial + ge;
        const float sum = wave_strided_sum(p, ba.data_grid, kLGbFloats, lane);
        if (lane == 0) ba.grad[legacy_grad_offset(wide_param(l) + 2 + which) + idx] = sum;
        return;
    }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kLegacyGradElements) return;
    int tensor, idx;
    legacy_locate(e, tensor, idx);
    int so;
    if (tensor == kDensityW) so = kLSlabHead + idx;                                        // row 0 of the density job
    else if (tensor == kDensityB) so = kLSlabB + kWide * kHidden;
    else if (tensor == kColorW) so = kLSlabHead + kOutPad * kHidden + (1 + idx / kHidden) * kHidden + idx % kHidden;
    else if (tensor == kColorB) so = kLSlabB + kWide * kHidden + kOutPad + 1 + idx;
    else {
        const int w = tensor < kDensityW ? tensor : tensor - 2;
        const int l = w / 4, which = w % 4;
        if (which >= 2) return;                   // gamma / beta: the blocks behind
        if (which == 1) so = kLSlabB + l * kHidden + idx;
        else {
            const int K = wide_inputs(l), row = idx / K, col = idx % K;
            if (l == 0) so = kLSlabEnc + row * kEncPad + encoding_column(col, kPosFreqs);
            else if (col < kHidden) so = kLSlabHid + (l - 1) * kHidden * kHidden + row * kHidden + col;
            else if (l == 4) so = kLSlabEnc + kHidden * kEncPad + row * kEncPad + encoding_column(col - kHidden, kPosFreqs);
            else so = kLSlabEnc + 2 * kHidden * kEncPad + row * kEncPad + encoding_column(col - kHidden, kDirFreqs);
        }
    }
    ba.grad[e] = strided_sum(ba.slabs + so, ba.splits, kLSlabFloats);
}

// Split-K factor of the weight gradient (nerf_backward.hip: choose_splits has the reasoning): one split per 24
// 32-sample tiles, but at most 64 here, not the main network's 128 — this network's partial slab is 2.7 MB and
// its 14 jobs already make 896 workgroups at 64 splits.  4096 rays x 64, weight-gradient + reduce kernel
// (scripts/ab_kernels.sh): 128 splits 1.315 + 0.085 ms, 96: 1.304 + 0.070, 64: 1.289 + 0.055, 48: 1.348 +
// 0.047, 32: 1.545 + 0.042.  (The main network measured the other way: 64 splits 0.708 + 0.025 against 0.643 +
// 0.038 at 128.)
constexpr int kLegacyMaxSplits = 64;
static_assert(kLegacyMaxSplits <= kMaxSplits, "the ring GEMM's split bookkeeping is sized for kMaxSplits");
int choose_splits(int64_t n_tiles) {
    int64_t s = n_tiles / 24;
    if (s < 1) s = 1;
    if (s > kLegacyMaxSplits) s = kLegacyMaxSplits;
    return (int)s;
}

}  // namespace

extern "C" {

size_t nerf_hip_legacy_backward_scratch_bytes(int64_t n_rays, int32_t num_samples) {
    if (n_rays <= 0 || num_samples < 2) return 0;
    const int chunks = (num_samples + kSamplesPerWave - 1) / kSamplesPerWave;
    const LegacyTrainLayout L = make_legacy_train_layout(n_rays, chunks);
    return ((size_t)kLegacyMaxSplits * kLSlabFloats + (size_t)kMaxDataGrid * (kLGbFloats + 16) + (size_t)L.bwd_total) *
           sizeof(float);
}

int nerf_hip_legacy_render_backward(const NerfHipLegacyBackwardArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: null args");
    const NerfHipRenderArgs& a = args->fwd.render;
    if (args->grad == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: grad is null");
    if (a.n_rays == 0)      // empty batch (an empty data-parallel shard): the gradient is zero
        return nerf_common::check_hip(hipMemsetAsync(args->grad, 0, (size_t)kLegacyGradElements * sizeof(float),
                                                     (hipStream_t)stream), "legacy_render_backward memset");
    if (args->scratch == nullptr || args->d_rgb == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: scratch / d_rgb is null");
    if (a.train_workspace == nullptr || a.packed == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: forward was not a training forward");
    if (a.n_rays < 0 || a.num_samples < 2 || a.num_samples > 4096)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: n_rays / num_samples out of range");
    if (a.precision != NERF_HIP_PRECISION_FP32 && a.precision != NERF_HIP_PRECISION_F16X3)
        return nerf_common::fail(NERF_HIP_EINVAL, "legacy_render_backward: unknown precision");
    const bool half = a.precision == NERF_HIP_PRECISION_F16X3;          // the arithmetic of the training forward
    hipStream_t st = (hipStream_t)stream;

    LBwdArgs ba;
    ba.a = a;
    ba.a.color_outputs = 3;             // (the shared compositing backward reads d_rgb with this many columns)
    ba.a.reserved = 0;
    ba.d_rgb = args->d_rgb;
    ba.samples = a.num_samples;
    ba.chunks = (a.num_samples + kSamplesPerWave - 1) / kSamplesPerWave;
    ba.L = make_legacy_train_layout(a.n_rays, ba.chunks);
    ba.groups = ba.L.mp / 16 / kWavesPerWg;                 // (padded ray, chunk) items / 4 waves
    const int64_t slots = ba.L.mp / 16 / ba.chunks;         // padded rays
    ba.grad = args->grad;
    ba.n_tiles = ba.L.mp / kKs;
    ba.splits = choose_splits(ba.n_tiles);
    ba.tiles_per_split = (ba.n_tiles + ba.splits - 1) / ba.splits;
    ba.slabs = args->scratch;
    ba.gb_partial = args->scratch + (size_t)kLegacyMaxSplits * kLSlabFloats;
    ba.dymax = ba.gb_partial + (size_t)kMaxDataGrid * kLGbFloats;
    ba.rows = ba.dymax + (size_t)kMaxDataGrid * 16;         // dY / dL/d(out) rows: L.bwd_total floats (16-byte aligned)

    int device = 0, cus = 0;
    int rc = nerf_common::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (rc) return rc;
    rc = nerf_common::check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device),
                                "hipDeviceGetAttribute");
    if (rc) return rc;
    static unsigned done_data = 0, done_wgrad = 0, done_data_h = 0, done_wgrad_h = 0;
    rc = half ? nerf_common::ensure_dynamic_lds((const void*)nerf_legacy_bwd_data_h_kernel, kLBwdLdsBytes + 64, device,
                                                &done_data_h)
              : nerf_common::ensure_dynamic_lds((const void*)nerf_legacy_bwd_data_kernel, kLBwdLdsBytes, device, &done_data);
    if (rc) return rc;
    rc = half ? nerf_common::ensure_dynamic_lds((const void*)nerf_legacy_wgrad_h_kernel, kRingSlots * kRingSlotBytes,
                                                device, &done_wgrad_h)
              : nerf_common::ensure_dynamic_lds((const void*)nerf_legacy_wgrad_kernel, kRingSlots * kRingSlotBytes, device,
                                                &done_wgrad);
    if (rc) return rc;
    int64_t grid = (int64_t)cus * 2;
    if (grid > ba.groups) grid = ba.groups;
    if (grid > kMaxDataGrid) grid = kMaxDataGrid;
    ba.data_grid = (int)grid;

    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_COMPOSITE_BACKWARD);
        hipLaunchKernelGGL(nerf_legacy_composite_bwd_kernel, dim3((unsigned)((slots + kWavesPerWg - 1) / kWavesPerWg)),
                           dim3(256), 0, st, ba);
    }
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_DATA_GRADIENT);
        if (half) hipLaunchKernelGGL(nerf_legacy_bwd_data_h_kernel, dim3((unsigned)grid), dim3(256), kLBwdLdsBytes + 64, st, ba);
        else hipLaunchKernelGGL(nerf_legacy_bwd_data_kernel, dim3((unsigned)grid), dim3(256), kLBwdLdsBytes, st, ba);
    }
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_WEIGHT_GRADIENT);
        if (half) hipLaunchKernelGGL(nerf_legacy_wgrad_h_kernel, dim3(ba.splits * kWgradJobs), dim3(256),
                                     kRingSlots * kRingSlotBytes, st, ba);
        else hipLaunchKernelGGL(nerf_legacy_wgrad_kernel, dim3(ba.splits * kWgradJobs), dim3(256),
                                kRingSlots * kRingSlotBytes, st, ba);
    }
    {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_REDUCE);
        hipLaunchKernelGGL(nerf_legacy_grad_reduce_kernel, dim3(kReduceDirectBlocks + kReduceGbBlocks),
                           dim3(kReduceThreads), 0, st, ba);
    }
    return nerf_common::check_hip(hipGetLastError(), "legacy_render_backward launch");
}

}  // extern "C"

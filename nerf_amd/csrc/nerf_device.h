// Device-side building blocks shared by the forward and backward kernels (gfx950 only):
// the LDS-DMA weight pipe, fp32 MFMA layer loops, DPP row reductions/scans, and the unfused
// fp32 front end (ray, fenceposts, frustum Gaussian, integrated positional encoding).
#ifndef NERF_DEVICE_H
#define NERF_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerf_hip.h"
#include "nerf_layout.h"
#include "nerf_common.h"

// wave priorities: a wave in a VALU phase (encoding, LayerNorm, compositing) above a wave inside an
// MFMA loop (the other orders were measured and lose: NOTES.md section 7)
#define NERF_PRIO_MFMA 0
#define NERF_PRIO_VALU 2

namespace nerf_device {

using namespace nerf_layout;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 3;                        // LDS ring slots (one 16 KiB stage each)
// The "small" image (bias, gamma, beta per layer, [g][T][r] arrays) in LDS: the four lane groups
// read the same tile of their own 256-byte row at once, and 256 bytes apart is the same four LDS
// banks (every read a 2-way conflict: 14 % of the LDS-active cycles of the render kernel).  In LDS
// the rows are therefore 272 bytes apart (one extra 16-byte column: four banks further per lane
// group); the global image keeps the dense layout.  LDS map of a workgroup: weight ring first (its
// LDS-DMA targets then stay below 48 KiB), the small image behind it.
constexpr int kSmallGStride = 68;               // floats between the lane groups' rows in LDS
constexpr int kSmallArrayLds = 4 * kSmallGStride;
constexpr int kSmallPerLayerLds = 3 * kSmallArrayLds;
constexpr int kSmallLdsFloats = 5 * kSmallPerLayerLds + kOutPad;
constexpr int kSmallLdsBytes = 16640;           // >= 4 * kSmallLdsFloats, multiple of 128
static_assert(kSmallLdsBytes >= 4 * kSmallLdsFloats, "small image does not fit");
constexpr int kRingBytes = kRing * kStageBytes;
constexpr int kWavesPerWg = 4;
constexpr int kSamplesPerWave = 16;

// index of element i of the packed (dense) small image in its padded LDS copy
__host__ __device__ inline int small_lds_index(int i) {
    if (i >= 5 * kSmallPerLayer) return 5 * kSmallPerLayerLds + (i - 5 * kSmallPerLayer);
    const int L = i / kSmallPerLayer, rem = i % kSmallPerLayer;
    const int which = rem / kHidden, q = rem % kHidden;
    return L * kSmallPerLayerLds + which * kSmallArrayLds + (q / 64) * kSmallGStride + (q % 64);
}
// global -> LDS copy by the whole workgroup (256 threads)
template <int kThreads = 256>
__device__ __forceinline__ void stage_small_image(const float* small_g, float* small_l) {
    for (int i = threadIdx.x; i < kSmallFloats; i += kThreads) small_l[small_lds_index(i)] = small_g[i];
}

// ---------------------------------------------------------------------------------------------
// saved-for-backward workspace (training forward writes it, backward reads/extends it)
// padded sample index sp = (ray * chunks + c) * 16 + j ; "tile" tensors are stored exactly in
// register order [chunk][T][lane][r] (1 KiB per store instruction), "row" tensors as
// [sp][feature] (what the weight-gradient GEMM stages through LDS).  The post-ReLU activations
// x = relu(gamma * x_hat + beta) are NOT saved: the weight-gradient kernel rebuilds them from
// x_hat while it reads its operands (two VALU ops per operand instead of 1 KiB per sample and
// layer of HBM writes in the forward).
// ---------------------------------------------------------------------------------------------
struct TrainLayout {
    int64_t mp;                 // padded samples = ceil4(n_rays) * chunks * 16
    int64_t h;                  // row  [mp, 96]   encoded inputs, kernel column order
    int64_t dy[5];              // row  [mp, 256]  grad wrt pre-LayerNorm output of layer L
    int64_t dy5;                // row  [mp, 64]   grad wrt padded network output
    int64_t xhat[5];            // row  [mp, 256]  normalised pre-affine activations of layer L
    int64_t rstd[5];            // [mp]
    int64_t out;                // tile [mp, 64]   padded network output
    int64_t comp;               // [mp, 4]         alpha, T_exclusive, dist, density(+noise)
    int64_t total;              // floats
};

// `width`: features per saved x_hat / dY row — 256, or 128 when a narrow network (hidden_size <= 128) trains at its
// own cost (both arithmetics; nerf_layout.h: Narrow<8>): half the bytes of the step's dominant tensors.  The
// workspace the caller allocates is always sized for 256 (nerf_hip_train_workspace_bytes has no shape argument).
__host__ __device__ inline TrainLayout make_train_layout(int64_t n_rays, int chunks, int width = kHidden) {
    TrainLayout t;
    const int64_t rays4 = (n_rays + 3) / 4 * 4;
    t.mp = rays4 * chunks * 16;
    int64_t off = 0;
    t.h = off; off += t.mp * kEncIn;
    for (int i = 0; i < 5; ++i) { t.dy[i] = off; off += t.mp * width; }
    t.dy5 = off; off += t.mp * kOutPad;
    for (int i = 0; i < 5; ++i) { t.xhat[i] = off; off += t.mp * width; }
    for (int i = 0; i < 5; ++i) { t.rstd[i] = off; off += t.mp; }
    t.out = off; off += t.mp * kOutPad;
    t.comp = off; off += t.mp * 4;
    t.total = off;
    return t;
}
// register tiles per sample the TRAINING kernels of a launch run at: 8 for a narrow network (hidden_size <= 128;
// <= 64 trains at 8 too — the weight gradient's 2 x 2 wave grid needs 4 x 4 accumulator tiles), else 16
__host__ __device__ inline int train_tiles(int hidden) { return hidden <= 128 ? 8 : 16; }      // (both arithmetics)
// ... and the register tiles the training forward and the data gradient COMPUTE at: 4 for hidden_size <= 64 in fp32
// arithmetic (a quarter of the MFMAs; tiles 4 .. 7 of the 128-wide saved rows are left unwritten: the weight gradient
// of such a network fetches tiles 0 .. 3 only and multiplies the others' places, zeroed once, in its ring slots —
// nerf_backward.hip: nerf_wgrad_n4_kernel), else train_tiles
__host__ __device__ inline int train_compute_tiles(int hidden, bool half) { return !half && hidden <= 64 ? 4 : train_tiles(hidden); }

// Saved 256-wide rows (x_hat of every hidden layer, dY of every layer; both networks) are TILE-MAJOR: the
// [16 samples][256 features] tile of a wave is stored as its 16 register tiles T, 1 KiB each — so one vector-memory
// instruction (a fixed T: lane (j, g) moves features 16 T + 4 g .. + 3 of sample j) touches ONE contiguous KiB
// instead of sixteen 64-byte segments 1 KiB apart.  Same bytes, same 1 KiB LDS-DMA pieces for the weight gradient
// (a 16-sample k-step is one 16 KiB tile either way) — but a CU's vector-memory path moves the contiguous form about
// twice as fast when it, not HBM, is what a burst waits for (round 4: a wave's 33-operation burst 1.9 against 3.9 us
// on an idle chip; the split-precision data gradient 0.74 against 0.86 ms).
// Inside a register tile the order is the LANE order of the wave that owns it — lane 16 g + j (sample j, features
// 4 g .. 4 g + 3) holds the 16 bytes at 16 * lane: [g][sample][4 features].  Which lane writes which 16 bytes of the
// KiB is free, and it matters: the vector-memory path looks up four consecutive lanes at a time, and with sample-major
// chunks ([sample][g]: a lane quad spans two 128-byte lines) the split-precision data gradient ran 0.74 ms, with the
// quad's chunks 256 bytes apart 0.85 ms — hardly better than row-major rows (0.86 ms, a quad over four lines 1 KiB
// apart).  The weight gradient's LDS-DMA re-orders the chunks of a piece on the way in (nerf_backward_common.h:
// SlotLayout), so its LDS layout does not constrain this one.
//   element (sample s of the tile, feature f) at  tile * 4096 + (f >> 4) * 256 + ((f & 15) >> 2) * 64 + s * 4 + (f & 3)
constexpr int kTileFloats = 16 * kHidden;       // one 16-sample tile of a 256-wide row tensor
constexpr int kTileT = 256;                     // floats between two register tiles T of a lane
__host__ __device__ inline int tile_lane_word(int s, int g) { return g * 64 + s * 4; }
__host__ __device__ inline int64_t tile_lane_base(int64_t sp, int g, int tile_floats = kTileFloats) {   // this lane's f32x4 of register tile 0
    return (sp >> 4) * tile_floats + tile_lane_word((int)(sp & 15), g);     // (tile_floats = 16 x row width)
}

// A lane's constant 32-bit offset behind an optimisation barrier, taken at every use: with a wave-uniform 64-bit base
// (scalar registers) + this offset a saved row needs no per-lane 64-bit pointer — otherwise loop-invariant code
// motion folds base + offset + tensor offset into one such pointer per saved tensor and parks them all across the
// layers (the legacy training forward: 30 spilled address registers; nerf_backward.hip uses the same form).
__device__ __forceinline__ uint32_t lane_offset(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
// ... of this lane's f32x4 inside a tile-major 16-sample tile (nerf_device.h: tile_lane_word(j, g)) and inside a
// [16] per-sample statistic, from the thread id alone (three integer operations per use, nothing kept)
__device__ __forceinline__ uint32_t row_lane_offset() {
    const uint32_t l = lane_offset(threadIdx.x);
    return ((l >> 4) & 3u) * 64u + (l & 15u) * 4u;
}
__device__ __forceinline__ uint32_t stat_lane_offset() { return lane_offset(threadIdx.x) & 15u; }

// The network's shape from an argument block (0 = the defaults 256 / 96), and what the LayerNorms need of it:
// they divide their sums by hidden_size, not by the 256 features the kernels carry — the padded ones are exactly
// zero before normalisation (nerf_layout.h: Shape) — and the second pass of the variance is written so that they add exactly nothing (nerf_fused.h).
__host__ __device__ inline Shape shape_of(const NerfHipRenderArgs& a) {
    return Shape{a.hidden > 0 ? a.hidden : kHidden, a.enc_inputs > 0 ? a.enc_inputs : kEncIn, a.num_outputs,
                 a.color_outputs > 0 ? a.color_outputs : 3};
}
struct NormDivisor {
    float inv_n;                // 1 / hidden_size (features beyond it are padding: exactly 0 before normalisation)
    int32_t real;               // hidden_size: feature f of a sample is real for f < real (the cold variance pass masks by it)
};
__host__ __device__ inline NormDivisor norm_divisor(int hidden) {
    return NormDivisor{1.0f / (float)hidden, hidden};
}
constexpr NormDivisor kFullWidth = {1.0f / 256.0f, 256};

// ---------------------------------------------------------------------------------------------
// weight stream: global -> LDS by LDS-DMA, two stages ahead of the MFMAs
// ---------------------------------------------------------------------------------------------
// kDepth: ring slots.  3 (every full-width kernel): the DMA of stage s + 2 is issued when stage s opens, the hand-over
// of stage s leaves its 4 pieces in flight.  2 (the narrow inference kernels, which then fit THREE workgroups of 48 KiB
// on a CU): the DMA of stage s + 1 is issued when stage s opens and has that stage's MFMAs to land — nothing of it may
// still fly at the next hand-over (vmcnt(0 + younger)).
// kSkippable: the stream can leave out ONE stage of the image per pass (skip_stage, set by the kernel before the first
// issue(); -1: none) — the 4-tile kernels run layer 0 of a network with few encoding scales from one stage of two.
template <int kStagesInImage, int kDepth = 3, bool kSkippable = false>
struct WeightPipe {
    static constexpr int kRingDepth = kDepth;
    int skip_stage;             // (kSkippable only)
    static constexpr int kPrioMfma = NERF_PRIO_MFMA, kPrioValu = NERF_PRIO_VALU;
    const char* blob;           // packed image, stage 0
    char* ring;                 // LDS ring base
    int issue_stage;            // next stage of the image to issue (cyclic)
    int issue_slot;             // ring slot it goes to
    int read_slot;              // ring slot of the stage being consumed
    int wave;                   // wave id in the workgroup (uniform)
    int lane;

    __device__ __forceinline__ void init(const void* image, char* lds_ring, int w, int l) {
        blob = (const char*)image;
        ring = lds_ring;
        issue_stage = issue_slot = read_slot = 0;
        skip_stage = -1;
        wave = w;
        lane = l;
    }

    // Unconditional: past the last stage a workgroup needs, the (cyclic) image is simply fetched
    // again into slots nobody reads — two wasted 16 KiB DMAs per workgroup at kernel end buy a
    // branch-free stage loop.  The kernel drains vmcnt before it exits.
    __device__ __forceinline__ void issue() {
        {
            // one address + one M0 base, the four 1 KiB pieces by the instruction's immediate
            // offset (it applies to the global and the LDS address alike).
            // Written as inline asm on purpose: once the compiler sees an LDS-DMA builtin in a
            // function, its wait-count pass stops counting LDS reads and drains them all
            // (s_waitcnt lgkmcnt(0)) before every use, which defeats issuing ds_reads ahead of
            // the MFMAs that consume them.  The DMA's own completion is tracked by hand anyway
            // (open_stage: vmcnt(4) + barrier); DMA ops the compiler does not know about only make
            // its own vmcnt waits more conservative, never too weak (vector memory returns in order).
            // Asm gets none of the compiler's hazard wait states: the s_nop covers "SGPR written by
            // a VALU (v_readlane of a spilled SGPR, v_readfirstlane) -> read by VMEM" (5) and
            // "m0 written -> LDS-DMA" (1); scripts/isa_hazards.py checks R2 / R3 on the emitted code.
            const uint32_t dst = (uint32_t)(uintptr_t)(ring + issue_slot * kStageBytes + wave * 4096);
            uint32_t m0_saved;
            // scalar base (uniform: image + stage + wave) + one 32-bit lane offset
            const char* base = blob + (size_t)issue_stage * kStageBytes + wave * 4096;
            const uint64_t base_u = (uint64_t)(uintptr_t)base;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
            const uint64_t sbase = ((uint64_t)hi << 32) | lo;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 2\n\t"             /* 5 wait states between any earlier VALU-written SGPR and the loads */
                "global_load_lds_dwordx4 %1, %3\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(m0_saved)
                : "v"(lane * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase)
                : "memory");
        }
        issue_stage = (issue_stage + 1 == kStagesInImage) ? 0 : issue_stage + 1;
        if (kSkippable && issue_stage == skip_stage) ++issue_stage;       // (never the image's last stage)
        issue_slot = (issue_slot + 1 == kDepth) ? 0 : issue_slot + 1;
    }

    // Top of a stage: this wave's DMA pieces of the stage have landed (the 4 youngest = the next
    // stage may still fly; any other younger vector-memory op only makes the wait conservative)
    // and its LDS reads have all returned (lgkmcnt(0): free in the MFMA loops, whose hand-over
    // sits behind the wait that retired the last reads of the old stage); every wave has passed
    // the barrier, so (a) all 16 pieces are visible and (b) every wave has RETIRED its reads of the
    // slot the next issue() overwrites: write-after-read safe without any timing argument.
    // The caller reads its first operands, THEN calls issue(): the reads' latency hides under the
    // previous stage's trailing MFMAs instead of behind the DMA address arithmetic.
    // kYounger: vector-memory operations (x_hat stores of the split-precision training forward) that
    // this wave is KNOWN to have issued after the DMA of the stage being opened, besides the 4 pieces
    // of the following stage.  vmcnt counts loads, stores and LDS-DMA together and retires them in
    // issue order (checked on the hardware: scripts/probes/vmcnt_order.hip), so without it the wait
    // also covers those stores (an HBM write latency per stage, which a 0.4 us split-precision stage
    // cannot hide); an under-count only makes the wait stricter, an over-count would let the stage be
    // read before it has landed.  WHICH operations are younger depends on the ring depth: with 3 slots the DMA of the
    // stage being opened was issued two hand-overs ago (the stores of the previous stage and of this one are younger),
    // with 2 slots one hand-over ago (only this stage's are) — callers count with Pipe::kRingDepth.  The wait carries
    // its ring depth as an assembler comment, which nerf_amd/isa_scan.py (rule R6) reads back from the emitted code.
    template <int kYounger = 0>
    __device__ __forceinline__ const f32x4* open_stage() {
        static_assert(kYounger >= 0 && 4 + kYounger <= 63, "vmcnt immediate");
        static_assert(kDepth == 3 || kDepth == 2, "ring depth");
        asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0) ; nerf_ring_depth=%c1" ::"n"((kDepth == 3 ? 4 : 0) + kYounger), "n"(kDepth) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const f32x4* p = (const f32x4*)(ring + read_slot * kStageBytes) + lane;
        read_slot = (read_slot + 1 == kDepth) ? 0 : read_slot + 1;
        return p;
    }
    __device__ __forceinline__ void prefetch_next() {
        asm volatile("" ::: "memory");
        issue();
    }
};

// The same interface over a weight image that is RESIDENT in LDS: a network of hidden_size <= 64 is 7 stages = 112 KiB
// (nerf_layout.h: Narrow<4>), so one workgroup of kWaves waves per CU loads it ONCE and every stage "opens" as a pointer
// into it — no DMA in the loop, no vmcnt wait, no barrier: the waves of the workgroup never meet again, each walks its
// own rays.  Worth 3 % on the hidden-64 frame (47.4 -> 45.9 ms), not more: that kernel's time is its MFMA cycles PLUS
// its VALU cycles (the exact-fp32 MFMA and the VALU exclude each other on this chip — scripts/probes/
// mfma32_valu_coexec.hip, NOTES.md section R6d), which no arrangement of waves changes.
template <int kStagesInImage, int kWaves>
struct ResidentPipe {
    static constexpr int kRingDepth = 0;          // (no ring: nothing for isa_scan rule R6 to count)
    // (measured: MFMA 0 / VALU 2 as in the ring kernels 45.8 ms on the hidden-64 frame, every other order 48.1)
    static constexpr int kPrioMfma = NERF_PRIO_MFMA, kPrioValu = NERF_PRIO_VALU;
    char* image;                // LDS copy of the packed image
    int read_stage;             // stage the next open_stage() hands out (cyclic)
    int skip_stage;             // a stage every pass leaves out (-1: none; WeightPipe: kSkippable)
    int lane;

    // Called by every wave of the workgroup; ends in a barrier.  4 KiB chunk c of the image (4 DMA pieces, as
    // WeightPipe::issue) by wave c % kWaves.
    __device__ __forceinline__ void init(const void* packed_image, char* lds, int w, int l) {
        image = lds;
        read_stage = 0;
        skip_stage = -1;
        lane = l;
        constexpr int kChunks = kStagesInImage * kStageBytes / 4096;
#pragma unroll 1
        for (int c = w; c < kChunks; c += kWaves) {
            const uint32_t dst = (uint32_t)(uintptr_t)(lds + c * 4096);
            const uint64_t base_u = (uint64_t)(uintptr_t)((const char*)packed_image + (size_t)c * 4096);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
            const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
            const uint64_t sbase = ((uint64_t)hi << 32) | lo;
            uint32_t m0_saved;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 2\n\t"
                "global_load_lds_dwordx4 %1, %3\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(m0_saved)
                : "v"(l * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase)
                : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __device__ __forceinline__ void issue() {}
    template <int kYounger = 0>
    __device__ __forceinline__ const f32x4* open_stage() {
        const f32x4* p = (const f32x4*)(image + read_stage * kStageBytes) + lane;
        read_stage = (read_stage + 1 == kStagesInImage) ? 0 : read_stage + 1;
        if (read_stage == skip_stage) ++read_stage;
        return p;
    }
    __device__ __forceinline__ void prefetch_next() {}
};

// relu(x) as an INTEGER maximum on the bits (a negative float is a negative integer, a positive one itself; -0.0 -> +0.0
// like fmax): one instruction.  fmaxf on a raw MFMA result costs two — the compiler canonicalises the operand first
// (v_max_f32 x, x, x: it cannot know the accumulator holds no signalling NaN), and so does every other floating-point
// way of writing it (v_med3, compare + select).
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct NoHook {
    __device__ __forceinline__ void operator()(int) const {}
};

// A 16-out-tile layer of KT stages (k-groups), one 16 KiB stage each: stage t holds, for every
// out tile T, the four A operands of k-group t; B operands are act[4 t .. 4 t + 3].
//
// Software pipeline over "groups" of two out tiles (8 MFMAs, 256 cycles): the two ds_read_b128 of
// group G+1 are issued right after the first MFMA of group G — i.e. right after the wait that
// retired group G's own reads — so every LDS read has a full group to land and the only
// lgkmcnt(0) in the loop never sees a freshly issued read.  When group G+1 opens a new stage, the
// stage hand-over (vmcnt wait, barrier, DMA issue for stage +2) happens at that same point, with
// 7 MFMAs of the old stage still to issue behind it.  At that barrier this wave has completed all
// its reads of the finished stage (they were retired by the wait above), so the DMA that is issued
// next may overwrite that ring slot: write-after-read safe by construction.
// hook(t) runs once per stage, after the stage's barrier.
template <int KT, class Pipe, class Hook = NoHook>
__device__ __forceinline__ void layer_wide(Pipe& pipe, f32x4 (&acc)[16], const float (&act)[64],
                                           Hook hook = Hook()) {
    f32x4 a[2][2];
    // Priority 0 inside the MFMA loop, 2 outside it: a wave in a VALU phase (encoding, LayerNorm,
    // compositing) wins issue arbitration against its SIMD partner's MFMA stream, finishes the
    // phase sooner and returns to feeding the matrix pipe (+0.7 % measured; MFMAs lose nothing,
    // they need one issue slot per 32 cycles).
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const f32x4* st = pipe.open_stage();
    a[0][0] = st[0];
    a[0][1] = st[64];
    pipe.prefetch_next();
    hook(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const float b0 = act[4 * t], b1 = act[4 * t + 1], b2 = act[4 * t + 2], b3 = act[4 * t + 3];
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
            const int cur = tp & 1, nxt = cur ^ 1;          // 8 groups per stage: parity carries over
            const f32x4 a0 = a[cur][0], a1 = a[cur][1];
            acc[2 * tp] = mfma4(a0.x, b0, acc[2 * tp]);
            __builtin_amdgcn_sched_barrier(0);   // the wait for this group's operands is above this line
            if (tp < 7) {
                a[nxt][0] = st[(2 * tp + 2) * 64];
                a[nxt][1] = st[(2 * tp + 3) * 64];
            } else if (t + 1 < KT) {
                st = pipe.open_stage();
                a[nxt][0] = st[0];
                a[nxt][1] = st[64];
                pipe.prefetch_next();
                hook(t + 1);
            }
            __builtin_amdgcn_sched_barrier(0);   // reads stay HERE (the scheduler would sink them)
            acc[2 * tp + 1] = mfma4(a1.x, b0, acc[2 * tp + 1]);
            acc[2 * tp] = mfma4(a0.y, b1, acc[2 * tp]);
            acc[2 * tp + 1] = mfma4(a1.y, b1, acc[2 * tp + 1]);
            acc[2 * tp] = mfma4(a0.z, b2, acc[2 * tp]);
            acc[2 * tp + 1] = mfma4(a1.z, b2, acc[2 * tp + 1]);
            acc[2 * tp] = mfma4(a0.w, b3, acc[2 * tp]);
            acc[2 * tp + 1] = mfma4(a1.w, b3, acc[2 * tp + 1]);
            // keep groups apart: merged groups would re-issue the reads just before their use
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// The same pipeline for a NARROW product (nerf_layout.h: Narrow<NT>): NT accumulator tiles, KG k-groups, a stage =
// 16 quads = 16 / NT k-groups; group p (two quads, 8 MFMAs) works on k-group p / (NT / 2) and the tile pair
// 2 (p % (NT / 2)).  layer_wide is this function at NT = 16 (kept as its own copy: the full-width kernels' code
// does not move).  hook(s) runs once per stage, after the stage's barrier.
template <int NT, int KG, class Pipe, class Hook = NoHook>
__device__ __forceinline__ void layer_wide_n(Pipe& pipe, f32x4 (&acc)[16], const float (&act)[64], Hook hook = Hook()) {
    constexpr int kPerK = NT / 2, kGroups = KG * kPerK;
    static_assert(kGroups % 8 == 0, "a product ends on a stage boundary");
    f32x4 a[2][2];
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const f32x4* st = pipe.open_stage();
    a[0][0] = st[0];
    a[0][1] = st[64];
    pipe.prefetch_next();
    hook(0);
#pragma unroll
    for (int p = 0; p < kGroups; ++p) {
        const int k = p / kPerK, lp = p % kPerK, tp = p % 8;
        const int T0 = 2 * lp, T1 = 2 * lp + 1;
        const float b0 = act[4 * k], b1 = act[4 * k + 1], b2 = act[4 * k + 2], b3 = act[4 * k + 3];
        const int cur = p & 1, nxt = cur ^ 1;
        const f32x4 a0 = a[cur][0], a1 = a[cur][1];
        acc[T0] = mfma4(a0.x, b0, acc[T0]);
        __builtin_amdgcn_sched_barrier(0);
        if (tp < 7) {
            a[nxt][0] = st[(2 * tp + 2) * 64];
            a[nxt][1] = st[(2 * tp + 3) * 64];
        } else if (p + 1 < kGroups) {
            st = pipe.open_stage();
            a[nxt][0] = st[0];
            a[nxt][1] = st[64];
            pipe.prefetch_next();
            hook((p + 1) / 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[T1] = mfma4(a1.x, b0, acc[T1]);
        acc[T0] = mfma4(a0.y, b1, acc[T0]);
        acc[T1] = mfma4(a1.y, b1, acc[T1]);
        acc[T0] = mfma4(a0.z, b2, acc[T0]);
        acc[T1] = mfma4(a1.z, b2, acc[T1]);
        acc[T0] = mfma4(a0.w, b3, acc[T0]);
        acc[T1] = mfma4(a1.w, b3, acc[T1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// Same for a short run of stages whose B operands are f32x4 values (backward, layer 5).
template <int KT, class Pipe, class Hook = NoHook>
__device__ __forceinline__ void layer_wide_v4(Pipe& pipe, f32x4 (&acc)[16], const f32x4 (&bv)[KT],
                                              Hook hook = Hook()) {
    float act[64];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        act[4 * t] = bv[t].x;
        act[4 * t + 1] = bv[t].y;
        act[4 * t + 2] = bv[t].z;
        act[4 * t + 3] = bv[t].w;
    }
    layer_wide<KT>(pipe, acc, act, hook);
}

// ---------------------------------------------------------------------------------------------
// split-precision ("f16x3") pieces shared by the forward and the data-gradient kernels
// ---------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 q2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_h(const h8& a, const h8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ h2 pack_rtz(float a, float b) {
    return __builtin_bit_cast(h2, (q2)__builtin_amdgcn_cvt_pkrtz(a, b));
}
// x - (float)pair[kHigh] in one instruction (v_fma_mix_f32 reads the f16 half directly; the compiler
// itself only emits v_cvt_f32_f16 + v_sub_f32 for this).
// Inline asm gets NONE of the wait states hipcc inserts around MFMAs (its hazard recognizer does not
// look inside asm), so the instruction may only touch registers that a compiler-visible VALU
// instruction wrote last (scripts/isa_hazards.py, rule R1): the result is tied to `x`'s register
// ("+v": x itself if it dies here, else a v_mov copy — either way written by a visible VALU after
// every MFMA that used the register), never a fresh temporary, which the allocator is free to take
// from the accumulators of MFMAs still in flight.
template <int kHigh>
__device__ __forceinline__ float residual(float x, const h2& pair) {
    float r = x;
    if (kHigh)
        asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pair));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pair));
    return r;
}

// (v | .) half of split8: one register tile -> its four hi and four lo halfs
__device__ __forceinline__ void split4(const f32x4& v, h2& hi0, h2& hi1, h2& lo0, h2& lo1) {
    hi0 = pack_rtz(v.x, v.y);
    hi1 = pack_rtz(v.z, v.w);
    lo0 = pack_rtz(residual<0>(v.x, hi0), residual<1>(v.y, hi0));
    lo1 = pack_rtz(residual<0>(v.z, hi1), residual<1>(v.w, hi1));
}
__device__ __forceinline__ h8 join8(const h2& a, const h2& b, const h2& c, const h2& d) {
    return h8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// (v0 | v1) -> hi, lo with hi + lo = v to ~22 bits.  Round-toward-zero never overflows to inf.
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, h8& hi, h8& lo) {
    h2 nh[4], nl[4];
    split4(v0, nh[0], nh[1], nl[0], nl[1]);
    split4(v1, nh[2], nh[3], nl[2], nl[3]);
    hi = join8(nh[0], nh[1], nh[2], nh[3]);
    lo = join8(nl[0], nl[1], nl[2], nl[3]);
}

// "1 MFMA, then `valu` VALU instructions", twice (the tail of a unit)
template <int kValu>
__device__ __forceinline__ void interleave_2() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);
    }
}

// A-operand register sets of the unit pipeline: a unit = one (out tile, k block) pair = two
// ds_read_b128 (hi slab, lo slab) and three MFMAs (48 cycles); the reads of unit U + kSets - 1 are
// issued right after the first MFMA of unit U, so an LDS read has kSets - 1 units to land (the
// compiler's counted lgkmcnt waits leave the younger reads in flight).  A stage's hand-over
// (vmcnt wait, barrier, DMA issue) therefore sits kSets - 1 units before the stage's first MFMA;
// at that barrier every wave has issued AND retired (lgkmcnt(0)) all reads of the stage it is
// still computing on, whose slot the DMA issued next overwrites.
constexpr int kSets = 4;       // 2: +2 % frame time; 3 and 5 defeat the unroller (dynamic register indexing)


// The data gradient's split-precision layer: acc[T] += sum over KB k blocks of 32, B operands
// already split into f16 pairs (bhi / blo: block m = the caller's register tiles 2m, 2m + 1).
// Image and stage order as in the forward's layer_fused_h: stage (half, m) holds out tiles
// 8 half .. 8 half + 7 of k block m as {hi slab, lo slab} pairs; same unit pipeline (kSets operand
// sets, hand-over kSets - 1 units ahead).  hook(t) runs once per stage right after its hand-over.
// kEntryYounger: vector-memory operations the caller is KNOWN to have issued after the DMA of this
// layer's stage 1 (saves of the LayerNorm backward, prefetches of the next x_hat tile): stages 0 and
// 1 were issued before them, so their counted waits leave those operations in flight.
// NT = 8 (a network that trains at 8 register tiles): one half, kStages = KB.
template <int KB, int kEntryYounger, int NT = 16, class Pipe, class Hook = NoHook>
__device__ __forceinline__ void layer_wide_h(Pipe& pipe, f32x4 (&acc)[16], const h8 (&bhi)[KB],
                                             const h8 (&blo)[KB], Hook hook = Hook()) {
    constexpr int kStages = (NT / 8) * KB, kUnits = 8 * kStages;
    h8 ah[kSets], al[kSets];
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = (const h8*)pipe.template open_stage<kEntryYounger>();
#pragma unroll
    for (int u = 0; u < kSets - 1; ++u) {
        ah[u] = st[(2 * u) * 64];
        al[u] = st[(2 * u + 1) * 64];
    }
    pipe.prefetch_next();
    hook(0);
#pragma unroll
    for (int s = 0; s < kStages; ++s) {
        const int half = s / KB, m = s % KB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int U = 8 * s + i, set = U % kSets;
            const int T = 8 * half + i;
            acc[T] = mfma_h(ah[set], bhi[m], acc[T]);
            __builtin_amdgcn_sched_barrier(0);
            if (U + kSets - 1 < kUnits) {
                const int ip = (i + kSets - 1) % 8, pset = (U + kSets - 1) % kSets;
                if (ip == 0) {
                    if (s == 0) st = (const h8*)pipe.template open_stage<kEntryYounger>();
                    else st = (const h8*)pipe.open_stage();
                }
                ah[pset] = st[(2 * ip) * 64];
                al[pset] = st[(2 * ip + 1) * 64];
                if (ip == 0) {
                    pipe.prefetch_next();
                    hook(s + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[T] = mfma_h(ah[set], blo[m], acc[T]);
            acc[T] = mfma_h(al[set], bhi[m], acc[T]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// ---------------------------------------------------------------------------------------------
// cross-lane: lane = 16 g + j ; rows of 16 lanes (one DPP row) are the samples, g the feature group
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float old, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                           CTRL, 0xf, 0xf, false));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;

// sum / max over the 16 lanes of a row, result in every lane of the row
__device__ __forceinline__ float row_sum(float v) {
    v += dpp<kQuadXor1>(0.f, v);
    v += dpp<kQuadXor2>(0.f, v);
    v += dpp<kRowHalfMirror>(0.f, v);
    v += dpp<kRowMirror>(0.f, v);
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = __builtin_fmaxf(v, dpp<kQuadXor1>(v, v));
    v = __builtin_fmaxf(v, dpp<kQuadXor2>(v, v));
    v = __builtin_fmaxf(v, dpp<kRowHalfMirror>(v, v));
    v = __builtin_fmaxf(v, dpp<kRowMirror>(v, v));
    return v;
}
// inclusive prefix product along the row (lane j gets x_0 * ... * x_j)
__device__ __forceinline__ float row_prefix_prod(float v) {
    v *= dpp<0x111>(1.0f, v);
    v *= dpp<0x112>(1.0f, v);
    v *= dpp<0x114>(1.0f, v);
    v *= dpp<0x118>(1.0f, v);
    return v;
}
// inclusive suffix sum along the row (lane j gets x_j + ... + x_15)
__device__ __forceinline__ float row_suffix_sum(float v) {
    v += dpp<0x101>(0.f, v);
    v += dpp<0x102>(0.f, v);
    v += dpp<0x104>(0.f, v);
    v += dpp<0x108>(0.f, v);
    return v;
}
// Reduce-scatter over the 16 lanes of a row, eight quantities at a time: from per-tile values v[T][i]
// (T = 0..15 tiles, i = 0..7 quantities, one register each) lane j of every row ends with the sums over the
// row's lanes of v[T = j][i].  Butterfly: distance 8 (lanes 0-7 keep tile t, lanes 8-15 tile t + 8:
// scatter_level8), distance 4 (banks 0, 2 keep tile t, banks 1, 3 tile t + 4 of their half: scatter_level4),
// then an all-reduce inside the quad, of which the lane with t = lane-in-quad keeps the result (scatter_take):
// per value 1 + 0.5 + 0.5 DPP adds and a quarter of a select, against 4 DPP adds + a select for a row_sum() of which one lane in
// sixteen keeps the result.  The levels can be applied as soon as their two inputs exist, so a caller that
// walks the tiles in the order t, t + 8, t + 4, t + 12 never holds more than a few tiles' values.
// (scripts/probes/row_scatter_sum.hip checks the lane semantics on the hardware.)
// The two halves of a level are bank-masked DPP adds into one register, which the compiler does not
// generate: inline asm.  Asm gets no hazard wait states from the compiler: `s_nop 1` covers "VALU write ->
// DPP read of the same VGPR" (2 wait states), and the inputs must not be raw results of MFMAs still in
// flight (nerf_amd/isa_scan.py rule R1) — callers pass values a visible VALU instruction produced.
#define NERF_DPP8(ctrl, mask, first)                                                               \
    "v_add_f32_dpp %0, %" #first ", %" #first " " ctrl " row_mask:0xf bank_mask:" mask "\n\t"
__device__ __forceinline__ void scatter_level8(const float (&lo)[8], const float (&hi)[8], float (&out)[8]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]),
          "=&v"(out[7])
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]),
          "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]));
}
// banks 0, 2 take `lo` from the lane 4 above (row_shl), banks 1, 3 `hi` from the lane 4 below (row_shr)
__device__ __forceinline__ void scatter_level4(const float (&lo)[8], const float (&hi)[8], float (&out)[8]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %13, %13 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %14, %14 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %15, %15 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %16, %16 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %17, %17 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %18, %18 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %19, %19 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %20, %20 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %21, %21 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %22, %22 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %23, %23 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]),
          "=&v"(out[7])
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]),
          "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]));
}
#undef NERF_DPP8
// The same level for FOUR values (the second half of a 16-value butterfly): out[i] = lo[i] + (lo[i] of the lane 4
// above) in banks 0, 2 and hi[i] + (hi[i] of the lane 4 below) in banks 1, 3.
__device__ __forceinline__ void scatter_level4(const float (&lo)[4], const float (&hi)[4], float (&out)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}
// Sum over the 16 lanes of a row of SIXTEEN per-lane values, scattered: lane j of every row gets the row's sum of
// v[j] — 16 + 8 + 8 DPP adds and 4 selects, against 64 DPP adds and 32 selects for sixteen row_sum()s of which one
// lane each keeps the result.  (Inputs: results of compiler-visible VALU instructions, isa_scan rule R1.)
__device__ __forceinline__ float row_scatter_sum16(const float (&v)[16], int lane) {
    float lo[8], hi[8], h8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = v[i], hi[i] = v[8 + i];
    scatter_level8(lo, hi, h8);          // lanes 0-7: v[i] of lanes j, j + 8; lanes 8-15: v[8 + i]
    const float a4[4] = {h8[0], h8[1], h8[2], h8[3]}, b4[4] = {h8[4], h8[5], h8[6], h8[7]};
    float q[4];
    scatter_level4(a4, b4, q);           // lane quads 0, 2: v[.. + i], quads 1, 3: v[.. + 4 + i]
    float kept = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float x = q[t];
        x += dpp<kQuadXor1>(0.f, x);
        x += dpp<kQuadXor2>(0.f, x);
        kept = (lane & 3) == t ? x : kept;
    }
    return kept;
}
// x[i]: the level-4 results of tile group t (tiles t, t + 4, t + 8, t + 12) -> all-reduce inside the quad; the
// lane whose position in its quad is t keeps them: after t = 0..3, kept[i] = the row's sum of tile
// 4 bank + (lane & 3) = tile (lane & 15)
__device__ __forceinline__ void scatter_take(float (&x)[8], int t, int lane, float (&kept)[8]) {
    const bool mine = (lane & 3) == t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] += dpp<kQuadXor1>(0.f, x[i]);
        x[i] += dpp<kQuadXor2>(0.f, x[i]);
        kept[i] = mine ? x[i] : kept[i];
    }
}
__device__ __forceinline__ float row_shift_up(float fill, float v) { return dpp<0x111>(fill, v); }    // from j-1
__device__ __forceinline__ float row_shift_down(float fill, float v) { return dpp<0x101>(fill, v); }  // from j+1

// Reductions over the 4 lane groups of a sample (lanes j, j+16, j+32, j+48) with the gfx950 row
// swaps instead of ds_bpermute: v_permlane16_swap exchanges the odd rows of its first operand with
// the even rows of its second, v_permlane32_swap the upper half with the lower half, so with both
// operands = v the two results hold (even-row value, odd-row value) resp. (lower-half value,
// upper-half value) in every lane — no LDS round trip in the exposed LayerNorm / softmax chains.
// (Inline asm: hipcc 7.2 miscompiles the sum of the two results of
// __builtin_amdgcn_permlane16_swap / _permlane32_swap into r0 + r0.  The s_nop covers the
// VALU-write -> permlane-swap wait states the compiler inserts for the builtin.)
__device__ __forceinline__ void rows_16(float v, float& even, float& odd) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    even = a;
    odd = b;
}
__device__ __forceinline__ void halves_32(float v, float& lower, float& upper) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lower = a;
    upper = b;
}
__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }
__device__ __forceinline__ float group_sum(float v) {      // over the 4 lane groups of a sample
    float a, b;
    rows_16(v, a, b);
    v = a + b;
    halves_32(v, a, b);
    return a + b;
}
__device__ __forceinline__ float group_max(float v) {
    float a, b;
    rows_16(v, a, b);
    v = __builtin_fmaxf(a, b);
    halves_32(v, a, b);
    return __builtin_fmaxf(a, b);
}

// Output slot n = 16 T + 4 g + reg of the padded last layer (nerf_layout.h: row_of_slot): 0 density, colors in
// registers 1..3 of tile 0 (three per lane group), the segmentation classes in the remaining slots up to the
// network's row count, the rest padding (nerf/model.py:591-592).  Tile 0's class slots come as a 16-bit mask the
// host derived for the launch (NerfHipRenderArgs.reserved in the library's own copy of the block: class_mask_tile0);
// beyond tile 0 no color lives, and a slot is a class iff it is below the row count.
__device__ __forceinline__ bool is_seg_slot(int T, int g, int reg, const NerfHipRenderArgs& a) {
    const int n = 16 * T + 4 * g + reg;
    return T == 0 ? ((a.reserved >> n) & 1) != 0 : n < a.num_outputs;
}
// what a launch's own copy of the argument block carries beyond the caller's: the normalised color count and the
// class mask of output tile 0
__host__ inline void derive_slot_constants(NerfHipRenderArgs& a) {
    const Shape s = shape_of(a);
    a.color_outputs = s.colors;
    a.reserved = class_mask_tile0(s);
}

// ---------------------------------------------------------------------------------------------
// front end: ray, fenceposts, Gaussian, IPE  (all fp32, unfused like the reference's ATen ops)
// ---------------------------------------------------------------------------------------------
struct Ray {
    float o[3], d[3];
};

__device__ __forceinline__ Ray load_ray(const NerfHipRenderArgs& a, int64_t local) {
#pragma clang fp contract(off)
    Ray r;
    if (a.rays_o != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = a.rays_o[local * 3 + k];
            r.d[k] = a.rays_d[local * 3 + k];
        }
    } else {
        // nerf/model.py:271-278 (pixel grid, ij indexing) and :367 (R . ray, summed left to right)
        const int64_t gid = a.ray_begin + local;
        const int64_t hw = (int64_t)a.image_h * a.image_w;
        const int64_t b = gid / hw;
        const int64_t pix = gid - b * hw;
        const int row = (int)(pix / a.image_w), col = (int)(pix - (int64_t)row * a.image_w);
        const float x = ((float)col - 0.5f * (float)(a.image_w - 1)) / a.focal_length;
        const float y = ((float)row - 0.5f * (float)(a.image_h - 1)) / a.focal_length;
        const float c0 = x, c1 = -y, c2 = -1.0f;
        const float* R = a.camera_r + b * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.d[k] = (R[3 * k] * c0 + R[3 * k + 1] * c1) + R[3 * k + 2] * c2;
            r.o[k] = a.camera_o[b * 3 + k];
        }
    }
    return r;
}

// Fencepost s of a ray (nerf/model.py:414-435), s clamped to the table.
// The launch's Philox offset: the argument block's, plus the device-resident word a graph-replayed launch is
// told apart by (include/nerf_hip.h: rng_counter).  A uniform address: one scalar load, only on drawing paths.
__device__ __forceinline__ uint64_t rng_offset_of(const NerfHipRenderArgs& a) {
    return a.rng_counter != nullptr ? a.rng_offset + *a.rng_counter : a.rng_offset;
}

__device__ __forceinline__ float fencepost(const NerfHipRenderArgs& a, int64_t local, int s) {
#pragma clang fp contract(off)
    const int S = a.num_samples;
    s = s < S - 1 ? s : S - 1;
    if (a.t_values != nullptr) return a.t_values[local * S + s];
    const float cur = a.t_table[s];
    float t = cur;
    const bool draw = (a.rng_mode & 1) != 0;
    if (a.u != nullptr || draw) {
        const float lower = s == 0 ? cur : 0.5f * (cur + a.t_table[s - 1]);
        const float upper = s == S - 1 ? cur : 0.5f * (a.t_table[s + 1] + cur);
        const float uu = a.u != nullptr ? a.u[local * S + s]
                                        : nerf_rng::uniform(a.rng_seed, rng_offset_of(a),
                                                            (uint64_t)(a.ray_begin + local), (uint32_t)s, 0u);
        t = lower + (upper - lower) * uu;
    }
    return t * a.t_scale;
}

// Fenceposts s .. s + N - 1 at once: the same values as N calls of fencepost() (same operations, selects for
// its branches), but every global load — table entries, their neighbours, the caller's draws — is issued before
// the first one is waited for.  Called one after the other, the stratified path's conditional loads wait one by
// one, a dozen dependent round trips at the head of every chunk (and each wait also drains the weight DMA).
template <int N>
__device__ __forceinline__ void fencepost_run(const NerfHipRenderArgs& a, int64_t local, int s, float (&t)[N]) {
#pragma clang fp contract(off)
    const int S = a.num_samples;
    int at[N];
#pragma unroll
    for (int k = 0; k < N; ++k) at[k] = s + k < S - 1 ? s + k : S - 1;
    if (a.t_values != nullptr) {
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = a.t_values[local * S + at[k]];
        return;
    }
    const bool draw = (a.rng_mode & 1) != 0;
    if (a.u == nullptr && !draw) {
#pragma unroll
        for (int k = 0; k < N; ++k) t[k] = a.t_table[at[k]] * a.t_scale;
        return;
    }
    float cur[N], prev[N], next[N], uu[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        cur[k] = a.t_table[at[k]];
        prev[k] = a.t_table[at[k] > 0 ? at[k] - 1 : 0];
        next[k] = a.t_table[at[k] < S - 1 ? at[k] + 1 : S - 1];
    }
    if (a.u != nullptr) {
#pragma unroll
        for (int k = 0; k < N; ++k) uu[k] = a.u[local * S + at[k]];
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
            uu[k] = nerf_rng::uniform(a.rng_seed, rng_offset_of(a), (uint64_t)(a.ray_begin + local), (uint32_t)at[k], 0u);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float lower = at[k] == 0 ? cur[k] : 0.5f * (cur[k] + prev[k]);
        const float upper = at[k] == S - 1 ? cur[k] : 0.5f * (next[k] + cur[k]);
        t[k] = (lower + (upper - lower) * uu[k]) * a.t_scale;
    }
}

struct Gaussian {
    float mean[3], cov[3];
};

// conical_frustum_to_gaussian(stable) + lift_gaussian(diag) + origin shift.
__device__ __forceinline__ Gaussian frustum(const Ray& r, float t0, float t1, float base_radius_sq) {
#pragma clang fp contract(off)
    const float c415 = (float)(4.0 / 15.0), c512 = (float)(5.0 / 12.0);
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    const float denom = 3.0f * mu2 + hw2;
    const float t_mean = mu + (2.0f * mu * hw2) / denom;
    const float t_var = hw2 / 3.0f - c415 * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
    const float r_var = base_radius_sq * ((mu2 / 4.0f + c512 * hw2) - (c415 * hw4) / denom);
    const float d0 = r.d[0] * r.d[0], d1 = r.d[1] * r.d[1], d2 = r.d[2] * r.d[2];
    const float mag = __builtin_fmaxf((d0 + d1) + d2, 1e-10f);
    const float dsq[3] = {d0, d1, d2};
    Gaussian g;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.mean[k] = r.d[k] * t_mean + r.o[k];
        g.cov[k] = t_var * dsq[k] + r_var * (1.0f - dsq[k] / mag);
    }
    return g;
}

// sin(y) for |y| up to ~2e5 rad (the encoding's largest scale times the far plane) to ~1 ulp
// (max |error| 1.3e-7 against fp64 over +-2e5): half-turn reduction n = rint(y / pi),
// r = y - n pi by two FMAs against pi split into two fp32 terms (Cody-Waite; each FMA rounds
// once, and the first one cancels exactly the leading bits, so r carries ~1e-7 absolute error —
// the same as the fp64 reduction it replaces, measured on 4e6 arguments; a third term of pi would add
// n 3.4e-15, 2e-10 at the top of the range), then an odd degree-9 minimax polynomial on [-pi/2, pi/2]
// (4.6e-9 fit error, 1.0e-7 evaluated in fp32 — the degree-11 one: 1.2e-7 —, as r + r^3 s(r^2) so the
// leading term is exact) and the (-1)^n sign.  13 issue slots against ~100+ for the general-purpose sinf
// with its Payne-Hanek path; every one of them is frame time in the fp32 kernels (NOTES.md section R6d).
// No fp64 anywhere in the kernels (the 24 reductions per sample used to be v_cvt / v_mul / v_rndne /
// v_fma _f64, quarter rate).
__device__ __forceinline__ float sin_reduced(float y) {
    const float n = __builtin_rintf(y * 0.318309886f);
    float r = __builtin_fmaf(-n, 3.1415927410125732f, y);
    r = __builtin_fmaf(-n, -8.742277657347586e-08f, r);
    // (-1)^n as a sign bit for an exclusive-or: a shift and a v_xor for the v_and, v_cmp and v_cndmask of a select
    const uint32_t sign = (uint32_t)(int)n << 31;
    const float u = r * r;
    float s = __builtin_fmaf(u, 2.5999029276135843e-06f, -0.00019806546333711594f);
    s = __builtin_fmaf(u, s, 0.008333016186952591f);
    s = __builtin_fmaf(u, s, -0.16666656732559204f);
    const float p = __builtin_fmaf(r * u, s, r);
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, p) ^ sign);
}

// 24 encoded features of this lane group (layout: nerf_layout.h).
__device__ __forceinline__ void encode(const Gaussian& gs, int g, float (&act)[64]) {
#pragma clang fp contract(off)
    const float base = __builtin_ldexpf(1.0f, 4 * g - 4);       // 2^(4g-4): scales 4g..4g+3 of -4..11
    const float half_pi = 1.5707963267948966f;
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        const float scale = base * (float)(1 << (p / 3));
        const float y = gs.mean[p % 3] * scale;
        const float yv = gs.cov[p % 3] * (scale * scale);
        const float damp = __builtin_amdgcn_exp2f(yv * -0.7213475204444817f);      // exp(-yv / 2): v_exp_f32 on -yv / (2 ln 2), |error| < 1 ulp + 2e-8
        act[p] = damp * sin_reduced(y);
        act[12 + p] = damp * sin_reduced(y + half_pi);
    }
}
// The narrow kernels' form: `per` scales per lane group (nerf_layout.h: scales_per_group — the scales the network HAS,
// spread over the four lane groups), the pairs of the scales beyond them skipped wave-uniformly and left 0 (their
// columns of layer 0 are zero in the narrow images): a network of encoding_size 16 evaluates 6 of the 12 pairs.
__device__ __forceinline__ void encode_n(const Gaussian& gs, int g, int per, float (&act)[64]) {
#pragma clang fp contract(off)
    const float base = __builtin_ldexpf(1.0f, per * g - 4);     // 2^(per g - 4): scales per g .. of -4..11
    const float half_pi = 1.5707963267948966f;
#pragma unroll
    for (int local = 0; local < 4; ++local) {
        // (results in scalars that merge behind the branch: written into act[] inside it, the array stays in memory)
        float sn[3] = {0.f, 0.f, 0.f}, cs[3] = {0.f, 0.f, 0.f};
        if (local < per) {                                      // (per: a launch constant)
            const float scale = base * (float)(1 << local);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float y = gs.mean[c] * scale;
                const float yv = gs.cov[c] * (scale * scale);
                const float damp = __builtin_amdgcn_exp2f(yv * -0.7213475204444817f);      // exp(-yv / 2): v_exp_f32 on -yv / (2 ln 2), |error| < 1 ulp + 2e-8
                sn[c] = damp * sin_reduced(y);
                cs[c] = damp * sin_reduced(y + half_pi);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            act[3 * local + c] = sn[c];
            act[12 + 3 * local + c] = cs[c];
        }
    }
}

// distance between the means of consecutive Gaussians (nerf/model.py:462-464)
__device__ __forceinline__ float mean_distance(const Gaussian& a, const Gaussian& b) {
#pragma clang fp contract(off)
    const float e0 = b.mean[0] - a.mean[0], e1 = b.mean[1] - a.mean[1], e2 = b.mean[2] - a.mean[2];
    return __builtin_sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
}


// ---------------------------------------------------------------------------------------------
// compositing of one 16-sample chunk (nerf/model.py:438-469, :660-663), shared by the fused
// inference kernel and the stand-alone compositing kernel of the training path
// ---------------------------------------------------------------------------------------------
constexpr float kSegBias = 100.0f;      // exponent bias of the segmentation sums (composite_chunk)
struct RayAccum {
    float carry;                // prod (alpha_i + 1e-10) over the finished chunks of the ray
    float rgb0, rgb1, rgb2;
    // running log-sum-exp over the ray's samples of ONE output slot per lane: lane (j, g) owns
    // slot i = j of its lane group, i.e. output n = 16 (j >> 2) + 4 g + (j & 3); in base 2: the value is
    // (seg_m + log2(seg_s) - kSegBias) ln 2 (composite_chunk)
    float seg_m, seg_s;
    __device__ __forceinline__ void reset() {
        carry = 1.0f;
        rgb0 = rgb1 = rgb2 = 0.f;
        seg_m = -__builtin_inff();
        seg_s = 0.f;
    }
};

// out[T][r] = padded network output n = 16 T + 4 g + r of sample s = 16 c + j (accumulator
// layout); returns the compositing weight of the sample.  `comp` (training): where this sample's
// (alpha, T_exclusive, dist, density + noise) goes.
// kSeg = false: a network without segmentation classes (the legacy network) — no class state is kept or updated.
template <bool kSave, bool kSeg = true>
__device__ __forceinline__ float composite_chunk(const NerfHipRenderArgs& a, int P, int64_t local, int s,
                                                 bool ok, int lane, const f32x4 (&out)[4], float dist,
                                                 RayAccum& acc, float* comp) {
#pragma clang fp contract(off)
    const int j = lane & 15, g = lane >> 4;
    float dens;                                  // out[0].x of lane group 0, to all four groups
    {
        float even, odd, lower, upper;
        rows_16(out[0].x, even, odd);
        halves_32(even, lower, upper);
        dens = lower;
    }
    if (a.noise != nullptr) {
        if (ok) dens = dens + a.noise[local * P + s] * a.density_noise_std;
    } else if (a.rng_mode & 2) {
        dens = dens + nerf_rng::normal(a.rng_seed, rng_offset_of(a), (uint64_t)(a.ray_begin + local),
                                       (uint32_t)s, 1u) * a.density_noise_std;
    }
    // (exp as v_exp_f32 on x log2(e): the product's rounding adds |x| 6e-8 of relative error, 2e-8 absolute at most)
    const float alpha = ok ? __builtin_amdgcn_exp2f(__builtin_fmaxf(dens, 0.f) * dist * -1.4426950408889634f) : 1.0f;
    const float prod = row_prefix_prod(ok ? alpha + 1e-10f : 1.0f);
    const float t_excl = acc.carry * row_shift_up(1.0f, prod);
    const float w = ok ? (1.0f - alpha) * t_excl : 0.f;
    acc.carry = acc.carry * __shfl(prod, (lane & 48) | 15);
    if (kSave && g == 0) *(f32x4*)comp = f32x4{alpha, t_excl, dist, dens};

    // colors: registers y, z, w of tile 0 are channels 3 g, 3 g + 1, 3 g + 2 of this lane group (nerf_layout.h:
    // color_slot) — all of them for the reference's 3 channels on lane group 0, harmless where the network has none
    // (sigmoid: v_rcp_f32, 1 ulp, for the IEEE division's ten instructions)
    const float cr = w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(out[0].y * -1.4426950408889634f));
    const float cg = w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(out[0].z * -1.4426950408889634f));
    const float cb = w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(out[0].w * -1.4426950408889634f));
    acc.rgb0 += row_sum(cr);
    acc.rgb1 += row_sum(cg);
    acc.rgb2 += row_sum(cb);

    if (kSeg && a.seg != nullptr) {
        // seg[class] = logsumexp over the ray's samples of log(w + 1e-10) + log_softmax(class logits) (nerf/model.py:
        // 660-663), kept per owning lane as a running (max, sum) in base 2.  Every VALU instruction of this kernel is
        // frame time (nothing executes beside an fp32 MFMA: NOTES.md section R6d), so the chunk's part is written for
        // few of them: logits times log2(e) once, v_exp / v_log directly, ONE stabiliser for the whole chunk — B = the
        // largest log2(w + 1e-10) of its samples, an upper bound of every term since log-probabilities are <= 0 —
        // instead of a row maximum per class, and the 16 sums over the samples as one reduce-scatter butterfly.  A
        // term enters as 2^(v - B + kSegBias): with the bias a class's sum underflows only when its probability is
        // below 2^-220 in every sample of weight (a logit gap of 150), where the reference, which takes the maximum
        // per class, would still return that -150; nothing a network trained with this loss produces.
        constexpr float kLog2e = 1.4426950408889634f;
        float mv[16];                               // class logits in base 2, -inf in the slots that hold none
        float m = -__builtin_inff();
        // (is_seg_slot with the lane group behind an optimisation barrier: the sixteen tests depend on nothing but the
        //  lane and the launch, and hoisted out of the ray loop they live as sixteen lane masks in scalar registers —
        //  spilled to vector-register lanes and read back with two v_readlane per use)
        int g4 = 4 * g;
        asm volatile("" : "+v"(g4));
        const uint32_t bits0 = (uint32_t)a.reserved >> g4;      // tile 0: the class bits of this lane group's four slots
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool is_class = T == 0 ? ((bits0 >> r) & 1u) != 0 : 16 * T + r + g4 < a.num_outputs;
                mv[4 * T + r] = is_class ? out[T][r] * kLog2e : -__builtin_inff();
                m = __builtin_fmaxf(m, mv[4 * T + r]);
            }
        m = group_max(m);
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) z += __builtin_amdgcn_exp2f(mv[i] - m);
        z = group_sum(z);
        const float lw = ok ? __builtin_amdgcn_logf(w + 1e-10f) : -__builtin_inff();       // v_log_f32: log2
        const float bound = row_max(lw);            // lane 0 of a chunk is always valid: finite
        const float shift = ((lw - m) - __builtin_amdgcn_logf(z)) + (kSegBias - bound);    // -inf on a padded sample
        float e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(mv[i] + shift);
        const float cs = row_scatter_sum16(e, lane);       // lane j: slot j of its lane group, over the chunk's samples
        const float nm = __builtin_fmaxf(acc.seg_m, bound);
        acc.seg_s = acc.seg_s * __builtin_amdgcn_exp2f(acc.seg_m - nm) + cs * __builtin_amdgcn_exp2f(bound - nm);
        acc.seg_m = nm;
    }
    return w;
}

// one coalesced store instruction per output row
template <bool kSeg = true>
__device__ __forceinline__ void store_ray(const NerfHipRenderArgs& a, int64_t local, bool ray_ok, int lane,
                                          const RayAccum& acc) {
    // lane 0 of a lane group stores its three sums (a per-lane select of acc.rgb0/1/2 by lane id makes the compiler
    // index the accumulator struct dynamically, which pins it to a scratch-memory stack object that is then
    // re-written every chunk): channels 3 g .. 3 g + 2, those the network has
    if (!kSeg) {                 // the legacy network: three channels, lane group 0's sums (its kernels' code does not move)
        if (ray_ok && lane == 0) {
            a.rgb[local * 3 + 0] = acc.rgb0;
            a.rgb[local * 3 + 1] = acc.rgb1;
            a.rgb[local * 3 + 2] = acc.rgb2;
        }
    } else if (ray_ok && (lane & 15) == 0) {
        const int C = a.color_outputs, c0 = 3 * (lane >> 4);
        if (c0 < C) a.rgb[local * C + c0] = acc.rgb0;
        if (c0 + 1 < C) a.rgb[local * C + c0 + 1] = acc.rgb1;
        if (c0 + 2 < C) a.rgb[local * C + c0 + 2] = acc.rgb2;
    }
    if (kSeg && a.seg != nullptr) {
        // the wave's 64 lanes cover n = 0..63 once: the class values leave in one store
        const int j = lane & 15, g = lane >> 4;
        const float mine = ((acc.seg_m + __builtin_amdgcn_logf(acc.seg_s)) - kSegBias) * 0.6931471805599453f;
        const int n = 16 * (j >> 2) + 4 * g + (j & 3);
        const Shape sh = shape_of(a);
        const int row = row_of_slot(n, sh);
        if (ray_ok && row > sh.colors) a.seg[local * sh.classes() + (row - 1 - sh.colors)] = mine;
    }
}

// Compositing of a training forward as a kernel body of its own (one wave per ray over the saved network
// outputs: `out_off` = padded outputs in tile format [tile][T][lane][r], `comp_off` = [sp][4] whose
// slot 2 the MLP kernel filled with the distance to the next sample): writes rgb / seg / out_weights
// and the compositing state (alpha, T_exclusive, dist, density + noise) the backward reads.
// P = samples per ray that are composited (S - 1 intervals, or S points for the legacy network).
__device__ __forceinline__ void composite_fwd_body(const NerfHipRenderArgs& a, int P, int chunks, int64_t out_off,
                                                   int64_t comp_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15;
    const int64_t local = (int64_t)blockIdx.x * kWavesPerWg + wave;
    if (local >= a.n_rays) return;
    float* const ws = a.train_workspace;
    RayAccum racc;
    racc.reset();
    // the chunks of a ray are a dependent chain (the transmittance carry), their loads are not: chunk c + 1 is
    // fetched while chunk c is composited (the loop was one exposed round trip to HBM per chunk)
    f32x4 next[4];
    float next_dist;
    auto fetch = [&](int c) {
        const int64_t tile = local * chunks + c;
        const float* otile = ws + out_off + tile * 1024 + lane * 4;
#pragma unroll
        for (int T = 0; T < 4; ++T) next[T] = *(const f32x4*)(otile + T * 256);
        next_dist = ws[comp_off + (tile * 16 + j) * 4 + 2];
    };
    fetch(0);
    for (int c = 0; c < chunks; ++c) {
        const int s = c * kSamplesPerWave + j;
        const bool ok = s < P;
        const int64_t sp = (local * chunks + c) * 16 + j;
        f32x4 out[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) out[T] = next[T];
        const float dist = next_dist;
        if (c + 1 < chunks) fetch(c + 1);
        float* comp = ws + comp_off + sp * 4;
        const float w = composite_chunk<true>(a, P, local, s, ok, lane, out, dist, racc, comp);
        if (a.out_weights != nullptr && ok && lane < 16) a.out_weights[local * P + s] = w;
    }
    store_ray(a, local, true, lane, racc);
}

}  // namespace nerf_device
#endif

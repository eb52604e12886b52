// Fused volume-render forward for gfx950 (MI355X): one persistent launch does
//   ray generation (nerf/model.py:243-278, :337-367)  ->  fenceposts (:369-435)
//   -> conical-frustum Gaussians (:33-45, :56-87, :112-136)
//   -> integrated positional encoding (:139-163, :24-30)
//   -> 6 x Linear with LayerNorm + ReLU on exact-fp32 MFMA (:525-542)
//   -> alpha compositing, RGB sum and segmentation log-sum-exp (:438-469, :660-663).
// Layout and the weight image are described in nerf_layout.h.  Written for gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nerf_device.h"
#include "nerf_fused.h"

using namespace nerf_layout;
using namespace nerf_device;
using namespace nerf_fused;

namespace {

constexpr int kLdsBytes = kRingBytes + kSmallLdsBytes;   // 64.25 KiB -> 2 workgroups / CU
// Split-precision kernel: per-lane state that only the front end and the compositing need (ray,
// running transmittance / RGB / segmentation sums, interval length) is parked in LDS while the MLP
// runs, instead of being spilled to scratch (= HBM writes) by the register allocator.
constexpr int kStashFloatsPerLane = 8;
constexpr int kStashBytes = kWavesPerWg * 64 * kStashFloatsPerLane * 4;   // 8 KiB
constexpr int kRayStashBytes = kWavesPerWg * 8 * 4;
constexpr int kLdsBytesHalf = kLdsBytes + kStashBytes + kRayStashBytes;   // 72.4 KiB -> still 2 / CU

struct KernelArgs {
    NerfHipRenderArgs a;
    int32_t intervals;          // P = S - 1
    int32_t chunks;             // ceil(P / 16)
    int64_t groups;             // ceil(n_rays / 4)
    TrainLayout save;           // offsets into a.train_workspace (training forward only)
    NormDivisor norm;           // 1 / hidden_size and the padded feature count of the LayerNorms (nerf_layout.h: Shape)
    int32_t enc_per;            // narrow kernels: encoding scales per lane group (nerf_layout.h: scales_per_group)
};

typedef WeightPipe<kNumStages> FwdPipe;

__device__ __forceinline__ void load_bias16(const float* small_l, int g, f32x4 (&acc)[16]) {
    const f32x4* b = (const f32x4*)(small_l + g * kSmallGStride);
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = b[T];
}

// One 16-out-tile layer: out += W . in over KT k-groups (the MFMA software pipeline of
// nerf_device.h: layer_wide), with `in` normalised lazily (kNormIn) and, in the last stage, the
// moments of `out` gathered and `out` copied back into `in` (also in the MFMA shadow): every
// layer then runs in -> out on the SAME two register tiles, so the four hidden layers share one
// instance of this code (two ping-ponged instances overflowed the instruction cache: +2 %).
template <int KT, bool kNormIn, bool kTrain>
__device__ __forceinline__ void layer_fused(FwdPipe& pipe, f32x4 (&in)[16], f32x4 (&out)[16],
                                            const LazyNorm& norm, Moments& mom) {
    if (kNormIn) normalize_tile<kTrain>(in[0], norm, 0);
    mom.reset();
    f32x4 a[2][2];
    f32x4 ga, be;               // gamma / beta of the tile being normalised next
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const f32x4* st = pipe.open_stage();
    a[0][0] = st[0];
    a[0][1] = st[64];
    pipe.prefetch_next();
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const float b0 = in[t].x, b1 = in[t].y, b2 = in[t].z, b3 = in[t].w;
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
            const int cur = tp & 1, nxt = cur ^ 1;
            const f32x4 a0 = a[cur][0], a1 = a[cur][1];
            out[2 * tp] = mfma4(a0.x, b0, out[2 * tp]);
            __builtin_amdgcn_sched_barrier(0);   // the wait for this group's operands is above this line
            if (tp < 7) {
                a[nxt][0] = st[(2 * tp + 2) * 64];
                a[nxt][1] = st[(2 * tp + 3) * 64];
                if (kNormIn && tp == 0 && t + 1 < KT) {      // a whole group ahead of their use
                    ga = norm.gam[t + 1];
                    be = norm.bet[t + 1];
                }
            } else if (t + 1 < KT) {
                // (training: the wait also covers this stage's and the previous stage's x_hat store.  Counting them out —
                //  open_stage<2>, as the split-precision layer does — was built and measured in round 6: forward / data
                //  gradient time ratio 0.971 against 0.965 - 0.985 before, i.e. nothing: an fp32 stage is 0.85 us of
                //  MFMAs, long enough for a store's acknowledgement)
                st = pipe.open_stage();
                a[nxt][0] = st[0];
                a[nxt][1] = st[64];
                pipe.prefetch_next();
            }
            __builtin_amdgcn_sched_barrier(0);   // reads stay HERE (the scheduler would sink them)
            out[2 * tp + 1] = mfma4(a1.x, b0, out[2 * tp + 1]);
            out[2 * tp] = mfma4(a0.y, b1, out[2 * tp]);
            out[2 * tp + 1] = mfma4(a1.y, b1, out[2 * tp + 1]);
            out[2 * tp] = mfma4(a0.z, b2, out[2 * tp]);
            out[2 * tp + 1] = mfma4(a1.z, b2, out[2 * tp + 1]);
            out[2 * tp] = mfma4(a0.w, b3, out[2 * tp]);
            out[2 * tp + 1] = mfma4(a1.w, b3, out[2 * tp + 1]);
            // VALU riding behind this group's MFMAs (same scheduling region; NOT dealt out over the MFMA gaps: nothing
            // executes beside an fp32 MFMA — NOTES.md section R6d — and bunched it costs fewer MFMA <-> VALU turn-arounds,
            // 360.4 against 361.3 ms per frame):
            if (kNormIn && tp == 1 && t + 1 < KT) {
                normalize_tile<kTrain>(in[t + 1], norm, t + 1, ga, be);
            }
            if (t == KT - 1 && tp >= 1) {         // tile pair finished one group ago
                mom.add(out[2 * tp - 2]);
                mom.add(out[2 * tp - 1]);
                in[2 * tp - 2] = out[2 * tp - 2];
                in[2 * tp - 1] = out[2 * tp - 1];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep groups apart (else reads re-issue just in time)
        }
    }
    mom.add(out[14]);
    mom.add(out[15]);
    in[14] = out[14];
    in[15] = out[15];
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// Layer 5 (256 -> 64 padded): 4 stages, each 4 k-groups x 4 out tiles; its input is normalised
// lazily like in layer_fused, one k-group (16 MFMAs) ahead.
template <bool kTrain>
__device__ __forceinline__ void layer_out(FwdPipe& pipe, f32x4 (&in)[16], f32x4 (&acc)[4],
                                          const LazyNorm& norm) {
    normalize_tile<kTrain>(in[0], norm, 0);
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
#pragma unroll
    for (int s = 0; s < kStagesL5; ++s) {
        const f32x4* st = pipe.open_stage();
        f32x4 q[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = st[i * 64];
        pipe.prefetch_next();
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int t = 4 * s + tl;
            if (tl + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) q[((tl + 1) & 1) * 4 + i] = st[(4 * (tl + 1) + i) * 64];
            }
            const f32x4 a0 = q[(tl & 1) * 4 + 0], a1 = q[(tl & 1) * 4 + 1], a2 = q[(tl & 1) * 4 + 2],
                        a3 = q[(tl & 1) * 4 + 3];
            const f32x4 b = in[t];
            acc[0] = mfma4(a0.x, b.x, acc[0]);
            acc[1] = mfma4(a1.x, b.x, acc[1]);
            acc[2] = mfma4(a2.x, b.x, acc[2]);
            acc[3] = mfma4(a3.x, b.x, acc[3]);
            acc[0] = mfma4(a0.y, b.y, acc[0]);
            acc[1] = mfma4(a1.y, b.y, acc[1]);
            acc[2] = mfma4(a2.y, b.y, acc[2]);
            acc[3] = mfma4(a3.y, b.y, acc[3]);
            acc[0] = mfma4(a0.z, b.z, acc[0]);
            acc[1] = mfma4(a1.z, b.z, acc[1]);
            acc[2] = mfma4(a2.z, b.z, acc[2]);
            acc[3] = mfma4(a3.z, b.z, acc[3]);
            acc[0] = mfma4(a0.w, b.w, acc[0]);
            acc[1] = mfma4(a1.w, b.w, acc[1]);
            acc[2] = mfma4(a2.w, b.w, acc[2]);
            acc[3] = mfma4(a3.w, b.w, acc[3]);
            if (t + 1 < 16) normalize_tile<kTrain>(in[t + 1], norm, t + 1);
        }
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// ---------------------------------------------------------------------------------------------
// NARROW instantiations (nerf_layout.h: Narrow<NT>, NT = 8 or 4 register tiles per sample): the same software
// pipeline — a stage = 8 groups of two quads = 64 MFMAs, next group's ds_read_b128 behind the first MFMA, stage
// hand-over at the last group — with NT / 2 groups per k-group instead of 8: group p of the layer works on k-group
// p / (NT / 2) and the out-tile pair 2 (p % (NT / 2)).  The lazy normalisation of the next k-group's tile and the
// moments of finished tile pairs ride in the MFMA shadow exactly as in layer_fused (which is this function at
// NT = 16; the full-width kernels keep their own copy so that their code does not move).
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void load_bias_n(const float* small_l, int g, f32x4 (&acc)[16]) {
    const f32x4* b = (const f32x4*)(small_l + g * kSmallGStride);
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] = b[T];
}

template <int NT, int KG, bool kNormIn, bool kTrain, class Pipe>
__device__ __forceinline__ void layer_fused_n(Pipe& pipe, f32x4 (&in)[16], f32x4 (&out)[16],
                                              const LazyNorm& norm, Moments& mom) {
    constexpr int kPerK = NT / 2;                 // groups (tile pairs) per k-group
    constexpr int kGroups = KG * kPerK;           // groups of the layer
    static_assert(kGroups % 8 == 0, "a layer ends on a stage boundary");
    if (kNormIn) normalize_tile<kTrain>(in[0], norm, 0);
    mom.reset();
    f32x4 a[2][2];
    f32x4 ga, be;
    __builtin_amdgcn_s_setprio(Pipe::kPrioMfma);
    const f32x4* st = pipe.open_stage();
    a[0][0] = st[0];
    a[0][1] = st[64];
    pipe.prefetch_next();
#pragma unroll
    for (int p = 0; p < kGroups; ++p) {
        const int k = p / kPerK, lp = p % kPerK, tp = p % 8;
        const int T0 = 2 * lp, T1 = 2 * lp + 1;
        const float b0 = in[k].x, b1 = in[k].y, b2 = in[k].z, b3 = in[k].w;
        const int cur = p & 1, nxt = cur ^ 1;
        const f32x4 a0 = a[cur][0], a1 = a[cur][1];
        out[T0] = mfma4(a0.x, b0, out[T0]);
        __builtin_amdgcn_sched_barrier(0);
        if (tp < 7) {
            a[nxt][0] = st[(2 * tp + 2) * 64];
            a[nxt][1] = st[(2 * tp + 3) * 64];
            if (kNormIn && lp == 0 && k + 1 < KG) {          // a whole group ahead of their use
                ga = norm.gam[k + 1];
                be = norm.bet[k + 1];
            }
        } else if (p + 1 < kGroups) {
            st = pipe.open_stage();
            a[nxt][0] = st[0];
            a[nxt][1] = st[64];
            pipe.prefetch_next();
        }
        __builtin_amdgcn_sched_barrier(0);
        out[T1] = mfma4(a1.x, b0, out[T1]);
        out[T0] = mfma4(a0.y, b1, out[T0]);
        out[T1] = mfma4(a1.y, b1, out[T1]);
        out[T0] = mfma4(a0.z, b2, out[T0]);
        out[T1] = mfma4(a1.z, b2, out[T1]);
        out[T0] = mfma4(a0.w, b3, out[T0]);
        out[T1] = mfma4(a1.w, b3, out[T1]);
        if (kNormIn && lp == 1 && k + 1 < KG) {
            normalize_tile<kTrain>(in[k + 1], norm, k + 1, ga, be);
        }
        if (k == KG - 1 && lp >= 1) {             // tile pair finished one group ago
            mom.add(out[2 * lp - 2]);
            mom.add(out[2 * lp - 1]);
            in[2 * lp - 2] = out[2 * lp - 2];
            in[2 * lp - 1] = out[2 * lp - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    mom.add(out[NT - 2]);
    mom.add(out[NT - 1]);
    in[NT - 2] = out[NT - 2];
    in[NT - 1] = out[NT - 1];
    __builtin_amdgcn_s_setprio(Pipe::kPrioValu);
}

// Layer 5 of a narrow network (16 NT -> 64 padded): NT / 4 stages of 4 k-groups x 4 out tiles.
template <int NT, bool kTrain, class Pipe>
__device__ __forceinline__ void layer_out_n(Pipe& pipe, f32x4 (&in)[16], f32x4 (&acc)[4], const LazyNorm& norm) {
    normalize_tile<kTrain>(in[0], norm, 0);
    __builtin_amdgcn_s_setprio(Pipe::kPrioMfma);
#pragma unroll
    for (int s = 0; s < NT / 4; ++s) {
        const f32x4* st = pipe.open_stage();
        f32x4 q[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = st[i * 64];
        pipe.prefetch_next();
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int t = 4 * s + tl;
            if (tl + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) q[((tl + 1) & 1) * 4 + i] = st[(4 * (tl + 1) + i) * 64];
            }
            const f32x4 a0 = q[(tl & 1) * 4 + 0], a1 = q[(tl & 1) * 4 + 1], a2 = q[(tl & 1) * 4 + 2],
                        a3 = q[(tl & 1) * 4 + 3];
            const f32x4 b = in[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = mfma4(a0[r], b[r], acc[0]);
                acc[1] = mfma4(a1[r], b[r], acc[1]);
                acc[2] = mfma4(a2[r], b[r], acc[2]);
                acc[3] = mfma4(a3[r], b[r], acc[3]);
            }
            if (t + 1 < NT) normalize_tile<kTrain>(in[t + 1], norm, t + 1);
        }
    }
    __builtin_amdgcn_s_setprio(Pipe::kPrioValu);
}


// Layer 5 (256 -> 64 padded) of the split-precision path: 4 stages of two k blocks x 4 out tiles;
// block m + 1 is built during the four units of block m.
template <bool kTrain, int NT = 16, class Pipe>
__device__ __forceinline__ void layer_out_h(Pipe& pipe, f32x4 (&in)[16], f32x4 (&acc)[4],
                                            const LazyNorm& norm) {
    constexpr int kStages5 = NT / 4, kBlocks = NT / 2;      // (k block, out tile) pairs: 8 to a stage
    constexpr int kUnits = 8 * kStages5;
    h8 bh[2], bl[2];
    normalize_tile<kTrain, kPackNorm>(in[0], norm, 0);
    normalize_tile<kTrain, kPackNorm>(in[1], norm, 1);
    split8(in[0], in[1], bh[0], bl[0]);
    h8 ah[kSets], al[kSets];
    f32x4 ga = norm.gam[2], be = norm.bet[2];
    h2 nh[4], nl[4];
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = (const h8*)pipe.open_stage();
#pragma unroll
    for (int u = 0; u < kSets - 1; ++u) {
        ah[u] = st[(2 * u) * 64];
        al[u] = st[(2 * u + 1) * 64];
    }
    pipe.prefetch_next();
#pragma unroll
    for (int s = 0; s < kStages5; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int U = 8 * s + i, set = U % kSets;
            const int m = 2 * s + (i >> 2), T = i & 3, q = i & 3;
            const int pb = m & 1;
            const int ta = 2 * m + 2, tb = 2 * m + 3;
            acc[T] = mfma_h(ah[set], bh[pb], acc[T]);
            __builtin_amdgcn_sched_barrier(0);
            if (U + kSets - 1 < kUnits) {
                const int ip = (i + kSets - 1) % 8, pset = (U + kSets - 1) % kSets;
                if (ip == 0) st = (const h8*)pipe.open_stage();
                ah[pset] = st[(2 * ip) * 64];
                al[pset] = st[(2 * ip + 1) * 64];
                if (ip == 0) pipe.prefetch_next();
            }
            if (q == 1 && m + 1 < kBlocks) {
                ga = norm.gam[tb];
                be = norm.bet[tb];
            }
            if (q == 3 && m + 2 < kBlocks) {
                ga = norm.gam[ta + 2];
                be = norm.bet[ta + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[T] = mfma_h(ah[set], bl[pb], acc[T]);
            acc[T] = mfma_h(al[set], bh[pb], acc[T]);
            if (m + 1 < kBlocks) {
                if (q == 0) normalize_tile<kTrain, kPackNorm>(in[ta], norm, ta, ga, be);
                if (q == 1) split4(in[ta], nh[0], nh[1], nl[0], nl[1]);
                if (q == 2) normalize_tile<kTrain, kPackNorm>(in[tb], norm, tb, ga, be);
                if (q == 3) {
                    split4(in[tb], nh[2], nh[3], nl[2], nl[3]);
                    bh[pb ^ 1] = join8(nh[0], nh[1], nh[2], nh[3]);
                    bl[pb ^ 1] = join8(nl[0], nl[1], nl[2], nl[3]);
                }
                interleave_2<4>();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// kPerSample: the instantiation that also writes the optional per-sample outputs (NeRF.forward's
// tensors, compositing weights for the hierarchical resampler, debug outputs); the render-only
// instantiations carry none of that code or its registers.
// Narrow kernels without per-sample outputs (NT < 16: render and training forward): a TWO-slot weight ring (48.25 KiB
// of LDS) and <= 168 registers, so that THREE workgroups share a CU — with a third of the MFMA work per chunk, the
// per-sample VALU phases (encoding, LayerNorm finishing, compositing) need a second partner to hide under.
template <bool kPerSample, int NT>
constexpr bool three_per_cu() { return NT < 16 && !kPerSample; }
// The render-only kernel at 4 register tiles keeps the WHOLE weight image in LDS (nerf_device.h: ResidentPipe): one
// workgroup of 16 waves per CU, 128.25 KiB, no barrier after the prologue; a wave owns a ray as before.
template <bool kTrain, bool kHalf, bool kPerSample, int NT>
constexpr bool resident_weights() { return NT == 4 && !kHalf && !kPerSample && !kTrain; }
constexpr int kResidentWaves = 16;
constexpr int kResidentLdsBytes = Narrow<4>::kStages * kStageBytes + kSmallLdsBytes;
static_assert(kResidentLdsBytes <= 160 * 1024, "the resident image and the small image share one CU's LDS");

template <bool kTrain, bool kHalf, bool kPerSample = false, int NT = 16>
__global__ __launch_bounds__((resident_weights<kTrain, kHalf, kPerSample, NT>() ? 64 * kResidentWaves : 256),
                             (resident_weights<kTrain, kHalf, kPerSample, NT>() ? 1 : three_per_cu<kPerSample, NT>() ? 3 : 2))
void nerf_render_fwd_kernel(const KernelArgs ka) {
    static_assert(!(kTrain && kPerSample), "the training forward has no per-sample outputs");
    static_assert(NT != 4 || !kHalf, "4 register tiles: fp32 arithmetic only (the split-precision arithmetic runs at 8 or 16)");
    // saved x_hat rows of a training forward: 128 wide for the narrow networks at EITHER compute width; a 4-tile forward
    // leaves tiles 4 .. 7 unwritten (its data and weight gradient read tiles 0 .. 3 only: nerf_wgrad_n4_kernel)
    constexpr int kSaveTiles = NT == 4 ? 8 : NT;
    typedef Narrow<NT> N;
    constexpr bool kResident = resident_weights<kTrain, kHalf, kPerSample, NT>();
    constexpr int kWaves = kResident ? kResidentWaves : kWavesPerWg;
    constexpr int kDepth = three_per_cu<kPerSample, NT>() ? 2 : 3;
    constexpr int kRingB = (kResident ? N::kStages : kDepth) * kStageBytes, kLdsB = kRingB + kSmallLdsBytes;
    // (the split-precision kernel's LDS stash of per-lane state exists for the 256-register full-width kernel; at
    //  8 register tiles the state stays in registers)
    constexpr bool kStash = kHalf && !kTrain && kDepth == 3;
    typedef typename std::conditional<kResident, ResidentPipe<N::kStages, kWaves>,
                                      WeightPipe<(kHalf && NT == 8 ? kNarrowH8Stages : N::kStages), kDepth, NT == 4>>::type Pipe;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ka.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int P = ka.intervals;
    const int chunks = ka.chunks;

    // small image -> LDS (once per workgroup)
    {
        stage_small_image<64 * kWaves>(a.packed + (kHalf ? kHSmallOffset : kBlobFloats), (float*)(smem + kRingB));
    }
    const float* small = (const float*)(smem + kRingB);

    Pipe pipe;
    pipe.init(a.packed + (kHalf ? (NT == 16 ? kHBlobOffset : kNarrowH8Offset)
                                : (NT == 16 ? 0 : (NT == 8 ? kNarrow8Offset : kNarrow4Offset))), smem, wave, lane);
    // 4 tiles, at most 8 encoding scales: layer 0 from ONE stage, the image's second stage left out (nerf_layout.h:
    // layer0_dense)
    const bool dense0 = NT == 4 && layer0_dense(4, ka.enc_per);
    if (NT == 4 && dense0) pipe.skip_stage = 1;
    if constexpr (!kResident) {
        pipe.issue();
        if (kDepth == 3) pipe.issue();
        __syncthreads();      // small image visible (this also drains the two DMA stages once)
    }                         // (resident: init() loaded the image and ended in the barrier)

    f32x4 X[16], Y[16];         // X: a layer's input tiles (B operands), Y: its accumulators (the first NT of them)
    float* const ws = a.train_workspace;

    // Inference: a wave owns a ray, walks its chunks in order and composites as it goes.
    // Training: compositing is a kernel of its own (nerf_composite_fwd_kernel), so the unit of
    // work here is one (ray, chunk) item per wave — a 512-ray batch then fills all 2,048 waves
    // instead of 512 of them — and the network outputs / distances are saved for it.
    for (int64_t grp = blockIdx.x; grp < ka.groups; grp += gridDim.x) {
        const int64_t unit = grp * kWaves + wave;
        const int64_t slot = kTrain ? unit / chunks : unit;      // padded ray slot (workspace rows)
        int64_t local = slot;
        const bool ray_ok = local < a.n_rays;
        if (!ray_ok) local = a.n_rays - 1;
        Ray ray = load_ray(a, local);
        RayAccum racc;
        racc.reset();
        float* const stash = (float*)(smem + kLdsB) + (wave * 64 + lane) * kStashFloatsPerLane;
        float* const ray_stash = (float*)(smem + kLdsB + kStashBytes) + wave * 8;
        if (kStash && lane == 0) {     // (a training item is one chunk: nothing to park)
            *(f32x4*)ray_stash = f32x4{ray.o[0], ray.o[1], ray.o[2], ray.d[0]};
            ray_stash[4] = ray.d[1];
            ray_stash[5] = ray.d[2];
        }
        if (kStash) asm volatile("" ::: "memory");    // the reads of the stash below stay below

        const int c_begin = kTrain ? (int)(unit - slot * chunks) : 0;
        const int c_end = kTrain ? c_begin + 1 : chunks;
        for (int c = c_begin; c < c_end; ++c) {
            const int s = c * kSamplesPerWave + j;
            const bool ok = s < P;
            const int64_t tile = slot * chunks + c;         // chunk index in the workspace
            const int64_t sp = tile * 16 + j;               // padded sample index
            if (kStash) {                         // the wave's ray, back from LDS (broadcast)
                const f32x4 r0 = *(const f32x4*)ray_stash;
                ray.o[0] = r0.x, ray.o[1] = r0.y, ray.o[2] = r0.z, ray.d[0] = r0.w;
                ray.d[1] = ray_stash[4], ray.d[2] = ray_stash[5];
            }
            float posts[3];
            fencepost_run<3>(a, local, s, posts);
            const float t0 = posts[0], t1 = posts[1], t2 = posts[2];
            const Gaussian gs = frustum(ray, t0, t1, a.base_radius_sq);
            float dist;
            {
                const Gaussian gn = frustum(ray, t1, t2, a.base_radius_sq);
                dist = s == P - 1 ? 1e10f : mean_distance(gs, gn);
            }
            {
                float feat[64];
                if constexpr (NT == 16) encode(gs, g, feat);
                else encode_n(gs, g, ka.enc_per, feat);
#pragma unroll
                for (int t = 0; t < kStagesL0; ++t)
                    X[t] = f32x4{feat[4 * t], feat[4 * t + 1], feat[4 * t + 2], feat[4 * t + 3]};
            }
            if (kTrain) {
                float* hrow = ws + ka.save.h + sp * kEncIn + 4 * g;
#pragma unroll
                for (int t = 0; t < kStagesL0; ++t) *(f32x4*)(hrow + 16 * t) = X[t];
            }
            float* const xrow = kTrain ? ws + tile_lane_base(sp, g, 256 * kSaveTiles) : nullptr;     // + ka.save.xhat[L] (tile-major)
            float* const rstd_p = kTrain ? ws + sp : nullptr;                      // + ka.save.rstd[L]

            LazyNorm norm;
            f32x4 out[4];
            if constexpr (kHalf) {
                HMoments mom;
                // split-precision MLP: X and Y swap roles layer by layer (no copy-back)
                const float eps = 1e-5f * (float)(1 << (kWScaleLog2 + kXScaleLog2)) *
                                  (float)(1 << (kWScaleLog2 + kXScaleLog2));
                // x_hat is scale-free; the saved 1/std is the one of the unscaled activations
                const float rs = (float)(1 << (kWScaleLog2 + kXScaleLog2));
                if (kStash) {
                    *(f32x4*)stash = f32x4{racc.carry, racc.rgb0, racc.rgb1, racc.rgb2};
                    *(f32x4*)(stash + 4) = f32x4{racc.seg_m, racc.seg_s, dist, 0.f};
                }
#pragma unroll
                for (int t = 0; t < kStagesL0; ++t) X[t] = X[t] * (float)(1 << kXScaleLog2);
                load_bias_n<NT>(small, g, Y);
                layer_fused_h<3, false, kTrain, kOrderNormRelu, NT>(pipe, X, Y, norm, mom);
                norm = finish_moments<kTrain, HMoments, NT>(mom, Y, small, g, xrow + ka.save.xhat[0],
                                                            rstd_p + ka.save.rstd[0], ka.norm, eps, rs);
#pragma unroll 1
                for (int L = 1; L <= 3; L += 2) {
                    const float* small_a = small + L * kSmallPerLayerLds;
                    load_bias_n<NT>(small_a, g, X);
                    layer_fused_h<NT / 2, true, kTrain, kOrderNormRelu, NT>(pipe, Y, X, norm, mom);
                    norm = finish_moments<kTrain, HMoments, NT>(mom, X, small_a, g, xrow + ka.save.xhat[L],
                                                                rstd_p + ka.save.rstd[L], ka.norm, eps, rs);
                    const float* small_b = small_a + kSmallPerLayerLds;
                    load_bias_n<NT>(small_b, g, Y);
                    layer_fused_h<NT / 2, true, kTrain, kOrderNormRelu, NT>(pipe, X, Y, norm, mom);
                    norm = finish_moments<kTrain, HMoments, NT>(mom, Y, small_b, g, xrow + ka.save.xhat[L + 1],
                                                                rstd_p + ka.save.rstd[L + 1], ka.norm, eps, rs);
                }
                {
                    const f32x4* b = (const f32x4*)(small + 5 * kSmallPerLayerLds) + g * 4;
#pragma unroll
                    for (int T = 0; T < 4; ++T) out[T] = b[T];
                }
                layer_out_h<kTrain, NT>(pipe, Y, out, norm);
#pragma unroll
                for (int T = 0; T < 4; ++T) out[T] = out[T] * (1.0f / rs);
                if (kStash) {
                    const f32x4 s0 = *(const f32x4*)stash, s1 = *(const f32x4*)(stash + 4);
                    racc.carry = s0.x, racc.rgb0 = s0.y, racc.rgb1 = s0.z, racc.rgb2 = s0.w;
                    racc.seg_m = s1.x, racc.seg_s = s1.y, dist = s1.z;
                }
            } else if constexpr (NT < 16) {
                // ---- a narrow network at its own cost (nerf_layout.h: Narrow<NT>) ----
                Moments mom;
                load_bias_n<NT>(small, g, Y);
                if (NT == 4 && dense0) {
                    // dense slots d = 4 t + r <- old slots (layer0_dense_source_slot): 0..5, then 12..17
                    X[1].z = X[3].x, X[1].w = X[3].y;
                    X[2] = f32x4{X[3].z, X[3].w, X[4].x, X[4].y};
                    X[3] = f32x4{0.f, 0.f, 0.f, 0.f};
                    layer_fused_n<NT, 4, false, kTrain>(pipe, X, Y, norm, mom);
                } else {
#pragma unroll
                    for (int t = kStagesL0; t < N::kKGroups0; ++t) X[t] = f32x4{0.f, 0.f, 0.f, 0.f};    // zero-padded k-groups
                    layer_fused_n<NT, N::kKGroups0, false, kTrain>(pipe, X, Y, norm, mom);
                }
                norm = finish_moments<kTrain, Moments, NT>(mom, X, small, g, xrow + ka.save.xhat[0], rstd_p + ka.save.rstd[0], ka.norm);
#pragma unroll 1
                for (int L = 1; L <= 4; ++L) {
                    const float* small_l = small + L * kSmallPerLayerLds;
                    load_bias_n<NT>(small_l, g, Y);
                    layer_fused_n<NT, NT, true, kTrain>(pipe, X, Y, norm, mom);
                    norm = finish_moments<kTrain, Moments, NT>(mom, X, small_l, g, xrow + ka.save.xhat[L],
                                                               rstd_p + ka.save.rstd[L], ka.norm);
                }
                {
                    const f32x4* b = (const f32x4*)(small + 5 * kSmallPerLayerLds) + g * 4;
#pragma unroll
                    for (int T = 0; T < 4; ++T) out[T] = b[T];
                }
                layer_out_n<NT, kTrain>(pipe, X, out, norm);
            } else {
            Moments mom;
            // ---- layer 0: 96 -> 256 ----
            load_bias16(small, g, Y);
            layer_fused<kStagesL0, false, kTrain>(pipe, X, Y, norm, mom);
            norm = finish_moments<kTrain, Moments>(mom, X, small, g, xrow + ka.save.xhat[0], rstd_p + ka.save.rstd[0], ka.norm);
            // ---- layers 1..4: 256 -> 256 ----
#pragma unroll 1
            for (int L = 1; L <= 4; ++L) {
                const float* small_l = small + L * kSmallPerLayerLds;
                load_bias16(small_l, g, Y);
                layer_fused<kStagesHidden, true, kTrain>(pipe, X, Y, norm, mom);
                norm = finish_moments<kTrain, Moments>(mom, X, small_l, g, xrow + ka.save.xhat[L],
                                              rstd_p + ka.save.rstd[L], ka.norm);
            }
            // ---- layer 5: 256 -> 54 (padded 64) ----
            {
                const f32x4* b = (const f32x4*)(small + 5 * kSmallPerLayerLds) + g * 4;
#pragma unroll
                for (int T = 0; T < 4; ++T) out[T] = b[T];
            }
            layer_out<kTrain>(pipe, X, out, norm);
            }
            if (kTrain) {
                float* otile = ws + ka.save.out + tile * 1024 + lane * 4;
#pragma unroll
                for (int T = 0; T < 4; ++T) *(f32x4*)(otile + T * 256) = out[T];
            }

            if (kTrain) {
                if (g == 0) *(f32x4*)(ws + ka.save.comp + sp * 4) = f32x4{0.f, 0.f, dist, 0.f};
                continue;
            }

            // ---- compositing (nerf/model.py:438-469, :660-663) ----
            const float w = composite_chunk<false>(a, P, local, s, ok, lane, out, dist, racc, nullptr);
            // optional per-sample outputs (NeRF.forward, nerf/model.py:553-594)
            if (kPerSample && ray_ok && ok) {
                const int64_t smp = local * P + s;
                if (a.out_raw != nullptr) {
                    // (the lane group behind an optimisation barrier: the columns of tile 0 depend on nothing but g
                    //  and the launch's constants, and hoisted out of the ray loop they live across the MLP as spills)
                    int gq = g;
                    asm volatile("" : "+v"(gq));
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (slot of the padded tile -> column: beyond tile 0 they coincide, nerf_layout.h)
                            const int n = 16 * T + 4 * g + r;
                            const int row = T == 0 ? row_of_tile0(gq, r, a.color_outputs, a.num_outputs) : (n < a.num_outputs ? n : -1);
                            if (row >= 0) a.out_raw[smp * a.num_outputs + row] = out[T][r];
                            if (T == 0) asm volatile("" : "+v"(gq) :: "memory");     // one column's address at a time
                        }
                }
                if (g == 0) {
                    if (a.out_mean != nullptr || a.out_cov != nullptr) {
                        // recomputed (same operations, same bits) rather than kept live across the MLP
                        const Gaussian gm = kHalf ? frustum(ray, fencepost(a, local, s), fencepost(a, local, s + 1),
                                                            a.base_radius_sq)
                                                  : gs;
                        if (a.out_mean != nullptr) {
                            a.out_mean[smp * 3 + 0] = gm.mean[0];
                            a.out_mean[smp * 3 + 1] = gm.mean[1];
                            a.out_mean[smp * 3 + 2] = gm.mean[2];
                        }
                        if (a.out_cov != nullptr) {
                            a.out_cov[smp * 3 + 0] = gm.cov[0];
                            a.out_cov[smp * 3 + 1] = gm.cov[1];
                            a.out_cov[smp * 3 + 2] = gm.cov[2];
                        }
                    }
                    if (a.out_weights != nullptr) a.out_weights[smp] = w;
                    if (a.out_t != nullptr) {           // recomputed: same operations, same bits
                        a.out_t[local * a.num_samples + s] = fencepost(a, local, s);
                        if (s == P - 1) a.out_t[local * a.num_samples + s + 1] = fencepost(a, local, s + 1);
                    }
                }
            }
        }
        if (!kTrain) store_ray(a, local, ray_ok, lane, racc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// Compositing of the training forward: one wave per ray over the saved network outputs.
__global__ __launch_bounds__(256) void nerf_composite_fwd_kernel(const KernelArgs ka) {
    composite_fwd_body(ka.a, ka.intervals, ka.chunks, ka.save.out, ka.save.comp);
}

// Per-sample outputs of a TRAINING forward (a differentiable NeRF.forward, nerf/model.py:553-594): the network
// outputs are already in the workspace (padded tiles), this copies the real ones out in the reference's
// [n_rays, S-1, num_outputs] layout; the thread of column 0 also recomputes the sample's Gaussian (the same
// operations as the render kernel's front end: same bits) for out_mean / out_cov.
__global__ __launch_bounds__(256) void nerf_field_outputs_kernel(const KernelArgs ka) {
    const NerfHipRenderArgs& a = ka.a;
    const int P = ka.intervals, n_out = a.num_outputs;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_rays * P * n_out) return;
    const int row = (int)(e % n_out), n = slot_of_row(row, shape_of(a));     // output row -> its slot of the padded tile
    const int64_t smp = e / n_out;
    const int64_t local = smp / P;
    const int s = (int)(smp - local * P);
    const int64_t tile = local * ka.chunks + (s >> 4);
    const int j = s & 15;
    if (a.out_raw != nullptr)
        a.out_raw[e] = a.train_workspace[ka.save.out + tile * 1024 + (n >> 4) * 256 + (((n & 15) >> 2) * 16 + j) * 4 + (n & 3)];
    if (row == 0 && (a.out_mean != nullptr || a.out_cov != nullptr)) {
        const Ray ray = load_ray(a, local);
        const Gaussian gm = frustum(ray, fencepost(a, local, s), fencepost(a, local, s + 1), a.base_radius_sq);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.out_mean != nullptr) a.out_mean[smp * 3 + k] = gm.mean[k];
            if (a.out_cov != nullptr) a.out_cov[smp * 3 + k] = gm.cov[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// parameter re-layout (state_dict order -> packed image)
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* p[NERF_HIP_NUM_PARAM_TENSORS];
    float* packed;
    int32_t n_out;              // rows of the last Linear (1 + colors + segmentation classes); padded to 64 with zeros
    int32_t colors;             // color channels: which slot of the padded tile a row lands in (nerf_layout.h: row_of_slot)
    __device__ __forceinline__ int row5(int slot) const { return row_of_slot(slot, Shape{hidden, enc_in, n_out, colors}); }
    int32_t hidden, enc_in;     // H and 6 x scales of the source tensors; the images are zero beyond them (nerf_layout.h)
    // element (out, in) of the source matrices, 0 in the padding: layer 0 by kernel slot (t, g, r), the hidden
    // layers L = 1..4, the last layer; element f of a per-feature vector (bias / gamma / beta: tensor index)
    // (per: scales per lane group of the image being written — 4 in the full-width images, scales_per_group in the narrow ones)
    __device__ __forceinline__ float w0(int out, int t, int g, int r, int per = 4) const {
        const int src = layer0_source_feature(t, g, r, enc_in / 6, per);
        return out < hidden && src >= 0 ? p[0][out * enc_in + src] : 0.f;
    }
    __device__ __forceinline__ float wh(int L, int out, int in) const {
        return out < hidden && in < hidden ? p[4 * L][out * hidden + in] : 0.f;
    }
    __device__ __forceinline__ float w5(int out, int in) const {
        const int row = row5(out);                  // `out`: slot of the padded output tile
        return row >= 0 && in < hidden ? p[20][row * hidden + in] : 0.f;
    }
    __device__ __forceinline__ float vec(int tensor, int f) const { return f < hidden ? p[tensor][f] : 0.f; }
};

__global__ void nerf_pack_kernel(const PackArgs pa) {
    if (blockIdx.x == 0) {
        // the bounds block (nerf_layout.h: kBoundsOffset; first, so that it runs beside the others): thread f = input
        // feature f of layer 1
        __shared__ float col[256], gam[256];
        const int f = threadIdx.x;
        // (loads at clamped addresses, masked afterwards: behind a bounds test per element the compiler emits a branch
        //  and a full wait per load — 256 dependent round trips, 50 us)
        const float* const w1 = pa.p[4];
        const int H = pa.hidden, fc = f < H ? f : H - 1;
        float c = 0.f;
        for (int o0 = 0; o0 < kHidden; o0 += 64) {
            float v[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int o = o0 + k < H ? o0 + k : H - 1;
                v[k] = w1[o * H + fc];
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) c += o0 + k < H ? __builtin_fabsf(v[k]) : 0.f;
        }
        if (f >= H) c = 0.f;
        col[f] = c;
        gam[f] = __builtin_fabsf(pa.vec(2, f));
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if (f < w) {
                col[f] = __builtin_fmaxf(col[f], col[f + w]);
                gam[f] = __builtin_fmaxf(gam[f], gam[f + w]);
            }
            __syncthreads();
        }
        if (f < 4) pa.packed[kBoundsOffset + f] = f == 0 ? 18.0f * 2097152.0f * 1.01f * gam[0] * col[0] : 0.f;
        return;
    }
    const int e = (blockIdx.x - 1) * blockDim.x + threadIdx.x;
    if (e >= kPackedFloats || (e >= kImageFloats && e < kWideFloats)) return;      // (the bounds block's four floats)
    float v = 0.f;
    if (e >= kNarrowBwd4Offset) {
        // transposed fp32 image at 4 register tiles (nerf_layout.h): hidden_size <= 64 training, fp32 arithmetic
        if (pa.hidden > 64) return;
        const int eb = e - kNarrowBwd4Offset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int i = lane & 15, g = lane >> 4;
        const int tout = quad / 4, tin = quad % 4;
        v = stage == 0 ? pa.w5(16 * tout + 4 * g + r, 16 * tin + i) : pa.wh(5 - stage, 16 * tout + 4 * g + r, 16 * tin + i);
    } else if (e >= kNarrowBwdH8Offset) {
        // transposed split-precision image at 8 register tiles (nerf_layout.h)
        if (pa.hidden > 128) return;
        const int eb = e - kNarrowBwdH8Offset;
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
            const int jj = 2 * word + k;
            const int m = stage < 2 ? stage : (stage - 2) % 4;
            const int out = 32 * m + 16 * (jj >> 2) + 4 * kg + (jj & 3);
            float w = stage < 2 ? pa.w5(out, 16 * pair + row) : pa.wh(4 - (stage - 2) / 4, out, 16 * pair + row);
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e >= kNarrowH8Offset) {
        // split-precision forward image at 8 register tiles (nerf_layout.h): two f16 of one slab per float slot
        if (pa.hidden > 128) return;
        const int eb = e - kNarrowH8Offset;
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
            const int jj = 2 * word + k;
            const int tl = jj >> 2, r = jj & 3;
            float w = 0.f;
            if (stage < 3) {
                w = pa.w0(16 * pair + row, 2 * stage + tl, kg, r, scales_per_group(pa.enc_in / 6));
            } else if (stage < 3 + 16) {
                const int L = 1 + (stage - 3) / 4, m = (stage - 3) % 4;
                w = pa.wh(L, 16 * pair + row, 32 * m + 16 * tl + 4 * kg + r);
            } else {
                const int m = 2 * (stage - 19) + (pair >> 2), T = pair & 3;
                w = pa.w5(16 * T + row, 32 * m + 16 * tl + 4 * kg + r);
            }
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e >= kNarrowBwd8Offset) {
        // transposed narrow image (nerf_layout.h): written for every network that can train at 8 register tiles
        if (pa.hidden > 128) return;
        const int eb = e - kNarrowBwd8Offset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int i = lane & 15, g = lane >> 4;
        if (stage < 2) {
            const int qq = stage * 16 + quad, tout = qq / 8, tin = qq % 8;
            v = pa.w5(16 * tout + 4 * g + r, 16 * tin + i);
        } else {
            const int L = 4 - (stage - 2) / 4;                       // 4, 3, 2, 1
            const int qq = ((stage - 2) % 4) * 16 + quad, tout = qq / 8, tin = qq % 8;
            v = pa.wh(L, 16 * tout + 4 * g + r, 16 * tin + i);
        }
    } else if (e >= kNarrow8Offset) {
        // the narrow fp32 images this network can run at (nerf_layout.h: Narrow<NT>): 8 register tiles for
        // hidden_size <= 128 (inference at 65 .. 128, training at <= 128), 4 for <= 64 (inference)
        const bool eight = e < kNarrow4Offset;
        const int nt = eight ? 8 : 4;
        if (pa.hidden > 16 * nt) return;
        const int eb = e - (eight ? kNarrow8Offset : kNarrow4Offset);
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int row = lane & 15, g = lane >> 4;
        const int s0 = eight ? Narrow<8>::kStages0 : Narrow<4>::kStages0;
        const int sh = eight ? Narrow<8>::kStagesHid : Narrow<4>::kStagesHid;
        if (stage < s0) {                                           // layer 0: quads numbered k-group * NT + out tile
            const int qq = stage * 16 + quad, k = qq / nt, T = qq % nt;
            const int per = scales_per_group(pa.enc_in / 6);
            if (layer0_dense(nt, per)) {                            // dense slots (nerf_layout.h): all of it in stage 0
                const int q = layer0_dense_source_slot(4 * k + r);
                v = q >= 0 ? pa.w0(16 * T + row, q / 4, g, q % 4, per) : 0.f;
            } else {
                v = k < kStagesL0 ? pa.w0(16 * T + row, k, g, r, per) : 0.f;
            }
        } else if (stage < s0 + 4 * sh) {                           // layers 1..4
            const int L = 1 + (stage - s0) / sh;
            const int qq = ((stage - s0) % sh) * 16 + quad, k = qq / nt, T = qq % nt;
            v = pa.wh(L, 16 * T + row, 16 * k + 4 * g + r);
        } else {                                                    // layer 5: stage s = k-groups 4 s .. 4 s + 3 x 4 out tiles
            const int s5 = stage - (s0 + 4 * sh);
            const int t = 4 * s5 + quad / 4, T = quad % 4;
            v = pa.w5(16 * T + row, 16 * t + 4 * g + r);
        }
    } else if (e < kBlobFloats) {
        const int stage = e / kStageFloats;
        const int in_stage = e - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int row = lane & 15, g = lane >> 4;
        if (stage < kStagesL0) {                                    // layer 0: W[256,96]
            const int out = 16 * quad + row;
            v = pa.w0(out, stage, g, r);
        } else if (stage < kStagesL0 + 4 * kStagesHidden) {         // layers 1..4: W[256,256]
            const int L = 1 + (stage - kStagesL0) / kStagesHidden;
            const int t = (stage - kStagesL0) % kStagesHidden;
            const int out = 16 * quad + row;
            v = pa.wh(L, out, 16 * t + 4 * g + r);
        } else {                                                    // layer 5: W[54,256]
            const int s = stage - (kStagesL0 + 4 * kStagesHidden);
            const int t = 4 * s + quad / 4, T = quad % 4;
            const int out = 16 * T + row;
            v = pa.w5(out, 16 * t + 4 * g + r);
        }
    } else if (e >= kBwdHBlobOffset) {
        // transposed split-precision image (nerf_layout.h): two f16 of one slab per float slot
        const int eb = e - kBwdHBlobOffset;
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
            const int jj = 2 * word + k;
            float w = 0.f;
            if (stage < kStagesL5) {
                const int half = stage / 2, m = stage % 2;
                const int out = 32 * m + 16 * (jj >> 2) + 4 * kg + (jj & 3);
                w = pa.w5(out, 16 * (8 * half + pair) + row);
            } else {
                const int L = 4 - (stage - kStagesL5) / kStagesHidden;      // 4, 3, 2, 1
                const int s = (stage - kStagesL5) % kStagesHidden;
                const int half = s / 8, m = s % 8;
                const int out = 32 * m + 16 * (jj >> 2) + 4 * kg + (jj & 3);
                w = pa.wh(L, out, 16 * (8 * half + pair) + row);
            }
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;              // round to nearest
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e >= kHSmallOffset) {
        // small image of the split-precision path: bias * 2^12, gamma * 2^4, beta * 2^4
        const int i = e - kHSmallOffset;
        const float sb = (float)(1 << (kWScaleLog2 + kXScaleLog2)), sx = (float)(1 << kXScaleLog2);
        if (i < 5 * kSmallPerLayer) {
            const int L = i / kSmallPerLayer, rem = i % kSmallPerLayer;
            const int which = rem / kHidden, q = rem % kHidden;
            const int g = q / 64, T = (q % 64) / 4, reg = q & 3;
            const int f = 16 * T + 4 * g + reg;
            const int tensor = which == 0 ? 4 * L + 1 : (which == 1 ? 4 * L + 2 : 4 * L + 3);
            v = pa.vec(tensor, f) * (which == 0 ? sb : sx);
        } else {
            const int q = i - 5 * kSmallPerLayer;
            const int g = q / 16, T = (q % 16) / 4, reg = q & 3;
            const int n = 16 * T + 4 * g + reg;
            if (pa.row5(n) >= 0) v = pa.p[21][pa.row5(n)] * sb;
        }
    } else if (e >= kHBlobOffset) {
        // split-precision image (nerf_layout.h): this float slot carries two f16 of one slab
        const int eb = e - kHBlobOffset;
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
            const int jj = 2 * word + k;
            const int tl = jj >> 2, r = jj & 3;           // tile 2 m + tl of the k block, register r
            float w = 0.f;
            if (stage < kStagesL0) {
                const int half = stage / 3, m = stage % 3;
                const int out = 16 * (8 * half + pair) + row;
                w = pa.w0(out, 2 * m + tl, kg, r);
            } else if (stage < kStagesL0 + 4 * kStagesHidden) {
                const int L = 1 + (stage - kStagesL0) / kStagesHidden;
                const int s = (stage - kStagesL0) % kStagesHidden;
                const int half = s / 8, m = s % 8;
                const int out = 16 * (8 * half + pair) + row;
                w = pa.wh(L, out, 32 * m + 16 * tl + 4 * kg + r);
            } else {
                const int s = stage - (kStagesL0 + 4 * kStagesHidden);
                const int m = 2 * s + (pair >> 2), T = pair & 3;
                const int out = 16 * T + row;
                w = pa.w5(out, 32 * m + 16 * tl + 4 * kg + r);
            }
            w = __builtin_fminf(__builtin_fmaxf(w * (float)(1 << kWScaleLog2), -65504.f), 65504.f);
            const _Float16 hi = (_Float16)w;              // round to nearest
            h[k] = is_lo ? (_Float16)(w - (float)hi) : hi;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        v = __builtin_bit_cast(float, h2v{h[0], h[1]});
    } else if (e >= kBwdBlobOffset) {
        // transposed image (nerf_layout.h): [lane (i, g)][r] = W[16 tout + 4 g + r][16 Tin + i]
        const int eb = e - kBwdBlobOffset;
        const int stage = eb / kStageFloats;
        const int in_stage = eb - stage * kStageFloats;
        const int tin = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int i = lane & 15, g = lane >> 4;
        if (stage < kStagesL5) {
            const int out = 16 * stage + 4 * g + r;
            v = pa.w5(out, 16 * tin + i);
        } else {
            const int L = 4 - (stage - kStagesL5) / kStagesHidden;      // 4, 3, 2, 1
            const int tout = (stage - kStagesL5) % kStagesHidden;
            v = pa.wh(L, 16 * tout + 4 * g + r, 16 * tin + i);
        }
    } else {
        const int i = e - kBlobFloats;
        if (i < 5 * kSmallPerLayer) {
            const int L = i / kSmallPerLayer, rem = i % kSmallPerLayer;
            const int which = rem / kHidden, q = rem % kHidden;     // which: bias, gamma, beta
            const int g = q / 64, T = (q % 64) / 4, reg = q & 3;
            const int f = 16 * T + 4 * g + reg;
            const int tensor = which == 0 ? 4 * L + 1 : (which == 1 ? 4 * L + 2 : 4 * L + 3);
            v = pa.vec(tensor, f);
        } else {
            const int q = i - 5 * kSmallPerLayer;                   // last bias [g][T(4)][reg]
            const int g = q / 16, T = (q % 16) / 4, reg = q & 3;
            const int n = 16 * T + 4 * g + reg;
            if (pa.row5(n) >= 0) v = pa.p[21][pa.row5(n)];
        }
    }
    pa.packed[e] = v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int nerf_hip_version(void) { return NERF_HIP_ABI_VERSION; }

const char* nerf_hip_last_error(void) { return nerf_common::last_error(); }

const char* nerf_hip_build_flags(void) { return nerf_common::build_flags(); }

size_t nerf_hip_packed_bytes(void) { return (size_t)kPackedFloats * sizeof(float); }

size_t nerf_hip_train_workspace_bytes(int64_t n_rays, int32_t num_samples) {
    if (n_rays <= 0 || num_samples < 2) return 0;
    const int chunks = (num_samples - 1 + kSamplesPerWave - 1) / kSamplesPerWave;
    return (size_t)make_train_layout(n_rays, chunks).total * sizeof(float);
}

size_t nerf_hip_grad_elements(int32_t hidden, int32_t enc_inputs, int32_t num_outputs) {
    const Shape s{hidden, enc_inputs, num_outputs, 1};       // (the count does not depend on how the rows split)
    return shape_ok(s) ? (size_t)grad_elements(s) : 0;
}

int nerf_hip_pack_weights(const float* const* params, int32_t hidden, int32_t enc_inputs, int32_t num_outputs,
                          int32_t color_outputs, float* packed, void* stream) {
    if (params == nullptr || packed == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "pack_weights: null pointer");
    if (color_outputs == 0) color_outputs = 3;
    if (!shape_ok(Shape{hidden, enc_inputs, num_outputs, color_outputs}))
        return nerf_common::fail(NERF_HIP_EUNSUPPORTED, "pack_weights: hidden must be 1 .. 256, enc_inputs a multiple of 6 in 6 .. 96, "
                                                        "color_outputs 1 .. 12, num_outputs 1 + color_outputs .. 64 (1 density + "
                                                        "color + segmentation classes)");
    PackArgs pa;
    pa.n_out = num_outputs;
    pa.colors = color_outputs;
    pa.hidden = hidden;
    pa.enc_in = enc_inputs;
    for (int i = 0; i < NERF_HIP_NUM_PARAM_TENSORS; ++i) {
        if (params[i] == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "pack_weights: null tensor");
        pa.p[i] = params[i];
    }
    pa.packed = packed;
    const int threads = 256, blocks = (kPackedFloats + threads - 1) / threads + 1;    // + the bounds block
    nerf_common::TimedLaunch timed((hipStream_t)stream, NERF_HIP_TIMING_PACK);
    hipLaunchKernelGGL(nerf_pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, pa);
    return nerf_common::check_hip(hipGetLastError(), "pack_weights launch");
}

int nerf_hip_render_forward(const NerfHipRenderArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: null args");
    const NerfHipRenderArgs& a = *args;
    if (a.n_rays == 0) return NERF_HIP_OK;
    if (a.n_rays < 0 || a.num_samples < 2 || a.num_samples > 4096)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: n_rays / num_samples out of range");
    // rgb may be NULL on a TRAINING forward that is asked for the per-sample outputs only (a differentiable
    // NeRF.forward, model.py:553-594: nothing is composited, its backward takes d_raw)
    const bool field_only = a.rgb == nullptr && a.train_workspace != nullptr && a.out_raw != nullptr &&
                            a.seg == nullptr && a.out_weights == nullptr;
    if (a.packed == nullptr || (a.rgb == nullptr && !field_only) || (a.t_table == nullptr && a.t_values == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: packed / rgb / t_table is null");
    if (!shape_ok(shape_of(a)))
        return nerf_common::fail(NERF_HIP_EUNSUPPORTED, "render_forward: hidden must be 1 .. 256, enc_inputs a multiple of 6 in 6 .. 96, "
                                                        "color_outputs 1 .. 12, num_outputs 1 + color_outputs .. 64 (1 density + "
                                                        "color + segmentation classes)");
    if (shape_of(a).classes() == 0 && a.seg != nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: seg given but the network has no segmentation classes");
    const bool arrays = a.rays_o != nullptr && a.rays_d != nullptr;
    const bool cameras = a.camera_o != nullptr && a.camera_r != nullptr && a.image_h > 0 &&
                         a.image_w > 0 && a.focal_length != 0.f;
    if (!arrays && !cameras)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: neither ray arrays nor cameras given");
    if ((a.rays_o == nullptr) != (a.rays_d == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: rays_o and rays_d must come together");

    KernelArgs ka;
    ka.a = a;
    derive_slot_constants(ka.a);
    ka.intervals = a.num_samples - 1;
    ka.chunks = (ka.intervals + kSamplesPerWave - 1) / kSamplesPerWave;
    const bool train = a.train_workspace != nullptr;
    const int tt = train_tiles(shape_of(a).hidden);       // training: 8 or 16 register tiles per sample
    ka.save = make_train_layout(a.n_rays, ka.chunks, 16 * tt);
    ka.norm = norm_divisor(shape_of(a).hidden);
    ka.enc_per = scales_per_group(shape_of(a).scales());
    if (a.precision != NERF_HIP_PRECISION_FP32 && a.precision != NERF_HIP_PRECISION_F16X3)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: unknown precision");
    if (train && a.out_t != nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: out_t is not produced by the training forward");
    // inference: one ray per wave; training: one (padded ray, chunk) item per wave
    ka.groups = train ? ka.save.mp / 16 / kWavesPerWg : (a.n_rays + kWavesPerWg - 1) / kWavesPerWg;

    int device = 0, cus = 0;
    int rc = nerf_common::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (rc) return rc;
    rc = nerf_common::check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device),
                                "hipDeviceGetAttribute");
    if (rc) return rc;
    const bool half = a.precision == NERF_HIP_PRECISION_F16X3;
    const bool per_sample = a.out_raw != nullptr || a.out_mean != nullptr || a.out_cov != nullptr ||
                            a.out_t != nullptr || a.out_weights != nullptr;
    typedef void (*Kernel)(const KernelArgs);
    // [train][half][per_sample]
    static const Kernel kernels[2][2][2] = {
        {{nerf_render_fwd_kernel<false, false, false>, nerf_render_fwd_kernel<false, false, true>},
         {nerf_render_fwd_kernel<false, true, false>, nerf_render_fwd_kernel<false, true, true>}},
        {{nerf_render_fwd_kernel<true, false, false>, nullptr},
         {nerf_render_fwd_kernel<true, true, false>, nullptr}}};
    // narrow networks at their own cost: fp32 inference [NT 8 | 4][per_sample]; split-precision inference at 8
    // (narrow_half: a network of hidden_size <= 64 runs zero-padded inside it); training forward at 8 in either
    // arithmetic (narrow_train[half])
    static const Kernel narrow[2][2] = {
        {nerf_render_fwd_kernel<false, false, false, 8>, nerf_render_fwd_kernel<false, false, true, 8>},
        {nerf_render_fwd_kernel<false, false, false, 4>, nerf_render_fwd_kernel<false, false, true, 4>}};
    static const Kernel narrow_half[2] = {nerf_render_fwd_kernel<false, true, false, 8>, nerf_render_fwd_kernel<false, true, true, 8>};
    static const Kernel narrow_train[2] = {nerf_render_fwd_kernel<true, false, false, 8>, nerf_render_fwd_kernel<true, true, false, 8>};
    static const Kernel narrow_train4 = nerf_render_fwd_kernel<true, false, false, 4>;      // hidden_size <= 64, fp32 arithmetic
    static unsigned done[2][2][2] = {}, done_narrow[2][2] = {}, done_narrow_half[2] = {}, done_narrow_train[2] = {}, done_narrow_train4 = 0;
    // training: compositing is its own kernel, which also writes out_weights
    const int ps = !train && per_sample;
    // register tiles per sample of this launch: training 8 / 16 (train_tiles); inference 4 / 8 / 16 in fp32
    // arithmetic, 8 / 16 in split-precision arithmetic
    int nt = train ? train_compute_tiles(shape_of(a).hidden, half) : tiles_for(shape_of(a).hidden);
    if (half && nt == 4) nt = 8;
    const bool is_narrow = nt < 16;
    const Kernel kernel = !is_narrow ? kernels[train][half][ps]
                          : train ? (nt == 4 ? narrow_train4 : narrow_train[half]) : (half ? narrow_half[ps] : narrow[nt == 4][ps]);
    unsigned* const done_mask = !is_narrow ? &done[train][half][ps]
                                : train ? (nt == 4 ? &done_narrow_train4 : &done_narrow_train[half])
                                        : (half ? &done_narrow_half[ps] : &done_narrow[nt == 4][ps]);
    // LDS: three-slot ring + small image (+ the split-precision kernel's stash); the render-only narrow kernels run a
    // two-slot ring without a stash, three workgroups per CU
    const bool three = is_narrow && !ps;
    // ... and the render-only kernel at 4 tiles keeps its whole image resident: ONE workgroup of 16 waves per CU, a ray
    // per wave (resident_weights above)
    const bool resident = nt == 4 && !half && !ps && !train;
    const int waves = resident ? kResidentWaves : kWavesPerWg;
    if (resident) ka.groups = (a.n_rays + waves - 1) / waves;
    const int lds_bytes = resident ? kResidentLdsBytes : three ? 2 * kStageBytes + kSmallLdsBytes : (half ? kLdsBytesHalf : kLdsBytes);
    rc = nerf_common::ensure_dynamic_lds((const void*)kernel, lds_bytes, device, done_mask);
    if (rc) return rc;
    int64_t grid = (int64_t)cus * (resident ? 1 : three ? 3 : 2);   // workgroups per CU (<= 72.4 KiB LDS, <= 256 VGPRs; narrow: 48.25 KiB, <= 168)
    if (grid > ka.groups) grid = ka.groups;
    hipStream_t st = (hipStream_t)stream;
    nerf_common::Timing::before(st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * waves), lds_bytes, st, ka);
    nerf_common::Timing::after(st, NERF_HIP_TIMING_FORWARD);
    if (train) {
        nerf_common::TimedLaunch timed(st, NERF_HIP_TIMING_COMPOSITE_FORWARD);
        const int64_t blocks = (a.n_rays + kWavesPerWg - 1) / kWavesPerWg;
        if (!field_only) hipLaunchKernelGGL(nerf_composite_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ka);
        if (a.out_raw != nullptr || a.out_mean != nullptr || a.out_cov != nullptr) {
            const int64_t elems = a.n_rays * ka.intervals * a.num_outputs;
            hipLaunchKernelGGL(nerf_field_outputs_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, ka);
        }
    }
    return nerf_common::check_hip(hipGetLastError(), "render_forward launch");
}

int nerf_hip_timing(int enable) { return nerf_common::Timing::enable(enable != 0); }

int nerf_hip_timing_read(int reset, double* avg_ms, int64_t* launches) {
    return nerf_common::Timing::read(reset != 0, avg_ms, launches);
}

int nerf_hip_timing_read_tagged(int reset, int32_t n_tags, double* avg_ms, int64_t* launches) {
    if (n_tags < 0 || n_tags > NERF_HIP_TIMING_TAGS)
        return nerf_common::fail(NERF_HIP_EINVAL, "timing_read_tagged: n_tags out of range");
    return nerf_common::Timing::read_tagged(reset != 0, n_tags, avg_ms, launches);
}

const char* nerf_hip_timing_tag_name(int32_t tag) {
    static const char* const names[NERF_HIP_TIMING_TAGS] = {
        "forward", "composite_forward", "composite_backward", "data_gradient", "weight_gradient",
        "reduce", "adam", "loss", "pack"};
    return tag >= 0 && tag < NERF_HIP_TIMING_TAGS ? names[tag] : nullptr;
}

}  // extern "C"

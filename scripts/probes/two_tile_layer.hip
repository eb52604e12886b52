// Timing probe for the split-precision ("f16x3") fused layer: does a wave that owns TWO 16-sample tiles
// (32 samples; one wave per SIMD, one workgroup per CU, A operands read once for six MFMAs, half the
// LDS-DMA bytes and half the stage hand-overs per sample) beat the product's structure (one tile per
// wave, two waves per SIMD, two workgroups per CU)?  Both kernels run the product's own fused hidden
// layer (nerf_fused.h: layer_fused_h + finish_moments_at) four times per item over the same cyclic
// 64-stage weight image; the two-tile kernel runs a two-tile copy of that loop (below).  Results are
// only summed into a sink: this is a TIMING probe, not a parity test.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize \
//     -I nerf_amd/csrc -I include scripts/probes/two_tile_layer.hip -o /tmp/two_tile && /tmp/two_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#include "nerf_fused.h"

using namespace nerf_fused;

constexpr int kImageStages = 64;                 // four hidden layers of 16 stages
typedef WeightPipe<kImageStages> ProbePipe;


// WeightPipe for ONE workgroup of eight waves per CU sharing the ring: every stage is fetched once for all
// eight (two 1 KiB pieces per wave), half the weight stream per sample of the product's two 4-wave workgroups.
struct Pipe8 {
    const char* blob;
    char* ring;
    int issue_stage, issue_slot, read_slot, wave, lane;
    __device__ __forceinline__ void init(const void* image, char* lds_ring, int w, int l) {
        blob = (const char*)image, ring = lds_ring, issue_stage = issue_slot = read_slot = 0, wave = w, lane = l;
    }
    __device__ __forceinline__ void issue() {
        const uint32_t dst = (uint32_t)(uintptr_t)(ring + issue_slot * kStageBytes + wave * 2048);
        uint32_t m0_saved;
        const char* base = blob + (size_t)issue_stage * kStageBytes + wave * 2048;
        const uint64_t base_u = (uint64_t)(uintptr_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
        const uint64_t sbase = ((uint64_t)hi << 32) | lo;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 2\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(m0_saved)
            : "v"(lane * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase)
            : "memory");
        issue_stage = (issue_stage + 1 == kImageStages) ? 0 : issue_stage + 1;
        issue_slot = (issue_slot + 1 == kRing) ? 0 : issue_slot + 1;
    }
    template <int kYounger = 0>
    __device__ __forceinline__ const f32x4* open_stage() {
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const f32x4* p = (const f32x4*)(ring + read_slot * kStageBytes) + lane;
        read_slot = (read_slot + 1 == kRing) ? 0 : read_slot + 1;
        return p;
    }
    __device__ __forceinline__ void prefetch_next() {
        asm volatile("" ::: "memory");
        issue();
    }
};

template <int kValu, int kMfma>
__device__ __forceinline__ void interleave_n() {
#pragma unroll
    for (int i = 0; i < kMfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);
    }
}

constexpr int kNoDma = 1, kNoBarrier = 2, kNoReads = 4, kNoBuild = 8, kNoMoments = 16, kSpreadDma = 32, kPairUnits = 64, kDmaToRegs = 128, kHalfReads = 256, kBoundary = 512;

// the same four 1 KiB fetches as plain loads into registers nobody reads: the texture path's share of a DMA's cost
template <class Pipe>
__device__ __forceinline__ void fetch_to_regs(Pipe& pipe) {
    const char* base = pipe.blob + (size_t)pipe.issue_stage * kStageBytes + pipe.wave * 4096 + pipe.lane * 16;
    f32x4 a, b, c, d;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(base) : "memory");
    pipe.issue_stage = (pipe.issue_stage + 1 == kImageStages) ? 0 : pipe.issue_stage + 1;
    pipe.issue_slot = (pipe.issue_slot + 1 == kRing) ? 0 : pipe.issue_slot + 1;
}

// one 1 KiB piece of the stage the pipe would issue next (WeightPipe::issue, piece by piece); the pipe's
// counters advance with the fourth
template <int kPiece, class Pipe>
__device__ __forceinline__ void issue_piece(Pipe& pipe) {
    const uint32_t dst = (uint32_t)(uintptr_t)(pipe.ring + pipe.issue_slot * kStageBytes + pipe.wave * 4096);
    uint32_t m0_saved;
    const char* base = pipe.blob + (size_t)pipe.issue_stage * kStageBytes + pipe.wave * 4096;
    const uint64_t base_u = (uint64_t)(uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 2\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:%c4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_saved)
        : "v"(pipe.lane * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase), "n"(kPiece * 1024)
        : "memory");
    if (kPiece == 3) {
        pipe.issue_stage = (pipe.issue_stage + 1 == kImageStages) ? 0 : pipe.issue_stage + 1;
        pipe.issue_slot = (pipe.issue_slot + 1 == kRing) ? 0 : pipe.issue_slot + 1;
    }
}

// layer_fused_h (inference, Linear -> LayerNorm -> ReLU order) for two sample tiles that share every
// A operand: unit = two ds_read_b128 and SIX MFMAs, the two tiles' MFMAs alternate.
template <int KB, bool kNormIn, int kFlags = 0, class Pipe>
__device__ __forceinline__ void layer_fused_h2(Pipe& pipe, f32x4 (&in)[2][16], f32x4 (&out)[2][16],
                                               const LazyNorm (&norm)[2], HMoments (&mom)[2]) {
    constexpr int kStages = 2 * KB, kUnits = 8 * kStages;
    h8 bhi[2][KB], blo[2][KB];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        if (kNormIn) {
            normalize_tile<false, kPackNorm>(in[v][0], norm[v], 0);
            normalize_tile<false, kPackNorm>(in[v][1], norm[v], 1);
        }
        split8(in[v][0], in[v][1], bhi[v][0], blo[v][0]);
        if (kFlags & kNoBuild) {
#pragma unroll
            for (int m = 1; m < KB; ++m) split8(in[v][2 * m], in[v][2 * m + 1], bhi[v][m], blo[v][m]);
        }
        mom[v].reset();
    }
    auto open = [&]() -> const h8* {
        if (kFlags & kNoBarrier) {
            const h8* p = (const h8*)((const f32x4*)(pipe.ring + pipe.read_slot * kStageBytes) + pipe.lane);
            pipe.read_slot = (pipe.read_slot + 1 == kRing) ? 0 : pipe.read_slot + 1;
            return p;
        }
        return (const h8*)pipe.open_stage();
    };
    auto next = [&]() {
        if (!(kFlags & kNoDma)) pipe.prefetch_next();
    };
    h8 ah[kSets], al[kSets];
    f32x4 ga, be;
    h2 nh[2][4], nl[2][4];
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = open();
#pragma unroll
    for (int u = 0; u < kSets - 1; ++u) {
        ah[u] = st[(2 * u) * 64];
        al[u] = st[(2 * u + 1) * 64];
    }
    next();
#pragma unroll
    for (int s = 0; s < kStages; ++s) {
        const int half = s / KB, m = s % KB;
        const bool build_next = !(kFlags & kNoBuild) && half == 0 && m + 1 < KB;
        const int ta = 2 * m + 2, tb = 2 * m + 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int U = 8 * s + i, set = U % kSets;
            const int T = 8 * half + i;
            out[0][T] = mfma_h(ah[set], bhi[0][m], out[0][T]);
            __builtin_amdgcn_sched_barrier(0);
            if (U + kSets - 1 < kUnits) {
                const int ip = (i + kSets - 1) % 8, pset = (U + kSets - 1) % kSets;
                if (ip == 0) st = open();
                if (!(kFlags & kNoReads)) {
                    ah[pset] = st[(2 * ip) * 64];
                    al[pset] = st[(2 * ip + 1) * 64];
                }
                if (ip == 0) next();
            }
            if (kNormIn && build_next && (i == 0 || i == 2)) {
                ga = norm[0].gam[i == 0 ? ta : tb];
                be = norm[0].bet[i == 0 ? ta : tb];
            }
            __builtin_amdgcn_sched_barrier(0);
            out[1][T] = mfma_h(ah[set], bhi[1][m], out[1][T]);
            out[0][T] = mfma_h(ah[set], blo[0][m], out[0][T]);
            out[1][T] = mfma_h(ah[set], blo[1][m], out[1][T]);
            out[0][T] = mfma_h(al[set], bhi[0][m], out[0][T]);
            out[1][T] = mfma_h(al[set], bhi[1][m], out[1][T]);
            if (build_next) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (kNormIn && i == 1) normalize_tile<false, kPackNorm>(in[v][ta], norm[v], ta, ga, be);
                    if (i == 2) split4(in[v][ta], nh[v][0], nh[v][1], nl[v][0], nl[v][1]);
                    if (kNormIn && i == 3) normalize_tile<false, kPackNorm>(in[v][tb], norm[v], tb, ga, be);
                    if (i == 4) {
                        split4(in[v][tb], nh[v][2], nh[v][3], nl[v][2], nl[v][3]);
                        bhi[v][m + 1] = join8(nh[v][0], nh[v][1], nh[v][2], nh[v][3]);
                        blo[v][m + 1] = join8(nl[v][0], nl[v][1], nl[v][2], nl[v][3]);
                    }
                }
                if (i >= 1 && i <= 4) interleave_n<4, 5>();
            }
            if (half == 1 && !(kFlags & kNoMoments)) {
#pragma unroll
                for (int T2 = 0; T2 < 8; ++T2) {
                    const int first = (8 * m + KB - 1) / KB;
                    if (T2 * KB / 8 == m && i == (s + 1 < kStages ? T2 - first : 0)) {
                        mom[0].add(out[0][T2]);
                        mom[1].add(out[1][T2]);
                        if (s + 1 < kStages) interleave_n<2, 5>();
                    }
                }
                if (s + 1 == kStages && i >= 1) {
                    mom[0].add(out[0][T - 1]);
                    mom[1].add(out[1][T - 1]);
                    interleave_n<2, 5>();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    mom[0].add(out[0][15]);
    mom[1].add(out[1][15]);
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

// The product's one-tile loop again, with timing-only switches (wrong results): which part of a unit costs what.
template <int KB, int kFlags, int kS = kSets, class Pipe>
__device__ __forceinline__ void layer_ablate(Pipe& pipe, f32x4 (&in)[16], f32x4 (&out)[16], const LazyNorm& norm,
                                             HMoments& mom) {
    constexpr int kStages = 2 * KB, kUnits = 8 * kStages;
    constexpr bool kBuild = !(kFlags & kNoBuild);
    h8 bhi[KB], blo[KB];
    normalize_tile<false, kPackNorm>(in[0], norm, 0);
    normalize_tile<false, kPackNorm>(in[1], norm, 1);
    split8(in[0], in[1], bhi[0], blo[0]);
    if (!kBuild) {
#pragma unroll
        for (int m = 1; m < KB; ++m) split8(in[2 * m], in[2 * m + 1], bhi[m], blo[m]);
    }
    mom.reset();
    h8 ah[kS], al[kS];
    f32x4 ga, be;
    h2 nh[4], nl[4];
    auto open = [&]() -> const h8* {
        if (kFlags & kNoBarrier) {
            const h8* p = (const h8*)((const f32x4*)(pipe.ring + pipe.read_slot * kStageBytes) + pipe.lane);
            pipe.read_slot = (pipe.read_slot + 1 == kRing) ? 0 : pipe.read_slot + 1;
            return p;
        }
        return (const h8*)pipe.open_stage();
    };
    auto next = [&]() {
        if (kFlags & kDmaToRegs) fetch_to_regs(pipe);
        else if (!(kFlags & kNoDma)) pipe.prefetch_next();
    };
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = open();
#pragma unroll
    for (int u = 0; u < kS - 1; ++u) {
        ah[u] = st[(2 * u) * 64];
        al[u] = st[(2 * u + 1) * 64];
    }
    next();
#pragma unroll
    for (int s = 0; s < kStages; ++s) {
        const int half = s / KB, m = s % KB;
        const bool build_next = kBuild && half == 0 && m + 1 < KB;
        const int ta = 2 * m + 2, tb = 2 * m + 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int U = 8 * s + i, set = U % kS;
            const int T = 8 * half + i;
            out[T] = mfma_h(ah[set], bhi[m], out[T]);
            __builtin_amdgcn_sched_barrier(0);
            if (U + kS - 1 < kUnits) {
                const int ip = (i + kS - 1) % 8, pset = (U + kS - 1) % kS;
                if (ip == 0) st = open();
                if (!(kFlags & kNoReads)) {
                    ah[pset] = st[(2 * ip) * 64];
                    if (kFlags & kHalfReads) al[pset] = ah[pset];
                    else al[pset] = st[(2 * ip + 1) * 64];
                }
                if (kFlags & kSpreadDma) {           // one piece behind each of the next four units' first MFMA
                    if (ip == 0) issue_piece<0>(pipe);
                    if (ip == 1) issue_piece<1>(pipe);
                    if (ip == 2) issue_piece<2>(pipe);
                    if (ip == 3) issue_piece<3>(pipe);
                } else if (ip == 0) next();
            }
            if (build_next && (i == 0 || i == 2)) {
                ga = norm.gam[i == 0 ? ta : tb];
                be = norm.bet[i == 0 ? ta : tb];
            }
            __builtin_amdgcn_sched_barrier(0);
            out[T] = mfma_h(ah[set], blo[m], out[T]);
            out[T] = mfma_h(al[set], bhi[m], out[T]);
            if (build_next) {
                if (i == 1) normalize_tile<false, kPackNorm>(in[ta], norm, ta, ga, be);
                if (i == 2) split4(in[ta], nh[0], nh[1], nl[0], nl[1]);
                if (i == 3) normalize_tile<false, kPackNorm>(in[tb], norm, tb, ga, be);
                if (i == 4) {
                    split4(in[tb], nh[2], nh[3], nl[2], nl[3]);
                    bhi[m + 1] = join8(nh[0], nh[1], nh[2], nh[3]);
                    blo[m + 1] = join8(nl[0], nl[1], nl[2], nl[3]);
                }
                if (i >= 1 && i <= 4) interleave_2<4>();
            }
            if (half == 1 && !(kFlags & kNoMoments)) {
#pragma unroll
                for (int T2 = 0; T2 < 8; ++T2) {
                    const int first = (8 * m + KB - 1) / KB;
                    if (T2 * KB / 8 == m && i == (s + 1 < kStages ? T2 - first : 0)) {
                        mom.add(out[T2]);
                        if (s + 1 < kStages) interleave_2<2>();
                    }
                }
                if (s + 1 == kStages && i >= 1) {
                    mom.add(out[T - 1]);
                    interleave_2<2>();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    mom.add(out[15]);
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}

struct Args {
    const float* image;         // kImageStages stages of 16 KiB
    const float* seed;          // [64] lane values
    float* sink;
    int items;                  // per workgroup
};

__device__ __forceinline__ void fill_small(float* small) {
    for (int i = threadIdx.x; i < 3 * kSmallArrayLds; i += 256)
        small[i] = i < kSmallArrayLds ? 0.01f : i < 2 * kSmallArrayLds ? 1.0f : 0.05f;      // bias, gamma, beta
}
__device__ __forceinline__ void bias16(const float* small, int g, f32x4 (&acc)[16]) {
    const f32x4* b = (const f32x4*)(small + g * kSmallGStride);
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = b[T];
}

__global__ __launch_bounds__(256, 2) void one_tile_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    float* small = (float*)(smem + kRingBytes);
    fill_small(small);
    ProbePipe pipe;
    pipe.init(a.image, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();
    const f32x4* gam = (const f32x4*)(small + kSmallArrayLds + g * kSmallGStride);
    const f32x4* bet = (const f32x4*)(small + 2 * kSmallArrayLds + g * kSmallGStride);
    const float eps = 1e-5f * 4096.f * 4096.f;
    float total = 0.f;
    for (int it = 0; it < a.items; ++it) {
        f32x4 X[16], Y[16];
        const float sv = a.seed[lane] + (float)it;
#pragma unroll
        for (int T = 0; T < 16; ++T) X[T] = f32x4{sv + T, sv - T, sv * 0.5f + T, sv * 0.25f - T};
        LazyNorm norm;
        norm.rstd = 0.01f, norm.shift = 0.f, norm.gam = gam, norm.bet = bet, norm.save_row = nullptr;
        HMoments mom;
#pragma unroll 1
        for (int L = 0; L < 2; ++L) {
            bias16(small, g, Y);
            layer_fused_h<8, true, false>(pipe, X, Y, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, Y, gam, bet, g, nullptr, nullptr, eps);
            bias16(small, g, X);
            layer_fused_h<8, true, false>(pipe, Y, X, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, X, gam, bet, g, nullptr, nullptr, eps);
        }
#pragma unroll
        for (int T = 0; T < 16; ++T) total += (X[T].x + X[T].y) * norm.rstd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total == 12345.678f) a.sink[threadIdx.x] = total;
}

template <int kFlags>
__global__ __launch_bounds__(256, 1) void two_tile_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    float* small = (float*)(smem + kRingBytes);
    fill_small(small);
    ProbePipe pipe;
    pipe.init(a.image, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();
    const f32x4* gam = (const f32x4*)(small + kSmallArrayLds + g * kSmallGStride);
    const f32x4* bet = (const f32x4*)(small + 2 * kSmallArrayLds + g * kSmallGStride);
    const float eps = 1e-5f * 4096.f * 4096.f;
    float total = 0.f;
    for (int it = 0; it < a.items; ++it) {
        f32x4 X[2][16], Y[2][16];
        const float sv = a.seed[lane] + (float)it;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int T = 0; T < 16; ++T) X[v][T] = f32x4{sv + T + v, sv - T, sv * 0.5f + T - v, sv * 0.25f - T};
        LazyNorm norm[2];
        HMoments mom[2];
#pragma unroll
        for (int v = 0; v < 2; ++v)
            norm[v].rstd = 0.01f, norm[v].shift = 0.f, norm[v].gam = gam, norm[v].bet = bet, norm[v].save_row = nullptr;
#pragma unroll 1
        for (int L = 0; L < 2; ++L) {
            bias16(small, g, Y[0]);
            bias16(small, g, Y[1]);
            layer_fused_h2<8, true, kFlags>(pipe, X, Y, norm, mom);
#pragma unroll
            for (int v = 0; v < 2; ++v)
                norm[v] = finish_moments_at<false, HMoments>(mom[v], Y[v], gam, bet, g, nullptr, nullptr, eps);
            bias16(small, g, X[0]);
            bias16(small, g, X[1]);
            layer_fused_h2<8, true, kFlags>(pipe, Y, X, norm, mom);
#pragma unroll
            for (int v = 0; v < 2; ++v)
                norm[v] = finish_moments_at<false, HMoments>(mom[v], X[v], gam, bet, g, nullptr, nullptr, eps);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int T = 0; T < 16; ++T) total += (X[v][T].x + X[v][T].y) * norm[v].rstd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total == 12345.678f) a.sink[threadIdx.x] = total;
}


template <int kFlags, int kS = kSets>
__global__ __launch_bounds__(256, 2) void ablate_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    float* small = (float*)(smem + kRingBytes);
    fill_small(small);
    ProbePipe pipe;
    pipe.init(a.image, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const f32x4* gam = (const f32x4*)(small + kSmallArrayLds + g * kSmallGStride);
    const f32x4* bet = (const f32x4*)(small + 2 * kSmallArrayLds + g * kSmallGStride);
    const float eps = 1e-5f * 4096.f * 4096.f;
    float total = 0.f;
    for (int it = 0; it < a.items; ++it) {
        f32x4 X[16], Y[16];
        const float sv = a.seed[lane] + (float)it;
#pragma unroll
        for (int T = 0; T < 16; ++T) X[T] = f32x4{sv + T, sv - T, sv * 0.5f + T, sv * 0.25f - T};
        LazyNorm norm;
        norm.rstd = 0.01f, norm.shift = 0.f, norm.gam = gam, norm.bet = bet, norm.save_row = nullptr;
        HMoments mom;
#pragma unroll 1
        for (int L = 0; L < 2; ++L) {
            bias16(small, g, Y);
            layer_ablate<8, kFlags, kS>(pipe, X, Y, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, Y, gam, bet, g, nullptr, nullptr, eps);
            bias16(small, g, X);
            layer_ablate<8, kFlags, kS>(pipe, Y, X, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, X, gam, bet, g, nullptr, nullptr, eps);
        }
#pragma unroll
        for (int T = 0; T < 16; ++T) total += (X[T].x + X[T].y) * norm.rstd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total == 12345.678f) a.sink[threadIdx.x] = total;
}

template <int kFlags, int kS = kSets>
int run_ablation(const Args& a, int lds, hipEvent_t e0, hipEvent_t e1, const char* what) {
    if (hipFuncSetAttribute((const void*)ablate_kernel<kFlags, kS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        ablate_kernel<kFlags, kS><<<512, 256, lds>>>(a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = 512.0 * 4 * a.items * 16 * 4 * 2.0 * 256 * 256 * 3;
    printf("  %-44s %.3f ms  (%.3f)\n", what, best, flop / (best * 1e-3) / 2.5166e15);
    return 0;
}


// The product's one-tile loop in one 8-wave workgroup per CU on a shared ring (Pipe8).
__global__ __launch_bounds__(512, 1) void shared_ring_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    float* small = (float*)(smem + kRingBytes);
    for (int i = threadIdx.x; i < 3 * kSmallArrayLds; i += 512)
        small[i] = i < kSmallArrayLds ? 0.01f : i < 2 * kSmallArrayLds ? 1.0f : 0.05f;
    Pipe8 pipe;
    pipe.init(a.image, smem, wave, lane);
    pipe.issue();
    pipe.issue();
    __syncthreads();
    const f32x4* gam = (const f32x4*)(small + kSmallArrayLds + g * kSmallGStride);
    const f32x4* bet = (const f32x4*)(small + 2 * kSmallArrayLds + g * kSmallGStride);
    const float eps = 1e-5f * 4096.f * 4096.f;
    float total = 0.f;
    for (int it = 0; it < a.items; ++it) {
        f32x4 X[16], Y[16];
        const float sv = a.seed[lane] + (float)it;
#pragma unroll
        for (int T = 0; T < 16; ++T) X[T] = f32x4{sv + T, sv - T, sv * 0.5f + T, sv * 0.25f - T};
        LazyNorm norm;
        norm.rstd = 0.01f, norm.shift = 0.f, norm.gam = gam, norm.bet = bet, norm.save_row = nullptr;
        HMoments mom;
#pragma unroll 1
        for (int L = 0; L < 2; ++L) {
            bias16(small, g, Y);
            layer_fused_h<8, true, false>(pipe, X, Y, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, Y, gam, bet, g, nullptr, nullptr, eps);
            bias16(small, g, X);
            layer_fused_h<8, true, false>(pipe, Y, X, norm, mom);
            norm = finish_moments_at<false, HMoments>(mom, X, gam, bet, g, nullptr, nullptr, eps);
        }
#pragma unroll
        for (int T = 0; T < 16; ++T) total += (X[T].x + X[T].y) * norm.rstd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total == 12345.678f) a.sink[threadIdx.x] = total;
}


// ---------------------------------------------------------------------------------------------
// The decomposition section 8 of DESIGN.md names as the way out: a workgroup's 64 samples share their B
// operands through LDS, each wave owns a 4 x 4 block of (output tile, sample tile) pairs — 8 + 8 ds_read_b128
// per 48 MFMAs instead of 32 — and only the A half it needs of every k block.  INNER LOOP ONLY: the B exchange
// buffer is filled once and never rewritten (no LayerNorm, no normalise / split, no cross-wave statistics), so
// this is an upper bound for such a kernel's layers, measured with the real weight stream (same image, a
// 4-slot ring: both 16 KiB halves of a k block are resident while the next block's are in flight).
// ---------------------------------------------------------------------------------------------
constexpr int kRing4 = 4;
template <int kWaves>
struct PipeN {
    const char* blob;
    char* ring;
    int issue_stage, issue_slot, read_slot, wave, lane;
    __device__ __forceinline__ void init(const void* image, char* lds_ring, int w, int l) {
        blob = (const char*)image, ring = lds_ring, issue_stage = issue_slot = read_slot = 0, wave = w, lane = l;
    }
    __device__ __forceinline__ void issue() {                    // one 16 KiB stage, four pieces per wave
        constexpr int kShare = kStageBytes / kWaves;          // 4 KiB (four pieces) or 2 KiB (two) per wave
        const uint32_t dst = (uint32_t)(uintptr_t)(ring + issue_slot * kStageBytes + wave * kShare);
        uint32_t m0_saved;
        const char* base = blob + (size_t)issue_stage * kStageBytes + wave * kShare;
        const uint64_t base_u = (uint64_t)(uintptr_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
        const uint64_t sbase = ((uint64_t)hi << 32) | lo;
        if (kWaves == 4)
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
                : "v"(lane * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase)
                : "memory");
        else
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 2\n\t"
                "global_load_lds_dwordx4 %1, %3\n\t"
                "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(m0_saved)
                : "v"(lane * 16), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(sbase)
                : "memory");
        issue_stage = (issue_stage + 1 == kImageStages) ? 0 : issue_stage + 1;
        issue_slot = (issue_slot + 1 == kRing4) ? 0 : issue_slot + 1;
    }
};

template <int kFlags, int kWaves = 4>
__global__ __launch_bounds__(64 * kWaves, kWaves == 4 ? 2 : 1) void gemm_block_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const bbuf = smem + kRing4 * kStageBytes;             // [2][kWaves sample tiles][hi | lo][1 KiB]
    // kBoundary: per-sample LayerNorm partials of every wave behind the exchange buffer: [wave][sample tile][j] (sum, sum of squares)
    // (aliased onto the exchange buffer: 80 KiB per 4-wave workgroup is exactly two per CU; this is a timing probe)
    float* const stats = (float*)bbuf;
    for (int i = threadIdx.x; i < 4096 * kWaves / 4; i += 64 * kWaves) ((float*)bbuf)[i] = a.seed[i & 63] * 0.001f;
    const float gam = a.seed[lane & 63] * 0.5f + 1.0f, bet = a.seed[(lane + 7) & 63] * 0.1f;
    f32x4 pending[2][4];                                         // second k block of this wave's outputs, built in the next layer's shadow
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) pending[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rstd_q[4] = {1.f, 1.f, 1.f, 1.f}, shift_q[4] = {0.f, 0.f, 0.f, 0.f};
    PipeN<kWaves> pipe;
    pipe.init(a.image, smem, wave, lane);
    pipe.issue();
    pipe.issue();                                                // k block 0: both halves
    __syncthreads();
    // this wave's out tiles: 8 half + 4 quad .. + 3; its four sample tiles: the workgroup's first or second four
    const int half = (wave >> 1) & 1, quad = wave & 1, side = wave >> 2;
    float total = 0.f;
    h8 ah[2][4], al[2][4], bh[2][4], bl[2][4];                   // operand sets of two k blocks (current | next)
    int blk = 0;                                                 // k blocks handed over so far
    // hand-over of the next k block — its two stages have landed (nothing else is in flight), every wave has its
    // operands of the block before in registers (so those slots may be refilled) — then that block's operands
    // into set `into` and the DMA of the block after it
    auto advance = [&](auto into_tag) {
        constexpr int into = decltype(into_tag)::value;
        if (!(kFlags & kNoBarrier)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        const h8* st = (const h8*)((const f32x4*)(pipe.ring + ((pipe.read_slot + half) & (kRing4 - 1)) * kStageBytes) + lane);
        pipe.read_slot = (pipe.read_slot + 2) & (kRing4 - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[into][t] = st[(2 * (4 * quad + t)) * 64];
            al[into][t] = st[(2 * (4 * quad + t) + 1) * 64];
        }
        const h8* bs = (const h8*)(bbuf + (blk & 1) * (2048 * kWaves) + side * 8192) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bh[into][q] = bs[(2 * q) * 64];
            bl[into][q] = bs[(2 * q + 1) * 64];
        }
        ++blk;
        if (!(kFlags & kNoDma)) {
            pipe.issue();
            pipe.issue();
        }
    };
    advance(std::integral_constant<int, 0>());
    for (int it = 0; it < a.items; ++it) {
        f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                constexpr int dummy = 0;
                const int cur = kb & 1;
                // the next block's hand-over and operand reads go out BEFORE this block's MFMAs
                if (cur == 0) advance(std::integral_constant<int, 1>());
                else advance(std::integral_constant<int, 0>());
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[t][q] = mfma_h(ah[cur][t], bh[cur][q], acc[t][q]);
                        acc[t][q] = mfma_h(ah[cur][t], bl[cur][q], acc[t][q]);
                        acc[t][q] = mfma_h(al[cur][t], bh[cur][q], acc[t][q]);
                    }
                if ((kFlags & kBoundary) && kb == 0) {
                    // the SECOND k block this wave produces for the layer that has just started (its out tiles 2, 3 of
                    // the layer before): normalise, ReLU, split, publish — in the shadow of k block 0's 48 MFMAs
                    h8* dst = (h8*)(bbuf + 2048 * kWaves + side * 8192) + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                v[t][r] = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(pending[t][q][r], rstd_q[q], shift_q[q]), gam, bet), 0.f);
                        h8 hi, lo;
                        split8(v[0], v[1], hi, lo);
                        dst[(2 * q) * 64] = hi;
                        dst[(2 * q + 1) * 64] = lo;
                    }
#pragma unroll
                    for (int m = 0; m < 48; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kFlags & kBoundary) {
                // ---- what a real layer boundary adds: LayerNorm needs a sample's 256 features, which sit in four waves.
                // (1) this wave's partial moments of its 64 features for its 64 samples, (2) through LDS + one barrier,
                // (3) normalise / ReLU / split of the FIRST k block it produces (out tiles 0, 1) and publish it: the next
                // layer's k block 0 needs it before the first MFMA; the second block waits in registers (`pending`)
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s1[q] += acc[t][q][r];
                            s2[q] = __builtin_fmaf(acc[t][q][r], acc[t][q][r], s2[q]);
                        }
                    s1[q] = group_sum(s1[q]);
                    s2[q] = group_sum(s2[q]);
                }
                const int j = lane & 15, g = lane >> 4;
                if (g == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(float2*)(stats + ((wave * 4 + q) * 16 + j) * 2) = float2{s1[q], s2[q]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float2 p = *(const float2*)(stats + (((side * 4 + w) * 4 + q) * 16 + j) * 2);
                        m1 += p.x;
                        m2 += p.y;
                    }
                    const float mean = m1 * (1.0f / 256.0f), var = m2 * (1.0f / 256.0f) - mean * mean;
                    const float rs = __builtin_amdgcn_rsqf(var + 1e-5f);
                    rstd_q[q] = rs;
                    shift_q[q] = -mean * rs;
                }
                h8* dst = (h8*)(bbuf + side * 8192) + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[t][r] = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(acc[t][q][r], rstd_q[q], shift_q[q]), gam, bet), 0.f);
                    h8 hi, lo;
                    split8(v[0], v[1], hi, lo);
                    dst[(2 * q) * 64] = hi;
                    dst[(2 * q + 1) * 64] = lo;
#pragma unroll
                    for (int t = 0; t < 2; ++t) pending[t][q] = acc[2 + t][q];
                }
                // (the publish barrier is the next k block's hand-over; bias = the accumulators' start value)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = f32x4{gam, bet, gam, bet};
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) total += acc[t][q].x + acc[t][q].y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (total == 12345.678f) a.sink[threadIdx.x] = total;
}

template <int kFlags>
int run_two(const Args& a, int lds, hipEvent_t e0, hipEvent_t e1, const char* what) {
    if (hipFuncSetAttribute((const void*)two_tile_kernel<kFlags>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0);
        two_tile_kernel<kFlags><<<256, 256, lds>>>(a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = 512.0 * 4 * a.items * 16 * 4 * 2.0 * 256 * 256 * 3;
    printf("  %-44s %.3f ms  (%.3f)\n", what, best, flop / (best * 1e-3) / 2.5166e15);
    return 0;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));                     \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main() {
    const size_t image_bytes = (size_t)kImageStages * kStageBytes;
    std::vector<_Float16> host(image_bytes / 2);
    srand(1);
    for (auto& h : host) h = (_Float16)(((rand() % 2001) - 1000) * (1.0f / 1000.0f) * 16.0f);     // |w| * 2^8 range
    std::vector<float> seed(64);
    for (int i = 0; i < 64; ++i) seed[i] = (float)((i * 37) % 64) * 0.03125f - 1.0f;
    void *image, *dseed, *sink;
    CK(hipMalloc(&image, image_bytes));
    CK(hipMalloc(&dseed, 256));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemcpy(image, host.data(), image_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dseed, seed.data(), 256, hipMemcpyHostToDevice));
    const int lds = kRingBytes + 3 * kSmallArrayLds * 4 + 256;
    CK(hipFuncSetAttribute((const void*)one_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)two_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // the same number of 16-sample tiles through both: 512 workgroups x 4 waves x items tiles,
    // or 256 workgroups x 4 waves x 2 tiles x items
    const int items = 400;
    for (int rep = 0; rep < 3; ++rep) {
        Args a{(const float*)image, (const float*)dseed, (float*)sink, items};
        float ms1, ms2;
        CK(hipEventRecord(e0));
        one_tile_kernel<<<512, 256, lds>>>(a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms1, e0, e1));
        CK(hipEventRecord(e0));
        two_tile_kernel<0><<<256, 256, lds>>>(a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms2, e0, e1));
        const double tiles = 512.0 * 4 * items;
        const double flop = tiles * 16 * 4 * 2.0 * 256 * 256 * 3;          // executed f16 MFMA flop
        printf("one tile per wave, 2 waves/SIMD: %.3f ms (%.3f of the f16 MFMA peak executed)   "
               "two tiles per wave, 1 wave/SIMD: %.3f ms (%.3f)\n",
               ms1, flop / (ms1 * 1e-3) / 2.5166e15, ms2, flop / (ms2 * 1e-3) / 2.5166e15);
    }
    {
        Args a{(const float*)image, (const float*)dseed, (float*)sink, items};
        printf("one-tile loop, timing-only ablations (wrong results):\n");
        run_ablation<0>(a, lds, e0, e1, "as the product");
        run_ablation<kNoDma>(a, lds, e0, e1, "no LDS-DMA issue");
        run_ablation<kNoBarrier>(a, lds, e0, e1, "no hand-over wait / barrier");
        run_ablation<kNoDma | kNoBarrier>(a, lds, e0, e1, "no DMA, no barrier");
        run_ablation<kNoReads>(a, lds, e0, e1, "no A-operand LDS reads");
        run_ablation<kNoBuild>(a, lds, e0, e1, "no lazy normalise / split in the loop");
        run_ablation<kNoBuild | kNoMoments>(a, lds, e0, e1, "no build, no moments");
        run_ablation<kNoDma | kNoBarrier | kNoReads>(a, lds, e0, e1, "no DMA, barrier, reads");
        run_ablation<kNoDma | kNoBarrier | kNoReads | kNoBuild | kNoMoments>(a, lds, e0, e1, "MFMAs only");
        run_ablation<kSpreadDma>(a, lds, e0, e1, "LDS-DMA pieces spread over four units");
        run_ablation<kDmaToRegs>(a, lds, e0, e1, "same fetches into registers, not LDS");
        run_ablation<0, 8>(a, lds, e0, e1, "eight A-operand sets (reads seven units ahead)");
        run_ablation<0, 2>(a, lds, e0, e1, "two A-operand sets (reads one unit ahead)");
        run_ablation<kHalfReads>(a, lds, e0, e1, "one of the two A-operand reads per unit");
        run_ablation<kHalfReads | kNoDma>(a, lds, e0, e1, "one read per unit, no DMA");
        run_ablation<kNoReads | kNoDma>(a, lds, e0, e1, "no reads, no DMA");
        {
            hipFuncSetAttribute((const void*)shared_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                float ms;
                hipEventRecord(e0);
                shared_ring_kernel<<<256, 512, lds>>>(a);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double flop = 512.0 * 4 * a.items * 16 * 4 * 2.0 * 256 * 256 * 3;
            printf("  %-44s %.3f ms  (%.3f)\n", "one 8-wave workgroup per CU, shared ring", best, flop / (best * 1e-3) / 2.5166e15);
        }
        {
            const int lds4 = kRing4 * kStageBytes + 16384;
            const double flop = 512.0 * 4 * a.items * 16 * 4 * 2.0 * 256 * 256 * 3;
            auto run = [&](auto kernel, int waves, const char* what) {
                const int bytes = kRing4 * kStageBytes + 4096 * waves;
                hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                Args b = a;
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    float ms;
                    hipEventRecord(e0);
                    kernel<<<waves == 4 ? 512 : 256, 64 * waves, bytes>>>(b);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("  %-44s %.3f ms  (%.3f)\n", what, best, flop / (best * 1e-3) / 2.5166e15);
            };
            // per item a 4-wave workgroup does 4 layers x 8 k blocks x 48 MFMAs x 4 waves = the work of 4 one-tile
            // items (an 8-wave one of 8): the same total through every variant
            run(gemm_block_kernel<0, 4>, 4, "4 x 4 blocks per wave, B through LDS (bare loop)");
            run(gemm_block_kernel<kNoDma, 4>, 4, "  the same without the weight stream");
            run(gemm_block_kernel<kNoDma | kNoBarrier, 4>, 4, "  without stream and hand-over");
            run(gemm_block_kernel<0, 8>, 8, "4 x 4 blocks, ONE 8-wave workgroup per CU");
            run(gemm_block_kernel<kNoDma, 8>, 8, "  the same without the weight stream");
            // ... and with what a real layer boundary adds (LayerNorm partials through LDS + barrier, normalise /
            // split / publish of the wave's first k block exposed, of its second in the next layer's shadow)
            run(gemm_block_kernel<kBoundary, 4>, 4, "4 x 4 blocks, 4 waves, WITH layer boundaries");
            run(gemm_block_kernel<kBoundary, 8>, 8, "4 x 4 blocks, 8 waves, WITH layer boundaries");
        }
        printf("two-tile loop (one wave per SIMD), same switches:\n");
        run_two<0>(a, lds, e0, e1, "as written");
        run_two<kNoDma>(a, lds, e0, e1, "no LDS-DMA issue");
        run_two<kNoBarrier>(a, lds, e0, e1, "no hand-over wait / barrier");
        run_two<kNoReads>(a, lds, e0, e1, "no A-operand LDS reads");
        run_two<kNoBuild | kNoMoments>(a, lds, e0, e1, "no build, no moments");
        run_two<kNoDma | kNoBarrier | kNoReads>(a, lds, e0, e1, "no DMA, barrier, reads");
        run_two<kNoDma | kNoBarrier | kNoReads | kNoBuild | kNoMoments>(a, lds, e0, e1, "MFMAs only");
    }
    return 0;
}

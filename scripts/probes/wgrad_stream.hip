// Memory-system probe (gfx950): the weight gradient's operand stream WITHOUT its arithmetic.
// nerf_wgrad_h_kernel streams 2.6 GB per 4096 x 64 batch through a 4-slot LDS ring: one 4-wave workgroup per CU,
// a k-step = 16 KiB of dY + 16 KiB of X as 32 LDS-DMA pieces of 1 KiB (8 per wave), the DMA of step t + 3 issued
// during step t, a counted vmcnt + one barrier per step.  The kernel runs 0.66-0.69 ms = 3.7-3.9 TB/s.  What does
// that stream reach by itself, and what moves it?
//   slots    ring slots (slots - 1 steps in flight per workgroup)
//   wgs      workgroups per CU (LDS per workgroup = slots x step bytes: must fit 160 KiB together)
//   step     KiB per k-step (32: the kernel's; 16: half of it, for two workgroups per CU)
//   barrier  1: the ring's per-step workgroup barrier; 0: every wave waits only for its own pieces
//   order    0: DMA lane l fetches chunk l of its piece (linear); 1: the product's re-ordered chunks
//            ([s & 3][s >> 2][g] in LDS from [g][s] in memory, odd pieces sample-swizzled); 2: sample-major chunks
//   mfma     v_mfma_f32_32x32x16_f16 per wave and step on constant operands (the kernel's hidden-layer step: 48),
//            each followed by `valu` dependent-free v_fma (the kernel converts with ~6 VALU per MFMA), and `lds`
//            ds_read2st64_b32 per step (the kernel: 32) — nothing depends on the streamed data: does the stream
//            run UNDER the arithmetic, or do the two add?
//   dma 0    the arithmetic alone (no DMA issued)
//   layout   0: every workgroup streams its own contiguous share of two separately allocated buffers;
//            1: the kernel's addresses — 768 workgroups = 6 jobs x 128 splits, three rounds on the chip; split s of
//               job j streams the 2 MiB [s * 2 MiB, (s + 1) * 2 MiB) of tensors that lie j * 2^28 bytes apart (the
//               workspace's dY / x_hat arrays are mp * 1 KiB = 2^28 bytes each): every stream of a round sits at the
//               same offset modulo 2 MiB, and the two operands of a workgroup 5 * 2^28 apart;
//            2: the same tensors, split s taking 32 KiB tiles s, s + 128, s + 256, ... (the 128 splits of a job sweep
//               one contiguous window together)
//   fly      steps that may still be in flight at a step's hand-over (the kernel: 1 — the DMA of step t + 3 is issued
//            in step t and must have landed by the END OF STEP t + 1, because step t + 2 converts its operands)
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/wgrad_stream.hip -o gpurun_out/wgrad_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// four consecutive 1 KiB pieces: global (uniform base + k KiB + lane offset) -> LDS (dst + k KiB + lane * 16)
__device__ __forceinline__ void dma4(const char* src, uint32_t dst, uint32_t off_even, uint32_t off_odd) {
    const uint64_t base_u = (uint64_t)(uintptr_t)src;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
    const uint32_t d = __builtin_amdgcn_readfirstlane(dst);
    uint32_t m0_saved;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 2\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %4, %3 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
        "global_load_lds_dwordx4 %4, %3 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(m0_saved)
        : "v"(off_even), "s"(d), "s"(sbase), "v"(off_odd)
        : "memory");
}

template <int kPieces>       // per wave and operand and step: 4 (32 KiB steps) or 2 (16 KiB steps)
__device__ __forceinline__ void dma_n(const char* src, uint32_t dst, uint32_t off_even, uint32_t off_odd) {
    if constexpr (kPieces == 4) {
        dma4(src, dst, off_even, off_odd);
    } else {
        const uint64_t base_u = (uint64_t)(uintptr_t)src;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
        const uint64_t sbase = ((uint64_t)hi << 32) | lo;
        const uint32_t d = __builtin_amdgcn_readfirstlane(dst);
        uint32_t m0_saved;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 2\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "global_load_lds_dwordx4 %4, %3 offset:1024\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(m0_saved)
            : "v"(off_even), "s"(d), "s"(sbase), "v"(off_odd)
            : "memory");
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N) : "memory");
}

// kSlots ring slots, kPieces pieces per wave per operand per step (step bytes = 2 operands x 4 waves x kPieces KiB)
template <int kSlots, int kPieces, bool kBarrier, int kMfma, int kValu, int kLds, bool kDma, int kFly>
__global__ __launch_bounds__(256) void stream(const char* a, const char* b, int64_t steps, int order, int layout, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kOperandBytes = 4 * kPieces * 1024;          // per step
    constexpr int kSlotBytes = 2 * kOperandBytes;
    constexpr int kAhead = kSlots - 1;
    constexpr int kPerStep = 2 * kPieces;                      // vector-memory instructions per wave and step
    const char* pa = a + (int64_t)blockIdx.x * steps * kOperandBytes + wave * kPieces * 1024;
    const char* pb = b + (int64_t)blockIdx.x * steps * kOperandBytes + wave * kPieces * 1024;
    int64_t pair_stride = 2 * kOperandBytes;                   // from one pair of steps (a 32-sample tile) to the next
    if (layout != 0) {
        const int job = blockIdx.x / 128, split = blockIdx.x % 128;
        const int64_t first = layout == 1 ? (int64_t)split * steps * kOperandBytes : (int64_t)split * 2 * kOperandBytes;
        pa = a + ((int64_t)(job + 5) << 28) + first + wave * kPieces * 1024;
        pb = a + ((int64_t)job << 28) + first + wave * kPieces * 1024;
        if (layout == 2) pair_stride = (int64_t)128 * 2 * kOperandBytes;
    }
    auto at = [&](const char* p, int64_t t) { return p + (t >> 1) * pair_stride + (t & 1) * kOperandBytes; };
    const int sample = 4 * ((lane >> 2) & 3) + (lane >> 4);
    uint32_t off_even = lane * 16, off_odd = lane * 16;
    if (order == 1) off_even = ((lane & 3) * 16 + sample) * 16, off_odd = off_even ^ 64;
    if (order == 2) off_even = (sample * 4 + (lane & 3)) * 16, off_odd = off_even ^ 256;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem + wave * kPieces * 1024;

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 opa, opb;
#pragma unroll
    for (int i = 0; i < 8; ++i) opa[i] = (_Float16)(lane * 0.001f + i), opb[i] = (_Float16)(i - lane * 0.002f);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    float got = 0.f;

    auto issue = [&](int64_t t, int slot) {
        if constexpr (kDma) {
            dma_n<kPieces>(at(pa, t), lds0 + slot * kSlotBytes, off_even, off_odd);
            dma_n<kPieces>(at(pb, t), lds0 + slot * kSlotBytes + kOperandBytes, off_even, off_odd);
        }
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t)
        if (t < steps) issue(t, t % kSlots);
    int slot = kAhead % kSlots, cur = 0;
    for (int64_t t = 0; t < steps; ++t) {
        if (t + kAhead < steps) issue(t + kAhead, slot);
        if constexpr (kLds > 0) {
            const uint32_t base = (uint32_t)(uintptr_t)smem + cur * kSlotBytes + lane * 4;
#pragma unroll
            for (int i = 0; i < kLds; ++i) {
                float2 x;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%c2 offset1:%c3" : "=v"(x) : "v"(base), "n"((2 * i) % 32), "n"((2 * i + 1) % 32));
                got += x.x;
            }
        }
#pragma unroll
        for (int m = 0; m < kMfma; ++m) {
            acc[m % 16] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opa, opb, acc[m % 16], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kValu; ++q) {
                float& x = v[(m * kValu + q) % 8];
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(1.0001f), "v"(0.5f));
            }
        }
        if (kLds > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + kAhead < steps) wait_vm<kFly * kPerStep>();         // the youngest kFly steps may still fly
        else wait_vm<0>();
        if (kBarrier) __builtin_amdgcn_s_barrier();
        slot = slot + 1 == kSlots ? 0 : slot + 1;
        cur = cur + 1 == kSlots ? 0 : cur + 1;
    }
    float keep = got;
#pragma unroll
    for (int i = 0; i < 16; ++i) keep += acc[i][0] + acc[i][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += v[i];
    if (keep == 12345.f) sink[0] = keep;
}

template <int kSlots, int kPieces, bool kBarrier, int kMfma = 0, int kValu = 0, int kLds = 0, bool kDma = true, int kFly = kSlots - 1>
static float run(const char* a, const char* b, int64_t total_steps32, int wgs_per_cu, int order, float* sink, int layout = 0) {
    const int grid = layout ? 768 : 256 * wgs_per_cu;
    // total_steps32 counts 32 KiB steps over the whole chip; a 16 KiB step is half of one
    const int64_t steps = layout ? 128 : total_steps32 * (4 / kPieces) / grid;      // (the kernel: 128 steps per split)
    const int lds = kSlots * 2 * 4 * kPieces * 1024;
    static_assert(kFly >= 1 && kFly <= kSlots - 1, "steps in flight at the hand-over");
    auto k = stream<kSlots, kPieces, kBarrier, kMfma, kValu, kLds, kDma, kFly>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, a, b, steps, order, layout, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double bytes = (double)steps * grid * 2 * 4 * kPieces * 1024;
    printf("layout %d slots %d fly %d step %2d KiB wgs/CU %d barrier %d order %d dma %d mfma %2d valu/mfma %d lds %2d: %.3f ms  %.2f TB/s\n", layout, kSlots, kFly,
           2 * 4 * kPieces, wgs_per_cu, (int)kBarrier, order, (int)kDma, kMfma, kValu, kLds, best, kDma ? bytes / best * 1e-9 : 0.0);
    fflush(stdout);
    return best;
}

int main() {
    const int64_t total_steps32 = 256 * 300;                  // 300 k-steps of 32 KiB per CU = 2.52 GB, the kernel's stream
    const size_t half = (size_t)total_steps32 * 16384;
    char *a, *b, *w;
    float* sink;
    CHECK(hipMalloc(&a, half));
    CHECK(hipMalloc(&b, half));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, half));
    CHECK(hipMemset(b, 2, half));
    CHECK(hipMalloc(&w, (size_t)11 << 28));              // the workspace's ten 2^28-byte arrays (layout 1 / 2)
    CHECK(hipMemset(w, 3, (size_t)11 << 28));
    // the stream alone: the kernel's ring on the probe's addresses, then on the kernel's
    run<4, 4, true>(a, b, total_steps32, 1, 0, sink);
    run<4, 4, true>(a, b, total_steps32, 1, 1, sink);
    run<4, 4, true>(w, w, total_steps32, 1, 0, sink, 1);
    run<4, 4, true>(w, w, total_steps32, 1, 1, sink, 1);
    run<4, 4, true>(w, w, total_steps32, 1, 0, sink, 2);
    run<4, 4, true>(w, w, total_steps32, 1, 1, sink, 2);
    run<4, 4, true, 0, 0, 0, true, 1>(w, w, total_steps32, 1, 0, sink, 1);
    run<4, 4, true, 0, 0, 0, true, 1>(w, w, total_steps32, 1, 0, sink, 2);
    // the arithmetic alone (every step a hidden-layer step, 300 per CU; the kernel runs 256 of them + 128 small ones
    // per CU — compare shapes, not milliseconds)
    run<4, 4, true, 48, 0, 0, false>(a, b, total_steps32, 1, 0, sink);
    run<4, 4, true, 48, 6, 0, false>(a, b, total_steps32, 1, 0, sink);
    run<4, 4, true, 48, 6, 32, false>(a, b, total_steps32, 1, 0, sink);
    // both
    run<4, 4, true, 48, 6, 0, true, 1>(a, b, total_steps32, 1, 0, sink);
    run<4, 4, true, 48, 6, 0, true, 1>(w, w, total_steps32, 1, 0, sink, 1);
    run<4, 4, true, 48, 6, 0, true, 1>(w, w, total_steps32, 1, 0, sink, 2);
    run<4, 4, true, 48, 6, 32, true, 1>(a, b, total_steps32, 1, 0, sink);
    run<4, 4, true, 48, 6, 32, true, 1>(w, w, total_steps32, 1, 0, sink, 1);
    run<4, 4, true, 48, 6, 32, true, 1>(w, w, total_steps32, 1, 0, sink, 2);
    run<4, 4, true, 48, 6, 32, true, 1>(w, w, total_steps32, 1, 1, sink, 2);
    run<4, 4, false, 48, 6, 32, true, 1>(w, w, total_steps32, 1, 0, sink, 1);
    run<4, 4, false, 48, 6, 32, true, 1>(w, w, total_steps32, 1, 0, sink, 2);
    return 0;
}

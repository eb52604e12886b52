// Does plain fp32 VALU work execute BESIDE the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles) the way it does
// beside the 16-bit ones?  (gfx950.)  The render kernels' counters say no: SQ_VALU_MFMA_BUSY_CYCLES + 4 x (non-MFMA
// SQ_ACTIVE_INST_VALU) add up to 96-99 % of the SIMD cycles in the headline kernel AND in the hidden-64 one, whatever
// the wave count, barriers or priorities.  This probe asks the hardware directly:
//   part 1 (one wave per SIMD): 1 MFMA + k independent v_fma_f32 per gap, k = 0..8 -> cycles per gap, for the fp32
//          16x16x4 MFMA and, as the control, the 32x32x16 f16 MFMA (where 4 fillers are free: valu_issue_cost.hip);
//   part 2 (two waves per SIMD): waves 0-3 stream MFMAs, waves 4-7 stream v_fma_f32; each side alone and both
//          together -> cycles for the same instruction counts;
//   part 3: the same with the VALU waves at s_setprio 2 (what the render kernels do) and 16 / 2 / 1 dependent chains.
// Result (profiles/r06_h_mfma32_valu_coexec_probe.log, NOTES.md section R6d): beside the fp32 MFMA stream the VALU
// wave makes NO progress at any priority (its time = the MFMA stream's + its own), beside the f16 stream it runs at
// 0.8 of its own rate; a VALU instruction between two of the wave's own fp32 MFMAs costs 15 ticks, then 4.4 each.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma32_valu_coexec.hip -o /tmp/mfma32_valu_coexec
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

template <bool kF32, int kFill>
__device__ __forceinline__ void gap_stream(int iters, float* sink) {
    float r[16];
    for (int c = 0; c < 16; ++c) r[c] = 1.0f + threadIdx.x * 1e-3f + c;
    const float k0 = 1.0000001f, k1 = 1e-9f;
    f32x4 acc4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x16 acc16;
    for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)(0.001f * (threadIdx.x + i));
        bh[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (kF32) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[g & 1]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc16) : "v"(ah), "v"(bh));
#pragma unroll
            for (int f = 0; f < kFill; ++f) {
                const int c = (g * kFill + f) & 15;
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(k0), "v"(k1));
            }
        }
    }
    float s = acc4[0].x + acc4[1].y + acc16[3];
    for (int c = 0; c < 16; ++c) s += r[c];
    *sink = s;
}

template <bool kF32, int kFill>
__global__ __launch_bounds__(256, 1) void gaps(float* out, unsigned long long* cycles, int iters, int slot) {
    float s;
    const unsigned long long t0 = __builtin_readcyclecounter();
    gap_stream<kF32, kFill>(iters, &s);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[slot] = t1 - t0;
}

// mode bit 0: waves 0-3 stream MFMAs; bit 1: waves 4-7 stream v_fma_f32 (16 independent chains)
template <bool kF32, int kValuPrio = 0, int kChains = 16>
__global__ __launch_bounds__(512, 1) void pair(float* out, unsigned long long* cycles, int iters, int mode, int slot) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) gap_stream<kF32, 0>(iters, &s);           // 16 MFMAs per iteration
    } else {
        if (mode & 2) {
            __builtin_amdgcn_s_setprio(kValuPrio);
            float r[16];
            for (int c = 0; c < 16; ++c) r[c] = 1.0f + threadIdx.x * 1e-3f + c;
            const float k0 = 1.0000001f, k1 = 1e-9f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 64; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[q % kChains]) : "v"(k0), "v"(k1));
            }
            for (int c = 0; c < 16; ++c) s += r[c];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cycles[slot + (threadIdx.x ? 1 : 0)] = t1 - t0;
}

template <bool kF32, int kFill>
int run_gap(float* out, unsigned long long* cyc, int cus, int iters, int slot) {
    hipLaunchKernelGGL((gaps<kF32, kFill>), dim3(cus), dim3(256), 0, 0, out, cyc, iters, slot);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 512 * 4));
    CHECK(hipMalloc(&cyc, 64 * 8));
    CHECK(hipMemset(cyc, 0, 64 * 8));
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {      // first pass warms up
        int bad = 0;
        bad |= run_gap<true, 0>(out, cyc, cus, iters, 0);
        bad |= run_gap<true, 1>(out, cyc, cus, iters, 1);
        bad |= run_gap<true, 2>(out, cyc, cus, iters, 2);
        bad |= run_gap<true, 3>(out, cyc, cus, iters, 3);
        bad |= run_gap<true, 4>(out, cyc, cus, iters, 4);
        bad |= run_gap<true, 6>(out, cyc, cus, iters, 5);
        bad |= run_gap<true, 8>(out, cyc, cus, iters, 6);
        bad |= run_gap<false, 0>(out, cyc, cus, iters, 8);
        bad |= run_gap<false, 2>(out, cyc, cus, iters, 9);
        bad |= run_gap<false, 4>(out, cyc, cus, iters, 10);
        bad |= run_gap<false, 6>(out, cyc, cus, iters, 11);
        bad |= run_gap<false, 8>(out, cyc, cus, iters, 12);
        if (bad) return 1;
        for (int f32 = 1; f32 >= 0; --f32)
            for (int mode = 1; mode <= 3; ++mode) {
                const int slot = 16 + (f32 ? 0 : 8) + 2 * (mode - 1);
                if (f32) hipLaunchKernelGGL((pair<true>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, slot);
                else hipLaunchKernelGGL((pair<false>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, slot);
                CHECK(hipDeviceSynchronize());
            }
    }
    // part 3: the VALU wave at s_setprio 2 (what the render kernels do), 16 / 2 / 1 independent chains
    for (int mode = 2; mode <= 3; ++mode) {
        const int o = 2 * (mode - 2);
        hipLaunchKernelGGL((pair<true, 2, 16>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 32 + o);
        hipLaunchKernelGGL((pair<true, 2, 2>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 36 + o);
        hipLaunchKernelGGL((pair<true, 2, 1>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 40 + o);
        hipLaunchKernelGGL((pair<false, 2, 16>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 44 + o);
        hipLaunchKernelGGL((pair<false, 2, 2>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 48 + o);
        hipLaunchKernelGGL((pair<false, 2, 1>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 52 + o);
        hipLaunchKernelGGL((pair<true, 0, 2>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, mode, 56 + o);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[64];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double gaps_n = (double)iters * 16;
    const int fills[7] = {0, 1, 2, 3, 4, 6, 8};
    printf("part 1, one wave per SIMD: clock ticks per MFMA gap (s_memtime ticks; 1 MFMA + k v_fma_f32)\n");
    for (int i = 0; i < 7; ++i) printf("  v_mfma_f32_16x16x4_f32  + %d fillers: %.1f\n", fills[i], h[i] / gaps_n);
    const int fills16[5] = {0, 2, 4, 6, 8};
    for (int i = 0; i < 5; ++i) printf("  v_mfma_f32_32x32x16_f16 + %d fillers: %.1f\n", fills16[i], h[8 + i] / gaps_n);
    printf("part 2, two waves per SIMD: wave 0 streams %d MFMAs, wave 4 streams %d v_fma_f32 (ticks for the whole stream)\n",
           iters * 16, iters * 64);
    for (int f32 = 1; f32 >= 0; --f32) {
        const int base = 16 + (f32 ? 0 : 8);
        printf("  %s: MFMA alone %llu | VALU alone %llu | together: MFMA wave %llu, VALU wave %llu\n",
               f32 ? "16x16x4_f32 " : "32x32x16_f16", h[base + 0], h[base + 2 + 1], h[base + 4], h[base + 4 + 1]);
    }
    printf("part 3, the VALU wave at s_setprio 2 with 16 / 2 / 1 independent chains (MFMA alone: see part 2)\n");
    const char* names[7] = {"16x16x4_f32  prio 2, 16 chains", "16x16x4_f32  prio 2,  2 chains", "16x16x4_f32  prio 2,  1 chain ",
                            "32x32x16_f16 prio 2, 16 chains", "32x32x16_f16 prio 2,  2 chains", "32x32x16_f16 prio 2,  1 chain ",
                            "16x16x4_f32  prio 0,  2 chains"};
    for (int v = 0; v < 7; ++v) {
        const int base = 32 + 4 * v;
        printf("  %s: VALU alone %llu | together: MFMA wave %llu, VALU wave %llu\n", names[v], h[base + 1], h[base + 2], h[base + 3]);
    }
    return 0;
}

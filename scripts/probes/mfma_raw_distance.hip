// Hardware-semantics probe (gfx950): hipcc separates an XDL MFMA from the first VALU instruction that
// reads its result by a FIXED number of wait states (8 for v_mfma_f32_16x16x32_f16: the instruction
// is modelled as 4 passes) and the hardware does not interlock that dependency.  Is that distance
// still enough when the SIMD's other wave keeps the matrix pipe busy, so that this wave's MFMA has
// to queue behind the partner's?  A result that is read too early shows in the LAST rows the MFMA
// writes: lanes 48-63 of the 16x16 accumulator tile.
//   v_mfma_f32_16x16x32_f16 v[100:103], a, b, c ; kDist x v_add_f32 filler ; v_mov out, v100..v103
//   reference: same MFMA, 64 idle wait states, then the reads
// mode 0: every wave runs the probe; mode 1: the second resident workgroup of every CU stream MFMAs back to back.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_raw_distance.hip -o /tmp/mfma_raw_distance
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int kDist>
__global__ __launch_bounds__(256) void probe(unsigned long long* counts, float* sink, int iters, int hog) {
    const int lane = threadIdx.x & 63;
    if (hog && blockIdx.x >= gridDim.x / 2) {   // second resident workgroup of every CU (round-robin dispatch)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        h8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * i);
        for (int it = 0; it < iters * 4; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3];
        return;
    }
    unsigned long long bad_lo = 0, bad_hi = 0;
    float sum = 0.f;
    for (int it = 0; it < iters; ++it) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)(0.125f * (float)((lane * 3 + i + it) & 15));
            b[i] = (_Float16)(0.25f * (float)((lane + 5 * i + 2 * it) & 7));
        }
        const f32x4 c = {(float)it, 1.f, 2.f, 3.f};
        f32x4 got, ref, old = {-7.f, -7.f, -7.f, -7.f};
        float filler = 0.f;
        asm volatile(
            "v_mov_b32 v100, %[old]\n\tv_mov_b32 v101, %[old]\n\tv_mov_b32 v102, %[old]\n\tv_mov_b32 v103, %[old]\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], %[a], %[b], %[c]\n\t"
            ".rept %c[n]\n\t"
            "v_add_f32 %[f], 1.0, %[f]\n\t"
            ".endr\n\t"
            "v_mov_b32 %[g0], v100\n\tv_mov_b32 %[g1], v101\n\tv_mov_b32 %[g2], v102\n\tv_mov_b32 %[g3], v103\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mfma_f32_16x16x32_f16 v[104:107], %[a], %[b], %[c]\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %[r0], v104\n\tv_mov_b32 %[r1], v105\n\tv_mov_b32 %[r2], v106\n\tv_mov_b32 %[r3], v107"
            : [g0] "=&v"(got.x), [g1] "=&v"(got.y), [g2] "=&v"(got.z), [g3] "=&v"(got.w),
              [r0] "=&v"(ref.x), [r1] "=&v"(ref.y), [r2] "=&v"(ref.z), [r3] "=&v"(ref.w), [f] "+&v"(filler)
            : [a] "v"(a), [b] "v"(b), [c] "v"(c), [old] "v"(old.x), [n] "n"(kDist)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
        for (int i = 0; i < 4; ++i)
            if (__float_as_uint(got[i]) != __float_as_uint(ref[i])) {
                if (lane >= 48) ++bad_hi; else ++bad_lo;
            }
        sum += filler + got.x + ref.w;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = sum;
    if (bad_lo) atomicAdd(counts, bad_lo);
    if (bad_hi) atomicAdd(counts + 1, bad_hi);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int kDist>
static int run(unsigned long long* cnt, float* sink, int iters, int grid, int hog) {
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL((probe<kDist>), dim3(grid), dim3(256), 0, 0, cnt, sink, iters, hog);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    printf("first read %2d VALU instructions behind the MFMA, %s: %llu stale values in lanes 0-47, %llu in lanes 48-63\n", kDist,
           hog ? "partner workgroups stream MFMAs" : "all waves run the probe", h[0], h[1]);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;
    unsigned long long* cnt;
    float* sink;
    CK(hipMalloc(&cnt, 16));
    CK(hipMalloc(&sink, (size_t)grid * 256 * 4));
    for (int hog = 0; hog < 2; ++hog) {
        if (run<2>(cnt, sink, iters, grid, hog)) return 1;
        if (run<4>(cnt, sink, iters, grid, hog)) return 1;
        if (run<6>(cnt, sink, iters, grid, hog)) return 1;
        if (run<7>(cnt, sink, iters, grid, hog)) return 1;
        if (run<8>(cnt, sink, iters, grid, hog)) return 1;
        if (run<10>(cnt, sink, iters, grid, hog)) return 1;
        if (run<12>(cnt, sink, iters, grid, hog)) return 1;
        if (run<16>(cnt, sink, iters, grid, hog)) return 1;
        if (run<24>(cnt, sink, iters, grid, hog)) return 1;
    }
    printf("done\n");
    return 0;
}

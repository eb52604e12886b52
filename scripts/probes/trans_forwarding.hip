// Hardware-semantics probe (gfx950): is the result of a transcendental VALU instruction (v_exp_f32:
// quarter rate, the last 16 lanes finish last) always visible to a dependent VALU instruction that
// issues `kDist` instructions later, also when the SIMD's OTHER wave keeps the transcendental unit
// busy?  hipcc only guarantees ONE wait state between a trans op and its consumer; the render
// kernels' front ends showed garbage in single encoded features, lanes 48-63 only, and only with two
// waves per SIMD.
//   every wave: x = v_exp_f32(a); kDist independent v_add_f32; y = v_mul_f32(x, 1.0)   (all in asm)
//   reference:  the same v_exp_f32 followed by s_nop 15 x 4 before it is read
// the second resident workgroup of every CU run a back-to-back stream of v_exp_f32 / v_rcp_f32 instead (pure contention).
// Prints mismatching results per distance, split into lanes 0-47 / 48-63.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/trans_forwarding.hip -o /tmp/trans_forwarding
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int kDist>
__global__ __launch_bounds__(256) void probe(unsigned long long* counts, float* sink, int iters, int hog) {
    const int lane = threadIdx.x & 63;
    unsigned long long bad_lo = 0, bad_hi = 0;
    float acc = 0.f;
    if (hog && blockIdx.x >= gridDim.x / 2) {   // second resident workgroup of every CU (round-robin dispatch)              // contention only: trans ops back to back
        float a = 0.001f * lane, b = 1.0f + 0.01f * lane;
        for (int it = 0; it < iters * 4; ++it) {
            asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\t"
                         "v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\t"
                         "v_mul_f32 %0, 0.001, %0" : "+v"(a), "+v"(b));
        }
        sink[blockIdx.x * 256 + threadIdx.x] = a + b;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        const float a = -3.0f + 6.0f * (float)((it * 64 + lane * 7919) & 1023) * (1.0f / 1024.0f);
        float x = 1234.5f, y, filler = (float)it, ref;
        asm volatile(
            "s_nop 7\n\t"
            "v_exp_f32 %[x], %[a]\n\t"
            ".rept %c[d]\n\t"
            "v_add_f32 %[f], 1.0, %[f]\n\t"
            ".endr\n\t"
            "v_mul_f32 %[y], 1.0, %[x]\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            "v_exp_f32 %[r], %[a]\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %[r], %[r]"
            : [x] "+&v"(x), [y] "=&v"(y), [f] "+&v"(filler), [r] "=&v"(ref)
            : [a] "v"(a), [d] "n"(kDist));
        if (__float_as_uint(y) != __float_as_uint(ref)) {
            if (lane >= 48) ++bad_hi; else ++bad_lo;
        }
        acc += filler + y;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
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
    printf("consumer %d instructions behind v_exp_f32, %s: %llu wrong in lanes 0-47, %llu wrong in lanes 48-63\n", kDist,
           hog ? "partner workgroups stream trans ops" : "all waves run the probe", h[0], h[1]);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;         // 2 workgroups per CU: 2 waves per SIMD
    unsigned long long* cnt;
    float* sink;
    CK(hipMalloc(&cnt, 16));
    CK(hipMalloc(&sink, (size_t)grid * 256 * 4));
    for (int hog = 0; hog < 2; ++hog) {
        if (run<0>(cnt, sink, iters, grid, hog)) return 1;
        if (run<1>(cnt, sink, iters, grid, hog)) return 1;
        if (run<2>(cnt, sink, iters, grid, hog)) return 1;
        if (run<3>(cnt, sink, iters, grid, hog)) return 1;
        if (run<5>(cnt, sink, iters, grid, hog)) return 1;
        if (run<8>(cnt, sink, iters, grid, hog)) return 1;
    }
    printf("done\n");
    return 0;
}

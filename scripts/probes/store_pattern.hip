// Memory-system probe (gfx950): what do the training kernels' ROW-MAJOR saves cost?  A wave owns 16
// samples; lane (j = lane & 15, g = lane >> 4) holds features 16 T + 4 g + r of sample j, so one
// global_store_dwordx4 writes sixteen 64-byte segments 1 KiB apart (half a 128-byte line each), and
// the sixteen stores of a row tile come back to back (data gradient) or spread out (forward).
//   mode 0  rows, 16 stores back to back            (nerf_bwd_data_kernel's dY saves)
//   mode 1  contiguous: each store writes 1 KiB      (a tile-major layout)
//   mode 2  rows, a pause after every store          (the training forward's x_hat saves)
//   mode 3  rows, sample pairs share a store so that every store writes whole 128-byte lines
// Prints time and rate per mode; run under `rocprofv3 --pmc WRITE_SIZE` for the bytes the L2 sends out.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/store_pattern.hip -o gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int kMode>
__global__ __launch_bounds__(256) void probe(float* dst, int64_t tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        float* base = dst + tile * 16 * 256;
        f32x4 v = {(float)tile, (float)lane, 1.f, 2.f};
        if (kMode == 1) {
#pragma unroll
            for (int T = 0; T < 16; ++T) *(f32x4*)(base + T * 256 + lane * 4) = v;
        } else if (kMode == 3) {
            // rows j and j ^ 1 pair up: one store writes the even row's whole line (tiles T, T + 1) with the
            // eight lanes of both rows, the next one the odd row's (in the kernel: 4 DPP moves per store)
#pragma unroll
            for (int T = 0; T < 16; T += 2) {
                const int off = T * 16 + 16 * (j & 1) + 4 * g;
                *(f32x4*)(base + (j & ~1) * 256 + off) = v;
                *(f32x4*)(base + (j | 1) * 256 + off) = v;
            }
        } else {
            float* row = base + j * 256 + 4 * g;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                *(f32x4*)(row + T * 16) = v;
                if (kMode == 2) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
}

template <int kMode>
static void run(float* dst, int64_t tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<kMode><<<512, 256>>>(dst, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<kMode><<<512, 256>>>(dst, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)tiles * 16 * 1024;
    printf("mode %d: %.3f ms per launch, %.2f GB written, %.0f GB/s\n", kMode, ms / 5, bytes / 1e9,
           bytes / (ms / 5 * 1e-3) / 1e9);
}

int main() {
    const int64_t tiles = 81920;                     // 1.34 GB = five dY saves of 4096 x 64 samples
    float* dst;
    if (hipMalloc(&dst, tiles * 16 * 1024) != hipSuccess) return 1;
    run<0>(dst, tiles);
    run<1>(dst, tiles);
    run<2>(dst, tiles);
    run<3>(dst, tiles);
    return 0;
}

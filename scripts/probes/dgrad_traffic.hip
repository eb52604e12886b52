// Memory-system probe (gfx950): the split-precision data gradient's HBM traffic WITHOUT its arithmetic.
// nerf_bwd_data_h_kernel moves, per 16-sample item and hidden layer, one saved x_hat tile in (16 loads of
// 16 B per lane + a 4-byte 1/std) and one dY tile out (16 stores), as bursts of a wave that then computes for
// microseconds; two 4-wave workgroups per CU (LDS-limited), 2,048 waves on the chip, 2.97 GB per 4096 x 64 batch.
// How long does that traffic take by itself, at that occupancy, in that pattern?
//   mode 0  rows: lane (j, g) touches 16 B at [sample j][16 T + 4 g] — sixteen 64-byte segments 1 KiB apart per
//           instruction (the product's row-major workspace)
//   mode 1  tile-contiguous: each instruction touches 1 KiB (what a tile-major workspace would give)
//   pause   s_sleep units between the burst of one layer and the next (0: back to back; the kernel computes ~20 us)
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/dgrad_traffic.hip -o gpurun_out/dgrad_traffic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// kMode 2 / 3: the kernel's order and coupling — the 16 saves (computed data) FIRST, then the 17 independent loads,
// consumed only after the pause; mode 3 also puts the workgroup's four waves through a barrier in front of every
// burst, as the weight ring's per-stage barrier does in the kernel (all four burst at the same instant)
template <int kMode>
__global__ __launch_bounds__(256, 2) void probe(const float* xhat, float* dy, int64_t items, int64_t mp, int pause,
                                                float* sink) {
    extern __shared__ char smem[];                  // 74 KiB: two workgroups per CU, like the kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < items; item += (int64_t)gridDim.x * 4) {
        for (int L = 4; L >= 0; --L) {
            const float* src = xhat + (int64_t)L * mp * 256 + item * 16 * 256;
            float* dst = dy + (int64_t)L * mp * 256 + item * 16 * 256;
            f32x4 x[16];
            if (kMode >= 2) {
                if (kMode == 3) __syncthreads();
#pragma unroll
                for (int T = 0; T < 16; ++T) *(f32x4*)(dst + j * 256 + 4 * g + T * 16) = keep + (float)T;
#pragma unroll
                for (int T = 0; T < 16; ++T) x[T] = *(const f32x4*)(src + j * 256 + 4 * g + T * 16);
                for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);
                keep = keep + x[3] + x[7] + x[11];
                continue;
            }
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                const int off = kMode == 0 ? j * 256 + 4 * g + T * 16 : T * 256 + lane * 4;
                x[T] = *(const f32x4*)(src + off);
            }
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                const int off = kMode == 0 ? j * 256 + 4 * g + T * 16 : T * 256 + lane * 4;
                *(f32x4*)(dst + off) = x[T] + keep;
            }
            keep = keep + x[3];
            for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);
        }
    }
    if (keep.x == 12345.f) sink[0] = keep.y;
    if (threadIdx.x == 0 && blockIdx.x == 0xffffff) smem[0] = 1;
}

template <int kMode>
static void run(const float* x, float* y, int64_t items, int64_t mp, int pause, float* sink, int grid = 512) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<kMode>, hipFuncAttributeMaxDynamicSharedMemorySize, 75776);
    probe<kMode><<<grid, 256, 75776>>>(x, y, items, mp, pause, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<kMode><<<grid, 256, 75776>>>(x, y, items, mp, pause, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)items * 5 * 2 * 16 * 1024;
    const double waves = grid * 4.0, bursts = (double)items * 5 / waves;     // bursts (one layer: 17 loads + 16 stores) per wave
    printf("mode %d pause %3d grid %4d: %.3f ms per launch, %.2f GB moved (half in, half out), %.0f GB/s, %.2f us per burst of a wave\n",
           kMode, pause, grid, ms / 5, bytes / 1e9, bytes / (ms / 5 * 1e-3) / 1e9, ms / 5 * 1e3 / bursts);
}

int main() {
    const int64_t items = 16384, mp = items * 16;     // 4096 rays x 4 chunks
    float *x, *y, *sink;
    if (hipMalloc(&x, mp * 256 * 4 * 5) != hipSuccess || hipMalloc(&y, mp * 256 * 4 * 5) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    hipMemset(x, 0, mp * 256 * 4 * 5);
    for (int pause : {0, 4, 16, 64}) {
        run<0>(x, y, items, mp, pause, sink);
        run<1>(x, y, items, mp, pause, sink);
    }
    // fewer workgroups (one per CU, one per 2, 4, 8, 32 CUs): how long ONE wave's burst takes when the memory
    // system is not saturated — the latency-bound floor of a burst
    for (int grid : {256, 128, 64, 32, 8}) {
        run<0>(x, y, items / 8, mp, 0, sink, grid);
        run<1>(x, y, items / 8, mp, 0, sink, grid);
    }
    // the kernel's regime: every wave pauses between bursts (it computes), so the memory system is not saturated and a
    // burst's duration is what the wave waits for: rows against tile-contiguous
    for (int pause : {1, 2, 4, 8}) {
        run<0>(x, y, items, mp, pause, sink);
        run<1>(x, y, items, mp, pause, sink);
    }
    for (int pause : {0, 2, 4, 6, 8}) {
        run<2>(x, y, items, mp, pause, sink);
        run<3>(x, y, items, mp, pause, sink);
    }
    return 0;
}

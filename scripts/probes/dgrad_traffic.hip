// Memory-system probe (gfx950): the split-precision data gradient's HBM traffic WITHOUT its arithmetic.
// nerf_bwd_data_h_kernel moves, per 16-sample item and hidden layer, one saved x_hat tile in (16 loads of
// 16 B per lane + a 4-byte 1/std) and one dY tile out (16 stores), as bursts of a wave that then computes for
// microseconds; two 4-wave workgroups per CU (LDS-limited), 2,048 waves on the chip, 2.97 GB per 4096 x 64 batch.
// How long does that traffic take by itself, at that occupancy, in that pattern?
//   mode 0  rows: lane (j, g) touches 16 B at [sample j][16 T + 4 g] — sixteen 64-byte segments 1 KiB apart per
//           instruction (the product's row-major workspace)
//   mode 1  tile-contiguous: each instruction touches 1 KiB (what a tile-major workspace would give)
//   mode 4  as mode 1 through buffer instructions (one descriptor per tensor, lane offset in a VGPR, tile base in an SGPR)
//   mode 5  as mode 1 with 8-byte accesses (twice the instructions): is a burst paid per instruction or per byte?
//   mode 6  as mode 1, the loads as LDS-DMA (global_load_lds_dwordx4) + ds_read_b128 instead of register loads
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
            if (kMode == 4) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xhat + (int64_t)L * mp * 256), 0, 0x7fffffff, 0x27000);
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)L * mp * 256), 0, 0x7fffffff, 0x27000);
                const int base = __builtin_amdgcn_readfirstlane((int)(item * 16 * 256 * 4));
                typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    x[T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + T * 1024, base, 0));
#pragma unroll
                for (int T = 0; T < 16; ++T)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, x[T] + keep), rd, lane * 16 + T * 1024, base, 0);
                keep = keep + x[3];
                for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);
                continue;
            }
            if (kMode == 5) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 h[32];
#pragma unroll
                for (int T = 0; T < 32; ++T) h[T] = *(const f32x2*)(src + T * 128 + lane * 2);
#pragma unroll
                for (int T = 0; T < 32; ++T) *(f32x2*)(dst + T * 128 + lane * 2) = h[T] + f32x2{keep.x, keep.y};
                keep.x += h[3].x;
                for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);
                continue;
            }
            if (kMode == 6) {
                // 16 KiB of this wave's LDS (4 waves x 16 KiB = 64 KiB of the 74 KiB block)
                char* mine = smem + (threadIdx.x >> 6) * 16384;
                const uint64_t sb = (uint64_t)(uintptr_t)src;
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
                const uint64_t sbase = ((uint64_t)hi << 32) | lo;
                for (int q = 0; q < 4; ++q) {
                    const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mine + q * 4096));
                    uint32_t saved;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\t"
                                 "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                                 "global_load_lds_dwordx4 %1, %3 offset:2048\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(saved) : "v"(lane * 16 + q * 4096), "s"(d), "s"(sbase) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int T = 0; T < 16; ++T) x[T] = *(const f32x4*)(mine + T * 1024 + lane * 16);
#pragma unroll
                for (int T = 0; T < 16; ++T) *(f32x4*)(dst + T * 256 + lane * 4) = x[T] + keep;
                keep = keep + x[3];
                for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(100);
                continue;
            }
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
    // round 4: other ways of moving the same contiguous KiB per instruction — one wave's burst on a quiet chip and in
    // the kernel's regime
    for (int grid : {256, 32, 8}) {
        run<1>(x, y, items / 8, mp, 0, sink, grid);
        run<4>(x, y, items / 8, mp, 0, sink, grid);
        run<5>(x, y, items / 8, mp, 0, sink, grid);
        run<6>(x, y, items / 8, mp, 0, sink, grid);
    }
    for (int pause : {0, 4}) {
        run<1>(x, y, items, mp, pause, sink);
        run<4>(x, y, items, mp, pause, sink);
        run<5>(x, y, items, mp, pause, sink);
        run<6>(x, y, items, mp, pause, sink);
    }
    return 0;
}

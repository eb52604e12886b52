// Issue-cost probe (gfx950): how long ONE wave is held by a global_load_lds_dwordx4 (LDS-DMA, 1 KiB per instruction),
// by the order in which its lanes fetch the 16-byte chunks of the KiB.  One wave per SIMD (one 256-thread workgroup per
// CU), the source resident in L2 (the same 64 KiB per wave over and over), s_memtime around 64 x 16 instructions, a
// vmcnt(0) every 16 so the queue never fills.
//   order 0  lane l fetches chunk l (linear)
//   order 1  the weight gradient's re-ordered chunks: LDS chunk (s & 3, s >> 2, g) <- memory chunk (g, s)
//   order 2  sample-major memory chunks (s, g): lane quads contiguous, quads scattered
//   order 3  lanes 256 B apart (every lane of a quad in another 128-byte line)
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/dma_issue_cost.hip -o /tmp/dma_issue_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void probe(const char* src, unsigned long long* cycles, int order, int iters, int wait_every) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sample = 4 * ((lane >> 2) & 3) + (lane >> 4);
    uint32_t off = lane * 16;
    if (order == 1) off = ((lane & 3) * 16 + sample) * 16;
    if (order == 2) off = (sample * 4 + (lane & 3)) * 16;
    if (order == 3) off = ((lane & 3) * 16 + (lane >> 2)) * 16;
    const char* p = src + ((size_t)blockIdx.x * 4 + wave) * 65536;
    const uint64_t base_u = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base_u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base_u >> 32));
    const uint64_t sbase = ((uint64_t)hi << 32) | lo;
    const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(smem + wave * 16384));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 2" ::"s"(d));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(off + (k >> 2) * 4096), "s"(sbase), "n"((k & 3) * 1024) : "memory");
            if (wait_every == 4 && (k & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
    char* src;
    unsigned long long* cyc;
    hipMalloc(&src, (size_t)256 * 4 * 65536);
    hipMemset(src, 1, (size_t)256 * 4 * 65536);
    hipMalloc(&cyc, 64);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int waves : {4, 1})                       // waves per CU issuing at the same time
        for (int grid : {1, 256})
            for (int wait_every : {16, 4})
                for (int order = 0; order < 4; ++order) {
                    const int iters = 64;
                    probe<<<grid, 64 * waves, 65536>>>(src, cyc, order, iters, wait_every);
                    hipDeviceSynchronize();
                    probe<<<grid, 64 * waves, 65536>>>(src, cyc, order, iters, wait_every);
                    hipDeviceSynchronize();
                    unsigned long long c;
                    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                    printf("waves per CU %d, workgroups %3d, vmcnt(0) every %2d, order %d: %7.1f ticks per DMA instruction (issue + its share of the wait)\n",
                           waves, grid, wait_every, order, (double)c / (iters * 16.0));
                }
    return 0;
}

// Hardware-semantics probe (gfx950): does `s_waitcnt vmcnt(K)` retire an OLDER load / LDS-DMA when
// the K younger operations are global STORES?  (The stage hand-overs of the forward kernels rely
// on vector-memory operations of one wave leaving the vmcnt queue in issue order, loads, stores
// and LDS-DMA alike; the removed f16x3 training forward used vmcnt(4 + k) with k younger stores.)
// The older op is a global_load_lds_dwordx4 (LDS-DMA), checked by reading the LDS bytes behind the wait.
// Each wave streams cold lines of a 1 GiB source buffer; K stores go to a small hot buffer.
// Prints the number of 16-byte pieces that did not hold the source pattern after the wait.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/vmcnt_order.hip -o gpurun_out/vmcnt_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pattern(uint32_t word_index) { return word_index * 2654435761u + 12345u; }

__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = pattern((uint32_t)i);
}

template <int kMode, int kStores>
__global__ __launch_bounds__(256) void probe(const uint32_t* src, uint32_t* hot, unsigned long long* bad,
                                              unsigned long long* checked, int iters, uint32_t src_words) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 256];      // 1 KiB per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    uint32_t* my_lds = lds + wave * 256;
    uint32_t* my_hot = hot + (size_t)gw * 64 * 4 * 8;                   // 8 KiB per wave, stays in L2
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // a different 1 KiB line group every iteration, far apart: misses L2 and mostly the MALL
        const uint32_t piece = (uint32_t)(((uint64_t)it * nw + gw) * 40503u % (src_words / 256));
        const uint32_t word0 = piece * 256 + lane * 4;
        const uint32_t lds_addr = (uint32_t)(uintptr_t)my_lds;
        *(u32x4*)(my_lds + lane * 4) = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
        __builtin_amdgcn_s_waitcnt(0);      // everything drained: the queue starts empty
        u32x4 got = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
        const u32x4 val = {(uint32_t)it, gw, (uint32_t)lane, 7u};
        const uint32_t off = word0 * 4;                                  // < 4 GiB
        uint32_t* st = my_hot + lane * 4;
        if (kMode == 0) {
            asm volatile(
                "s_mov_b32 m0, %[m]\n\t"
                "s_nop 4\n\t"
                "global_load_lds_dwordx4 %[off], %[base]\n\t"
                ".rept %c[k]\n\t"
                "global_store_dwordx4 %[st], %[val], off\n\t"
                ".endr\n\t"
                "s_waitcnt vmcnt(%c[k])\n\t"
                "ds_read_b128 %[got], %[la]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_waitcnt vmcnt(0)"
                : [got] "=&v"(got)
                : [m] "s"(__builtin_amdgcn_readfirstlane(lds_addr)), [off] "v"(off), [base] "s"(src),
                  [st] "v"(st), [val] "v"(val), [la] "v"(lds_addr + lane * 16), [k] "n"(kStores)
                : "memory");
        }
        (void)lds_addr;
        if (kMode == 0) {
            const bool ok = got.x == pattern(word0) && got.y == pattern(word0 + 1) &&
                            got.z == pattern(word0 + 2) && got.w == pattern(word0 + 3);
            nbad += ok ? 0 : 1;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (lane == 0) atomicAdd(checked, (unsigned long long)iters * 64);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int kStores>
static int run(const uint32_t* src, uint32_t* hot, unsigned long long* cnt, int iters, uint32_t words, int grid) {
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL((probe<0, kStores>), dim3(grid), dim3(256), 0, 0, src, hot, cnt, cnt + 1, iters, words);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    printf("LDS-DMA then %d stores, s_waitcnt vmcnt(%d): %llu of %llu pieces not landed\n", kStores, kStores, h[0], h[1]);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = argc > 2 ? atoi(argv[2]) : 1024;
    const size_t words = (size_t)1 << 28;                                // 1 GiB
    uint32_t *src, *hot;
    unsigned long long* cnt;
    CK(hipMalloc(&src, words * 4));
    CK(hipMalloc(&hot, (size_t)grid * 4 * 8192 * 4));
    CK(hipMalloc(&cnt, 16));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, src, words);
    CK(hipDeviceSynchronize());
    // control: vmcnt(0) semantics (k = 0) must always pass
    if (run<0>(src, hot, cnt, iters, (uint32_t)words, grid)) return 1;
    if (run<1>(src, hot, cnt, iters, (uint32_t)words, grid)) return 1;
    if (run<2>(src, hot, cnt, iters, (uint32_t)words, grid)) return 1;
    if (run<4>(src, hot, cnt, iters, (uint32_t)words, grid)) return 1;
    if (run<8>(src, hot, cnt, iters, (uint32_t)words, grid)) return 1;
    printf("done\n");
    return 0;
}

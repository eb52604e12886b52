// Hardware-semantics probe (gfx950): can the DATA registers of a global_store_dwordx4 be overwritten
// by an LDS read RETURN (ds_read_b128 into the same VGPRs, issued right behind the store) before the
// store has read them?  hipcc's hazard recognizer only covers a VALU write one wait state behind a
// >64-bit store; its register allocator freely gives a store's data registers to the next ds_read.
// The split-precision training forward does exactly that (x_hat / h / out stores, operands re-read
// from LDS) and showed run-to-run differences in the STORED tensors, 16 lanes x 1 dword at a time.
//
// Every wave: fill six 4-register sets with pattern A (VALU), issue six global_store_dwordx4 of them
// back to back, then immediately overwrite the sets
//   mode 0: with ds_read_b128 (LDS holds pattern B)        <- the case in question
//   mode 1: with v_mov_b32 (pattern B)                      <- covered by the compiler's hazard rule
//   mode 2: not at all                                      <- control
// drain, read the stored bytes back and count dwords that are not pattern A (and how many of those
// are pattern B).  `dma` > 0 adds that many 1 KiB LDS-DMA loads per iteration in front of the
// stores (vector-memory back-pressure like in the render kernel).
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/store_data_hazard.hip -o /tmp/store_data_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kMode, int kDma>
__global__ __launch_bounds__(256) void probe(uint32_t* out, const uint32_t* src, unsigned long long* counts,
                                              int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 6 * 256 + 4 * 256 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wave;
    uint32_t* my_lds = lds + wave * 6 * 256;                 // 6 KiB per wave: pattern B
    uint32_t* dma_lds = lds + 4 * 6 * 256 + wave * 8 * 256;  // 8 KiB per wave: LDS-DMA landing zone
    for (int k = 0; k < 6; ++k)
        *(u32x4*)(my_lds + k * 256 + lane * 4) = u32x4{0xB0000000u | k, 0xB1000000u | k, 0xB2000000u | k, 0xB3000000u | k};
    __syncthreads();
    uint32_t* mine = out + (size_t)gw * 6 * 256;             // 6 KiB per wave
    unsigned long long bad = 0, bad_b = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 d[6];
        for (int k = 0; k < 6; ++k) {
            const uint32_t a = 0xA0000000u | (it << 12) | (k << 8) | lane;
            d[k] = u32x4{a, a + 0x40, a + 0x80, a + 0xC0};
        }
        const uint32_t lds_addr = (uint32_t)(uintptr_t)my_lds + lane * 16;
        uint32_t* p = mine + lane * 4;
        if (kDma > 0) {
            const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dma_lds);
            const uint32_t off = (uint32_t)((((uint64_t)it * gridDim.x * 4 + gw) * 40503u % (1u << 18)) * 1024u) + lane * 16;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\t"
                         ".rept %c3\n\tglobal_load_lds_dwordx4 %1, %2\n\t.endr"
                         :: "s"(m), "v"(off), "s"(src), "n"(kDma) : "memory");
        }
        if (kMode == 0) {
            asm volatile(
                "global_store_dwordx4 %[p], %[d0], off\n\t"
                "global_store_dwordx4 %[p], %[d1], off offset:1024\n\t"
                "global_store_dwordx4 %[p], %[d2], off offset:2048\n\t"
                "global_store_dwordx4 %[p], %[d3], off offset:3072\n\t"
                "global_store_dwordx4 %[q], %[d4], off\n\t"
                "global_store_dwordx4 %[q], %[d5], off offset:1024\n\t"
                "s_nop 1\n\t"
                "ds_read_b128 %[d5], %[l] offset:5120\n\t"
                "ds_read_b128 %[d0], %[l]\n\t"
                "ds_read_b128 %[d1], %[l] offset:1024\n\t"
                "ds_read_b128 %[d3], %[l] offset:3072\n\t"
                "ds_read_b128 %[d4], %[l] offset:4096\n\t"
                "ds_read_b128 %[d2], %[l] offset:2048\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5])
                : [p] "v"(p), [q] "v"(p + 1024), [l] "v"(lds_addr)
                : "memory");
        } else if (kMode == 1) {
            asm volatile(
                "global_store_dwordx4 %[p], v[100:103], off\n\t"
                "global_store_dwordx4 %[p], v[104:107], off offset:1024\n\t"
                "global_store_dwordx4 %[p], v[108:111], off offset:2048\n\t"
                "global_store_dwordx4 %[p], v[112:115], off offset:3072\n\t"
                "global_store_dwordx4 %[q], v[116:119], off\n\t"
                "global_store_dwordx4 %[q], v[120:123], off offset:1024\n\t"
                "s_nop 1\n\t"
                "v_mov_b32 v120, 0xB0000005\n\tv_mov_b32 v121, 0xB1000005\n\tv_mov_b32 v122, 0xB2000005\n\tv_mov_b32 v123, 0xB3000005\n\t"
                "v_mov_b32 v100, 0xB0000000\n\tv_mov_b32 v101, 0xB1000000\n\tv_mov_b32 v102, 0xB2000000\n\tv_mov_b32 v103, 0xB3000000\n\t"
                "v_mov_b32 v104, 0xB0000001\n\tv_mov_b32 v105, 0xB1000001\n\tv_mov_b32 v106, 0xB2000001\n\tv_mov_b32 v107, 0xB3000001\n\t"
                "v_mov_b32 v112, 0xB0000003\n\tv_mov_b32 v113, 0xB1000003\n\tv_mov_b32 v114, 0xB2000003\n\tv_mov_b32 v115, 0xB3000003\n\t"
                "v_mov_b32 v116, 0xB0000004\n\tv_mov_b32 v117, 0xB1000004\n\tv_mov_b32 v118, 0xB2000004\n\tv_mov_b32 v119, 0xB3000004\n\t"
                "v_mov_b32 v108, 0xB0000002\n\tv_mov_b32 v109, 0xB1000002\n\tv_mov_b32 v110, 0xB2000002\n\tv_mov_b32 v111, 0xB3000002"
                : "+{v[100:103]}"(d[0]), "+{v[104:107]}"(d[1]), "+{v[108:111]}"(d[2]), "+{v[112:115]}"(d[3]),
                  "+{v[116:119]}"(d[4]), "+{v[120:123]}"(d[5])
                : [p] "v"(p), [q] "v"(p + 1024)
                : "memory");
        } else {
            for (int k = 0; k < 6; ++k) *(u32x4*)(p + k * 256) = d[k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < 6; ++k) {
            const u32x4 got = __builtin_nontemporal_load((const u32x4*)(p + k * 256));
            const uint32_t a = 0xA0000000u | (it << 12) | (k << 8) | lane;
            const uint32_t want[4] = {a, a + 0x40, a + 0x80, a + 0xC0};
            for (int c = 0; c < 4; ++c)
                if (got[c] != want[c]) {
                    ++bad;
                    if ((got[c] >> 28) == 0xB) ++bad_b;
                    if (lane >= 48) ++bad_hi;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (bad) {
        atomicAdd(counts, bad);
        atomicAdd(counts + 1, bad_b);
        atomicAdd(counts + 2, bad_hi);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int kMode, int kDma>
static int run(uint32_t* out, const uint32_t* src, unsigned long long* cnt, int iters, int grid) {
    CK(hipMemset(cnt, 0, 32));
    hipLaunchKernelGGL((probe<kMode, kDma>), dim3(grid), dim3(256), 0, 0, out, src, cnt, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost));
    const char* what[] = {"ds_read_b128 into the store data registers", "v_mov_b32 into the store data registers", "no overwrite (control)"};
    printf("mode %d (%s), %d LDS-DMA per iteration: %llu of %llu stored dwords wrong (%llu hold the overwriting pattern, %llu in lanes 48-63)\n",
           kMode, what[kMode], kDma, h[0], (unsigned long long)grid * 4 * 64 * 24 * iters, h[1], h[2]);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;
    uint32_t *out, *src;
    unsigned long long* cnt;
    CK(hipMalloc(&out, (size_t)grid * 4 * 6 * 1024));
    CK(hipMalloc(&src, (size_t)1 << 28));
    CK(hipMemset(src, 0x5a, (size_t)1 << 28));
    CK(hipMalloc(&cnt, 32));
    if (run<2, 0>(out, src, cnt, iters, grid)) return 1;
    if (run<1, 0>(out, src, cnt, iters, grid)) return 1;
    if (run<0, 0>(out, src, cnt, iters, grid)) return 1;
    if (run<1, 8>(out, src, cnt, iters, grid)) return 1;
    if (run<0, 8>(out, src, cnt, iters, grid)) return 1;
    printf("done\n");
    return 0;
}

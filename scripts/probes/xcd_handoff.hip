// Memory-system probe (gfx950): how fast can one workgroup hand 16 KiB tiles to ANOTHER workgroup through memory?
// DESIGN.md section 4b costs a producer / consumer form of the training step (data-gradient workgroups hand their dY
// and x_hat tiles to weight-gradient workgroups that keep a layer's accumulators resident) whose HBM traffic is
// 2.9 GB per step instead of 7.5 — IF the hand-off stays on the chip.  This probe measures that one assumption:
// `pairs` producer workgroups each stream `iters` tiles of `tile` KiB to their consumer through a ring of `slots`
// tiles with one flag word per slot (release store / acquire load, agent scope) and one acknowledgement counter
// per pair; every consumer sums what it reads (and checks it: producer p writes the value p * 2048 + i mod 2048 into
// all of tile i).
//   mode 0   ring in ordinary (coarse-grained) device memory: the agent-scope release / acquire the compiler emits
//            write back / invalidate the XCD's L2
//   mode 1   ring in fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)): coherent
//            across the eight XCDs without cache maintenance
//   place 0  consumer of producer block b is block b + 1 (the neighbouring XCD: workgroups go round-robin over XCDs)
//   place 1  consumer is block b + 8 (the SAME XCD, another CU)
// Every spin is bounded (a stuck flag sets an error word and the workgroup leaves), so the grid always drains.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/xcd_handoff.hip -o gpurun_out/xcd_handoff
// Run:   gpurun_out/xcd_handoff [mode] [place] [pairs] [tile KiB] [slots] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
    float* ring;            // [pairs][slots][tile floats]
    uint32_t* flags;        // [pairs][slots]: sequence number + 1 of the tile the slot holds
    uint32_t* acks;         // [pairs]: tiles the consumer has finished reading
    double* sums;           // [pairs]
    uint32_t* error;        // != 0: a spin ran out or a tile held the wrong value
    int pairs, slots, iters, tile_floats, place;
};

constexpr uint32_t kMaxSpins = 2000000u;       // x ~1 us: seconds, far beyond any real wait

__device__ __forceinline__ bool wait_ge(const uint32_t* word, uint32_t want, uint32_t* error) {
    // lane 0 spins (agent-scope acquire), the workgroup follows through the barrier behind it
    __shared__ int ok;
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        ok = 1;
        while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kMaxSpins) {
                ok = 0;
                atomicExch(error, 1u);
                break;
            }
        }
    }
    __syncthreads();
    const bool good = ok != 0;
    __syncthreads();
    return good;
}

__global__ __launch_bounds__(256) void handoff_kernel(const Args a) {
    // blocks [0, 2 * pairs): place 0: even block = producer, the next block its consumer;
    // place 1: blocks are grouped in sixteens — the first eight produce, block + 8 consumes
    int pair, role;
    if (a.place == 0) {
        pair = blockIdx.x >> 1;
        role = blockIdx.x & 1;
    } else {
        const int grp = blockIdx.x >> 4, in = blockIdx.x & 15;
        pair = grp * 8 + (in & 7);
        role = in >> 3;
    }
    if (pair >= a.pairs) return;
    float* ring = a.ring + (size_t)pair * a.slots * a.tile_floats;
    uint32_t* flags = a.flags + (size_t)pair * a.slots;
    uint32_t* ack = a.acks + pair;
    const int per_thread = a.tile_floats / 4 / 256;               // f32x4 per thread and tile
    if (role == 0) {
        for (int i = 0; i < a.iters; ++i) {
            const int slot = i % a.slots;
            if (i >= a.slots && !wait_ge(ack, (uint32_t)(i - a.slots + 1), a.error)) return;     // slot free again
            const float v = (float)((pair << 11) | (i & 2047));        // < 2^18: the consumer's sums stay exact
            f32x4* dst = (f32x4*)(ring + (size_t)slot * a.tile_floats);
            for (int k = 0; k < per_thread; ++k) dst[k * 256 + threadIdx.x] = f32x4{v, v, v, v};
            __syncthreads();                                       // every thread's stores issued ...
            if (threadIdx.x == 0) {
                __threadfence();                                   // ... and visible to the agent before the flag
                __hip_atomic_store(flags + slot, (uint32_t)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        double sum = 0.0;
        for (int i = 0; i < a.iters; ++i) {
            const int slot = i % a.slots;
            if (!wait_ge(flags + slot, (uint32_t)(i + 1), a.error)) return;
            const f32x4* src = (const f32x4*)(ring + (size_t)slot * a.tile_floats);
            float part = 0.f;
            for (int k = 0; k < per_thread; ++k) {
                const f32x4 x = src[k * 256 + threadIdx.x];
                part += (x.x + x.y) + (x.z + x.w);
            }
            if (part != 4.0f * per_thread * (float)((pair << 11) | (i & 2047))) atomicExch(a.error, 2u);
            sum += part;
            __syncthreads();                                       // every thread has read the slot
            if (threadIdx.x == 0) __hip_atomic_store(ack, (uint32_t)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) a.sums[pair] = sum;
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, place = argc > 2 ? atoi(argv[2]) : 0;
    const int pairs = argc > 3 ? atoi(argv[3]) : 128, tile_kib = argc > 4 ? atoi(argv[4]) : 16;
    const int slots = argc > 5 ? atoi(argv[5]) : 8, iters = argc > 6 ? atoi(argv[6]) : 2000;
    if (pairs < 1 || pairs > 128 || tile_kib < 4 || tile_kib % 4 || slots < 2 || iters < 1 || (place == 1 && pairs % 8)) {
        printf("usage: xcd_handoff [mode 0|1] [place 0|1] [pairs <= 128 (multiple of 8 for place 1)] [tile KiB, multiple of 4] "
               "[slots >= 2] [iters]\n");
        return 1;
    }
    Args a;
    a.pairs = pairs, a.slots = slots, a.iters = iters, a.tile_floats = tile_kib * 256, a.place = place;
    const size_t ring_bytes = (size_t)pairs * slots * tile_kib * 1024;
    if (mode == 1) {
        CHECK(hipExtMallocWithFlags((void**)&a.ring, ring_bytes, hipDeviceMallocFinegrained));
        CHECK(hipExtMallocWithFlags((void**)&a.flags, (size_t)pairs * slots * 4, hipDeviceMallocFinegrained));
        CHECK(hipExtMallocWithFlags((void**)&a.acks, (size_t)pairs * 4, hipDeviceMallocFinegrained));
    } else {
        CHECK(hipMalloc((void**)&a.ring, ring_bytes));
        CHECK(hipMalloc((void**)&a.flags, (size_t)pairs * slots * 4));
        CHECK(hipMalloc((void**)&a.acks, (size_t)pairs * 4));
    }
    CHECK(hipMalloc((void**)&a.sums, (size_t)pairs * 8));
    CHECK(hipMalloc((void**)&a.error, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // 2 x pairs <= 256 workgroups of 256 threads, one per CU: all resident at once (the producers and consumers wait
    // for each other)
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(a.flags, 0, (size_t)pairs * slots * 4));
        CHECK(hipMemset(a.acks, 0, (size_t)pairs * 4));
        CHECK(hipMemset(a.error, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(handoff_kernel, dim3(2 * pairs), dim3(256), 0, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t err = 0;
        CHECK(hipMemcpy(&err, a.error, 4, hipMemcpyDeviceToHost));
        const double bytes = (double)pairs * iters * tile_kib * 1024.0;
        printf("mode %d (%s) place %d (%s) pairs %d tile %d KiB slots %d iters %d: %.3f ms, %.2f TB/s handed over "
               "(%.1f us per tile and pair)%s\n",
               mode, mode ? "fine-grained" : "coarse-grained + agent fences", place, place ? "same XCD" : "next XCD",
               pairs, tile_kib, slots, iters, ms, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / iters,
               err ? (err == 1 ? "  ERROR: a spin ran out" : "  ERROR: a tile held the wrong value") : "");
    }
    return 0;
}

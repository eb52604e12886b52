// Host-side helpers of the C ABI (error text, HIP-event timing) and the counter-based RNG the
// kernels use for the in-kernel stochastic path.
#ifndef NERF_COMMON_H
#define NERF_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "nerf_hip.h"

namespace nerf_common {

inline char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline const char* last_error() { return error_buffer(); }

inline int fail(int code, const char* what) {
    snprintf(error_buffer(), 512, "%s", what);
    return code;
}

inline int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return NERF_HIP_OK;
    snprintf(error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
    return NERF_HIP_EHIP;
}

// Kernel timing with HIP events recorded on the SAME stream the kernel is launched on
// (torch.cuda.Event would only see torch's current stream).  Off unless enabled.
struct Timing {
    struct Pair {
        hipEvent_t start, stop;
        int tag;                // NERF_HIP_TIMING_* (include/nerf_hip.h): which launch of a step the pair brackets
    };
    static std::mutex& mu() {
        static std::mutex m;
        return m;
    }
    static std::atomic<bool>& on() {      // read on every launch without the mutex
        static std::atomic<bool> v{false};
        return v;
    }
    static std::vector<Pair>& pairs() {
        static std::vector<Pair> v;
        return v;
    }
    static hipEvent_t& pending() {
        static thread_local hipEvent_t e = nullptr;
        return e;
    }
    static int enable(bool v) {
        on().store(v, std::memory_order_relaxed);
        return NERF_HIP_OK;
    }
    static constexpr size_t kMaxPairs = 8192;   // un-read launches beyond this are not timed
    static void before(hipStream_t st) {
        pending() = nullptr;
        if (!on().load(std::memory_order_relaxed)) return;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;       // a captured region is replayed, not timed
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        {
            std::lock_guard<std::mutex> lk(mu());
            if (pairs().size() >= kMaxPairs) return;
        }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, st);
        pending() = e;
    }
    static void after(hipStream_t st, int tag = NERF_HIP_TIMING_FORWARD) {
        if (pending() == nullptr) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            (void)hipEventDestroy(pending());
            pending() = nullptr;
            return;
        }
        (void)hipEventRecord(e, st);
        std::lock_guard<std::mutex> lk(mu());
        pairs().push_back(Pair{pending(), e, tag});
        pending() = nullptr;
    }
    // averages per tag over the pairs recorded since the last reset; avg_ms / launches: [n_tags]
    static int read_tagged(bool reset, int n_tags, double* avg_ms, int64_t* launches) {
        std::lock_guard<std::mutex> lk(mu());
        for (int t = 0; t < n_tags; ++t) {
            if (avg_ms) avg_ms[t] = 0.0;
            if (launches) launches[t] = 0;
        }
        std::vector<double> total((size_t)(n_tags > 0 ? n_tags : 0), 0.0);
        std::vector<int64_t> n((size_t)(n_tags > 0 ? n_tags : 0), 0);
        for (auto& p : pairs()) {
            if (p.tag < 0 || p.tag >= n_tags) continue;
            if (hipEventSynchronize(p.stop) != hipSuccess) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
                total[(size_t)p.tag] += ms;
                ++n[(size_t)p.tag];
            }
        }
        for (int t = 0; t < n_tags; ++t) {
            if (avg_ms) avg_ms[t] = n[(size_t)t] ? total[(size_t)t] / (double)n[(size_t)t] : 0.0;
            if (launches) launches[t] = n[(size_t)t];
        }
        if (reset) {
            for (auto& p : pairs()) {
                (void)hipEventDestroy(p.start);
                (void)hipEventDestroy(p.stop);
            }
            pairs().clear();
        }
        return NERF_HIP_OK;
    }
    // the forward (render) kernel alone: what bench.py's roofline divides by
    static int read(bool reset, double* avg_ms, int64_t* launches) {
        double a[NERF_HIP_TIMING_TAGS];
        int64_t n[NERF_HIP_TIMING_TAGS];
        const int rc = read_tagged(reset, NERF_HIP_TIMING_TAGS, a, n);
        if (avg_ms) *avg_ms = a[NERF_HIP_TIMING_FORWARD];
        if (launches) *launches = n[NERF_HIP_TIMING_FORWARD];
        return rc;
    }
};
// One launch (or a short run of launches) between a pair of events, when timing is on
struct TimedLaunch {
    hipStream_t st;
    int tag;
    TimedLaunch(hipStream_t s, int t) : st(s), tag(t) { Timing::before(st); }
    ~TimedLaunch() { Timing::after(st, tag); }
};

// Name of the experiment this library was built as ("" for the product build).  The product sources
// carry no experiment switches (settled ones live on the `experiments-r02` branch and under
// scripts/probes/); a throw-away variant built with -DNERF_HIP_EXPERIMENT=name says so here, and the
// loader (nerf_amd/_lib.py) refuses such a library unless it was asked for by path.
#define NERF_HIP_STR2(x) #x
#define NERF_HIP_STR(x) NERF_HIP_STR2(x)
inline const char* build_flags() {
#ifdef NERF_HIP_EXPERIMENT
    return NERF_HIP_STR(NERF_HIP_EXPERIMENT);
#else
    return "";
#endif
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): function attributes
// are per device, and a process may drive more than one.
inline int ensure_dynamic_lds(const void* fn, int bytes, int device, unsigned* done_mask) {
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    if (device >= 0 && device < 32 && (*done_mask & (1u << device))) return NERF_HIP_OK;
    const int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes),
                             "hipFuncSetAttribute");
    if (rc == NERF_HIP_OK && device >= 0 && device < 32) *done_mask |= (1u << device);
    return rc;
}

}  // namespace nerf_common

// Philox4x32-10.  Key = seed XOR offset, counter = (ray id lo, ray id hi, sample block, stream):
// the per-launch `offset` (call count, with the data-parallel rank in its high bits) selects an
// independent KEY, so the draws of successive launches and of different ranks share no counter
// block (with the offset added to the block index, launch k + 1 would replay launch k's blocks
// shifted by one).  Used only when the caller asks the kernel to draw u / noise itself (rng_mode);
// the parity path takes the draws as inputs.
namespace nerf_rng {

__device__ __forceinline__ void round_(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

__device__ __forceinline__ void philox(uint64_t seed, uint64_t offset, uint64_t ray, uint32_t block,
                                       uint32_t stream, uint32_t (&c)[4]) {
    c[0] = (uint32_t)ray;
    c[1] = (uint32_t)(ray >> 32);
    c[2] = block;
    c[3] = stream;
    uint32_t k0 = (uint32_t)seed ^ (uint32_t)offset, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        round_(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// uniform in [0, 1) with 24 bits, like torch.rand for fp32
__device__ __forceinline__ float uniform(uint64_t seed, uint64_t offset, uint64_t ray, uint32_t s,
                                         uint32_t stream) {
    uint32_t c[4];
    philox(seed, offset, ray, s >> 2, stream, c);
    const uint32_t x = c[s & 3];
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// standard normal by Box-Muller on two 24-bit uniforms of one Philox block
__device__ __forceinline__ float normal(uint64_t seed, uint64_t offset, uint64_t ray, uint32_t s,
                                        uint32_t stream) {
    uint32_t c[4];
    philox(seed, offset, ray, s >> 1, stream, c);
    const uint32_t a = c[(s & 1) * 2], b = c[(s & 1) * 2 + 1];
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return __builtin_sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

}  // namespace nerf_rng

#endif

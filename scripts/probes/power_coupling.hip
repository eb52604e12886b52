// Probe (gfx950): does matrix work on one half of the chip slow the memory system for the other half?
// The training kernels' compute time and memory time ADD instead of overlapping, whatever the instruction order
// (NOTES.md R4).  One explanation is the board's power cap: a chip at its cap with dense f16 MFMAs gives the memory
// system what is left.  Here workgroups with an even index stream (the data gradient's loads + saves, back to back)
// and workgroups with an odd index either sleep or run a dense register-resident v_mfma_f32_16x16x32_f16 loop on
// random data for the whole launch; the streaming half's throughput is measured both ways.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/power_coupling.hip -o gpurun_out/power_coupling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void probe(const float* xhat, float* dy, int64_t items, int64_t mp, int mfma_on,
                                                int stream_on, volatile int* stop, float* sink, unsigned long long* done) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    if (blockIdx.x & 1) {
        // matrix half: runs until the streaming half has finished (or a fixed count when nothing streams)
        h8 a, b;
        for (int k = 0; k < 8; ++k) {
            a[k] = (_Float16)(((lane * 37 + k * 11) % 97) * 0.02f - 1.0f);
            b[k] = (_Float16)(((lane * 53 + k * 7) % 89) * 0.02f - 0.9f);
        }
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned long long n = 0;
        while (true) {
            if (mfma_on) {
#pragma unroll
                for (int r = 0; r < 32; ++r)
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
                n += 256;
            } else {
                __builtin_amdgcn_s_sleep(100);
            }
            if (__builtin_amdgcn_readfirstlane(*stop)) break;
            if (!stream_on && n >= 256ull * 40000) break;
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) s += acc[t].x;
        if (s == 12345.f) sink[1] = s;
        if (lane == 0) atomicAdd(done, n);
        return;
    }
    if (!stream_on) return;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const int64_t first = (int64_t)(blockIdx.x >> 1) * 4 + wave, stride = (int64_t)(gridDim.x >> 1) * 4;
    for (int64_t item = first; item < items; item += stride) {
        for (int L = 4; L >= 0; --L) {
            const float* src = xhat + (int64_t)L * mp * 256 + item * 16 * 256;
            float* dst = dy + (int64_t)L * mp * 256 + item * 16 * 256;
            f32x4 x[16];
#pragma unroll
            for (int T = 0; T < 16; ++T) *(f32x4*)(dst + j * 256 + 4 * g + T * 16) = keep + (float)T;
#pragma unroll
            for (int T = 0; T < 16; ++T) x[T] = *(const f32x4*)(src + j * 256 + 4 * g + T * 16);
            keep = keep + x[3] + x[7] + x[11];
        }
    }
    if (keep.x == 12345.f) sink[0] = keep.y;
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd((int*)stop + 1, 1) == (int)(gridDim.x >> 1) - 1) *stop = 1;   // last streamer stops the others
}

int main() {
    const int64_t items = 16384, mp = items * 16;
    float *x, *y, *sink;
    int* stop;
    unsigned long long* done;
    if (hipMalloc(&x, mp * 256 * 4 * 5) != hipSuccess || hipMalloc(&y, mp * 256 * 4 * 5) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    hipMalloc(&stop, 64);
    hipMalloc(&done, 64);
    hipMemset(x, 0, mp * 256 * 4 * 5);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 75776);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            const int mfma_on = mode >= 1, stream_on = mode <= 1;
            hipMemset(stop, 0, 64);
            hipMemset(done, 0, 64);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            probe<<<512, 256, 75776>>>(x, y, items, mp, mfma_on, stream_on, stop, sink, done);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long n = 0;
            hipMemcpy(&n, done, 8, hipMemcpyDeviceToHost);
            const double bytes = (double)items * 5 * 2 * 16 * 1024;
            const double tflops = (double)n * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12;
            if (stream_on)
                printf("streaming half %s: %.3f ms for 2.68 GB = %.0f GB/s; matrix half %.0f TFLOP/s\n",
                       mfma_on ? "beside dense f16 MFMAs on the other half" : "beside sleeping workgroups          ", ms,
                       bytes / (ms * 1e-3) / 1e9, tflops);
            else
                printf("matrix half alone: %.3f ms, %.0f TFLOP/s\n", ms, tflops);
        }
    return 0;
}

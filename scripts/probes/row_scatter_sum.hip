// Hardware check of nerf_device.h: scatter_level8 / scatter_level4 / scatter_take (the reduce-scatter butterfly
// of the gamma / beta gradient sums): lane j of every 16-lane row must end with the sum over its row's lanes of
// tile T = j, for each of eight quantities, bit for bit what row_sum() gives (integer-valued inputs: the order
// of the adds does not matter).
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -I nerf_amd/csrc -I include \
//     scripts/probes/row_scatter_sum.hip -o /tmp/row_scatter_sum && /tmp/row_scatter_sum
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "nerf_device.h"

__global__ void probe(const float* in, float* out, float* ref) {
    using namespace nerf_device;
    const int lane = threadIdx.x;
    float v[16][8];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[t][i] = in[t * 64 + lane] * (float)(i + 1);     // visible VALU results, as in the kernels
    float kept[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float w0[8], w1[8], x[8];
        scatter_level8(v[t], v[t + 8], w0);
        scatter_level8(v[t + 4], v[t + 12], w1);
        scatter_level4(w0, w1, x);
        scatter_take(x, t, lane, kept);
    }
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float s = row_sum(v[t][1]);
        if ((lane & 15) == t) keep = s;
    }
    out[lane] = kept[1];
    ref[lane] = keep;
    // the other seven quantities are multiples of the same sums
    bool same = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) same = same && kept[i] * 2.0f == kept[1] * (float)(i + 1);
    if (!same) out[lane] = -12345.0f;
}

int main() {
    float h[16 * 64], o[64], r[64];
    for (int t = 0; t < 16; ++t)
        for (int l = 0; l < 64; ++l) h[t * 64 + l] = (float)((t * 7 + l * 13 + t * l) % 31) - 15.f;
    float *d, *e, *f;
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&e, sizeof o) != hipSuccess || hipMalloc(&f, sizeof r) != hipSuccess) return 2;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, e, f);
    (void)hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r, f, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float want = 0;
        for (int m = 0; m < 16; ++m) want += 2.0f * h[(l % 16) * 64 + (l / 16) * 16 + m];
        if (want != o[l] || want != r[l]) {
            ++bad;
            printf("lane %d: scatter %g, row_sum %g, host %g\n", l, o[l], r[l], want);
        }
    }
    printf("row_scatter_sum: %s (%d of 64 lanes wrong)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}

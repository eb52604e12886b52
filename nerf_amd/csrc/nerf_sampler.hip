// On-device batched ray/pixel sampler: the GPU replacement of PixelRayDataset.__getitem__ +
// DataLoader collation (nerf/dataset.py:246-316), which caps the reference at ~1.7e4 rays/s.
// One thread per example; pure gather, HBM/latency bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerf_hip.h"
#include "nerf_common.h"

namespace {

__global__ void nerf_gather_kernel(const NerfHipGatherArgs ga) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ga.n) return;
    int64_t id = ga.index[i];
    const int64_t w = id % ga.image_w;                 // dataset.py:283-291
    id /= ga.image_w;
    const int64_t h = id % ga.image_h;
    id /= ga.image_h;
    const int64_t b = id % ga.batch;
    const int64_t pix = (b * ga.image_h + h) * ga.image_w + w;
    const float* px = ga.images + pix * 3;
    ga.pixels[i * 3 + 0] = px[0];
    ga.pixels[i * 3 + 1] = px[1];
    ga.pixels[i * 3 + 2] = px[2];
    if (ga.label != nullptr && ga.segmentation != nullptr) ga.label[i] = ga.segmentation[pix];
    // camera-frame ray of pixel (h, w): nerf/model.py:271-278
    const float x = ((float)w - 0.5f * (float)(ga.image_w - 1)) / ga.focal_length;
    const float y = ((float)h - 0.5f * (float)(ga.image_h - 1)) / ga.focal_length;
    const float c0 = x, c1 = -y, c2 = -1.0f;
    ga.rays[i * 3 + 0] = c0;
    ga.rays[i * 3 + 1] = c1;
    ga.rays[i * 3 + 2] = c2;
    const float* pose = ga.poses + b * 16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ga.rays_o[i * 3 + k] = pose[4 * k + 3];
        ga.rays_d[i * 3 + k] = (pose[4 * k] * c0 + pose[4 * k + 1] * c1) + pose[4 * k + 2] * c2;
    }
    if (ga.image_wi != nullptr) ga.image_wi[i] = w;
    if (ga.image_hi != nullptr) ga.image_hi[i] = h;
    if (ga.image_bi != nullptr) ga.image_bi[i] = b;
}

}  // namespace

extern "C" int nerf_hip_gather_pixel_rays(const NerfHipGatherArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "gather_pixel_rays: null args");
    const NerfHipGatherArgs& g = *args;
    if (g.n == 0) return NERF_HIP_OK;
    if (g.n < 0 || g.index == nullptr || g.images == nullptr || g.poses == nullptr || g.pixels == nullptr ||
        g.rays == nullptr || g.rays_o == nullptr || g.rays_d == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "gather_pixel_rays: null pointer or negative n");
    if (g.batch <= 0 || g.image_h <= 0 || g.image_w <= 0 || g.focal_length == 0.f)
        return nerf_common::fail(NERF_HIP_EINVAL, "gather_pixel_rays: bad image geometry");
    const int threads = 256;
    const int64_t blocks = (g.n + threads - 1) / threads;
    hipLaunchKernelGGL(nerf_gather_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, g);
    return nerf_common::check_hip(hipGetLastError(), "gather_pixel_rays launch");
}

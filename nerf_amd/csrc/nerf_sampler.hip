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

// ---------------------------------------------------------------------------------------------
// inverse-CDF resampling for the hierarchical (coarse -> fine) configuration
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kMaxPosts = 1024;

// One wave per ray.  CDF by a wave-wide shuffle scan over the intervals (64 per step with a
// carry), kept in LDS; every fine sample then binary-searches it; the union is a merge of two
// sorted lists done by rank: rank(fine k) = k + #coarse posts <= it, rank(coarse i) = i + #fine
// samples < it, both read off the same monotone map u <-> t.
__global__ __launch_bounds__(256) void nerf_resample_kernel(const NerfHipResampleArgs ra) {
#pragma clang fp contract(off)                     // t0 + frac * (t1 - t0) as torch evaluates it: multiply, then add
    __shared__ float cdf_s[4][kMaxPosts];
    __shared__ float uf_s[4][kMaxPosts];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= ra.n_rays) return;
    const int Sc = ra.num_coarse, Sf = ra.num_fine, P = Sc - 1;
    const float* w = ra.weights + ray * P;
    const float* tc = ra.t_coarse + ray * Sc;
    float* cdf = cdf_s[wave];
    float* uf = uf_s[wave];

    // inclusive scan of (w + floor) -> cdf[i + 1].  The spec (oracle/nerf_oracle.py: resample_fenceposts) is
    // torch.cumsum on the CPU, which accumulates float32 input in DOUBLE and rounds every prefix to float: the
    // scan runs in double too (the association differs from torch's sequential loop, but a double sum of <= 1023
    // floats rounds to the same float except for one prefix in ~1e9) — the only fp64 arithmetic in the library,
    // 7 adds per interval of a kernel that takes microseconds.
    double carry = 0.0;
    for (int base = 0; base < P; base += 64) {
        const int i = base + lane;
        double v = i < P ? (double)(w[i] + ra.pdf_floor) : 0.0;       // w + floor: one fp32 add, as in the oracle
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(v, d);
            if (lane >= d) v += up;
        }
        if (i < P) cdf[i + 1] = (float)(carry + v);
        carry += __shfl(v, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const float total = (float)carry;             // = cdf[P], the float the oracle divides by
    for (int i = lane; i <= P; i += 64) cdf[i] = i == P ? 1.0f : cdf[i] / total;
    for (int k = lane; k < Sf; k += 64)
        uf[k] = ra.u != nullptr ? ra.u[ray * Sf + k] : ((float)k + 0.5f) / (float)Sf;
    __builtin_amdgcn_wave_barrier();

    float* out = ra.t_union + ray * (int64_t)(Sc + Sf);
    // fine samples: interval idx with cdf[idx] <= u < cdf[idx + 1]
    for (int k = lane; k < Sf; k += 64) {
        const float u = uf[k];
        int lo = 0, hi = P;                       // invariant: cdf[lo] <= u < cdf[hi] (cdf[P] = 1 > u)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid; else hi = mid;
        }
        const float c0 = cdf[lo], c1 = cdf[lo + 1];
        const float den = c1 - c0;
        const float frac = den > 0.f ? (u - c0) / den : 0.f;
        const float t0 = tc[lo], t1 = tc[lo + 1];
        out[k + lo + 1] = t0 + frac * (t1 - t0);
    }
    // coarse fenceposts: rank = i + #fine samples with u < cdf[i]
    for (int i = lane; i < Sc; i += 64) {
        const float c = cdf[i];
        int lo = 0, hi = Sf;                      // first k in [0, Sf] with uf[k] >= c
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uf[mid] < c) lo = mid + 1; else hi = mid;
        }
        out[i + lo] = tc[i];
    }
}

}  // namespace

extern "C" int nerf_hip_resample_pdf(const NerfHipResampleArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "resample_pdf: null args");
    const NerfHipResampleArgs& r = *args;
    if (r.n_rays == 0) return NERF_HIP_OK;
    if (r.n_rays < 0 || r.t_coarse == nullptr || r.weights == nullptr || r.t_union == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "resample_pdf: null pointer or negative n_rays");
    if (r.num_coarse < 2 || r.num_coarse > kMaxPosts || r.num_fine < 1 || r.num_fine > kMaxPosts)
        return nerf_common::fail(NERF_HIP_EINVAL, "resample_pdf: sample counts out of range");
    const int64_t blocks = (r.n_rays + 3) / 4;
    hipLaunchKernelGGL(nerf_resample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, r);
    return nerf_common::check_hip(hipGetLastError(), "resample_pdf launch");
}

// ---------------------------------------------------------------------------------------------
// Adam as ONE launch over all parameter tensors: the update of torch.optim.Adam(lr, betas, eps) without
// weight decay / amsgrad — what the reference's scripts construct (train_conditional_nerf.py:106-107,
// examples/example.ipynb cell 7).  torch's own fused kernel walks a tensor list in 64 K-element chunks: for this
// model (22 or 44 tensors, 0.3-0.6 M parameters) that is ~20 workgroups and 43 us of a 0.4 ms training step
// at 512 rays per GPU; here one thread owns one parameter (1,190 workgroups, a few microseconds).
//   m <- b1 m + (1 - b1) g ;  v <- b2 v + (1 - b2) g^2 ;  p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// t = step[b] + 1 with one copy of the count per workgroup b in DEVICE memory, so a captured launch replays
// correctly and no workgroup reads a count that another one has already advanced; the workgroups stride over
// the parameters (NERF_HIP_ADAM_STEP_SLOTS of them at most).
namespace {

struct AdamKernelArgs {
    NerfHipAdamArgs a;
};

__global__ __launch_bounds__(256) void nerf_adam_kernel(const AdamKernelArgs ka) {
#pragma clang fp contract(off)
    const NerfHipAdamArgs& a = ka.a;
    const float t = a.step[blockIdx.x] + 1.0f;
    const float bc1 = 1.0f - powf(a.beta1, t), bc2 = 1.0f - powf(a.beta2, t);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.total; e += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = a.num_tensors;               // tensor t with offsets[t] <= e < offsets[t + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.offsets[mid] <= e) lo = mid; else hi = mid;
        }
        const int64_t i = e - a.offsets[lo];
        const float g = a.grads[lo][i];
        const float m = a.beta1 * a.exp_avg[e] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.exp_avg_sq[e] + (1.0f - a.beta2) * (g * g);
        a.exp_avg[e] = m;
        a.exp_avg_sq[e] = v;
        const float denom = __builtin_sqrtf(v) / __builtin_sqrtf(bc2) + a.eps;
        a.params[lo][i] = a.params[lo][i] - (a.lr / bc1) * (m / denom);
    }
    __syncthreads();                                  // every thread of this workgroup holds t
    if (threadIdx.x == 0) a.step[blockIdx.x] = t;
}

// mean((pred - target)^2) and its gradient: one workgroup of 1,024 threads, strided partial sums, a fixed
// butterfly per wave, the sixteen waves in order (one association whatever the launch: reproducible).
struct MseKernelArgs {
    NerfHipMseArgs a;
};

__global__ __launch_bounds__(1024) void nerf_mse_kernel(const MseKernelArgs ka) {
#pragma clang fp contract(off)
    __shared__ float part[16];
    const NerfHipMseArgs& a = ka.a;
    const int ch = a.channels > 0 ? a.channels : 3;
    const int64_t per_ray = (int64_t)a.stages * ch, count = a.n_rays * per_ray;
    const float inv = 1.0f / (float)(count > 0 ? count : 1);      // autograd: d loss / d sum = 1 / count ...
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < count; e += 1024) {
        // element e = (ray, stage, channel): the target's is (ray, channel); one stage: the same index
        const int64_t te = a.stages == 1 ? e : (e / per_ray) * ch + (e % ch);
        const float x = a.pred[e] - a.target[te];
        acc += x * x;
        a.grad[e] = inv * (2.0f * x);                              // ... times d x^2 / d x = 2 x, one rounding
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);   // a fixed butterfly per wave ...
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {                                        // ... and the 16 waves in order
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += part[w];
        a.loss[0] = sum / (float)(count > 0 ? count : 1);
    }
}

}  // namespace

extern "C" int nerf_hip_adam_step(const NerfHipAdamArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "adam_step: null args");
    const NerfHipAdamArgs& a = *args;
    if (a.num_tensors < 1 || a.num_tensors > NERF_HIP_ADAM_MAX_TENSORS || a.total < 0 || a.step == nullptr ||
        a.exp_avg == nullptr || a.exp_avg_sq == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "adam_step: tensor count / state pointers out of range");
    if (a.offsets[0] != 0 || a.offsets[a.num_tensors] != a.total)
        return nerf_common::fail(NERF_HIP_EINVAL, "adam_step: offsets must run from 0 to total");
    for (int t = 0; t < a.num_tensors; ++t)
        if (a.params[t] == nullptr || a.grads[t] == nullptr || a.offsets[t + 1] < a.offsets[t])
            return nerf_common::fail(NERF_HIP_EINVAL, "adam_step: null tensor or decreasing offsets");
    if (a.total == 0) return NERF_HIP_OK;
    AdamKernelArgs ka;
    ka.a = a;
    // every launch uses ALL the slots (idle workgroups only count), so the copies stay equal whatever `total` is
    nerf_common::TimedLaunch timed((hipStream_t)stream, NERF_HIP_TIMING_ADAM);
    hipLaunchKernelGGL(nerf_adam_kernel, dim3(NERF_HIP_ADAM_STEP_SLOTS), dim3(256), 0, (hipStream_t)stream, ka);
    return nerf_common::check_hip(hipGetLastError(), "adam_step launch");
}

__global__ void nerf_rng_advance_kernel(uint64_t* counter, uint64_t delta) { *counter += delta; }

extern "C" int nerf_hip_rng_advance(uint64_t* counter, uint64_t delta, void* stream) {
    if (counter == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "rng_advance: null counter");
    hipLaunchKernelGGL(nerf_rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
    return nerf_common::check_hip(hipGetLastError(), "rng_advance launch");
}

extern "C" int nerf_hip_mse_loss(const NerfHipMseArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "mse_loss: null args");
    const NerfHipMseArgs& a = *args;
    if (a.n_rays < 0 || a.stages < 1 || a.loss == nullptr || a.channels < 0 || a.channels > NERF_HIP_MAX_COLORS)
        return nerf_common::fail(NERF_HIP_EINVAL, "mse_loss: n_rays / stages / loss pointer out of range");
    if (a.n_rays > 0 && (a.pred == nullptr || a.target == nullptr || a.grad == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "mse_loss: null tensor");
    MseKernelArgs ka;
    ka.a = a;
    nerf_common::TimedLaunch timed((hipStream_t)stream, NERF_HIP_TIMING_LOSS);
    hipLaunchKernelGGL(nerf_mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ka);
    return nerf_common::check_hip(hipGetLastError(), "mse_loss launch");
}

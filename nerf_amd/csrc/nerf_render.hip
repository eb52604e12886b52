// Fused volume-render forward for gfx950 (MI355X): one persistent launch does
//   ray generation (nerf/model.py:243-278, :337-367)  ->  fenceposts (:369-435)
//   -> conical-frustum Gaussians (:33-45, :56-87, :112-136)
//   -> integrated positional encoding (:139-163, :24-30)
//   -> 6 x Linear with LayerNorm + ReLU on exact-fp32 MFMA (:525-542)
//   -> alpha compositing, RGB sum and segmentation log-sum-exp (:438-469, :660-663).
// Layout and the weight image are described in nerf_layout.h.  Written for gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nerf_hip.h"
#include "nerf_layout.h"
#include "nerf_common.h"

using namespace nerf_layout;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 3;                        // LDS ring slots (one stage each)
constexpr int kSmallLdsBytes = 16384;           // small image (15,616 B) padded
constexpr int kLdsBytes = kSmallLdsBytes + kRing * kStageBytes;   // 64 KiB -> 2 workgroups / CU
constexpr int kWavesPerWg = 4;
constexpr int kSamplesPerWave = 16;

struct KernelArgs {
    NerfHipRenderArgs a;
    int32_t intervals;          // P = S - 1
    int32_t chunks;             // ceil(P / 16)
    int64_t groups;             // ceil(n_rays / 4)
};

// ---------------------------------------------------------------------------------------------
// weight stream: global -> LDS by LDS-DMA, two stages ahead of the MFMAs
// ---------------------------------------------------------------------------------------------
struct WeightPipe {
    const char* blob;           // packed image, stage 0
    char* ring;                 // LDS ring base
    int64_t to_issue;           // stages still to be issued by this workgroup
    int issue_stage;            // next stage of the image to issue (0..73, cyclic)
    int issue_slot;             // ring slot it goes to
    int read_slot;              // ring slot of the stage being consumed
    int wave;                   // wave id in the workgroup (uniform)
    int lane;

    __device__ __forceinline__ void issue() {
        if (to_issue > 0) {
            const char* src = blob + (size_t)issue_stage * kStageBytes + wave * 4096 + lane * 16;
            char* dst = ring + issue_slot * kStageBytes + wave * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + i * 1024),
                    (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
            --to_issue;
        }
        issue_stage = (issue_stage + 1 == kNumStages) ? 0 : issue_stage + 1;
        issue_slot = (issue_slot + 1 == kRing) ? 0 : issue_slot + 1;
    }

    // Top of a stage: own DMA pieces of this stage have landed (the 4 youngest = next stage
    // may still fly), every wave has passed the barrier, so (a) all 16 pieces are visible and
    // (b) nobody still reads the slot the next issue overwrites.
    __device__ __forceinline__ const f32x4* begin_stage() {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const f32x4* p = (const f32x4*)(ring + read_slot * kStageBytes) + lane;
        read_slot = (read_slot + 1 == kRing) ? 0 : read_slot + 1;
        return p;
    }
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 16 KiB stage of a 16-out-tile layer: k-group t against all 16 out tiles.
__device__ __forceinline__ void stage_wide(const f32x4* st, f32x4 (&acc)[16], float b0, float b1,
                                           float b2, float b3) {
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
        const f32x4 a0 = st[(2 * tp) * 64];
        const f32x4 a1 = st[(2 * tp + 1) * 64];
        acc[2 * tp] = mfma4(a0.x, b0, acc[2 * tp]);
        acc[2 * tp + 1] = mfma4(a1.x, b0, acc[2 * tp + 1]);
        acc[2 * tp] = mfma4(a0.y, b1, acc[2 * tp]);
        acc[2 * tp + 1] = mfma4(a1.y, b1, acc[2 * tp + 1]);
        acc[2 * tp] = mfma4(a0.z, b2, acc[2 * tp]);
        acc[2 * tp + 1] = mfma4(a1.z, b2, acc[2 * tp + 1]);
        acc[2 * tp] = mfma4(a0.w, b3, acc[2 * tp]);
        acc[2 * tp + 1] = mfma4(a1.w, b3, acc[2 * tp + 1]);
    }
}

template <int KT>
__device__ __forceinline__ void layer_wide(WeightPipe& pipe, f32x4 (&acc)[16],
                                           const float (&act)[64]) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const f32x4* st = pipe.begin_stage();
        stage_wide(st, acc, act[4 * t], act[4 * t + 1], act[4 * t + 2], act[4 * t + 3]);
    }
}

// Layer 5 (256 -> 64 padded): 4 stages, each 4 k-groups x 4 out tiles.
__device__ __forceinline__ void layer_out(WeightPipe& pipe, f32x4 (&acc)[4],
                                          const float (&act)[64]) {
#pragma unroll
    for (int s = 0; s < kStagesL5; ++s) {
        const f32x4* st = pipe.begin_stage();
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int t = 4 * s + tl;
            const f32x4 a0 = st[(4 * tl + 0) * 64];
            const f32x4 a1 = st[(4 * tl + 1) * 64];
            const f32x4 a2 = st[(4 * tl + 2) * 64];
            const f32x4 a3 = st[(4 * tl + 3) * 64];
            acc[0] = mfma4(a0.x, act[4 * t], acc[0]);
            acc[1] = mfma4(a1.x, act[4 * t], acc[1]);
            acc[2] = mfma4(a2.x, act[4 * t], acc[2]);
            acc[3] = mfma4(a3.x, act[4 * t], acc[3]);
            acc[0] = mfma4(a0.y, act[4 * t + 1], acc[0]);
            acc[1] = mfma4(a1.y, act[4 * t + 1], acc[1]);
            acc[2] = mfma4(a2.y, act[4 * t + 1], acc[2]);
            acc[3] = mfma4(a3.y, act[4 * t + 1], acc[3]);
            acc[0] = mfma4(a0.z, act[4 * t + 2], acc[0]);
            acc[1] = mfma4(a1.z, act[4 * t + 2], acc[1]);
            acc[2] = mfma4(a2.z, act[4 * t + 2], acc[2]);
            acc[3] = mfma4(a3.z, act[4 * t + 2], acc[3]);
            acc[0] = mfma4(a0.w, act[4 * t + 3], acc[0]);
            acc[1] = mfma4(a1.w, act[4 * t + 3], acc[1]);
            acc[2] = mfma4(a2.w, act[4 * t + 3], acc[2]);
            acc[3] = mfma4(a3.w, act[4 * t + 3], acc[3]);
        }
    }
}

__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }

// LayerNorm(256, eps 1e-5, affine, biased variance) + ReLU on the accumulator tile, result
// written as the next layer's B operands.  A sample's 256 features sit in 64 registers of
// each of the 4 lanes {j, j+16, j+32, j+48}.
__device__ __forceinline__ void layer_norm_relu(const float* small_l, int g, const f32x4 (&acc)[16],
                                                float (&act)[64]) {
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < 16; ++T) s += (acc[T].x + acc[T].y) + (acc[T].z + acc[T].w);
    s += xor16(s);
    s += xor32(s);
    const float mean = s * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int T = 0; T < 16; ++T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = acc[T][r] - mean;
            v = __builtin_fmaf(d, d, v);
        }
    }
    v += xor16(v);
    v += xor32(v);
    const float rstd = 1.0f / __builtin_sqrtf(v * (1.0f / 256.0f) + 1e-5f);
    const f32x4* gam = (const f32x4*)(small_l + kHidden) + g * 16;
    const f32x4* bet = (const f32x4*)(small_l + 2 * kHidden) + g * 16;
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        const f32x4 ga = gam[T], be = bet[T];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y = __builtin_fmaf((acc[T][r] - mean) * rstd, ga[r], be[r]);
            act[4 * T + r] = __builtin_fmaxf(y, 0.f);
        }
    }
}

__device__ __forceinline__ void load_bias16(const float* small_l, int g, f32x4 (&acc)[16]) {
    const f32x4* b = (const f32x4*)small_l + g * 16;
#pragma unroll
    for (int T = 0; T < 16; ++T) acc[T] = b[T];
}

// ---------------------------------------------------------------------------------------------
// front end: ray, fenceposts, Gaussian, IPE  (all fp32, unfused like the reference's ATen ops)
// ---------------------------------------------------------------------------------------------
struct Ray {
    float o[3], d[3];
};

__device__ __forceinline__ Ray load_ray(const NerfHipRenderArgs& a, int64_t local) {
#pragma clang fp contract(off)
    Ray r;
    if (a.rays_o != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = a.rays_o[local * 3 + k];
            r.d[k] = a.rays_d[local * 3 + k];
        }
    } else {
        // nerf/model.py:271-278 (pixel grid, ij indexing) and :367 (R . ray, summed left to right)
        const int64_t gid = a.ray_begin + local;
        const int64_t hw = (int64_t)a.image_h * a.image_w;
        const int64_t b = gid / hw;
        const int64_t pix = gid - b * hw;
        const int row = (int)(pix / a.image_w), col = (int)(pix - (int64_t)row * a.image_w);
        const float x = ((float)col - 0.5f * (float)(a.image_w - 1)) / a.focal_length;
        const float y = ((float)row - 0.5f * (float)(a.image_h - 1)) / a.focal_length;
        const float c0 = x, c1 = -y, c2 = -1.0f;
        const float* R = a.camera_r + b * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.d[k] = (R[3 * k] * c0 + R[3 * k + 1] * c1) + R[3 * k + 2] * c2;
            r.o[k] = a.camera_o[b * 3 + k];
        }
    }
    return r;
}

// Fencepost s of a ray (nerf/model.py:414-435), s clamped to the table.
__device__ __forceinline__ float fencepost(const NerfHipRenderArgs& a, int64_t local, int s) {
#pragma clang fp contract(off)
    const int S = a.num_samples;
    s = s < S - 1 ? s : S - 1;
    if (a.t_values != nullptr) return a.t_values[local * S + s];
    const float cur = a.t_table[s];
    float t = cur;
    const bool draw = (a.rng_mode & 1) != 0;
    if (a.u != nullptr || draw) {
        const float lower = s == 0 ? cur : 0.5f * (cur + a.t_table[s - 1]);
        const float upper = s == S - 1 ? cur : 0.5f * (a.t_table[s + 1] + cur);
        const float uu = a.u != nullptr ? a.u[local * S + s]
                                        : nerf_rng::uniform(a.rng_seed, a.rng_offset,
                                                            (uint64_t)(a.ray_begin + local), (uint32_t)s, 0u);
        t = lower + (upper - lower) * uu;
    }
    return t * a.t_scale;
}

struct Gaussian {
    float mean[3], cov[3];
};

// conical_frustum_to_gaussian(stable) + lift_gaussian(diag) + origin shift.
__device__ __forceinline__ Gaussian frustum(const Ray& r, float t0, float t1, float base_radius_sq) {
#pragma clang fp contract(off)
    const float c415 = (float)(4.0 / 15.0), c512 = (float)(5.0 / 12.0);
    const float mu = (t0 + t1) / 2.0f;
    const float hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    const float denom = 3.0f * mu2 + hw2;
    const float t_mean = mu + (2.0f * mu * hw2) / denom;
    const float t_var = hw2 / 3.0f - c415 * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
    const float r_var = base_radius_sq * ((mu2 / 4.0f + c512 * hw2) - (c415 * hw4) / denom);
    const float d0 = r.d[0] * r.d[0], d1 = r.d[1] * r.d[1], d2 = r.d[2] * r.d[2];
    const float mag = __builtin_fmaxf((d0 + d1) + d2, 1e-10f);
    const float dsq[3] = {d0, d1, d2};
    Gaussian g;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.mean[k] = r.d[k] * t_mean + r.o[k];
        g.cov[k] = t_var * dsq[k] + r_var * (1.0f - dsq[k] / mag);
    }
    return g;
}

// 24 encoded features of this lane group (layout: nerf_layout.h).
__device__ __forceinline__ void encode(const Gaussian& gs, int g, float (&act)[64]) {
#pragma clang fp contract(off)
    const float base = __builtin_ldexpf(1.0f, 4 * g - 4);       // 2^(4g-4): scales 4g..4g+3 of -4..11
    const float half_pi = 1.5707963267948966f;
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        const float scale = base * (float)(1 << (p / 3));
        const float y = gs.mean[p % 3] * scale;
        const float yv = gs.cov[p % 3] * (scale * scale);
        const float damp = expf(-0.5f * yv);
        act[p] = damp * sinf(y);
        act[12 + p] = damp * sinf(y + half_pi);
    }
}

__device__ __forceinline__ float row_shfl_up(float v, int d) { return __shfl_up(v, d, 16); }
__device__ __forceinline__ float row_shfl_xor(float v, int d) { return __shfl_xor(v, d, 16); }

__device__ __forceinline__ float row_sum(float v) {
    v += row_shfl_xor(v, 1);
    v += row_shfl_xor(v, 2);
    v += row_shfl_xor(v, 4);
    v += row_shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = __builtin_fmaxf(v, row_shfl_xor(v, 1));
    v = __builtin_fmaxf(v, row_shfl_xor(v, 2));
    v = __builtin_fmaxf(v, row_shfl_xor(v, 4));
    v = __builtin_fmaxf(v, row_shfl_xor(v, 8));
    return v;
}

// Output slot n = 16 T + 4 g + reg of the padded last layer: 0 density, 1..3 color,
// 4..53 segmentation classes, 54..63 padding (nerf/model.py:591-592).
__device__ __forceinline__ bool is_seg_slot(int T, int g, int reg) {
    const int n = 16 * T + 4 * g + reg;
    return n >= 4 && n < kOut;
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void nerf_render_fwd_kernel(const KernelArgs ka) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NerfHipRenderArgs& a = ka.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int P = ka.intervals;
    const int chunks = ka.chunks;

    // small image -> LDS (once per workgroup)
    {
        const float* small_g = a.packed + kBlobFloats;
        float* small_l = (float*)smem;
        for (int i = threadIdx.x; i < kSmallFloats; i += 256) small_l[i] = small_g[i];
    }
    const float* small = (const float*)smem;

    const int64_t my_groups = ka.groups > (int64_t)blockIdx.x
                                  ? (ka.groups - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    WeightPipe pipe;
    pipe.blob = (const char*)a.packed;
    pipe.ring = smem + kSmallLdsBytes;
    pipe.to_issue = my_groups * chunks * kNumStages;
    pipe.issue_stage = 0;
    pipe.issue_slot = 0;
    pipe.read_slot = 0;
    pipe.wave = wave;
    pipe.lane = lane;
    pipe.issue();
    pipe.issue();
    __syncthreads();          // small image visible (this also drains the two DMA stages once)

    float act[64];
    f32x4 acc[16];

    for (int64_t grp = blockIdx.x; grp < ka.groups; grp += gridDim.x) {
        int64_t local = grp * kWavesPerWg + wave;
        const bool ray_ok = local < a.n_rays;
        if (!ray_ok) local = a.n_rays - 1;
        const Ray ray = load_ray(a, local);

        float carry = 1.0f;                     // prod (alpha_i + 1e-10) over finished chunks
        float rgb0 = 0.f, rgb1 = 0.f, rgb2 = 0.f;
        float segM[16], segS[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            segM[i] = -__builtin_inff();
            segS[i] = 0.f;
        }

        for (int c = 0; c < chunks; ++c) {
            const int s = c * kSamplesPerWave + j;
            const bool ok = s < P;
            const float t0 = fencepost(a, local, s);
            const float t1 = fencepost(a, local, s + 1);
            const float t2 = fencepost(a, local, s + 2);
            const Gaussian gs = frustum(ray, t0, t1, a.base_radius_sq);
            const Gaussian gn = frustum(ray, t1, t2, a.base_radius_sq);
            encode(gs, g, act);

            // ---- layer 0: 96 -> 256 ----
            load_bias16(small, g, acc);
            layer_wide<kStagesL0>(pipe, acc, act);
            layer_norm_relu(small, g, acc, act);
            // ---- layers 1..4: 256 -> 256 ----
#pragma unroll 1
            for (int L = 1; L <= 4; ++L) {
                const float* small_l = small + L * kSmallPerLayer;
                load_bias16(small_l, g, acc);
                layer_wide<kStagesHidden>(pipe, acc, act);
                layer_norm_relu(small_l, g, acc, act);
            }
            // ---- layer 5: 256 -> 54 (padded 64) ----
            f32x4 out[4];
            {
                const f32x4* b = (const f32x4*)(small + 5 * kSmallPerLayer) + g * 4;
#pragma unroll
                for (int T = 0; T < 4; ++T) out[T] = b[T];
            }
            layer_out(pipe, out, act);

            // ---- compositing (nerf/model.py:438-469, :660-663) ----
            {
#pragma clang fp contract(off)
                float dens = __shfl(out[0].x, j);
                if (a.noise != nullptr) {
                    if (ok) dens = dens + a.noise[local * P + s] * a.density_noise_std;
                } else if (a.rng_mode & 2) {
                    dens = dens + nerf_rng::normal(a.rng_seed, a.rng_offset,
                                                   (uint64_t)(a.ray_begin + local), (uint32_t)s, 1u)
                                      * a.density_noise_std;
                }
                const float e0 = gn.mean[0] - gs.mean[0], e1 = gn.mean[1] - gs.mean[1],
                            e2 = gn.mean[2] - gs.mean[2];
                float dist = __builtin_sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
                if (s == P - 1) dist = 1e10f;
                const float alpha = ok ? expf(-__builtin_fmaxf(dens, 0.f) * dist) : 1.0f;
                float prod = ok ? alpha + 1e-10f : 1.0f;      // inclusive scan over the 16 lanes
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const float up = row_shfl_up(prod, d);
                    if (j >= d) prod *= up;
                }
                float excl = row_shfl_up(prod, 1);
                if (j == 0) excl = 1.0f;
                const float w = ok ? (1.0f - alpha) * (carry * excl) : 0.f;
                carry = carry * __shfl(prod, (lane & 48) | 15);

                // RGB: valid on lane group 0, harmless elsewhere
                const float cr = w * (1.0f / (1.0f + expf(-out[0].y)));
                const float cg = w * (1.0f / (1.0f + expf(-out[0].z)));
                const float cb = w * (1.0f / (1.0f + expf(-out[0].w)));
                rgb0 += row_sum(cr);
                rgb1 += row_sum(cg);
                rgb2 += row_sum(cb);

                if (a.seg != nullptr) {
                    // log_softmax over the 50 class logits of this sample
                    float m = -__builtin_inff();
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (is_seg_slot(T, g, r)) m = __builtin_fmaxf(m, out[T][r]);
                    m = __builtin_fmaxf(m, xor16(m));
                    m = __builtin_fmaxf(m, xor32(m));
                    float z = 0.f;
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (is_seg_slot(T, g, r)) z += expf(out[T][r] - m);
                    z += xor16(z);
                    z += xor32(z);
                    const float logz = logf(z);
                    const float lw = logf(w + 1e-10f);
                    // online log-sum-exp over the samples this lane sees
#pragma unroll
                    for (int T = 0; T < 4; ++T)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * T + r;
                            const float v = lw + ((out[T][r] - m) - logz);
                            if (ok) {
                                const float nm = __builtin_fmaxf(segM[i], v);
                                segS[i] = segS[i] * expf(segM[i] - nm) + expf(v - nm);
                                segM[i] = nm;
                            }
                        }
                }

                // optional per-sample outputs (NeRF.forward, nerf/model.py:553-594)
                if (ray_ok && ok) {
                    const int64_t smp = local * P + s;
                    if (a.out_raw != nullptr) {
#pragma unroll
                        for (int T = 0; T < 4; ++T)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int n = 16 * T + 4 * g + r;
                                if (n < kOut) a.out_raw[smp * kOut + n] = out[T][r];
                            }
                    }
                    if (g == 0) {
                        if (a.out_mean != nullptr) {
                            a.out_mean[smp * 3 + 0] = gs.mean[0];
                            a.out_mean[smp * 3 + 1] = gs.mean[1];
                            a.out_mean[smp * 3 + 2] = gs.mean[2];
                        }
                        if (a.out_weights != nullptr) a.out_weights[smp] = w;
                    }
                }
            }
        }

        // ---- ray epilogue: one coalesced store instruction per output row ----
        if (ray_ok && lane < 3) {
            const float v = lane == 0 ? rgb0 : (lane == 1 ? rgb1 : rgb2);
            a.rgb[local * 3 + lane] = v;
        }
        if (a.seg != nullptr) {
            // after the row reductions every lane of a row holds the same 16 values; lane (j, g)
            // keeps slot i = j, i.e. output n = 16 (j >> 2) + 4 g + (j & 3), so the wave's 64 lanes
            // cover n = 0..63 once and the 50 class values leave in one store instruction.
            float mine = 0.f;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * T + r;
                    const float mx = row_max(segM[i]);
                    const float sm = row_sum(segS[i] * expf(segM[i] - mx));
                    const float val = mx + logf(sm);
                    if (j == i) mine = val;
                }
            const int n = 16 * (j >> 2) + 4 * g + (j & 3);
            if (ray_ok && n >= 4 && n < kOut) a.seg[local * kSegClasses + (n - 4)] = mine;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// parameter re-layout (state_dict order -> packed image)
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* p[NERF_HIP_NUM_PARAM_TENSORS];
    float* packed;
};

__global__ void nerf_pack_kernel(const PackArgs pa) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kPackedFloats) return;
    float v = 0.f;
    if (e < kBlobFloats) {
        const int stage = e / kStageFloats;
        const int in_stage = e - stage * kStageFloats;
        const int quad = in_stage / kQuadFloats;
        const int lane = (in_stage % kQuadFloats) / 4, r = in_stage & 3;
        const int row = lane & 15, g = lane >> 4;
        if (stage < kStagesL0) {                                    // layer 0: W[256,96]
            const int out = 16 * quad + row;
            v = pa.p[0][out * kEncIn + layer0_source_feature(stage, g, r)];
        } else if (stage < kStagesL0 + 4 * kStagesHidden) {         // layers 1..4: W[256,256]
            const int L = 1 + (stage - kStagesL0) / kStagesHidden;
            const int t = (stage - kStagesL0) % kStagesHidden;
            const int out = 16 * quad + row;
            v = pa.p[4 * L][out * kHidden + 16 * t + 4 * g + r];
        } else {                                                    // layer 5: W[54,256]
            const int s = stage - (kStagesL0 + 4 * kStagesHidden);
            const int t = 4 * s + quad / 4, T = quad % 4;
            const int out = 16 * T + row;
            if (out < kOut) v = pa.p[20][out * kHidden + 16 * t + 4 * g + r];
        }
    } else {
        const int i = e - kBlobFloats;
        if (i < 5 * kSmallPerLayer) {
            const int L = i / kSmallPerLayer, rem = i % kSmallPerLayer;
            const int which = rem / kHidden, q = rem % kHidden;     // which: bias, gamma, beta
            const int g = q / 64, T = (q % 64) / 4, reg = q & 3;
            const int f = 16 * T + 4 * g + reg;
            const int tensor = which == 0 ? 4 * L + 1 : (which == 1 ? 4 * L + 2 : 4 * L + 3);
            v = pa.p[tensor][f];
        } else {
            const int q = i - 5 * kSmallPerLayer;                   // last bias [g][T(4)][reg]
            const int g = q / 16, T = (q % 16) / 4, reg = q & 3;
            const int n = 16 * T + 4 * g + reg;
            if (n < kOut) v = pa.p[21][n];
        }
    }
    pa.packed[e] = v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int nerf_hip_version(void) { return NERF_HIP_ABI_VERSION; }

const char* nerf_hip_last_error(void) { return nerf_common::last_error(); }

size_t nerf_hip_packed_bytes(void) { return (size_t)kPackedFloats * sizeof(float); }

int nerf_hip_pack_weights(const float* const* params, float* packed, void* stream) {
    if (params == nullptr || packed == nullptr)
        return nerf_common::fail(NERF_HIP_EINVAL, "pack_weights: null pointer");
    PackArgs pa;
    for (int i = 0; i < NERF_HIP_NUM_PARAM_TENSORS; ++i) {
        if (params[i] == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "pack_weights: null tensor");
        pa.p[i] = params[i];
    }
    pa.packed = packed;
    const int threads = 256, blocks = (kPackedFloats + threads - 1) / threads;
    hipLaunchKernelGGL(nerf_pack_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, pa);
    return nerf_common::check_hip(hipGetLastError(), "pack_weights launch");
}

int nerf_hip_render_forward(const NerfHipRenderArgs* args, void* stream) {
    if (args == nullptr) return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: null args");
    const NerfHipRenderArgs& a = *args;
    if (a.n_rays == 0) return NERF_HIP_OK;
    if (a.n_rays < 0 || a.num_samples < 2 || a.num_samples > 4096)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: n_rays / num_samples out of range");
    if (a.packed == nullptr || a.rgb == nullptr || (a.t_table == nullptr && a.t_values == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: packed / rgb / t_table is null");
    const bool arrays = a.rays_o != nullptr && a.rays_d != nullptr;
    const bool cameras = a.camera_o != nullptr && a.camera_r != nullptr && a.image_h > 0 &&
                         a.image_w > 0 && a.focal_length != 0.f;
    if (!arrays && !cameras)
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: neither ray arrays nor cameras given");
    if ((a.rays_o == nullptr) != (a.rays_d == nullptr))
        return nerf_common::fail(NERF_HIP_EINVAL, "render_forward: rays_o and rays_d must come together");

    KernelArgs ka;
    ka.a = a;
    ka.intervals = a.num_samples - 1;
    ka.chunks = (ka.intervals + kSamplesPerWave - 1) / kSamplesPerWave;
    ka.groups = (a.n_rays + kWavesPerWg - 1) / kWavesPerWg;

    int device = 0, cus = 0;
    int rc = nerf_common::check_hip(hipGetDevice(&device), "hipGetDevice");
    if (rc) return rc;
    rc = nerf_common::check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device),
                                "hipDeviceGetAttribute");
    if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        rc = nerf_common::check_hip(
            hipFuncSetAttribute((const void*)nerf_render_fwd_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes),
            "hipFuncSetAttribute");
        if (rc) return rc;
        attr_set = true;
    }
    int64_t grid = (int64_t)cus * 2;              // 2 workgroups per CU (64 KiB LDS, <= 256 VGPRs)
    if (grid > ka.groups) grid = ka.groups;
    hipStream_t st = (hipStream_t)stream;
    nerf_common::Timing::before(st);
    hipLaunchKernelGGL(nerf_render_fwd_kernel, dim3((unsigned)grid), dim3(256), kLdsBytes, st, ka);
    rc = nerf_common::check_hip(hipGetLastError(), "render_forward launch");
    nerf_common::Timing::after(st);
    return rc;
}

int nerf_hip_timing(int enable) { return nerf_common::Timing::enable(enable != 0); }

int nerf_hip_timing_read(int reset, double* avg_ms, int64_t* launches) {
    return nerf_common::Timing::read(reset != 0, avg_ms, launches);
}

}  // extern "C"

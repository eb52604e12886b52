// The split-precision ("f16x3") fused layer shared by the two networks' forward kernels (gfx950 only): the
// unit-pipelined MFMA loop over an LDS-DMA weight ring with the LayerNorm fused into it — the next layer's
// B operands are normalised lazily and split into f16 pairs in the MFMA shadow of the stage before the one
// that consumes them, and the moments of the outputs are gathered while the second half of the layer runs.
#ifndef NERF_FUSED_H
#define NERF_FUSED_H

#include "nerf_device.h"

namespace nerf_fused {

using namespace nerf_layout;
using namespace nerf_device;


// ---------------------------------------------------------------------------------------------
// LayerNorm(256, eps 1e-5, affine, biased variance) + ReLU, fused INTO the MFMA loops around it.
//
// A sample's 256 features sit in 64 registers of each of the 4 lanes {j, j+16, j+32, j+48}; the
// accumulator tile T (f32x4) of a layer is, after normalisation, the B operand of k-group T of the
// next layer.  So the normalisation is deferred and applied in place, tile by tile, one stage
// ahead of the stage that consumes the tile, and the moments are accumulated tile pair by tile
// pair during the layer's last stage, one MFMA group behind the group that finishes the pair:
// both passes issue in the shadow of MFMAs (an MFMA leaves ~6 VALU issue slots) instead of in a
// VALU-only phase between two layers.  What stays exposed per layer: the moments of the last tile
// pair, two cross-lane-group sums, the rsqrt, and the normalisation of tile 0.
// ---------------------------------------------------------------------------------------------
// Two layer orders share this code.  kOrderNormRelu: Linear -> LayerNorm -> ReLU (the network of nerf/model.py):
// moments over the raw outputs y, input of the next layer = relu(gamma y_hat + beta).  kOrderReluNorm: Linear
// -> ReLU -> LayerNorm (the legacy network of examples/nerf.pth): moments over relu(y), input of the next
// layer = gamma a_hat + beta with a_hat = (relu(y) - mean) / std.
constexpr int kOrderNormRelu = 0, kOrderReluNorm = 1;

struct LazyNorm {
    float rstd, shift;          // x_hat = fma(x, rstd, shift), shift = -mean * rstd
    const f32x4* gam;           // this lane group's gamma / beta in LDS: tile T at [T]
    const f32x4* bet;
    float* save_row;            // training: this lane's x_hat of register tile 0 (tile T at + kTileT T), else unused.
                                // kOrderReluNorm (the legacy network's kernels): the wave's UNIFORM tile base, the
                                // lane's offset inside the tile is taken at every use (nerf_device.h: row_lane_offset)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Moments {
    float s, q;
    __device__ __forceinline__ void reset() { s = q = 0.f; }
    __device__ __forceinline__ float sum() const { return s; }
    __device__ __forceinline__ float sum_sq() const { return q; }
    template <int kOrder = kOrderNormRelu>
    __device__ __forceinline__ void add(const f32x4& raw) {
        const f32x4 v = kOrder == kOrderReluNorm ? __builtin_elementwise_max(raw, f32x4{0.f, 0.f, 0.f, 0.f}) : raw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s += v[r];
            q = __builtin_fmaf(v[r], v[r], q);
        }
    }
};

// Same with two partial sums each, so that the adds and fmas go out as packed fp32 instructions
// (v_pk_add_f32 / v_pk_fma_f32: two values per issue slot); used where VALU issue is the limit
// (the split-precision path).  Packed operands need aligned register pairs, which costs the
// fp32 kernels more registers than it saves them issue slots.
struct MomentsPk {
    f32x2 s, q;
    __device__ __forceinline__ void reset() { s = q = f32x2{0.f, 0.f}; }
    __device__ __forceinline__ float sum() const { return s.x + s.y; }
    __device__ __forceinline__ float sum_sq() const { return q.x + q.y; }
    template <int kOrder = kOrderNormRelu>
    __device__ __forceinline__ void add(const f32x4& raw) {
        const f32x4 v = kOrder == kOrderReluNorm ? __builtin_elementwise_max(raw, f32x4{0.f, 0.f, 0.f, 0.f}) : raw;
        const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
        s += a;
        q = a * a + q;
        s += b;
        q = b * b + q;
    }
};

template <bool kTrain, bool kPacked = false, int kOrder = kOrderNormRelu>
__device__ __forceinline__ void normalize_tile(f32x4& x, const LazyNorm& n, int T, const f32x4& ga,
                                               const f32x4& be) {
    f32x4 xh;
    if (kOrder == kOrderReluNorm) {
        // training: a_hat is saved; the backward reads the ReLU gate as a_hat > shift (the a_hat of a == 0),
        // so an OPEN gate whose a_hat rounds onto `shift` is moved one ulp up (nerf_legacy.hip: relu_layer_norm)
        const float above = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, n.shift) - 1u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = __builtin_fmaf(__builtin_fmaxf(x[r], 0.f), n.rstd, n.shift);
            if (kTrain) xh[r] = (x[r] > 0.f && xh[r] <= n.shift) ? above : xh[r];
            x[r] = __builtin_fmaf(xh[r], ga[r], be[r]);
        }
        if (kTrain) *(f32x4*)(n.save_row + row_lane_offset() + T * kTileT) = xh;
        return;
    }
    if (kPacked) {
        xh = x * n.rstd + n.shift;              // packed fp32 fmas, two values per instruction
        x = __builtin_elementwise_max(xh * ga + be, f32x4{0.f, 0.f, 0.f, 0.f});
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = __builtin_fmaf(x[r], n.rstd, n.shift);
            x[r] = __builtin_fmaxf(__builtin_fmaf(xh[r], ga[r], be[r]), 0.f);
        }
    }
    if (kTrain) *(f32x4*)(n.save_row + T * kTileT) = xh;
}
template <bool kTrain, bool kPacked = false, int kOrder = kOrderNormRelu>
__device__ __forceinline__ void normalize_tile(f32x4& x, const LazyNorm& n, int T) {
    normalize_tile<kTrain, kPacked, kOrder>(x, n, T, n.gam[T], n.bet[T]);
}

// Moments -> the deferred normalisation of `raw` (the layer's finished accumulators).
// var = E[x^2] - mean^2 cancels when |mean| >> std, so whenever the mean carries more than 3/4 of
// the second moment in ANY sample of the wave, a second pass (the two-pass variance) is taken instead
// (wave-uniform branch; pre-LayerNorm activations of this network have |mean| well below std, so
// it is cold).  1/sqrt: hardware estimate (1 ulp) + one Newton step.
template <bool kTrain, class Mom, int kOrder = kOrderNormRelu, int NT = 16>
__device__ __forceinline__ LazyNorm finish_moments_at(const Mom& m, const f32x4 (&raw)[16], const f32x4* gam,
                                                      const f32x4* bet, int g, float* save_row,
                                                      float* save_rstd, float eps = 1e-5f,
                                                      float save_scale = 1.0f, float* save_shift = nullptr,
                                                      const NormDivisor nd = kFullWidth) {
    // (kOrderReluNorm: save_row / save_rstd / save_shift are wave-uniform bases of the sample tile, the lane's offsets
    //  are added where they are used; else per-lane pointers)
    const float mean = group_sum(m.sum()) * nd.inv_n;
    const float ex2 = group_sum(m.sum_sq()) * nd.inv_n;
    float var = ex2 - mean * mean;
    if (__builtin_amdgcn_ballot_w64(mean * mean > 0.75f * ex2) != 0) {
        // second pass, the true two-pass form sum (x - mean)^2 over the REAL features: an error d in the rounded mean
        // enters as hidden * d^2 (second order: the cross term is d * sum (x - mean) = 0), so the variance keeps its
        // ~eps relative error however large |mean| / std is — sum (x - mean) x, which this branch used in round 5, is
        // first-order in d (error d * mean * hidden: the amplification mean^2 / var of the one-pass form again;
        // ADVICE r5).  A narrower network's padded features are exactly 0 here and would each add mean^2, so they are
        // masked by feature index (lane group g holds features 16 T + 4 g + r): two more VALU per element, in a
        // branch that is cold — the pre-LayerNorm activations of this network have |mean| well below std.
        // (the mask as integer arithmetic — sign bits of 16 T + r - (real - 4 g), ANDed onto the difference — not as
        //  compares: 4 NT lane masks in scalar registers are what the narrow kernels, at their register limit, lack)
        // (... and `lim` behind an optimisation barrier IN the branch: the compiler turns the sign-bit arithmetic back
        //  into 4 NT compares, which depend on nothing but the lane and the launch and are hoisted to the top of the
        //  kernel as 64 lane masks = 128 scalar registers, parked in vector-register lanes and crowding the values the
        //  hot path needs out with them: 187 v_writelane in the prologue, ~95 v_readlane per 16-sample chunk)
        // (the legacy network's split-precision kernels, at 256 registers, keep the plain form: with the barrier their
        //  allocation moves a reload between two layers)
        int lim = nd.real - 4 * g;
        if (kOrder != kOrderReluNorm) asm volatile("" : "+v"(lim));
        float v = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = kOrder == kOrderReluNorm ? __builtin_fmaxf(raw[T][r], 0.f) : raw[T][r];
                const int real_bits = __builtin_amdgcn_sbfe(16 * T + r - lim, 31, 1);      // -1 for a real feature, else 0
                const float d = __builtin_bit_cast(float, __builtin_bit_cast(int, x - mean) & real_bits);
                v = __builtin_fmaf(d, d, v);
            }
        }
        var = group_sum(v) * nd.inv_n;
    }
    const float ve = var + eps;
    float rstd = __builtin_amdgcn_rsqf(ve);
    rstd = rstd * __builtin_fmaf(-0.5f * ve * rstd, rstd, 1.5f);
    if (kTrain && g == 0) {
        if (kOrder == kOrderReluNorm) save_rstd[stat_lane_offset()] = rstd * save_scale;
        else *save_rstd = rstd * save_scale;
    }
    LazyNorm n;
    n.rstd = rstd;
    n.shift = -mean * rstd;
    if (kTrain && kOrder == kOrderReluNorm && g == 0) save_shift[stat_lane_offset()] = n.shift;   // (x_hat, so scale-free)
    n.gam = gam;
    n.bet = bet;
    n.save_row = save_row;
    return n;
}

// gamma / beta of a layer at their place in the main network's padded LDS image (nerf_device.h)
template <bool kTrain, class Mom, int NT = 16>
__device__ __forceinline__ LazyNorm finish_moments(const Mom& m, const f32x4 (&raw)[16],
                                                   const float* small_l, int g, float* save_row,
                                                   float* save_rstd, const NormDivisor nd, float eps = 1e-5f,
                                                   float save_scale = 1.0f) {
    return finish_moments_at<kTrain, Mom, kOrderNormRelu, NT>(m, raw, (const f32x4*)(small_l + kSmallArrayLds + g * kSmallGStride),
                                          (const f32x4*)(small_l + 2 * kSmallArrayLds + g * kSmallGStride), g,
                                          save_row, save_rstd, eps, save_scale, nullptr, nd);
}

// ---------------------------------------------------------------------------------------------
// Split-precision ("f16x3") layers of the forward (inference, and the training forward on request):
// every fp32 operand is an f16 pair
// (hi, lo) and a product is three v_mfma_f32_16x16x32_f16 (hi.hi + hi.lo + lo.hi, fp32
// accumulate; the dropped lo.lo term is ~2^-22 relative), 5.3x the fp32-MFMA rate per product.
// Image and scalings: nerf_layout.h.  A k block m = register tiles 2m, 2m+1 of the input; its
// B operands are built (normalise lazily like layer_fused, then split) one stage ahead.
// ---------------------------------------------------------------------------------------------
// packed fp32 in the normalisation costs aligned register pairs: at this register pressure it
// spills inside the layer loops (150 ms per frame against 136), so only the moments are packed
constexpr bool kPackNorm = false;
typedef MomentsPk HMoments;
// A 16-out-tile layer over KB k blocks.  Stage order (half, m): out tiles 0..7 are complete after
// the first KB stages, so their moments ride in the second half; the B operands of block m + 1 are
// built (normalise tile by tile, then split) during stage (0, m).
// NT = 8 (a narrow network, nerf_layout.h: kNarrowH8Offset): ONE half — a stage = the eight out tiles of k block m —
// so every tile completes in the last stage and the moments ride there, a unit behind.
// layer_fused_hb: the same with the B operands (the f16 pairs of the normalised input, k block by k block) in the
// CALLER's registers, for a caller that multiplies the same input once more behind the layer (the legacy network's
// density head on x'_7): it then reads the pairs instead of keeping the 16 fp32 input tiles alive across this layer.
template <int KB, bool kNormIn, bool kTrain, int kOrder = kOrderNormRelu, int NT = 16, class Pipe>
__device__ __forceinline__ void layer_fused_hb(Pipe& pipe, f32x4 (&in)[16], f32x4 (&out)[16],
                                               const LazyNorm& norm, HMoments& mom, h8 (&bhi)[KB], h8 (&blo)[KB]) {
    static_assert(NT == 16 || NT == 8, "out tiles in halves of eight");
    constexpr int kHalves = NT / 8;
    constexpr int kStages = kHalves * KB, kUnits = 8 * kStages;
    if (kNormIn) {
        normalize_tile<kTrain, kPackNorm, kOrder>(in[0], norm, 0);
        normalize_tile<kTrain, kPackNorm, kOrder>(in[1], norm, 1);
    }
    split8(in[0], in[1], bhi[0], blo[0]);
    mom.reset();
    h8 ah[kSets], al[kSets];
    f32x4 ga, be;
    h2 nh[4], nl[4];                                     // halves of the block being built
    __builtin_amdgcn_s_setprio(NERF_PRIO_MFMA);
    const h8* st = (const h8*)pipe.open_stage();
#pragma unroll
    for (int u = 0; u < kSets - 1; ++u) {
        ah[u] = st[(2 * u) * 64];
        al[u] = st[(2 * u + 1) * 64];
    }
    pipe.prefetch_next();
#pragma unroll
    for (int s = 0; s < kStages; ++s) {
        const int half = s / KB, m = s % KB;
        const bool build_next = half == 0 && m + 1 < KB;
        const int ta = 2 * m + 2, tb = 2 * m + 3;        // tiles of block m + 1
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int U = 8 * s + i, set = U % kSets;
            const int T = 8 * half + i;
            out[T] = mfma_h(ah[set], bhi[m], out[T]);
            __builtin_amdgcn_sched_barrier(0);
            if (U + kSets - 1 < kUnits) {
                const int ip = (i + kSets - 1) % 8, pset = (U + kSets - 1) % kSets;
                if (ip == 0) {
                    // training: the x_hat stores of this stage (units 1 and 3, before this hand-over) are younger
                    // than the DMA of the stage being opened — and so are those of the previous stage when that DMA
                    // was issued two hand-overs ago (3-slot ring); with 2 slots it was issued at the previous
                    // hand-over, BEHIND the previous stage's stores, which therefore must not be counted
                    constexpr bool kStores = kTrain && kNormIn;
                    const bool mine = kStores && build_next;
                    const bool prev = Pipe::kRingDepth == 3 && kStores && s >= 1 && s - 1 < KB - 1;
                    if (mine && prev) st = (const h8*)pipe.template open_stage<4>();
                    else if (mine || prev) st = (const h8*)pipe.template open_stage<2>();
                    else st = (const h8*)pipe.open_stage();
                }
                ah[pset] = st[(2 * ip) * 64];
                al[pset] = st[(2 * ip + 1) * 64];
                if (ip == 0) pipe.prefetch_next();
            }
            if (kNormIn && build_next && (i == 0 || i == 2)) {   // a unit ahead of their use
                ga = norm.gam[i == 0 ? ta : tb];
                be = norm.bet[i == 0 ? ta : tb];
            }
            __builtin_amdgcn_sched_barrier(0);
            out[T] = mfma_h(ah[set], blo[m], out[T]);
            out[T] = mfma_h(al[set], bhi[m], out[T]);
            // VALU riding in the shadow of this unit's MFMAs
            if (build_next) {
                if (kNormIn && i == 1) normalize_tile<kTrain, kPackNorm, kOrder>(in[ta], norm, ta, ga, be);
                if (i == 2) split4(in[ta], nh[0], nh[1], nl[0], nl[1]);
                if (kNormIn && i == 3) normalize_tile<kTrain, kPackNorm, kOrder>(in[tb], norm, tb, ga, be);
                if (i == 4) {
                    split4(in[tb], nh[2], nh[3], nl[2], nl[3]);
                    bhi[m + 1] = join8(nh[0], nh[1], nh[2], nh[3]);
                    blo[m + 1] = join8(nl[0], nl[1], nl[2], nl[3]);
                }
                if (i >= 1 && i <= 4) interleave_2<4>();
            }
            if (kHalves == 2 && half == 1) {
                // tiles 0..7 (finished in the first half), spread over the second half's stages
#pragma unroll
                for (int T2 = 0; T2 < 8; ++T2) {
                    const int first = (8 * m + KB - 1) / KB;             // first tile of stage m
                    if (T2 * KB / 8 == m && i == (s + 1 < kStages ? T2 - first : 0)) {
                        mom.template add<kOrder>(out[T2]);
                        if (s + 1 < kStages) interleave_2<2>();
                    }
                }
            }
            if (s + 1 == kStages && i >= 1) {            // tile finished one unit ago
                mom.template add<kOrder>(out[T - 1]);
                interleave_2<2>();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    mom.template add<kOrder>(out[NT - 1]);
    __builtin_amdgcn_s_setprio(NERF_PRIO_VALU);
}
template <int KB, bool kNormIn, bool kTrain, int kOrder = kOrderNormRelu, int NT = 16, class Pipe>
__device__ __forceinline__ void layer_fused_h(Pipe& pipe, f32x4 (&in)[16], f32x4 (&out)[16],
                                              const LazyNorm& norm, HMoments& mom) {
    h8 bhi[KB], blo[KB];
    layer_fused_hb<KB, kNormIn, kTrain, kOrder, NT>(pipe, in, out, norm, mom, bhi, blo);
}

}  // namespace nerf_fused
#endif

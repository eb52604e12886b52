// Shared constants of the LEGACY (generation-A) 8 x 256 network of examples/nerf.pth: the packed
// parameter images its kernels stream, the saved-for-backward workspace, and the flat gradient vector.
// PARITY UNPINNED (no source of this network is in the reference repository): SURVEY.md section 2.3,
// oracle/legacy_oracle.py.
//
// Network (structure read off the checkpoint's 44 tensors):
//   wide layers L = 0..9, each  y = W x + b ;  a = relu(y) ;  x' = LayerNorm(a) = gamma a_hat + beta
//     L0     : x = PE(position / normalize_position)        60 -> 256     block_0.0
//     L1..L3 : x = x'_{L-1}                                256 -> 256     block_0.{3,6,9}
//     L4     : x = [x'_3 | PE(position)]                   316 -> 256     block_1.0
//     L5..L7 :                                             256 -> 256     block_1.{3,6,9}
//     density head on x'_7                                 256 -> 1       density
//     L8     : x = [x'_7 | PE(direction)]                  292 -> 256     block_2.0
//     L9     :                                             256 -> 256     block_2.3
//     color head on x'_9                                   256 -> 3       color
// Parameter tensor order everywhere (pack routine, flat gradient, LegacyNeRF8x256.parameters()):
//   [W, b, gamma, beta] of L0..L7 (tensors 0..31), density W, b (32, 33), [W, b, gamma, beta] of L8, L9
//   (34..41), color W, b (42, 43).
#ifndef NERF_LEGACY_LAYOUT_H
#define NERF_LEGACY_LAYOUT_H

#include <stdint.h>

#include "nerf_layout.h"

namespace nerf_legacy {

using namespace nerf_layout;

constexpr int kPosFreqs = 10, kDirFreqs = 6;
constexpr int kPosFeatures = 3 * 2 * kPosFreqs;          // 60: 15 per lane group, padded to 16
constexpr int kDirFeatures = 3 * 2 * kDirFreqs;          // 36:  9 per lane group, padded to 12
constexpr int kPosPerGroup = 15, kDirPerGroup = 9;
constexpr int kPosTiles = 4, kDirTiles = 3;
constexpr int kEncPad = 64;                              // an encoding as a padded row: column 16 t + 4 g + r
constexpr int kWide = 10;                                // LayerNorm layers: block_0 x4, block_1 x4, block_2 x2

// ---- forward image, fp32 (nerf_layout.h stage / quad format) -----------------------------------
// stages per wide layer (k-groups of 16 input features; the concatenated encodings add 4 / 3)
__host__ __device__ constexpr int wide_stages(int L) {
    return L == 0 ? kPosTiles : (L == 4 ? 16 + kPosTiles : (L == 8 ? 16 + kDirTiles : 16));
}
// consumption order: L0..L7, density head, L8, L9, color head
constexpr int kLegacyStages = 4 + 3 * 16 + 20 + 3 * 16 + 1 + 19 + 16 + 1;      // 157
constexpr int kLegacyBlobFloats = kLegacyStages * kStageFloats;
constexpr int kLegacySmallPerLayer = 3 * kHidden;                              // bias, gamma, beta
constexpr int kHeadBiasFloats = 32;                                            // density [16], color [16]
constexpr int kLegacySmallFloats = kWide * kLegacySmallPerLayer + kHeadBiasFloats;   // 7,712
// ---- forward image, f16 pairs (slab format of the main kernel's split-precision image; weights x 2^8,
// activations enter x 2^4): wide layer L has KB = inputs / 32 k blocks (2 for layer 0, 8, or 10 with a
// concatenated encoding padded to two blocks) = 2 KB stages of 8 (out tile, k block) pairs; a head is
// one stage = the 8 k blocks of its single out tile.
__host__ __device__ constexpr int wide_blocks(int L) { return L == 0 ? 2 : ((L == 4 || L == 8) ? 10 : 8); }
constexpr int kLegacyHStages = 4 + 3 * 16 + 20 + 3 * 16 + 1 + 20 + 16 + 1;     // 158
constexpr int kLegacyHBlobFloats = kLegacyHStages * kStageFloats;
constexpr int kLegacyHOffset = kLegacyBlobFloats + kLegacySmallFloats;
constexpr int kLegacyHSmallOffset = kLegacyHOffset + kLegacyHBlobFloats;
// ---- transposed fp32 image for the data gradient (dX = W^T dY), in the order the chain consumes it:
//   stage 0        color head      (k = its 16 padded outputs; slot (g 0, r 1..3) = color rows, rest 0)
//   stages 1..16   L9              stage = k-group `tout` of 16 OUT features, quad = in tile Tin,
//   stages 17..32  L8, hidden part   [lane (i, g)][r] = W[16 tout + 4 g + r][16 Tin + i]
//   stage 33       density head    (slot (g 0, r 0) = the density row)
//   stages 34..145 L7, L6, L5, L4 (hidden part), L3, L2, L1: 16 stages each
// (the encodings receive no gradient: rays are not differentiated, so the concatenated columns of L4 / L8
// and all of L0 do not appear)
constexpr int kLegacyBwdStages = 1 + 16 + 16 + 1 + 7 * 16;                      // 146
constexpr int kLegacyBwdBlobFloats = kLegacyBwdStages * kStageFloats;
constexpr int kLegacyBwdOffset = kLegacyHSmallOffset + kLegacySmallFloats;
constexpr int kBwdColorStage = 0, kBwdDensityStage = 33;
// ---- the same chain as f16 pairs (slab format: stage (half, m) = out tiles 8 half .. 8 half + 7 of k block m,
// here "out" = the forward layer's IN features and k = its OUT features): a head is one k block (its 16 padded
// outputs, zeros beyond) = 2 stages, a wide layer 8 k blocks = 16 stages
//   stages 0..1 color head, 2..17 L9, 18..33 L8 (hidden columns), 34..35 density head, 36..147 L7 .. L1
//   element (lane (row, kg), jj) of pair i = 2^kWScaleLog2 * W[k = 32 m + 16 (jj >> 2) + 4 kg + (jj & 3)][16 (8 half + i) + row]
constexpr int kLegacyBwdHStages = 2 + 16 + 16 + 2 + 7 * 16;                    // 148
constexpr int kLegacyBwdHBlobFloats = kLegacyBwdHStages * kStageFloats;
constexpr int kLegacyBwdHOffset = kLegacyBwdOffset + kLegacyBwdBlobFloats;
constexpr int kLegacyPackedFloats = kLegacyBwdHOffset + kLegacyBwdHBlobFloats;
// wide layer of stage s of the f16 backward image (-1: color head, -2: density head); `local` = stage within it
__host__ __device__ inline int bwd_h_layer_of_stage(int s, int& local) {
    if (s < 2) { local = s; return -1; }
    if (s < 34) { local = (s - 2) % 16; return 9 - (s - 2) / 16; }
    if (s < 36) { local = s - 34; return -2; }
    local = (s - 36) % 16;
    return 7 - (s - 36) / 16;
}
// wide layer whose transposed weights stage s of the backward image carries (-1: a head stage)
__host__ __device__ inline int bwd_layer_of_stage(int s, int& tout) {
    if (s == kBwdColorStage || s == kBwdDensityStage) { tout = 0; return -1; }
    const int w = s < kBwdDensityStage ? s - 1 : s - 2;       // wide stages in order: L9, L8, L7 .. L1
    tout = w % 16;
    return 9 - w / 16;
}

// f16 image, consumption order: L0 .. L7, L8 (hidden blocks, then its two encoding blocks), density head,
// L9, color head (the density head runs AFTER L8's loops there: L8's fused loop is what normalises x'_7)
__host__ __device__ inline int h_stage_of_layer(int L) {    // first stage of wide layer L in the f16 image
    int s = 0;
    for (int i = 0; i < L; ++i) s += 2 * wide_blocks(i);
    return s + (L >= 9 ? 1 : 0);
}
constexpr int kHDensityStage = 4 + 3 * 16 + 20 + 3 * 16 + 20;     // 140
constexpr int kHColorStage = kLegacyHStages - 1;
__host__ __device__ inline int stage_of_layer(int L) {      // first stage of wide layer L
    int s = 0;
    for (int i = 0; i < L; ++i) s += wide_stages(i);
    return s + (L >= 8 ? 1 : 0);                             // the density head sits before layer 8
}
constexpr int kDensityStage = 4 + 3 * 16 + 20 + 3 * 16;      // 120
constexpr int kColorStage = kLegacyStages - 1;

__host__ __device__ inline int wide_param(int L) { return L < 8 ? 4 * L : 34 + 4 * (L - 8); }
__host__ __device__ inline int wide_inputs(int L) {
    return L == 0 ? kPosFeatures : (L == 4 ? kHidden + kPosFeatures : (L == 8 ? kHidden + kDirFeatures : kHidden));
}
constexpr int kDensityW = 32, kDensityB = 33, kColorW = 42, kColorB = 43;

// Which encoding features a lane group computes.  The reference's layout (nerf/model.py:233-240) is, per
// coordinate, [sin f_0 .. sin f_{F-1}, cos f_0 .. cos f_{F-1}], coordinate-major: feature = coord * 2F +
// cos * F + k.  Lane group g takes, of every coordinate, the trig function g >> 1 (sine for groups 0, 1,
// cosine for 2, 3) and the frequencies k = H (g & 1) + 0 .. H - 1 (H = F / 2: 5 position, 3 direction
// frequencies), in slots q = H' coord + (k mod H) of its 15 (9) — so that coordinate and frequency offset of
// a slot are compile-time constants and only ONE base frequency and ONE sin / cos choice depend on the lane
// (a slot -> feature map that needs a division per lane makes the compiler keep two dozen lane-dependent
// frequencies and flags alive across every loop of the kernels).
__host__ __device__ inline int encoding_feature_of(int g, int q, int freqs) {     // slot q of lane group g
    const int half = freqs / 2;
    return (q / half) * 2 * freqs + (g >> 1) * freqs + half * (g & 1) + q % half;
}
// ... and back: column of feature f in a padded encoding row (lane group g, slot q = 4 t + r -> 16 t + 4 g + r)
__host__ __device__ inline int encoding_column(int f, int freqs) {
    const int half = freqs / 2;
    const int coord = f / (2 * freqs), within = f % (2 * freqs);
    const int cosine = within / freqs, k = within % freqs;
    const int g = 2 * cosine + k / half, q = half * coord + k % half;
    return 16 * (q / 4) + 4 * g + (q % 4);
}

// ---- saved-for-backward workspace (training forward writes it, backward reads it) --------------------
// padded sample sp = (ray slot * chunks + c) * 16 + j, chunks = ceil(S / 16); "row" tensors [sp][feature].
// What only the BACKWARD writes and reads — dY of every wide layer, dL/d(out) — is not part of it: those rows
// live in the backward's scratch buffer (offsets dy[] / dy5 below are relative to THAT area, `bwd_total` floats),
// which the host caches across steps; the forward's allocation, held from forward to backward, is the first
// `total` floats only (11.1 instead of 21.6 KB per padded sample).
struct LegacyTrainLayout {
    int64_t mp;                 // padded samples = ceil4(n_rays) * chunks * 16
    int64_t pos, dir;           // row  [mp, 64]   encoded position / direction (padded columns)
    int64_t xhat[kWide];        // row  [mp, 256]  a_hat = (relu(y) - mean) / std of wide layer L
    int64_t rstd[kWide];        // [mp]
    int64_t shift[kWide];       // [mp]            a_hat of a closed ReLU gate: a_hat > shift <=> y > 0
    int64_t out;                // tile [mp, 64]   (density, r, g, b) in the compositing kernels' tile format
    int64_t comp;               // [mp, 4]         alpha, T_exclusive, dist, density(+noise)
    int64_t total;              // floats of the workspace
    // backward-owned rows, offsets into the scratch buffer's row area
    int64_t dy[kWide];          // row  [mp, 256]  grad wrt the Linear output y of wide layer L
    int64_t dy5;                // row  [mp, 64]   grad wrt (density, r, g, b) in columns 0..3, rest 0
    int64_t bwd_total;          // floats of that area
};

__host__ __device__ inline LegacyTrainLayout make_legacy_train_layout(int64_t n_rays, int chunks) {
    LegacyTrainLayout t;
    const int64_t rays4 = (n_rays + 3) / 4 * 4;
    t.mp = rays4 * chunks * 16;
    int64_t off = 0;
    t.pos = off; off += t.mp * kEncPad;
    t.dir = off; off += t.mp * kEncPad;
    for (int i = 0; i < kWide; ++i) { t.xhat[i] = off; off += t.mp * kHidden; }
    for (int i = 0; i < kWide; ++i) { t.rstd[i] = off; off += t.mp; }
    for (int i = 0; i < kWide; ++i) { t.shift[i] = off; off += t.mp; }
    t.out = off; off += t.mp * kOutPad;
    t.comp = off; off += t.mp * 4;
    t.total = off;
    off = 0;
    for (int i = 0; i < kWide; ++i) { t.dy[i] = off; off += t.mp * kHidden; }
    t.dy5 = off; off += t.mp * kOutPad;
    t.bwd_total = off;
    return t;
}

// ---- flat gradient vector: the 44 tensors in the order above, PyTorch layouts ------------------------
__host__ __device__ inline int legacy_tensor_elements(int t) {
    if (t == kDensityW) return kHidden;
    if (t == kDensityB) return 1;
    if (t == kColorW) return 3 * kHidden;
    if (t == kColorB) return 3;
    const int w = t < kDensityW ? t : t - 2;                 // index among the wide layers' 40 tensors
    const int L = w / 4, which = w % 4;
    return which == 0 ? kHidden * wide_inputs(L) : kHidden;
}
__host__ __device__ inline int legacy_grad_offset(int tensor) {
    int off = 0;
    for (int i = 0; i < tensor; ++i) off += legacy_tensor_elements(i);
    return off;
}
constexpr int kLegacyGradElements = kHidden * (kPosFeatures + 3 * kHidden + (kHidden + kPosFeatures) + 3 * kHidden +
                                               (kHidden + kDirFeatures) + kHidden) +
                                    kWide * 3 * kHidden + kHidden + 1 + 3 * kHidden + 3;     // 638,468

}  // namespace nerf_legacy
#endif

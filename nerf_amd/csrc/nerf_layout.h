// Shared constants: the packed parameter image streamed by the render kernels.
//
// MFMA used: v_mfma_f32_16x16x4_f32 (exact fp32, D = A[16x4] * B[4x16] + C).
//   A operand: lane l holds A[row = l & 15][k = l >> 4]        -> weights
//   B operand: lane l holds B[k = l >> 4][col = l & 15]        -> activations, col = sample
//   C/D      : lane l, register reg holds D[row = 4 * (l >> 4) + reg][col = l & 15]
// A wave owns 16 samples (the lane's `j = l & 15`) and ALL features of them: lane group
// g = l >> 4 holds features f = 16 * t + 4 * g + r in register 4 * t + r.  With that map the
// accumulator of one layer IS the B operand of the next (k index of MFMA step (t, r) on lane
// group g = feature 16 t + 4 g + r), so activations never leave registers; only weights move.
//
// Weight image: 74 stages of 16 KiB.  A stage is 16 "quads" of 1 KiB; quad = the four A
// operands (r = 0..3) of one (out-tile T, k-group t) pair, laid out [lane][r] so that one
// ds_read_b128 per lane fetches them conflict-free and one global_load_lds_dwordx4 per wave
// moves a whole quad.
//   stages  0.. 5 : layer 0 (96 -> 256), stage = k-group t, quad = out-tile T (16 of them)
//   stages  6..69 : layers 1..4 (256 -> 256), 16 stages each, same shape
//   stages 70..73 : layer 5 (256 -> 64 padded), stage s holds k-groups 4s..4s+3,
//                   quad = (t - 4 s) * 4 + T, T = 0..3
// followed by the "small" image: per hidden layer L = 0..4 bias[256], gamma[256], beta[256],
// each in [g][T][reg] order (feature 16 T + 4 g + reg), then the padded last bias [g][T(4)][reg].
#ifndef NERF_LAYOUT_H
#define NERF_LAYOUT_H

namespace nerf_layout {

constexpr int kHidden = 256;
constexpr int kEncIn = 96;
constexpr int kOutPad = 64;      // the last layer's outputs padded to four 16-row tiles; the number in use
                                 // (1 density + color_outputs + segmentation classes, <= 64) is a launch argument
constexpr int kMinOutputs = 2;   // density + one color channel, no segmentation classes
constexpr int kMaxColors = 12;   // color channels (run-time count): three per lane group of output tile 0

constexpr int kStageBytes = 16384;
constexpr int kStageFloats = kStageBytes / 4;
constexpr int kQuadFloats = 256;
constexpr int kStagesL0 = kEncIn / 16;            // 6
constexpr int kStagesHidden = kHidden / 16;       // 16
constexpr int kStagesL5 = 4;
constexpr int kNumStages = kStagesL0 + 4 * kStagesHidden + kStagesL5;   // 74
constexpr int kBlobFloats = kNumStages * kStageFloats;

constexpr int kSmallPerLayer = 3 * kHidden;       // bias, gamma, beta
constexpr int kSmallFloats = 5 * kSmallPerLayer + kOutPad;   // 3904
// Transposed image for the data-gradient chain (dX = W^T dY), consumed last layer first:
//   stages 0..3   : layer 5, stage = k-group of OUT features tout (64 padded outs)
//   stages 4..67  : layers 4, 3, 2, 1, 16 stages each, stage = tout
// quad = in-tile Tin (16 of them), [lane (i, g)][r] = W[16 tout + 4 g + r][16 Tin + i].
constexpr int kBwdStages = kStagesL5 + 4 * kStagesHidden;               // 68
constexpr int kBwdBlobFloats = kBwdStages * kStageFloats;
constexpr int kBwdBlobOffset = kBlobFloats + kSmallFloats;              // 16-byte aligned
// Split-precision ("f16x3") image of the inference path: the same 74 stages and the same bytes,
// but every weight as an f16 pair (hi, lo) with hi + lo = 2^kWScaleLog2 * w to ~22 bits, for
// v_mfma_f32_16x16x32_f16 (A: lane l holds A[row l & 15][k = 8 (l >> 4) + jj], jj = 0..7).
// K is walked in blocks m of 32 features = two register tiles: MFMA k slot (kg = l >> 4, jj) is
// feature 32 m + 16 (jj >> 2) + 4 kg + (jj & 3), i.e. lane group kg's registers of tiles 2m, 2m+1,
// so the fp32 accumulator layout still chains layer to layer.  A stage is 8 (out tile, k block)
// pairs x {hi slab, lo slab}; slab = 1 KiB [lane][8 halfs]; pair i at slabs 2i (hi), 2i+1 (lo).
//   wide layers (KB = 3 for layer 0, 8 for layers 1..4): stage s of the layer = half * KB + m
//                   (half = s / KB), pair i = out tile 8 * half + i of k block m
//   layer 5:        stage s: pair i = (k block 2 s + (i >> 2), out tile i & 3)
// Activations enter the MFMAs scaled by 2^kXScaleLog2 (folded into gamma/beta and the encoding),
// so accumulators hold 2^12 * (W x + b); the "small" image of this path carries bias * 2^12,
// gamma * 2^4, beta * 2^4 and LayerNorm runs with eps * 2^24 (all exact power-of-two scalings).
constexpr int kWScaleLog2 = 8;
constexpr int kXScaleLog2 = 4;
constexpr int kHBlobOffset = kBwdBlobOffset + kBwdBlobFloats;
constexpr int kHSmallOffset = kHBlobOffset + kBlobFloats;
// Transposed split-precision image for the data gradient (dX = W^T dY on v_mfma_f32_16x16x32_f16):
// the 68 stages of the transposed fp32 image in the slab format above.  A rows = IN features of the
// forward layer (out tiles of the product), k = its OUT features in blocks m of 32:
//   stages 0..3   : layer 5 (64 padded outs = 2 k blocks): stage s = half * 2 + m
//   stages 4..67  : layers 4, 3, 2, 1, 16 stages each:     stage s = half * 8 + m
//   pair i of a stage = in tile 8 * half + i; element (lane (row, kg), jj) of its slabs
//   = 2^kWScaleLog2 * W[32 m + 16 (jj >> 2) + 4 kg + (jj & 3)][16 (8 half + i) + row]
constexpr int kBwdHBlobOffset = kHSmallOffset + kSmallFloats;
constexpr int kImageFloats = kBwdHBlobOffset + kBwdBlobFloats;          // the six images above
// ... followed by four floats of per-launch constants derived from the weights:
//   [0] K0 = 18 * 2^21 * max|gamma_0| * max_f sum_out |W_1[out][f]| * 1.01: with it the split-precision data gradient
//       bounds a sample's |dL/dy_0| from two scalars it holds (nerf_backward.hip: nerf_bwd_data_h_kernel) — the
//       layer-0 weight-gradient job's f16 scale; [1..3] unused
constexpr int kBoundsOffset = kImageFloats;
constexpr int kWideFloats = kImageFloats + 4;

// NARROW networks at their own cost (hidden_size <= 128 or <= 64): a second fp32 image of the inference path,
// behind the six full-width ones, for kernels instantiated at NT = 8 or 4 register tiles per sample instead of 16.
// Same 16 KiB stages of 16 quads, same quad format; what changes is how many k-groups a stage holds.  With NT out
// tiles per k-group a layer's quads are numbered qq = k-group * NT + out tile and cut into stages of 16:
//   layer 0:   KG0 k-groups (6, the 96 padded encoding inputs; 8 for NT = 4 so that the layer ends on a stage
//              boundary — k-groups 6, 7 are zero) x NT out tiles
//   layers 1-4: NT k-groups x NT out tiles  = NT^2 / 16 stages each
//   layer 5:   NT / 4 stages of 4 k-groups x 4 out tiles (quad = (t - 4 s) * 4 + T, as in the full-width image)
// The MFMA loop is the full-width one (a stage = 8 groups of 2 quads = 64 MFMAs); only the (stage, group) ->
// (k-group, out-tile pair) map differs.  bias / gamma / beta come from the full-width small image.
template <int NT>
struct Narrow {
    static_assert(NT == 16 || NT == 8 || NT == 4, "register tiles per sample");
    static constexpr int kTiles = NT;
    static constexpr int kKGroups0 = NT == 4 ? 8 : 6;                       // k-groups of layer 0 (zero-padded)
    static constexpr int kStages0 = kKGroups0 * NT / 16;
    static constexpr int kStagesHid = NT * NT / 16;
    static constexpr int kStages5 = NT / 4;
    static constexpr int kStages = kStages0 + 4 * kStagesHid + kStages5;     // 74 / 21 / 7
    static constexpr int kFloats = kStages * kStageFloats;
};
static_assert(Narrow<16>::kStages == kNumStages, "NT = 16 is the full-width image");
constexpr int kNarrow8Offset = kWideFloats;                                  // 16-byte aligned (kWideFloats % 4 == 0)
constexpr int kNarrow4Offset = kNarrow8Offset + Narrow<8>::kFloats;
// ... and the TRANSPOSED narrow fp32 image for the data gradient of a network that trains at 8 register tiles
// (hidden_size <= 128, fp32 arithmetic): as the full-width transposed image, last layer first, k = OUT features,
// accumulator tiles = IN tiles; quads numbered k-group * 8 + in tile, 16 to a stage:
//   stages 0..1  : layer 5 (4 k-groups of padded outputs x 8 in tiles)
//   stages 2..17 : layers 4, 3, 2, 1 (8 k-groups x 8 in tiles = 4 stages each)
// quad (k-group tout, in tile Tin): [lane (i, g)][r] = W[16 tout + 4 g + r][16 Tin + i]
constexpr int kNarrowBwd8Stages = 2 + 4 * 4;
constexpr int kNarrowBwd8Offset = kNarrow4Offset + Narrow<4>::kFloats;
// ... and the split-precision ("f16x3") forward image at 8 register tiles (inference of a network with hidden_size
// <= 128): the slab format of the full-width f16 image with ONE half — wide layers: stage = k block m (3 for layer 0,
// 4 for layers 1..4), pair i = out tile i; layer 5: stage s, pair i = (k block 2 s + (i >> 2), out tile i & 3), 2 stages
constexpr int kNarrowH8Stages = 3 + 4 * 4 + 2;
constexpr int kNarrowH8Offset = kNarrowBwd8Offset + kNarrowBwd8Stages * kStageFloats;
// ... and its transposed counterpart for the split-precision data gradient at 8 register tiles: the slab format of
// kBwdHBlobOffset with ONE half (pair i = in tile i): layer 5 = 2 stages (k blocks m = 0, 1 of the 64 padded outputs),
// layers 4, 3, 2, 1 = 4 stages each (k block m of the 128 out features);
// element (lane (row, kg), jj) = 2^kWScaleLog2 * W[32 m + 16 (jj >> 2) + 4 kg + (jj & 3)][16 i + row]
constexpr int kNarrowBwdH8Stages = 2 + 4 * 4;
constexpr int kNarrowBwdH8Offset = kNarrowH8Offset + kNarrowH8Stages * kStageFloats;
// ... and the transposed fp32 image at 4 register tiles (hidden_size <= 64 training in fp32 arithmetic: forward and
// data gradient COMPUTE at 4 tiles, the saved rows stay 128 wide, their tiles 4 .. 7 unused): quads numbered
// k-group * 4 + in tile: layer 5 = 4 k-groups of padded outputs x 4 in tiles = 1 stage, layers 4, 3, 2, 1 = 4 k-groups
// x 4 in tiles = 1 stage each
constexpr int kNarrowBwd4Stages = 1 + 4;
constexpr int kNarrowBwd4Offset = kNarrowBwdH8Offset + kNarrowBwdH8Stages * kStageFloats;
constexpr int kPackedFloats = kNarrowBwd4Offset + kNarrowBwd4Stages * kStageFloats;
static_assert(kWideFloats % 4 == 0, "narrow images start 16-byte aligned");
// register tiles a network of `hidden` features needs, rounded up to an instantiated width
__host__ __device__ inline int tiles_for(int hidden) { return hidden <= 64 ? 4 : (hidden <= 128 ? 8 : 16); }

// Network shape at run time (include/nerf_hip.h: hidden / enc_inputs / num_outputs of the argument blocks): the kernels always compute the compiled-in
// 256 / 96 / 64 widths; a narrower network (hidden_size H <= 256, encoding_size with S = enc / 2 <= 16 scales,
// i.e. 6 S <= 96 inputs) runs ZERO-PADDED inside them, which is exact:
//   * padded rows / columns of every weight matrix, padded biases, gamma and beta are zero in the packed images,
//     so a padded feature carries 0 into every product and the padded scales of the encoding meet zero weights;
//   * LayerNorm divides its sums by H, not 256 (the padded pre-activations are exactly 0 and add nothing to the
//     sum or the sum of squares; the mean-shifted second pass of the variance, sum (x - mean) x, gets exactly 0 from them);
//   * a padded feature's normalised value is not zero, but gamma = beta = 0 makes its activation, its ReLU gate
//     and every gradient that reaches a REAL parameter through it exactly zero; what the backward computes for
//     padded parameters is never copied into the flat gradient.
// The cost is that of the full-width network.
struct Shape {
    int hidden;                 // H
    int enc_in;                 // 6 S: inputs of layer 0, [sin: scale-major x coord-minor | shifted: same] (model.py:158-163)
    int n_out;                  // rows of the last Linear = 1 + colors + segmentation classes
    int colors = 3;             // color_outputs (nerf/model.py:471, :541-542, :591-592, :660)
    __host__ __device__ int scales() const { return enc_in / 6; }
    __host__ __device__ int classes() const { return n_out - 1 - colors; }
};
__host__ __device__ inline bool shape_ok(const Shape& s) {
    return s.hidden >= 1 && s.hidden <= kHidden && s.enc_in >= 6 && s.enc_in <= kEncIn && s.enc_in % 6 == 0 &&
           s.colors >= 1 && s.colors <= kMaxColors && s.n_out >= 1 + s.colors && s.n_out <= kOutPad;
}

// ROWS of the last Linear (state_dict order: density | colors | classes, nerf/model.py:591-592) against SLOTS of the
// kernels' padded 64-row output tile (slot n = 16 T + 4 g + r lives in register r of tile T of lane group g).
// Compositing keeps THREE running color sums per lane, registers y, z, w of tile 0 — so color channel c sits at slot
// 4 (c / 3) + 1 + c % 3: lane group g holds channels 3 g .. 3 g + 2, up to 12 of them, at no cost to the kernels
// (the sums were always computed on all four lane groups; only lane group 0's were stored).  Density is slot 0; the
// classes fill the remaining slots in increasing order.  For the reference's 3 channels this is the identity map
// (density 0, colors 1..3, classes 4..), which is what every fixture and profile of rounds 1-5 ran with.
__host__ __device__ inline int color_slot(int c) { return 4 * (c / 3) + 1 + c % 3; }
__host__ __device__ inline bool is_color_slot(int n, int colors) {
    return n < 16 && (n & 3) != 0 && 3 * (n >> 2) + (n & 3) - 1 < colors;
}
__host__ __device__ inline int colors_below(int n, int colors) {      // color slots among 0 .. n - 1
    if (n >= 16) return colors;
    const int g = n >> 2, r = n & 3;
    const int full = 3 * g < colors ? 3 * g : colors;                 // lane groups below g
    const int left = colors - 3 * g > 0 ? colors - 3 * g : 0, mine = r > 1 ? r - 1 : 0;
    return full + (mine < left ? mine : left);
}
// row of the last Linear held by slot n, or -1 for a padding slot
__host__ __device__ inline int row_of_slot(int n, int colors, int n_out) {
    if (n == 0) return 0;
    if (is_color_slot(n, colors)) return 1 + 3 * (n >> 2) + (n & 3) - 1;
    const int row = 1 + colors + (n - 1 - colors_below(n, colors));
    return row < n_out ? row : -1;
}
__host__ __device__ inline int row_of_slot(int n, const Shape& s) { return row_of_slot(n, s.colors, s.n_out); }
// the same for register r of output tile 0 on lane group g (slot 4 g + r), in the few operations a kernel that is
// short of registers can afford: the group holds `left` = clamp(colors - 3 g, 0, 3) color channels in registers
// 1 .. left, its other slots are classes counted from `base` = the row of the group's register 0
__host__ __device__ inline int row_of_tile0(int g, int r, int colors, int n_out) {
    const int c0 = 3 * g, below = c0 < colors ? c0 : colors;
    const int left = colors - below;                       // (> 3 is as good as 3: r - 1 <= 2)
    const int base = colors + 4 * g - below;
    const int row = r == 0 ? (g == 0 ? 0 : base) : (r - 1 < left ? c0 + r : base + r - (left < 3 ? left : 3));
    return row < n_out ? row : -1;
}
__host__ __device__ inline int slot_of_row(int row, const Shape& s) {
    if (row == 0) return 0;
    if (row <= s.colors) return color_slot(row - 1);
    for (int n = 1; n < kOutPad; ++n)
        if (!is_color_slot(n, s.colors) && 1 + s.colors + (n - 1 - colors_below(n, s.colors)) == row) return n;
    return -1;
}
// bit n (n < 16) = slot n of output tile 0 holds a segmentation class (slots >= 16: class iff n < n_out, no color
// lives there): computed by the host per launch, tested per element by the compositing code
__host__ __device__ inline int class_mask_tile0(const Shape& s) {
    int m = 0;
    for (int n = 1; n < 16; ++n) {
        const int row = row_of_slot(n, s);
        if (row > s.colors) m |= 1 << n;
    }
    return m;
}

// flat gradient vector: the 22 tensors in state_dict order, PyTorch layouts
// (304,438 elements for the reference's defaults: hidden 256, 96 inputs, 1 + 3 + 50 outputs)
__host__ __device__ inline int tensor_elements(int tensor, const Shape& s) {
    // tensor index in state_dict order: 4 L + {0 W, 1 b, 2 gamma, 3 beta} for L < 5; 20 W5, 21 b5
    const int L = tensor / 4, which = tensor % 4;
    if (which == 0) return L == 0 ? s.hidden * s.enc_in : (L == 5 ? s.n_out * s.hidden : s.hidden * s.hidden);
    return L == 5 ? s.n_out : s.hidden;
}
__host__ __device__ inline int grad_offset(int tensor, const Shape& s) {
    int off = 0;
    for (int i = 0; i < tensor; ++i) off += tensor_elements(i, s);
    return off;
}
__host__ __device__ inline int grad_elements(const Shape& s) { return grad_offset(22, s); }

// Input-feature permutation of layer 0: lane group g computes, for the Gaussian of its sample,
// the 12 (scale, coord) pairs with scale index 4 g .. 4 g + 3; local slot q = 4 t + r:
//   q < 12 : sin part of pair q        q >= 12 : sin(. + pi/2) part of pair q - 12
// Reference layout (nerf/model.py:158-163): [sin: scale-major x coord-minor | shifted: same].
__host__ __device__ inline int layer0_source_feature(int t, int g, int r) {
    const int q = 4 * t + r;
    const int part = q / 12, p = q % 12;
    const int scale = 4 * g + p / 3, coord = p % 3;
    return part * 48 + 3 * scale + coord;
}
// The same slot for a network with `scales` <= 16 scales (encoding_size = 2 scales): its source feature in the
// reference's [sin: scales x 3 | shifted: scales x 3] order, or -1 for a scale the network does not have.
// `per` = scales per lane group: lane group g evaluates scales per g .. per g + per - 1 in its first 3 per (scale,
// coordinate) pairs.  4 for the full-width kernels (whatever the network: their front end always evaluates 16 scales);
// the NARROW kernels (nerf_layout.h: Narrow<NT>), whose frame time is front-end VALU work, spread the scales a
// network HAS over the four lane groups — scales_per_group — and skip the pairs beyond them.
__host__ __device__ inline int scales_per_group(int scales) { return (scales + 3) / 4; }
// DENSE layer 0 of the kernels at 4 register tiles, for networks of at most 8 encoding scales (scales_per_group <= 2):
// the 12 slots a lane group then fills — 6 sine slots q = 0..5 and 6 shifted ones q = 12..17 of the layout above —
// move together into dense slots d = 0..11 = three k-groups, and layer 0 is ONE stage of four k-groups (the fourth
// zero) where the sparse layout needs the two stages of eight (nerf_layout.h: Narrow<4>; the second is skipped).
// Old slot of dense slot d, or -1 for the padding slots d >= 12:
__host__ __device__ inline bool layer0_dense(int nt, int per) { return nt == 4 && per <= 2; }
__host__ __device__ inline int layer0_dense_source_slot(int d) { return d < 6 ? d : (d < 12 ? d + 6 : -1); }
__host__ __device__ inline int layer0_source_feature(int t, int g, int r, int scales, int per = 4) {
    const int q = 4 * t + r;
    const int part = q / 12, p = q % 12;
    const int local = p / 3, coord = p % 3;
    const int scale = per * g + local;
    return local < per && scale < scales ? part * 3 * scales + 3 * scale + coord : -1;
}
// ... and back: kernel column (16 t + 4 g + r) of source feature f
__host__ __device__ inline int layer0_kernel_column(int f, int scales, int per = 4) {
    const int part = f / (3 * scales), rem = f % (3 * scales);
    const int scale = rem / 3, coord = rem % 3;
    const int g = scale / per, q = part * 12 + (scale % per) * 3 + coord;
    return 16 * (q / 4) + 4 * g + (q % 4);
}

}  // namespace nerf_layout
#endif

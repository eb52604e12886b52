// Hardware probe (gfx950 / MI355X, ROCm 7.2): packed-fp32 VALU results are WRONG in lanes 48-63 while
// the SIMD's other wave executes certain MFMAs.
//
// Found while hunting run-to-run differences of the split-precision training forward: with two
// workgroups per CU (two waves per SIMD), one wave in its VALU front end and its SIMD partner in the
// MFMA loop, single encoded features came out as if an operand had been read as 0 — always lane
// group 3 (lanes 48-63), always the result of one v_pk_mul_f32.  This probe isolates it: the FIRST
// resident workgroup of every CU runs a packed-fp32 instruction in a loop and checks both halves of
// its result; the SECOND resident workgroup (blockIdx >= gridDim / 2 under round-robin dispatch)
// streams one kind of MFMA.  Reported per (packed form, MFMA kind): wrong results per lane quarter.
//
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/pk_vs_mfma_coexec.hip -o /tmp/pk_vs_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

enum { MFMA_NONE, MFMA_16x16x32_F16, MFMA_16x16x16_F16, MFMA_16x16x4_F32, MFMA_32x32x16_F16, MFMA_16x16x32_BF16, VALU_FMA, kNumPartners };
static const char* kPartnerName[] = {"idle partner (exits)", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32",
                                     "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_bf16", "v_fma_f32 stream"};
enum { PK_MUL_PLAIN, PK_MUL_CROSS, PK_MUL_BCAST, PK_FMA_PLAIN, PK_ADD_PLAIN, PK_MUL_SGPR, SCALAR_MUL, PK_MUL_SEL0, PK_MUL_SEL01, PK_FMA_SEL1, PK_ADD_SEL1, PK_MUL_HI_ONLY, kNumForms };
static const char* kFormName[] = {"v_pk_mul_f32 d, a, b", "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 d, a, b op_sel_hi:[1,0]",
                                  "v_pk_fma_f32 d, a, b, c", "v_pk_add_f32 d, a, b", "v_pk_mul_f32 d, a, s[n:n+1] op_sel_hi:[1,0]",
                                  "v_mul_f32 x2 (control)", "v_pk_mul_f32 d, a, b op_sel:[1,0]", "v_pk_mul_f32 d, a, b op_sel:[1,1] op_sel_hi:[0,0]",
                                  "v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]", "v_pk_add_f32 d, a, b op_sel:[0,1]",
                                  "v_pk_mul_f32 d, a, b op_sel_hi:[0,0]"};

template <int kPartner>
__device__ void partner_stream(float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.02f * i);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc16;
    for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
    float x = 0.5f;
    for (int it = 0; it < iters * 4; ++it) {
        for (int r = 0; r < 4; ++r) {
            if (kPartner == MFMA_16x16x32_F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            if (kPartner == MFMA_16x16x16_F16) {
                const h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
            }
            if (kPartner == MFMA_16x16x4_F32) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[1], acc, 0, 0, 0);
            if (kPartner == MFMA_32x32x16_F16) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16, 0, 0, 0);
            if (kPartner == MFMA_16x16x32_BF16) {
                typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
            }
            if (kPartner == VALU_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(x));
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3] + acc16[0] + acc16[15] + x;
}

template <int kForm, int kPartner>
__global__ __launch_bounds__(256) void probe(unsigned long long* counts, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= gridDim.x / 2) {
        if (kPartner != MFMA_NONE) partner_stream<kPartner>(sink, iters);
        return;
    }
    unsigned long long bad[4] = {0, 0, 0, 0}, zero_like = 0, lo_bad = 0, hi_bad = 0;
    float keep = 0.f;
    const f32x2 sc = {3.0f, 7.0f};
    for (int it = 0; it < iters; ++it) {
        const f32x2 a = {1.0f + 0.001f * (float)((it + lane) & 255), 3.0f + 0.002f * (float)(lane & 31)};
        const f32x2 b = {5.0f + 0.004f * (float)(it & 63), 2.0f + 0.003f * (float)((it * 3 + lane) & 127)};
        const f32x2 c = {0.25f, -0.5f};
        f32x2 d, want;
        if (kForm == PK_MUL_PLAIN) {
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x * b.x, a.y * b.y};
        } else if (kForm == PK_MUL_CROSS) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x * b.y, a.y * b.x};
        } else if (kForm == PK_MUL_BCAST) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x * b.x, a.y * b.x};
        } else if (kForm == PK_FMA_PLAIN) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            want = f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
        } else if (kForm == PK_ADD_PLAIN) {
            asm volatile("v_pk_add_f32 %0, %1, %2\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x + b.x, a.y + b.y};
        } else if (kForm == PK_MUL_SGPR) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "s"(sc));
            want = f32x2{a.x * sc.x, a.y * sc.x};
        } else if (kForm == PK_MUL_SEL0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.y * b.x, a.y * b.y};
        } else if (kForm == PK_MUL_SEL01) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.y * b.y, a.x * b.x};
        } else if (kForm == PK_FMA_SEL1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
            want = f32x2{__builtin_fmaf(a.x, b.y, c.x), __builtin_fmaf(a.y, b.y, c.y)};
        } else if (kForm == PK_ADD_SEL1) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x + b.y, a.y + b.y};
        } else if (kForm == PK_MUL_HI_ONLY) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]\n\ts_nop 3" : "=&v"(d) : "v"(a), "v"(b));
            want = f32x2{a.x * b.x, a.x * b.x};
        } else {
            asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5\n\ts_nop 3"
                         : "=&v"(d.x), "=&v"(d.y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
            want = f32x2{a.x * b.x, a.y * b.y};
        }
        const bool wl = __float_as_uint(d.x) != __float_as_uint(want.x), wh = __float_as_uint(d.y) != __float_as_uint(want.y);
        if (wl || wh) {
            ++bad[lane >> 4];
            lo_bad += wl, hi_bad += wh;
            if ((wl && (d.x == 0.f || d.x == c.x)) || (wh && (d.y == 0.f || d.y == c.y))) ++zero_like;
        }
        keep += d.x + d.y;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = keep;
    for (int q = 0; q < 4; ++q)
        if (bad[q]) atomicAdd(counts + q, bad[q]);
    if (lo_bad) atomicAdd(counts + 4, lo_bad);
    if (hi_bad) atomicAdd(counts + 5, hi_bad);
    if (zero_like) atomicAdd(counts + 6, zero_like);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int kForm, int kPartner>
static void run(unsigned long long* cnt, float* sink, int iters, int grid) {
    CK(hipMemset(cnt, 0, 64));
    hipLaunchKernelGGL((probe<kForm, kPartner>), dim3(grid), dim3(256), 0, 0, cnt, sink, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost));
    const unsigned long long per_quarter = (unsigned long long)(grid / 2) * 4 * 16 * iters;
    printf("%-52s | partner %-26s | wrong per lane quarter %llu %llu %llu %llu of %llu each; lo half %llu, hi half %llu; as if an operand were 0: %llu\n",
           kFormName[kForm], kPartnerName[kPartner], h[0], h[1], h[2], h[3], per_quarter, h[4], h[5], h[6]);
    fflush(stdout);
}

template <int kPartner>
static void forms(unsigned long long* cnt, float* sink, int iters, int grid) {
    run<PK_MUL_PLAIN, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_CROSS, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_BCAST, kPartner>(cnt, sink, iters, grid);
    run<PK_FMA_PLAIN, kPartner>(cnt, sink, iters, grid);
    run<PK_ADD_PLAIN, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_SGPR, kPartner>(cnt, sink, iters, grid);
    run<SCALAR_MUL, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_SEL0, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_SEL01, kPartner>(cnt, sink, iters, grid);
    run<PK_FMA_SEL1, kPartner>(cnt, sink, iters, grid);
    run<PK_ADD_SEL1, kPartner>(cnt, sink, iters, grid);
    run<PK_MUL_HI_ONLY, kPartner>(cnt, sink, iters, grid);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int grid = argc > 2 ? atoi(argv[2]) : 512;        // 2 workgroups per CU
    unsigned long long* cnt;
    float* sink;
    CK(hipMalloc(&cnt, 64));
    CK(hipMalloc(&sink, (size_t)grid * 256 * 4));
    forms<MFMA_16x16x32_F16>(cnt, sink, iters, grid);
    forms<MFMA_16x16x32_BF16>(cnt, sink, iters, grid);
    forms<MFMA_16x16x16_F16>(cnt, sink, iters, grid);
    forms<MFMA_16x16x4_F32>(cnt, sink, iters, grid);
    forms<MFMA_32x32x16_F16>(cnt, sink, iters, grid);
    forms<VALU_FMA>(cnt, sink, iters, grid);
    forms<MFMA_NONE>(cnt, sink, iters, grid);
    printf("done\n");
    return 0;
}

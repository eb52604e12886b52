// Issue-cost probe (gfx950): cycles per instruction of the VALU forms the kernels' conversion code is
// made of, alone and in the gaps of an MFMA stream.  One wave per SIMD (one 256-thread workgroup per
// CU), 16 independent register chains per form, s_memtime around 64 x 256 instructions.
//   part 1: each form alone                          -> cycles per instruction
//   part 2: 1 MFMA (32x32x16 f16, 32 cycles) + k fillers, k = 0..8  -> cycles per MFMA gap
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/valu_issue_cost.hip -o /tmp/valu_issue_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// one instruction of form F on chain c (registers r[c], q[c]; constants k0, k1)
#define I_ADD(c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[c]) : "v"(k0));
#define I_FMA(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(k0), "v"(k1));
#define I_MUL(c) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[c]) : "v"(k0));
#define I_MAX(c) asm volatile("v_max_f32 %0, 0, %0" : "+v"(r[c]));
#define I_MOV(c) asm volatile("v_mov_b32 %0, %1" : "=v"(q[c]) : "v"(r[c]));
#define I_AND(c) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(q[c]) : "v"(r[c]));
#define I_SUB(c) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q[c]) : "v"(r[c]), "v"(k0));
#define I_PERM(c) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(q[c]) : "v"(r[c]), "v"(k0), "s"(sel));
#define I_PKRTZ(c) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(q[c]) : "v"(r[c]), "v"(k0));
#define I_MIX(c) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r[c]) : "v"(q[c]));
#define I_MIXHI(c) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[c]) : "v"(q[c]));
#define I_PKBF(c) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(q[c]) : "v"(r[c]), "v"(k0));
#define I_DPPADD(c) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(q[c]) : "v"(r[c]));
#define I_CNDMASK(c) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(q[c]) : "v"(r[c]), "v"(k0) : "vcc");
#define I_PKFMA(c) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[c]) : "v"(kk));
#define I_EXP(c) asm volatile("v_exp_f32 %0, %1" : "=v"(q[c]) : "v"(r[c]));
#define I_CND3(c) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(q[c]) : "v"(r[c]), "v"(k0), "s"(mask64));
#define I_CMPCND(c) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %1, vcc" : "=v"(q[c]) : "v"(r[c]), "v"(k1) : "vcc");
#define I_CMPCND3(c) asm volatile("v_cmp_gt_f32_e64 %3, %1, %2\n\tv_cndmask_b32_e64 %0, 0, %1, %3" : "=v"(q[c]), "+s"(mask64) : "v"(r[c]), "v"(k1));
#define I_MASKAND(c) asm volatile("v_sub_u32 %0, 0, %1\n\tv_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %2" : "=&v"(q[c]) : "v"(r[c]), "v"(k0));
#define I_MULCLAMP(c) asm volatile("v_mul_f32_e64 %0, %1, %2 clamp\n\tv_mul_f32 %0, %0, %3" : "=&v"(q[c]) : "v"(r[c]), "v"(k0), "v"(k1));

// round 4: the candidates for cheaper conversions
#define I_PKMUL(c) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p2[c]) : "v"(kk));
#define I_PKADD(c) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[c]) : "v"(kk));
#define I_MIXLO16(c) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(r[c]) : "v"(q[c]));
#define I_MIXHI16(c) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[c]) : "v"(q[c]), "v"(k0));
// one value PAIR of the weight gradient's dY conversion as it ships (2 add, 2 mul, pkrtz, 2 fma_mix, pkrtz) and with
// packed scale / bias add + f16-result residuals (pk_add, pk_mul, pkrtz, mixlo, mixhi)
#define I_PAIR8(c) I_ADD(c) I_ADD((c + 1) & 15) I_MUL(c) I_MUL((c + 1) & 15) I_PKRTZ(c) I_MIX(c) I_MIXHI((c + 1) & 15) I_PKRTZ((c + 1) & 15)
#define I_PAIR5(c) I_PKADD(c) I_PKMUL(c) I_PKRTZ(c) I_MIXLO16(c) I_MIXHI16(c)
#define I_PAIR6(c) I_PKADD(c) I_PKMUL(c) I_PKRTZ(c) I_MIX(c) I_MIXHI((c + 1) & 15) I_PKRTZ((c + 1) & 15)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int kForm>
__device__ __forceinline__ void sixteen(float (&r)[16], float (&q)[16], f32x2 (&p2)[16], float k0, float k1, f32x2 kk,
                                        uint32_t sel) {
    uint64_t mask64 = 0x5555555555555555ull;
    if (kForm == 0) { REP16(I_ADD) }
    if (kForm == 1) { REP16(I_FMA) }
    if (kForm == 2) { REP16(I_MUL) }
    if (kForm == 3) { REP16(I_MAX) }
    if (kForm == 4) { REP16(I_MOV) }
    if (kForm == 5) { REP16(I_AND) }
    if (kForm == 6) { REP16(I_SUB) }
    if (kForm == 7) { REP16(I_PERM) }
    if (kForm == 8) { REP16(I_PKRTZ) }
    if (kForm == 9) { REP16(I_MIX) }
    if (kForm == 10) { REP16(I_MIXHI) }
    if (kForm == 11) { REP16(I_PKBF) }
    if (kForm == 12) { REP16(I_DPPADD) }
    if (kForm == 13) { REP16(I_CNDMASK) }
    if (kForm == 14) { REP16(I_PKFMA) }
    if (kForm == 15) { REP16(I_EXP) }
    if (kForm == 16) { REP16(I_CND3) }
    if (kForm == 17) { REP16(I_CMPCND) }
    if (kForm == 18) { REP16(I_MASKAND) }
    if (kForm == 19) { REP16(I_MULCLAMP) }
    // all sixteen results stay live to here: distinct destination registers, no WAW chain on one
#define KEEP(c) asm volatile("" ::"v"(q[c]));
    REP16(KEEP)
}

static const char* kNames[20] = {"v_add_f32", "v_fma_f32", "v_mul_f32", "v_max_f32", "v_mov_b32", "v_and_b32 literal",
                                 "v_sub_f32", "v_perm_b32", "v_cvt_pkrtz_f16_f32", "v_fma_mix_f32 lo", "v_fma_mix_f32 hi",
                                 "v_cvt_pk_bf16_f32", "v_add_f32_dpp", "v_cndmask_b32 (vcc)", "v_pk_fma_f32", "v_exp_f32",
                                 "v_cndmask_b32_e64 (sgpr pair)", "v_cmp + v_cndmask (2 instr)", "sub+ashr+and (3 instr)",
                                 "mul clamp + mul (2 instr)"};

template <int kForm>
__global__ __launch_bounds__(256, 1) void alone(float* out, unsigned long long* cycles, int iters) {
    float r[16], q[16];
    f32x2 p2[16];
    for (int c = 0; c < 16; ++c) {
        r[c] = 1.0f + threadIdx.x * 1e-3f + c;
        q[c] = 0.f;
        p2[c] = f32x2{r[c], r[c]};
    }
    const float k0 = 1.0000001f, k1 = 1e-9f;
    const f32x2 kk = {k0, k0};
    const uint32_t sel = 0x07060302u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        sixteen<kForm>(r, q, p2, k0, k1, kk, sel);
        sixteen<kForm>(r, q, p2, k0, k1, kk, sel);
        sixteen<kForm>(r, q, p2, k0, k1, kk, sel);
        sixteen<kForm>(r, q, p2, k0, k1, kk, sel);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 16; ++c) s += r[c] + q[c] + p2[c].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[kForm] = t1 - t0;   // (two-instruction forms: per PAIR/TRIPLE)
}

// 1 MFMA + kFill fillers of form kForm per gap
template <int kForm, int kFill>
__global__ __launch_bounds__(256, 1) void gaps(float* out, unsigned long long* cycles, int iters) {
    float r[16], q[16];
    f32x2 p2[16];
    for (int c = 0; c < 16; ++c) {
        r[c] = 1.0f + threadIdx.x * 1e-3f + c;
        q[c] = 0.f;
        p2[c] = f32x2{r[c], r[c]};
    }
    const float k0 = 1.0000001f, k1 = 1e-9f;
    const f32x2 kk = {k0, k0};
    const uint32_t sel = 0x07060302u;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.001f * (threadIdx.x + i));
        b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i));
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < kFill; ++f) {
                const int c = (g * kFill + f) & 15;
                if (kForm == 0) { I_ADD(c) }
                if (kForm == 8) { I_PKRTZ(c) }
                if (kForm == 9) { I_MIX(c) }
                if (kForm == 7) { I_PERM(c) }
                if (kForm == 20) { I_PKMUL(c) }
                if (kForm == 21) { I_PKADD(c) }
                if (kForm == 22) { I_MIXLO16(c) }
                if (kForm == 23) { I_MIXHI16(c) }
                if (kForm == 24) { I_PAIR8(c) }           // (kFill counts PAIRS for forms 24..26)
                if (kForm == 25) { I_PAIR5(c) }
                if (kForm == 26) { I_PAIR6(c) }
            }
        }
        REP16(KEEP)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 16; ++c) s += r[c] + q[c] + p2[c].x;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int kForm>
static void run_alone(float* out, unsigned long long* cyc) {
    const int iters = 256;
    alone<kForm><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    alone<kForm><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc + kForm, 8, hipMemcpyDeviceToHost);
    printf("%-24s %6.2f shader-clock ticks per instruction\n", kNames[kForm], (double)c / (iters * 64.0));
}

template <int kForm, int kFill>
static void run_gap(float* out, unsigned long long* cyc, const char* name) {
    const int iters = 256;
    gaps<kForm, kFill><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    gaps<kForm, kFill><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("MFMA + %d x %-20s %6.2f ticks per gap\n", kFill, name, (double)c / (iters * 16.0));
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 32 * 8);
    printf("(ticks of __builtin_readcyclecounter = s_memtime; compare forms with each other)\n");
    run_alone<0>(out, cyc); run_alone<1>(out, cyc); run_alone<2>(out, cyc); run_alone<3>(out, cyc);
    run_alone<4>(out, cyc); run_alone<5>(out, cyc); run_alone<6>(out, cyc); run_alone<7>(out, cyc);
    run_alone<8>(out, cyc); run_alone<9>(out, cyc); run_alone<10>(out, cyc); run_alone<11>(out, cyc);
    run_alone<12>(out, cyc); run_alone<13>(out, cyc); run_alone<14>(out, cyc); run_alone<15>(out, cyc);
    run_alone<16>(out, cyc); run_alone<17>(out, cyc); run_alone<18>(out, cyc); run_alone<19>(out, cyc);
    run_gap<0, 0>(out, cyc, "-");
    run_gap<0, 2>(out, cyc, "v_add_f32"); run_gap<0, 4>(out, cyc, "v_add_f32"); run_gap<0, 6>(out, cyc, "v_add_f32");
    run_gap<0, 8>(out, cyc, "v_add_f32");
    run_gap<8, 2>(out, cyc, "v_cvt_pkrtz"); run_gap<8, 4>(out, cyc, "v_cvt_pkrtz"); run_gap<8, 6>(out, cyc, "v_cvt_pkrtz");
    run_gap<9, 2>(out, cyc, "v_fma_mix"); run_gap<9, 4>(out, cyc, "v_fma_mix"); run_gap<9, 6>(out, cyc, "v_fma_mix");
    run_gap<7, 2>(out, cyc, "v_perm_b32"); run_gap<7, 4>(out, cyc, "v_perm_b32"); run_gap<7, 6>(out, cyc, "v_perm_b32");
    run_gap<20, 2>(out, cyc, "v_pk_mul_f32 bcast"); run_gap<20, 4>(out, cyc, "v_pk_mul_f32 bcast"); run_gap<20, 6>(out, cyc, "v_pk_mul_f32 bcast");
    run_gap<21, 2>(out, cyc, "v_pk_add_f32"); run_gap<21, 4>(out, cyc, "v_pk_add_f32"); run_gap<21, 6>(out, cyc, "v_pk_add_f32");
    run_gap<22, 2>(out, cyc, "v_fma_mixlo_f16"); run_gap<22, 4>(out, cyc, "v_fma_mixlo_f16"); run_gap<22, 6>(out, cyc, "v_fma_mixlo_f16");
    run_gap<23, 2>(out, cyc, "v_fma_mixhi_f16"); run_gap<23, 4>(out, cyc, "v_fma_mixhi_f16"); run_gap<23, 6>(out, cyc, "v_fma_mixhi_f16");
    // whole conversion pairs per MFMA gap (the kernel: 0.75 pairs per gap = 6 VALU)
    run_gap<24, 1>(out, cyc, "pair, 8 VALU (ships)"); run_gap<25, 1>(out, cyc, "pair, 5 VALU (pk + mixlo/hi)");
    run_gap<26, 1>(out, cyc, "pair, 6 VALU (pk + mix + pkrtz)");
    return 0;
}

// ubench_valu.hip -- measures the per-SIMD issue cost (cycles per wave64 instruction) of the VALU
// instructions the DDM kernels are made of, on the GPU it runs on.  The numbers feed the
// VALU-bound ceiling quoted in DESIGN.md / bench.py (the path has no HBM or MFMA roofline that binds).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 16384;
constexpr int UNROLL = 16;

// each kernel: 8 independent register chains, UNROLL instructions per loop trip
#define DEF_KERNEL(NAME, DECL, BODY, SINK)                                                      \
    __global__ __launch_bounds__(256) void NAME(float *out, uint64_t *clk)                      \
    {                                                                                           \
        DECL;                                                                                   \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();     \
        for (int i = 0; i < ITERS; ++i) { BODY BODY }                                          \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();     \
        SINK;                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }        \
    }

#define F8 float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; float c = 1.0001f
#define U8 uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; uint32_t c = 0xD2511F53u
#define SINKF out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7
#define SINKU out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define OP8(INS) \
    asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(c));
#define OP8_1(INS) \
    asm volatile(INS " %0, %0" : "+v"(a0)); asm volatile(INS " %0, %0" : "+v"(a1)); \
    asm volatile(INS " %0, %0" : "+v"(a2)); asm volatile(INS " %0, %0" : "+v"(a3)); \
    asm volatile(INS " %0, %0" : "+v"(a4)); asm volatile(INS " %0, %0" : "+v"(a5)); \
    asm volatile(INS " %0, %0" : "+v"(a6)); asm volatile(INS " %0, %0" : "+v"(a7));
#define OP8_FMA \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a7) : "v"(c));
// v_mad_u64_u32 vdst[2], sdst(carry), a, b, c64
#define MAD1(A) { uint64_t w; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(A), "v"(c) : "vcc"); A = (uint32_t)(w >> 32) ^ (uint32_t)w; }
#define MAD1N(A) { uint64_t w; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(A), "v"(c) : "vcc"); asm volatile("" : "=v"(A) : "0"((uint32_t)(w >> 32))); }
#define OP8_MAD MAD1N(a0) MAD1N(a1) MAD1N(a2) MAD1N(a3) MAD1N(a4) MAD1N(a5) MAD1N(a6) MAD1N(a7)

DEF_KERNEL(k_fma, F8, OP8_FMA, SINKF)
DEF_KERNEL(k_add, F8, OP8("v_add_f32"), SINKF)
DEF_KERNEL(k_mul, F8, OP8("v_mul_f32"), SINKF)
DEF_KERNEL(k_xor, U8, OP8("v_xor_b32"), SINKU)
DEF_KERNEL(k_add_u32, U8, OP8("v_add_u32"), SINKU)
DEF_KERNEL(k_mul_lo, U8, OP8("v_mul_lo_u32"), SINKU)
DEF_KERNEL(k_mul_hi, U8, OP8("v_mul_hi_u32"), SINKU)
DEF_KERNEL(k_mul_u24, U8, OP8("v_mul_u32_u24"), SINKU)
DEF_KERNEL(k_mul_hi_u24, U8, OP8("v_mul_hi_u32_u24"), SINKU)
DEF_KERNEL(k_mad64, U8, OP8_MAD, SINKU)
DEF_KERNEL(k_log, F8, OP8_1("v_log_f32"), SINKF)
DEF_KERNEL(k_sqrt, F8, OP8_1("v_sqrt_f32"), SINKF)
DEF_KERNEL(k_sin, F8, OP8_1("v_sin_f32"), SINKF)
DEF_KERNEL(k_cos, F8, OP8_1("v_cos_f32"), SINKF)
DEF_KERNEL(k_rcp, F8, OP8_1("v_rcp_f32"), SINKF)
DEF_KERNEL(k_cvt_f32_u32, F8, OP8_1("v_cvt_f32_u32"), SINKF)
#define OP8_S(INS, SV) \
    asm volatile(INS " %0, %1, %0" : "+v"(a0) : "s"(SV)); asm volatile(INS " %0, %1, %0" : "+v"(a1) : "s"(SV)); \
    asm volatile(INS " %0, %1, %0" : "+v"(a2) : "s"(SV)); asm volatile(INS " %0, %1, %0" : "+v"(a3) : "s"(SV)); \
    asm volatile(INS " %0, %1, %0" : "+v"(a4) : "s"(SV)); asm volatile(INS " %0, %1, %0" : "+v"(a5) : "s"(SV)); \
    asm volatile(INS " %0, %1, %0" : "+v"(a6) : "s"(SV)); asm volatile(INS " %0, %1, %0" : "+v"(a7) : "s"(SV));
#define OP8_3(INS) \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a7) : "v"(c));
#define OP8_ALIGN \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a0) : "s"(sk)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a1) : "s"(sk)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a2) : "s"(sk)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a3) : "s"(sk)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a4) : "s"(sk)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a5) : "s"(sk)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a6) : "s"(sk)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a7) : "s"(sk));
#define OP8_CNDS \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "s"(sm)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "s"(sm)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "s"(sm)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "s"(sm));
#define US8 U8; uint32_t sk = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u + 12345u); unsigned long long sm = __builtin_amdgcn_readfirstlane(blockIdx.x) * 0x9E3779B97F4A7C15ull + 77ull
DEF_KERNEL(k_xor_sgpr, US8, OP8_S("v_xor_b32", sk), SINKU)
DEF_KERNEL(k_add3, U8, OP8_3("v_add3_u32"), SINKU)
DEF_KERNEL(k_alignbit, US8, OP8_ALIGN, SINKU)
DEF_KERNEL(k_cndmask_s, US8, OP8_CNDS, SINKU)
DEF_KERNEL(k_cmp_e64, U8, asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %0\n v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %0\n v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %0\n v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %0" :: "v"(a0), "v"(c) : "s20", "s21", "s22", "s23");, SINKU)
#define P8 double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; double c = 1.0001
#define SINKP out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define OP8_PK(INS) \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
    asm volatile(INS " %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a7) : "v"(c));
#define OP8_PK2(INS) \
    asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(c));
DEF_KERNEL(k_pk_fma, P8, OP8_PK("v_pk_fma_f32"), SINKP)
DEF_KERNEL(k_pk_mul, P8, OP8_PK2("v_pk_mul_f32"), SINKP)
DEF_KERNEL(k_pk_add, P8, OP8_PK2("v_pk_add_f32"), SINKP)
DEF_KERNEL(k_cmp, F8, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0" :: "v"(a0), "v"(c) : "vcc");, SINKF)

// mixed streams: does the transcendental pipe overlap with ordinary VALU issue?  (2 trans + 6 simple per 8)
#define M8 float f0 = threadIdx.x + 2.f, f1 = f0 + 1.f; uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13; uint32_t c = 0xD2511F53u
#define SINKM out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5)
#define OP8_MIX(T, INS) \
    asm volatile(T " %0, %0" : "+v"(f0)); asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(T " %0, %0" : "+v"(f1)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(c));
DEF_KERNEL(k_mix_log_xor, M8, OP8_MIX("v_log_f32", "v_xor_b32"), SINKM)
DEF_KERNEL(k_mix_sin_mullo, M8, OP8_MIX("v_sin_f32", "v_mul_lo_u32"), SINKM)
#define OP8_B3 \
    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a0) : "v"(c)); asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a1) : "v"(c)); \
    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a2) : "v"(c)); asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a3) : "v"(c)); \
    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a4) : "v"(c)); asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a5) : "v"(c)); \
    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a6) : "v"(c)); asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a7) : "v"(c));
DEF_KERNEL(k_bitop3, U8, OP8_B3, SINKU)
// the same three-operand instructions with THREE DIFFERENT source registers (the form the Philox rounds use): naming one
// register twice, as the rows above do, costs the instruction an extra ~1.5 cycles
#define OP8_3D(INS, SUF) \
    asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a0) : "v"(c), "v"(d)); asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a1) : "v"(c), "v"(d)); \
    asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a2) : "v"(c), "v"(d)); asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a3) : "v"(c), "v"(d)); \
    asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a4) : "v"(c), "v"(d)); asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a5) : "v"(c), "v"(d)); \
    asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a6) : "v"(c), "v"(d)); asm volatile(INS " %0, %0, %1, %2" SUF : "+v"(a7) : "v"(c), "v"(d));
#define U8D U8; uint32_t d = 0xCD9E8D57u ^ threadIdx.x
#define F8D F8; float d = 0.999f + 1e-6f * threadIdx.x
DEF_KERNEL(k_bitop3_3r, U8D, OP8_3D("v_bitop3_b32", " bitop3:0x96"), SINKU)
DEF_KERNEL(k_add3_3r, U8D, OP8_3D("v_add3_u32", ""), SINKU)
DEF_KERNEL(k_fma_3r, F8D, OP8_3D("v_fma_f32", ""), SINKF)
// v_fmamk_f32 d, a, K, b (d = a * K + b with a 32-bit literal): the form the uniform's scaling takes in the step loop
#define OP8_FMAMK \
    asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a0) : "v"(c)); asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a1) : "v"(c)); \
    asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a2) : "v"(c)); asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a3) : "v"(c)); \
    asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a4) : "v"(c)); asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a5) : "v"(c)); \
    asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a6) : "v"(c)); asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(a7) : "v"(c));
DEF_KERNEL(k_fmamk, F8, OP8_FMAMK, SINKF)
// v_alignbit_b32 with its high word in a VGPR instead of an SGPR
#define OP8_ALIGNV \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a0) : "v"(c)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a1) : "v"(c)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a2) : "v"(c)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a3) : "v"(c)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a4) : "v"(c)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a5) : "v"(c)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a6) : "v"(c)); asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(a7) : "v"(c));
DEF_KERNEL(k_alignbit_v, U8, OP8_ALIGNV, SINKU)
// ... and with the shift count in a VGPR too (three different source registers)
#define OP8_ALIGNVV \
    asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a0) : "v"(c), "v"(d)); asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a1) : "v"(c), "v"(d)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a2) : "v"(c), "v"(d)); asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a3) : "v"(c), "v"(d)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a4) : "v"(c), "v"(d)); asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a5) : "v"(c), "v"(d)); \
    asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a6) : "v"(c), "v"(d)); asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a7) : "v"(c), "v"(d));
#define U8S U8; uint32_t d = 9u + (threadIdx.x & 0u)
DEF_KERNEL(k_alignbit_vv, U8S, OP8_ALIGNVV, SINKU)
// v_cmpx (writes EXEC as well): an always-true compare, so the lanes stay on
DEF_KERNEL(k_cmpx, F8, asm volatile("v_cmpx_le_f32 vcc, %0, %0\n v_cmpx_le_f32 vcc, %1, %1\n v_cmpx_le_f32 vcc, %0, %0\n v_cmpx_le_f32 vcc, %1, %1\n v_cmpx_le_f32 vcc, %0, %0\n v_cmpx_le_f32 vcc, %1, %1\n v_cmpx_le_f32 vcc, %0, %0\n v_cmpx_le_f32 vcc, %1, %1" :: "v"(a0), "v"(c) : "vcc", "exec");, SINKF)
// v_mad_u64_u32 with the multiplier in an SGPR, as the Philox rounds have it
#define MADS(A) { uint64_t w; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(A), "s"(sk) : "vcc"); asm volatile("" : "=v"(A) : "0"((uint32_t)(w >> 32))); }
#define OP8_MADS MADS(a0) MADS(a1) MADS(a2) MADS(a3) MADS(a4) MADS(a5) MADS(a6) MADS(a7)
DEF_KERNEL(k_mad64_s, US8, OP8_MADS, SINKU)
DEF_KERNEL(k_fmac, F8, OP8("v_fmac_f32"), SINKF)
DEF_KERNEL(k_add_co, U8, asm volatile("v_add_co_u32_e64 %0, s[20:21], 1, %0\n v_add_co_u32_e64 %1, s[22:23], 1, %1\n v_add_co_u32_e64 %2, s[20:21], 1, %2\n v_add_co_u32_e64 %3, s[22:23], 1, %3\n v_add_co_u32_e64 %4, s[20:21], 1, %4\n v_add_co_u32_e64 %5, s[22:23], 1, %5\n v_add_co_u32_e64 %6, s[20:21], 1, %6\n v_add_co_u32_e64 %7, s[22:23], 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23");, SINKU)

// float64 rows (NDDM_STATE_F64: the reference's double recurrence): add / mul / fma on 64-bit register pairs, the f32 -> f64
// conversion, a double compare
#define D8 double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; double c = 1.0000001; double d = 0.9999999
#define SINKD out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
DEF_KERNEL(k_add_f64, D8, OP8("v_add_f64"), SINKD)
DEF_KERNEL(k_mul_f64, D8, OP8("v_mul_f64"), SINKD)
DEF_KERNEL(k_fma_f64, D8, OP8_3D("v_fma_f64", ""), SINKD)
#define OP8_CVT64 \
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a0) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a1) : "v"(f1)); \
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a2) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a3) : "v"(f1)); \
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a4) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a5) : "v"(f1)); \
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a6) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a7) : "v"(f1));
#define D8F D8; float f0 = threadIdx.x + 0.5f, f1 = f0 + 1.0f
DEF_KERNEL(k_cvt_f64_f32, D8F, OP8_CVT64, SINKD)
DEF_KERNEL(k_cmp_f64, D8, asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0" :: "v"(a0), "v"(c) : "vcc");, SINKD)

typedef void (*kern_t)(float *, uint64_t *);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s  CUs=%d  clockRate=%d kHz\n", prop.name, cus, prop.clockRate);
    float *out; uint64_t *clk;
    const int wpc_list[] = {4, 8, 16, 32};   // waves per CU (1, 2, 4, 8 per SIMD)
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(float)));
    CHECK(hipMalloc(&clk, 16));
    Entry es[] = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_xor_b32", k_xor},
                  {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
                  {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u24}, {"v_mad_u64_u32", k_mad64},
                  {"v_log_f32", k_log}, {"v_sqrt_f32", k_sqrt}, {"v_sin_f32", k_sin}, {"v_cos_f32", k_cos},
                  {"v_rcp_f32", k_rcp}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cmp_lt_f32", k_cmp}, {"v_xor_b32 (sgpr)", k_xor_sgpr},
                  {"v_add3_u32", k_add3}, {"v_alignbit_b32", k_alignbit}, {"v_cndmask_e64 (s)", k_cndmask_s},
                  {"v_cmp_lt_u32_e64", k_cmp_e64}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul},
                  {"v_pk_add_f32", k_pk_add}, {"v_bitop3_b32 (xor3)", k_bitop3}, {"v_fmac_f32", k_fmac}, {"v_add_co_u32_e64", k_add_co},
                  {"v_bitop3_b32 (3 regs)", k_bitop3_3r}, {"v_add3_u32 (3 regs)", k_add3_3r}, {"v_fma_f32 (3 regs)", k_fma_3r},
                  {"v_fmamk_f32", k_fmamk}, {"v_mad_u64_u32 (sgpr)", k_mad64_s}, {"v_cmpx_le_f32", k_cmpx},
                  {"v_alignbit_b32 (vgpr, imm)", k_alignbit_v}, {"v_alignbit_b32 (3 vgprs)", k_alignbit_vv},
                  {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64 (3 regs)", k_fma_f64},
                  {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cmp_lt_f64", k_cmp_f64},
                  {"2 log + 6 xor", k_mix_log_xor},
                  {"2 sin + 6 mul_lo", k_mix_sin_mullo}};
    printf("%-28s", "instr \\ waves/SIMD");
    for (int w : wpc_list) printf("  %6d", w / 4);
    printf("   (SIMD cycles per wave64 instruction = wall time x in-kernel clock / instructions per SIMD)\n");
    for (auto &e : es) {
        printf("%-28s", e.name);
        for (int wpc : wpc_list) {
            const int blocks = cus * wpc / 4;    // 256-thread blocks = 4 waves
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            // wall-clock view: every SIMD hosts wpc/4 waves, each issuing ITERS*UNROLL instructions;
            // clock = the in-kernel shader clock of this very kernel (s_memtime / s_memrealtime * 100 MHz)
            const double ghz = (double)h[0] / (double)h[1] * 0.1;
            const double instr_per_simd = (double)(wpc / 4) * ITERS * UNROLL;
            const double per_simd = (ms * 1e-3) * ghz * 1e9 / instr_per_simd;
            printf("  %6.2f", per_simd);
            if (wpc == 32) printf("   clock %.3f GHz, wall %.3f ms", ghz, ms);
        }
        printf("\n");
    }
    return 0;
}

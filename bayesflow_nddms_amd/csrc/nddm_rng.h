// nddm_rng.h -- device-side random stream of the DDM simulators (gfx950).
//
// Philox4x32-10 (Salmon et al., SC'11), counter-based: the stream of a trial is a pure
// function of (seed, set, trial, draw), so results do not depend on launch geometry.
// Two Gaussian transforms on top of it:
//   exact : Box-Muller from IEEE add / mul / fma / sqrt only -- reproducible bit for bit on
//           a CPU; this is what tests compare element-wise
//   fast  : Box-Muller on the CDNA transcendental unit (v_log_f32, v_sqrt_f32, v_sin_f32,
//           v_cos_f32; sin/cos take their argument in turns, so no 2*pi range reduction)
// Of a pair's two words the first is the radius uniform, the second the angle: its low 23 bits, as turns.
// This translation unit is compiled with -ffp-contract=off: every fma below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nddm {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;   // wave-uniform: stays on the scalar unit
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// a ^ b ^ c in one instruction: gfx950's v_bitop3_b32 with truth table 0x96 (3.8 issue cycles against 2 x 2.3 for two
// v_xor_b32; profiles/r1_ubench_valu.txt).  All three operands must be VGPRs here.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// a ? b : c bitwise, in one instruction: v_bitop3_b32 with truth table 0xCA.  With three different source registers it
// issues at full rate (2.3 cycles; profiles/r2_ubench_valu.txt), where v_alignbit_b32 / v_and_or_b32 take 4.2.
__device__ __forceinline__ uint32_t mux3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The path stream's Philox block, specialised.  Counter layout of every per-trial stream:
//     (c0, c1, c2, c3) = (set_lo, trial, set_hi28 | stream << 28, draw)
// i.e. the words that are constant within a set sit where round 0 MULTIPLIES (c0, c2) and the one that changes from
// block to block (draw = index of the block's first step) where round 0 only XORs.  Following the constants through
// the rounds, round 0 needs no multiply at all per block, rounds 1 and 2 need one each, and six of the round keys
// fold into per-set / per-trial constants: 16 v_mad_u64_u32 + 13 v_bitop3 + 5 v_xor per block instead of 20 + 20.
//   per set   (PathSet, computed when a tile opens):  P0 = M0*set_lo, P1 = M1*c2
//   per trial (PathCtr, computed at hand-out, 2 multiplies): Q0 = M0*c0[1], S1 = M1*c2[2]      ([r] = after round r-1)
//   per block: see philox4x32_10_path
// The remaining round keys come from LDS: on gfx950 an SGPR operand costs VALU issue time, the LDS pipe is otherwise
// idle in the step loop, so it delivers the (wave-uniform) keys as broadcast reads, prefetched one round ahead.
// LDS bytes [0, 80) hold key pair r at 8r; kbase is a VGPR holding LDS byte address 0.
constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u, PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

struct PathSet {
    uint32_t cA, cB, hP1k, X1;
    __device__ __forceinline__ void init(uint32_t set_lo, uint32_t c2, uint32_t k0, uint32_t k1)
    {
        const uint64_t P0 = (uint64_t)PHILOX_M0 * set_lo, P1 = (uint64_t)PHILOX_M1 * c2;
        cA = (uint32_t)(P0 >> 32) ^ k1;                       // round 0: c2[1] = cA ^ draw
        cB = (uint32_t)P1 ^ (k0 + PHILOX_W0);                 // round 1: c0[2] = hi(Q1) ^ cB
        hP1k = (uint32_t)(P1 >> 32) ^ k0;                     // round 0: c0[1] = hP1k ^ trial
        X1 = (uint32_t)P0 ^ (k1 + PHILOX_W1);                 // round 1: c2[2] = hi(Q0) ^ X1
    }
};

struct PathCtr {
    uint32_t cA, cB, cC, cD, cE;
    // kC = k0 + 2 W0, kD = k1 + 2 W1, kE = k0 + 3 W0: the round keys that fold into the per-trial constants (the kernel
    // keeps them in its LDS header next to the key table: a hand-out must not cost a scalar-memory round trip)
    __device__ __forceinline__ void init(uint32_t sA, uint32_t sB, uint32_t hP1k, uint32_t X1, uint32_t trial,
                                         uint32_t kC, uint32_t kD, uint32_t kE)
    {
        cA = sA; cB = sB;
        const uint64_t Q0 = (uint64_t)PHILOX_M0 * (hP1k ^ trial);                  // round 1, constant half
        const uint64_t S1 = (uint64_t)PHILOX_M1 * ((uint32_t)(Q0 >> 32) ^ X1);     // round 2, constant half
        cC = (uint32_t)(S1 >> 32) ^ kC;                       // round 2: c0[3] = lo(Q1) ^ cC
        cD = (uint32_t)Q0 ^ kD;                               // round 2: c2[3] = hi(S0) ^ cD
        cE = (uint32_t)S1 ^ kE;                               // round 3: c0[4] = hi(T1) ^ cE
    }
};

__device__ __forceinline__ u32x4 philox4x32_10_path(uint32_t draw, const PathCtr &pc, uint32_t kbase)
{
    uint32_t ka, kb;
    uint64_t kn;
    asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(kn) : "v"(kbase));          // keys of round 3 (only k1 + 3 W1 is used)
    const uint64_t Q1 = (uint64_t)PHILOX_M1 * (pc.cA ^ draw);                        // rounds 0 + 1
    const uint64_t S0 = (uint64_t)PHILOX_M0 * ((uint32_t)(Q1 >> 32) ^ pc.cB);        // round 2
    const uint64_t T0 = (uint64_t)PHILOX_M0 * ((uint32_t)Q1 ^ pc.cC);                // round 3
    const uint64_t T1 = (uint64_t)PHILOX_M1 * ((uint32_t)(S0 >> 32) ^ pc.cD);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
    kb = (uint32_t)(kn >> 32);
    uint32_t c0 = (uint32_t)(T1 >> 32) ^ pc.cE, c1 = (uint32_t)T1, c2 = xor3((uint32_t)(T0 >> 32), (uint32_t)S0, kb), c3 = (uint32_t)T0;
    asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(kn) : "v"(kbase));          // keys of round 4
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
    ka = (uint32_t)kn; kb = (uint32_t)(kn >> 32);
#define NDDM_ROUND(NEXT_OFF, LAST)                                                                        \
    {                                                                                                     \
        if (!(LAST)) asm volatile("ds_read_b64 %0, %1 offset:" #NEXT_OFF : "=v"(kn) : "v"(kbase));        \
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;                                                     \
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;                                                     \
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, ka);                                           \
        const uint32_t n1 = (uint32_t)p1;                                                                 \
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, kb);                                           \
        const uint32_t n3 = (uint32_t)p0;                                                                 \
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;                                                               \
        if (!(LAST)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn)); ka = (uint32_t)kn; kb = (uint32_t)(kn >> 32); } \
    }
    NDDM_ROUND(40, false) NDDM_ROUND(48, false) NDDM_ROUND(56, false) NDDM_ROUND(64, false) NDDM_ROUND(72, false)
    NDDM_ROUND(0, true)
#undef NDDM_ROUND
    return {c0, c1, c2, c3};
}

// The same block with the round keys of rounds 3..9 held in VGPRs for the whole kernel (13 registers) instead of read from
// LDS every block: identical VALU work -- the xor3 operands are VGPRs either way -- but no LDS instruction and no wait in
// the step loop.  For the kernels that have the registers to spare (<= 64 VGPRs keeps 8 waves per SIMD); it matters most
// when a wave has the SIMD to itself (small launches), where every LDS round trip is exposed latency.
struct PathKeys {
    uint32_t b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9;
    __device__ __forceinline__ void init(uint32_t k0, uint32_t k1)
    {
        auto v = [](uint32_t s) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s)); return r; };   // opaque: stays a VGPR
        b3 = v(k1 + 3u * PHILOX_W1);
        a4 = v(k0 + 4u * PHILOX_W0); b4 = v(k1 + 4u * PHILOX_W1); a5 = v(k0 + 5u * PHILOX_W0); b5 = v(k1 + 5u * PHILOX_W1);
        a6 = v(k0 + 6u * PHILOX_W0); b6 = v(k1 + 6u * PHILOX_W1); a7 = v(k0 + 7u * PHILOX_W0); b7 = v(k1 + 7u * PHILOX_W1);
        a8 = v(k0 + 8u * PHILOX_W0); b8 = v(k1 + 8u * PHILOX_W1); a9 = v(k0 + 9u * PHILOX_W0); b9 = v(k1 + 9u * PHILOX_W1);
    }
};

__device__ __forceinline__ u32x4 philox4x32_10_path(uint32_t draw, const PathCtr &pc, const PathKeys &K)
{
    const uint64_t Q1 = (uint64_t)PHILOX_M1 * (pc.cA ^ draw);                        // rounds 0 + 1
    const uint64_t S0 = (uint64_t)PHILOX_M0 * ((uint32_t)(Q1 >> 32) ^ pc.cB);        // round 2
    const uint64_t T0 = (uint64_t)PHILOX_M0 * ((uint32_t)Q1 ^ pc.cC);                // round 3
    const uint64_t T1 = (uint64_t)PHILOX_M1 * ((uint32_t)(S0 >> 32) ^ pc.cD);
    uint32_t c0 = (uint32_t)(T1 >> 32) ^ pc.cE, c1 = (uint32_t)T1, c2 = xor3((uint32_t)(T0 >> 32), (uint32_t)S0, K.b3), c3 = (uint32_t)T0;
#define NDDM_ROUND(KA, KB)                                                                                \
    {                                                                                                     \
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;                                                     \
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;                                                     \
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, KA);                                           \
        const uint32_t n1 = (uint32_t)p1;                                                                 \
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, KB);                                           \
        const uint32_t n3 = (uint32_t)p0;                                                                 \
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;                                                               \
    }
    NDDM_ROUND(K.a4, K.b4) NDDM_ROUND(K.a5, K.b5) NDDM_ROUND(K.a6, K.b6) NDDM_ROUND(K.a7, K.b7) NDDM_ROUND(K.a8, K.b8)
    NDDM_ROUND(K.a9, K.b9)
#undef NDDM_ROUND
    return {c0, c1, c2, c3};
}

// General counter, all ten key pairs from LDS (bridge-correction uniforms)
__device__ __forceinline__ u32x4 philox4x32_10_lds(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t kbase)
{
    uint32_t ka, kb;
    uint64_t kn;
    asm volatile("ds_read_b64 %0, %1" : "=v"(kn) : "v"(kbase));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn));
    ka = (uint32_t)kn; kb = (uint32_t)(kn >> 32);
#define NDDM_ROUND(NEXT_OFF, LAST)                                                                        \
    {                                                                                                     \
        if (!(LAST)) asm volatile("ds_read_b64 %0, %1 offset:" #NEXT_OFF : "=v"(kn) : "v"(kbase));        \
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;                                                   \
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;                                                   \
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, ka);                                           \
        const uint32_t n1 = (uint32_t)p1;                                                                 \
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, kb);                                           \
        const uint32_t n3 = (uint32_t)p0;                                                                 \
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;                                                               \
        if (!(LAST)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kn)); ka = (uint32_t)kn; kb = (uint32_t)(kn >> 32); } \
    }
    NDDM_ROUND(8, false) NDDM_ROUND(16, false) NDDM_ROUND(24, false) NDDM_ROUND(32, false) NDDM_ROUND(40, false)
    NDDM_ROUND(48, false) NDDM_ROUND(56, false) NDDM_ROUND(64, false) NDDM_ROUND(72, false) NDDM_ROUND(0, true)
#undef NDDM_ROUND
    return {c0, c1, c2, c3};
}

// General counter, all ten key pairs held in VGPRs for the whole kernel (20 registers): the three-input XORs of the LDS form without
// its ten LDS round trips per block -- for a kernel whose residency is bounded by something else anyway (the exact sampler: SGPRs)
struct AllKeys {
    uint32_t a[10], b[10];
    __device__ __forceinline__ void init(uint32_t k0, uint32_t k1)
    {
        auto v = [](uint32_t s) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s)); return r; };   // opaque: stays a VGPR
#pragma unroll
        for (int r = 0; r < 10; ++r) { a[r] = v(k0 + (uint32_t)r * PHILOX_W0); b[r] = v(k1 + (uint32_t)r * PHILOX_W1); }
    }
};
__device__ __forceinline__ u32x4 philox4x32_10_vkeys(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const AllKeys &K)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, K.a[r]);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, K.b[r]);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    return {c0, c1, c2, c3};
}

// ---------------------------------------------------------------- exact transform
// ln(u), u in [2^-33, 1]; Cephes logf polynomial, every rounding spelled out.
__device__ __forceinline__ float exact_logf(float u)
{
    const uint32_t ix = __float_as_uint(u);
    int e = (int)(ix >> 23) - 126;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f000000u);   // [0.5, 1)
    if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, m, -1.1514610310e-1f);
    y = __builtin_fmaf(y, m, 1.1676998740e-1f);
    y = __builtin_fmaf(y, m, -1.2420140846e-1f);
    y = __builtin_fmaf(y, m, 1.4249322787e-1f);
    y = __builtin_fmaf(y, m, -1.6668057665e-1f);
    y = __builtin_fmaf(y, m, 2.0000714765e-1f);
    y = __builtin_fmaf(y, m, -2.4999993993e-1f);
    y = __builtin_fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

// correctly rounded sqrt for s >= 0: v_sqrt_f32 (<= 1 ulp) + one fma residual fix-up -- the scheme LLVM uses for
// its IEEE f32 sqrt lowering.  +-0 comes back unchanged (NaN residuals compare false), as from an IEEE sqrt.
__device__ __forceinline__ float exact_sqrtf(float s)
{
    float r = __builtin_amdgcn_sqrtf(s);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u);
    const float r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, s);
    const float e_up = __builtin_fmaf(-r_up, r, s);
    r = (e_dn <= 0.0f) ? r_dn : r;
    r = (e_up > 0.0f) ? r_up : r;
    return r;
}

// sin, cos of 2*pi*x/2^32: quadrant from the integer, Cephes sinf/cosf kernels on [-pi/4, pi/4)
__device__ __forceinline__ void exact_sincos_turn(uint32_t x, float &sn, float &cs)
{
    const uint32_t y = x + 0x20000000u;
    const uint32_t q = y >> 30;
    const int32_t rem = (int32_t)(y & 0x3fffffffu) - 0x20000000;
    const float t = (float)rem * 1.4629180792671596e-9f;   // 2*pi / 2^32
    const float t2 = t * t;
    float s = -1.9515295891e-4f;
    s = __builtin_fmaf(s, t2, 8.3321608736e-3f);
    s = __builtin_fmaf(s, t2, -1.6666654611e-1f);
    s = __builtin_fmaf(s * t2, t, t);
    float c = 2.443315711809948e-5f;
    c = __builtin_fmaf(c, t2, -1.388731625493765e-3f);
    c = __builtin_fmaf(c, t2, 4.166664568298827e-2f);
    c = __builtin_fmaf(c * t2, t2, __builtin_fmaf(-0.5f, t2, 1.0f));
    const float a = (q & 1u) ? c : s;     // |sin|
    const float b = (q & 1u) ? s : c;     // |cos|
    sn = (q & 2u) ? -a : a;
    cs = ((q + 1u) & 2u) ? -b : b;
}

// exp(x) for x <= 0, Cephes expf with every rounding spelled out; 0 below -87
__device__ __forceinline__ float exact_expf_neg(float x)
{
    if (x < -87.0f) return 0.0f;
    const float z = __builtin_floorf(__builtin_fmaf(1.44269504088896341f, x, 0.5f));
    const int n = (int)z;
    x = __builtin_fmaf(z, -0.693359375f, x);
    x = __builtin_fmaf(z, 2.12194440e-4f, x);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, x, 1.3981999507e-3f);
    p = __builtin_fmaf(p, x, 8.3334519073e-3f);
    p = __builtin_fmaf(p, x, 4.1665795894e-2f);
    p = __builtin_fmaf(p, x, 1.6666665459e-1f);
    p = __builtin_fmaf(p, x, 5.0000001201e-1f);
    p = __builtin_fmaf(p * x, x, x) + 1.0f;
    return p * __uint_as_float((uint32_t)(n + 127) << 23);
}

// The Box-Muller radius comes out in NOISE UNITS so that no per-pair multiply by sigma is needed: the simulators carry
// their state divided by  S = noise_unit<FAST>(sigma),
//   exact: r = sqrt(-2 ln u),  S = sigma                (r * cos is a standard normal)
//   fast : r = sqrt(-log2 u),  S = sigma * sqrt(2 ln 2) (v_log_f32 is log2; the constant moves into S)
// and an Euler-Maruyama step is  w = fma(r, cos|sin, w) + mu*dt/S.
template <bool FAST>
__device__ __forceinline__ float noise_unit(float sigma)
{
    return FAST ? sigma * 1.1774100225154747f : sigma;
}

// (r, cos, sin) of one Box-Muller pair from two u32, radius in noise units.  HOT: the step loop's form of the fast transform,
// which holds two constants in VGPRs for the whole kernel; the once-per-trial uses (auxiliary normals) build the same bits
// with a shift + v_alignbit_b32 and no registers.
template <bool FAST, bool HOT = false>
__device__ __forceinline__ void polar_pair(uint32_t xa, uint32_t xb, float &r, float &cs, float &sn)
{
    const float u = __builtin_fmaf((float)xa, 2.3283064365386963e-10f, 1.1641532182693481e-10f);   // (0, 1]
    if constexpr (FAST) {
        r = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));
        // v_sin/v_cos take turns and reduce the integer part themselves: feed [1, 2) whose mantissa is the angle word's
        // low 23 bits, spliced under the exponent by ONE full-rate instruction
        const float ang = HOT ? __uint_as_float(mux3(0x007fffffu, xb, 0x3f800000u))    // 0x3f800000 | (xb & 0x7fffff)
                              : __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, xb << 9, 9u));
        cs = __builtin_amdgcn_cosf(ang);
        sn = __builtin_amdgcn_sinf(ang);
    } else {
        r = exact_sqrtf(-2.0f * exact_logf(u));
        exact_sincos_turn(xb << 9, sn, cs);      // the same 23 bits as turns
    }
}

// PACKED layout: one u32 per Box-Muller pair, so a Philox block yields FOUR pairs = 8 normals and the generator's cost
// per Euler-Maruyama step halves.  The word's high 16 bits are the radius uniform, its low 16 bits the angle:
//   u   = 2 - [1,2)-float built from x >> 9   (the 16 random bits on top, dithered by the angle's top 7 bits below
//         them: u is uniform on a 2^-16 grid whose offset inside a cell depends on the angle), u in [2^-23, 1]
//   ang = (x & 0xffff) / 2^16 turns
// Against the 32 + 23 bit layout above: P(r > r0) is reproduced on a 2^-16 grid, |z| <= 5.65 (exact: 5.65; the mass
// beyond 4.7 sigma, 2^-16 per pair, follows the dither rather than fresh bits).  Opt-in (NDDM_GAUSS_PACKED); KS of the
// first-passage distributions against the reference is at the two-sample noise floor like the default layout's.
template <bool FAST>
__device__ __forceinline__ void polar_pair_packed(uint32_t x, float &r, float &cs, float &sn)
{
    const float u = 2.0f - __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, x, 9u));      // (0x3f800000 | x >> 9) in [1, 2)
    if constexpr (FAST) {
        r = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(u));
        const float ang = __uint_as_float(mux3(0x007fff80u, x << 7, 0x3f800000u));         // [1, 2): the 16 angle bits on top of the mantissa
        cs = __builtin_amdgcn_cosf(ang);
        sn = __builtin_amdgcn_sinf(ang);
    } else {
        r = exact_sqrtf(-2.0f * exact_logf(u));
        exact_sincos_turn(x << 16, sn, cs);
    }
}

// 4 standard normals of one Philox block
template <bool FAST>
__device__ __forceinline__ void normals4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1, float (&z)[4])
{
    const u32x4 x = philox4x32_10(c0, c1, c2, c3, k0, k1);
    const float sc = noise_unit<FAST>(1.0f);
    float r, cs, sn;
    polar_pair<FAST>(x.x, x.y, r, cs, sn);
    r *= sc;
    z[0] = r * cs; z[1] = r * sn;
    polar_pair<FAST>(x.z, x.w, r, cs, sn);
    r *= sc;
    z[2] = r * cs; z[3] = r * sn;
}

// uniform in (0,1) from one u32 (prior sampler)
__device__ __forceinline__ float uniform01(uint32_t x)
{
    return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

}  // namespace nddm

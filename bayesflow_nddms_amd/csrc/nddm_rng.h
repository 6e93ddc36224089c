// nddm_rng.h -- device-side random stream of the DDM simulators (gfx950).
//
// Philox4x32-10 (Salmon et al., SC'11), counter-based: the stream of a trial is a pure
// function of (seed, set, trial, draw), so results do not depend on launch geometry.
// Two Gaussian transforms on top of it:
//   exact : Box-Muller from IEEE add / mul / fma / sqrt only -- reproducible bit for bit on
//           a CPU; this is what tests compare element-wise
//   fast  : Box-Muller on the CDNA transcendental unit (v_log_f32, v_sqrt_f32, v_sin_f32,
//           v_cos_f32; sin/cos take their argument in turns, so no 2*pi range reduction)
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

// correctly rounded sqrt for s in [0, 46]: v_sqrt_f32 (<= 1 ulp) + one fma residual fix-up
__device__ __forceinline__ float exact_sqrtf(float s)
{
    float r = __builtin_amdgcn_sqrtf(s);
    // candidates one ulp below / above; pick by the sign of the fma residuals (the scheme LLVM
    // uses for its IEEE f32 sqrt lowering).  s == 0 -> r == 0, residuals 0, r kept.
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u);
    const float r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, s);
    const float e_up = __builtin_fmaf(-r_up, r, s);
    r = (e_dn <= 0.0f) ? r_dn : r;
    r = (e_up > 0.0f) ? r_up : r;
    return (s == 0.0f) ? 0.0f : r;
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

template <bool FAST>
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float &z0, float &z1)
{
    const float u = __builtin_fmaf((float)xa, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    if constexpr (FAST) {
        // -2 ln u = -2 ln2 * log2 u
        const float r = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);
        const float ang = (float)xb * 2.3283064365386963e-10f;   // turns
        z0 = r * __builtin_amdgcn_cosf(ang);
        z1 = r * __builtin_amdgcn_sinf(ang);
    } else {
        const float r = exact_sqrtf(-2.0f * exact_logf(u));
        float sn, cs;
        exact_sincos_turn(xb, sn, cs);
        z0 = r * cs;
        z1 = r * sn;
    }
}

template <bool FAST>
__device__ __forceinline__ void normals4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1, float (&z)[4])
{
    const u32x4 x = philox4x32_10(c0, c1, c2, c3, k0, k1);
    box_muller<FAST>(x.x, x.y, z[0], z[1]);
    box_muller<FAST>(x.z, x.w, z[2], z[3]);
}

// uniform in (0,1) from one u32 (prior sampler)
__device__ __forceinline__ float uniform01(uint32_t x)
{
    return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}

}  // namespace nddm

// nddm_prepass.h -- the small kernels around the simulator (gfx950): the pre-pass that writes the hand-out records in
// longest-first order, the summary finalisation, the on-device prior sampler and the test aid.  Included by
// nddm_kernels.hip (one translation unit).
#pragma once
#include "nddm_sim.h"

namespace nddm {

// ------------------------------------------------------------------------------------------------
// Longest-first scheduling.  Expected trial length differs by two orders of magnitude across the prior (a set of
// wide-boundary, zero-drift trials keeps a wave busy ~100x longer than a fast one), so pulling sets in the given order
// leaves a long tail at the end of a launch and mixes fast and slow trials in one wave.  A counting sort by the
// expected number of Euler-Maruyama steps (closed-form mean first-passage time of the DDM, half-octave buckets,
// slowest bucket first) removes both: +8 % at 1M sets, +40 % at 100k.  Only the ORDER of processing changes; outputs
// stay at their set's position and do not depend on it.
constexpr int ORDER_BUCKETS = 32;

__device__ __forceinline__ int duration_bucket(int model, const float *p, float dt, int max_k)
{
    float v, a, beta, sg;
    switch (model) {
    case NDDM_BASIC_DDM_DC: v = p[0]; a = p[1]; beta = p[2]; sg = p[4]; break;
    case NDDM_SINGLE_TRIAL: v = p[0]; a = p[1]; beta = p[2]; sg = p[5]; break;
    case NDDM_SINGLE_TRIAL_ALT: v = p[0]; a = p[1]; beta = p[2]; sg = p[5]; break;
    case NDDM_ALPHA_NOT_SCALED: v = p[0]; a = p[1]; beta = p[2]; sg = p[5]; break;
    default: v = p[0]; a = 1.0f; beta = p[1]; sg = p[3]; break;
    }
    const float s2 = sg * sg, av = fabsf(v);
    const float z = v >= 0.0f ? a * beta : a - a * beta;   // mirror negative drift: same mean time, no exp overflow
    float et;                                           // mean first-passage time, seconds
    if (av * a < 1e-3f * s2) et = z * (a - z) / s2;
    else et = (a * (1.0f - __expf(-2.0f * av * z / s2)) / (1.0f - __expf(-2.0f * av * a / s2)) - z) / av;
    float steps = et / dt;
    if (!(steps >= 1.0f)) steps = 1.0f;                 // also catches NaN
    if (steps > (float)max_k) steps = (float)max_k;
    int b = (int)(2.0f * __log2f(steps));               // half-octave buckets
    b = b < 0 ? 0 : (b > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : b);
    return ORDER_BUCKETS - 1 - b;                       // bucket 0 = slowest
}

// ws[0..31] histogram, ws[32..63] cursors (both zeroed before the launch by zero_words_kernel)
__global__ void order_hist_kernel(int model, const float *params, int P, int B, float dt, int max_k, int *ws)
{
    __shared__ int h[ORDER_BUCKETS];
    if (threadIdx.x < ORDER_BUCKETS) h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x)
        atomicAdd(&h[duration_bucket(model, params + (long long)i * P, dt, max_k)], 1);
    __syncthreads();
    if (threadIdx.x < ORDER_BUCKETS && h[threadIdx.x]) atomicAdd(&ws[threadIdx.x], h[threadIdx.x]);
}

// set_offset + (off_dev ? *off_dev : 0) = global index of row 0 (nddm.h: nddm_simulate_indirect)
__device__ __forceinline__ unsigned long long global_offset(unsigned long long set_offset, const unsigned long long *off_dev)
{
    return set_offset + (off_dev ? *off_dev : 0ull);
}

// mode: bit 0 = fast Gaussian transform, bit 1 = NDDM_STATE_F64 (make_record)
__global__ void order_scatter_kernel(int model, int mode, const float *params, int P, int B, float dt, float sqrt_dt, int max_k,
                                     unsigned long long set_offset, const unsigned long long *off_dev, int *ws, uint32_t *recs)
{
    const unsigned long long g0 = global_offset(set_offset, off_dev);
    __shared__ int start[ORDER_BUCKETS], lh[ORDER_BUCKETS], lbase[ORDER_BUCKETS];
    if (threadIdx.x == 0) { int acc = 0; for (int b = 0; b < ORDER_BUCKETS; ++b) { start[b] = acc; acc += ws[b]; } }
    for (int base = blockIdx.x * blockDim.x; base < B; base += gridDim.x * blockDim.x) {
        if (threadIdx.x < ORDER_BUCKETS) lh[threadIdx.x] = 0;
        __syncthreads();
        const int i = base + threadIdx.x;
        int b = -1, r = 0;
        if (i < B) {
            b = duration_bucket(model, params + (long long)i * P, dt, max_k);
            r = atomicAdd(&lh[b], 1);                   // rank within this block's share of the bucket (LDS)
        }
        __syncthreads();
        if (threadIdx.x < ORDER_BUCKETS && lh[threadIdx.x])     // one global cursor bump per bucket per block
            lbase[threadIdx.x] = atomicAdd(&ws[ORDER_BUCKETS + threadIdx.x], lh[threadIdx.x]);
        __syncthreads();
        if (i < B) {
            const int q = start[b] + lbase[b] + r;      // position of set i in the processing order
            make_record(model, (mode & 1) != 0, (mode & 2) != 0, params + (long long)i * P, dt, sqrt_dt, i, g0 + (unsigned long long)i, recs + (long long)q * REC);
        }
        __syncthreads();
    }
}

// launches too small to be worth sorting: records in the given order
__global__ void prep_kernel(int model, int mode, const float *params, int P, int B, float dt, float sqrt_dt,
                            unsigned long long set_offset, const unsigned long long *off_dev, uint32_t *recs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B)
        make_record(model, (mode & 1) != 0, (mode & 2) != 0, params + (long long)i * P, dt, sqrt_dt, i,
                    global_offset(set_offset, off_dev) + (unsigned long long)i, recs + (long long)i * REC);
}

// ------------------------------------------------------------------------------------------------
// sets split into several tiles: add the tiles' integer partial sums up and finalise the summary row (one thread per
// set; exact integer arithmetic, so the result equals the single-tile path bit for bit)
__global__ void combine_partials_kernel(const unsigned long long *partials, int pw, const float *params, int P, int tau_idx,
                                        long long B, int tiles_per_set, int n_total, float tscale, float *out_summary)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    long long n_up = 0, n_lo = 0, n_miss = 0, sz = 0, szz = 0;
    unsigned long long sk = 0, sk2 = 0, sk_up = 0, sk2_up = 0;
    for (int t = 0; t < tiles_per_set; ++t) {
        const unsigned long long *q = partials + (b * tiles_per_set + t) * pw;       // layout: partial_words()
        n_up += (long long)(q[0] & 0x1fffffull); n_lo += (long long)((q[0] >> 21) & 0x1fffffull); n_miss += (long long)(q[0] >> 42);
        sk += q[1]; sk2 += q[2]; sk_up += q[3]; sk2_up += q[4];
        if (pw > 5) { sz += (long long)q[5]; szz += (long long)q[6]; }
    }
    finalize_summary(out_summary + b * NDDM_SUMMARY_K, (int)n_up, (int)n_lo, (int)n_miss, sk, sk2, sk_up, sk2_up, sz, szz,
                     n_total, tscale, params[b * P + tau_idx]);
}

// ------------------------------------------------------------------------------------------------
// the 2-byte wire format of a trial (step index | code << 14; nddm.h: nddm_simulate_codes) back to the float pair the
// simulator would have written -- the same arithmetic as flush_set: rt = fma(float(k), dt, tau); basic: (rt, choice),
// alpha_not_scaled: (choice * rt, (choice + 1) / 2).  One thread per trial; 2 B read, 8 B written.
__global__ void decode_codes_kernel(int model, const uint16_t *codes, const float *params, int P, int tau_idx, long long B,
                                    int n_trials, float dt, float2 *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (long long)n_trials) return;
    const long long b = i / n_trials;
    const uint32_t v = codes[i];
    const uint32_t k = v & 0x3fffu, code = v >> 14;
    const float ch = code == 1u ? 1.0f : (code == 2u ? -1.0f : 0.0f);
    const float rt = __builtin_fmaf((float)k, dt, params[b * P + tau_idx]);
    float2 o;
    if (model == NDDM_BASIC_DDM_DC) { o.x = rt; o.y = ch; }
    else { o.x = ch * rt; o.y = 0.5f * (ch + 1.0f); }
    out[i] = o;
}

// ------------------------------------------------------------------------------------------------
// debugging kernel for the parity tests: 4 normals per counter
template <bool FAST>
__global__ void debug_normals_kernel(const uint32_t *ctr, long long n, uint32_t k0, uint32_t k1, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float z[4];
    normals4<FAST>(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, z);
    out[4 * i] = z[0]; out[4 * i + 1] = z[1]; out[4 * i + 2] = z[2]; out[4 * i + 3] = z[3];
}

// ------------------------------------------------------------------------------------------------
// on-device draw_prior(): basic_ddm_dc.py:62-80 / single_trial_alpha_not_scaled.py:78-102 (+ _alt :889-913,
// _scale :1205-1232 share the marginals).  Stream 2 of the row; one thread per row.
struct PriorStream {
    uint32_t k0, k1, row_lo, c3, draw;
    float z[4];
    uint32_t u[4];
    int nz, nu;
    __device__ PriorStream(uint32_t k0_, uint32_t k1_, uint64_t row)
        : k0(k0_), k1(k1_), row_lo((uint32_t)row), c3(((uint32_t)(row >> 32) & 0x0fffffffu) | 0x20000000u),
          draw(0), nz(0), nu(0) {}
    __device__ float normal()
    {
        if (nz == 0) { normals4<false>(draw++, 0u, row_lo, c3, k0, k1, z); nz = 4; }
        const int j = 4 - nz; nz--;
        return j == 0 ? z[0] : (j == 1 ? z[1] : (j == 2 ? z[2] : z[3]));
    }
    __device__ float uniform()
    {
        if (nu == 0) { const u32x4 x = philox4x32_10(draw++, 1u, row_lo, c3, k0, k1); u[0] = x.x; u[1] = x.y; u[2] = x.z; u[3] = x.w; nu = 4; }
        const int j = 4 - nu; nu--;
        return uniform01(j == 0 ? u[0] : (j == 1 ? u[1] : (j == 2 ? u[2] : u[3])));
    }
    // N(mean, sd) truncated to [low, upp] by rejection (truncnorm_better, basic_ddm_dc.py:55-57)
    __device__ float truncnorm(float mean, float sd, float low, float upp)
    {
        float v = mean;
        for (int i = 0; i < 256; ++i) {
            v = __builtin_fmaf(sd, normal(), mean);
            if (v >= low && v <= upp) break;
        }
        return fminf(fmaxf(v, low), upp);
    }
    // Beta(2,2) = the median of three uniforms (order statistic U_(2:3))
    __device__ float beta22()
    {
        const float a = uniform(), b = uniform(), c = uniform();
        return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
    }
};

__global__ void prior_kernel(int model, long long B, uint32_t k0, uint32_t k1, unsigned long long set_offset,
                             const unsigned long long *off_dev, float gamma, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    PriorStream s(k0, k1, set_offset + (off_dev ? *off_dev : 0ull) + (unsigned long long)i);
    if (model == NDDM_BASIC_DDM_DC) {
        float *o = out + i * 5;
        o[0] = 2.0f * s.normal();                       // drift ~ N(0, 2)          basic_ddm_dc.py:65
        o[1] = s.truncnorm(1.0f, 0.5f, 0.0f, 10.0f);    // alpha ~ TN(1,.5; 0,10)    :68
        o[2] = s.beta22();                              // beta ~ Beta(2,2)          :71
        o[3] = s.truncnorm(0.5f, 0.25f, 0.0f, 1.5f);    // ter ~ TN(.5,.25; 0,1.5)   :74
        o[4] = s.truncnorm(1.0f, 0.5f, 0.0f, 10.0f);    // dc ~ TN(1,.5; 0,10)       :77
    } else {   // single-trial family: same marginals for base / _alt / _scale
        float *o = out + i * 8;
        o[0] = 2.0f * s.normal();                       // single_trial_alpha_not_scaled.py:81
        o[1] = s.truncnorm(1.0f, 0.5f, 0.0f, 10.0f);    // mu_alpha                  :84
        o[2] = s.beta22();                              //                           :87
        o[3] = s.truncnorm(0.5f, 0.25f, 0.0f, 1.5f);    //                           :90
        o[4] = s.truncnorm(1.0f, 0.5f, 0.0f, 3.0f);     // std_alpha ~ TN(1,.5; 0,3) :93
        o[5] = s.truncnorm(1.0f, 0.5f, 0.0f, 10.0f);    // dc                        :96
        o[6] = 5.0f * s.uniform();                      // sigma1 ~ U(0,5)           :99
        o[7] = gamma >= 0.0f ? gamma : 2.0f * s.uniform();   // gamma ~ U(0,2) (:1229) when gamma < 0 is passed
    }
}

}  // namespace nddm

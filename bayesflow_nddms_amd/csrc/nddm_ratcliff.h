// nddm_ratcliff.h -- simulratcliff (pyhddmjagsutils.py:47-176), the EXACT first-passage sampler that the reference's
// alpha_not_scaled.py:95-108 generates its data with, as a gfx950 kernel: no Euler-Maruyama grid, no dt.  Included by
// nddm_kernels.hip (one translation unit); entry point nddm_simulratcliff (include/nddm.h).
//
// The algorithm (Tuerlinckx et al. 2001): from the current position the process leaves the largest sphere that fits between the
// boundaries -- radius = distance to the nearer one -- upwards with probability 1 / (1 + exp(-radius mu / D)), after a time drawn by
// rejection from the sphere's exit-time law (a theta-function series); a step that reaches the nearer boundary ends the trial, a step
// away moves the position by the radius.  Every rounding of the exact mode is spelled out below (float32, contraction off: the test
// suite's CPU checker restates this file operation by operation -- its section D -- and the exact mode equals it bit for bit):
//   mu = fmaf(Eta, z, Nu) with Nu clipped to +-5; D = Varsigma^2 / 2; per sphere of radius r = min(du, dl):
//   lambda = (0.25 mu^2) / D + (0.25 D pi^2) / r^2;  F = 1 / (1 + G^2), G = r mu / (D pi) (the reference's F0^2 / (1 + F0^2), defined
//   at mu = 0);  P(up) = 1 / (1 + e) or e / (1 + e), e = exp(-|r mu / D|);  an attempt (s2, s1) is accepted when
//   s2 e^-a <= e^-a + sum_{k = 3, 5, ..} (-+) k e^{-a k^2},  a = -F ln s1  (the reference's s2 <= 1 + s1^-F * series, without a positive
//   exponent); the series stops when a term no longer changes the float32 sum (<= 64 terms); a < 2^-6 is rejected without it (the bound
//   is below 2^-33, the smallest uniform); the sphere's time is -ln(s1) / lambda.  Caps that make every loop finite: 4096 attempts
//   (then accepted), 4096 spheres (then ended on the nearer boundary).  Laid out for 64-lane waves:
//   * one single-wave workgroup per tile (= one parameter set of <= 512 trials), PERSISTENT LANES inside it: the trial loop, the
//     sphere loop and the rejection loop of the reference are flattened into ONE loop whose trip is one rejection attempt, and a lane
//     whose trial ended takes the set's next trial at the top of the next trip (ballot + mbcnt), so the three data-dependent loop
//     counts (1-10 spheres, 1-5 attempts, 2-18 series terms) do not multiply into idle lanes;
//   * randomness: counter-based like every other stream of the library -- the per-trial drift is auxiliary normal 0 of the trial
//     (stream 1: the very draw the Euler-Maruyama form of the model uses), the uniforms are stream 3 of the trial, consumed in order;
//   * results are staged in LDS as one float per trial (the decision time with the response as its sign bit) and flushed as whole
//     float2 (y, acc) lines with the fused summary reduction (integer sums of the decision time in 2^-16 s: bit-reproducible).
// The path is VALU / transcendental bound (two exp per series term); 8 B are written per trial.
#pragma once
#include "nddm_sim.h"

namespace nddm {

constexpr int RATCLIFF_MAX_TERMS = 64, RATCLIFF_MAX_ATTEMPTS = 4096, RATCLIFF_MAX_SPHERES = 4096;

struct RatArgs {
    const float *params;            // [B, 6]: Nu, Alpha, Beta, Tau, Eta, Varsigma
    float *out_trials;              // [B, N, 2] (y, acc) or null
    float *out_summary;             // [B, K] or null (tiles_per_set == 1: written here; else through partials + combine_partials_kernel)
    float *out_ext;                 // [B] or null
    unsigned long long *partials;   // [B * tiles_per_set, 5] or null
    long long n_vsets;              // B * tiles_per_set
    int n_trials;                   // trials per tile
    int n_total;                    // trials per set
    int tiles_per_set;
    uint32_t k0, k1;
    unsigned long long set_offset;
    float ext_sigma;
    int ext_mode;
};

template <bool FAST> __device__ __forceinline__ float rat_neg_log(float u)        // -ln u, u in (0, 1]
{
    if constexpr (FAST) return -0.693147180559945309f * __builtin_amdgcn_logf(u);
    else return -exact_logf(u);
}
template <bool FAST> __device__ __forceinline__ float rat_exp_neg(float y)        // exp(-y), y >= 0
{
    if constexpr (FAST) return __builtin_amdgcn_exp2f(-1.44269504088896341f * y);
    else return exact_expf_neg(-y);
}

// uniform j of (set, trial): word j & 3 of block j >> 2 of stream 3
struct UnifStream {
    uint32_t set_lo, trial, c2, blk, q;
    u32x4 x;
    __device__ __forceinline__ void init(uint32_t set_lo_, uint32_t set_hi28, uint32_t trial_)
    {
        set_lo = set_lo_; trial = trial_; c2 = set_hi28 | 0x30000000u; blk = 0xffffffffu; q = 0u;
        x = {0u, 0u, 0u, 0u};
    }
    __device__ __forceinline__ float next(uint32_t k0, uint32_t k1)
    {
        const uint32_t b = q >> 2;
        if (b != blk) { x = philox4x32_10(set_lo, trial, c2, b, k0, k1); blk = b; }
        const uint32_t j = q & 3u;
        const uint32_t w = j == 0u ? x.x : (j == 1u ? x.y : (j == 2u ? x.z : x.w));
        q++;
        return uniform01(w);
    }
};

template <bool FAST>
__global__ __launch_bounds__(WAVE) void ratcliff_kernel(const RatArgs A)
{
    extern __shared__ uint32_t lds_raw[];
    float *const staged = reinterpret_cast<float *>(lds_raw);          // [n_trials]: copysign(decision time, response)
    const int lane = threadIdx.x;
    for (long long vset = blockIdx.x; vset < A.n_vsets; vset += gridDim.x) {
        const int TPS = A.tiles_per_set;
        const long long set = TPS == 1 ? vset : vset / TPS;
        const int t0 = TPS == 1 ? 0 : (int)(vset - set * TPS) * A.n_trials;
        const int n_here = (A.n_total - t0) < A.n_trials ? (A.n_total - t0) : A.n_trials;
        // ---- per-set constants
        const float *p = A.params + set * 6;
        float Nu = p[0];
        if (Nu < -5.0f || Nu > 5.0f) Nu = Nu > 0.0f ? 5.0f : -5.0f;                    // pyhddmjagsutils.py:102-103
        const float Alpha = p[1], Beta = p[2], Tau = p[3], Eta = p[4], Vs = p[5];
        const float D = (Vs * Vs) * 0.5f, inv_D = 1.0f / D;                             // :117
        const float c_lam2 = (0.25f * D) * 9.86960440108935862f;                        // 0.25 D pi^2
        const float zz = Beta * Alpha, du0 = Alpha - zz, dl0 = zz;
        const unsigned long long gset = A.set_offset + (unsigned long long)set;
        const uint32_t set_lo = (uint32_t)gset, set_hi = (uint32_t)(gset >> 32) & 0x0fffffffu;

        // ---- per-lane trial state
        bool has = false;
        int slot = 0;                                     // the trial's index within the tile
        float du = 0.0f, dl = 0.0f, total = 0.0f, lam1 = 0.0f, g1 = 0.0f, x1 = 0.0f, lam = 1.0f, F = 0.0f;
        bool up = false;
        int sphere = 0, att = 0;
        UnifStream us;
        us.init(0u, 0u, 0u);
        int next = 0;                                     // wave-uniform: the tile's next unassigned trial

        // the sphere that starts at the current position: its constants and its direction (one uniform) -- or the end of the
        // trial, when the position lies on a boundary (Beta = 0 or 1) or the safety cap is reached; returns "the trial goes on"
        auto setup_sphere = [&]() -> bool {
            const float radius = fminf(du, dl);
            if (!(radius > 0.0f) || sphere >= RATCLIFF_MAX_SPHERES) {
                staged[slot] = copysignf(total, du <= dl ? 1.0f : -1.0f);
                return false;
            }
            lam = lam1 + c_lam2 / (radius * radius);                                     // :138
            const float G = radius * g1;
            F = 1.0f / __builtin_fmaf(G, G, 1.0f);                                       // :140-141, F0^2 / (1 + F0^2) with F0 = 1 / G
            const float x = radius * x1;
            const float e = rat_exp_neg<FAST>(__builtin_fabsf(x));
            const float p_up = (x >= 0.0f) ? 1.0f / (1.0f + e) : e / (1.0f + e);         // :143-144
            up = us.next(A.k0, A.k1) < p_up;                                             // :145
            att = 0;
            return true;
        };

        while (true) {
            // ---- hand out the tile's next trials to the lanes that hold none
            const unsigned long long want = __builtin_amdgcn_ballot_w64(!has);
            if (next < n_here && want) {
                const int tr = next + (int)lane_rank(want);
                const bool take = !has && tr < n_here;
                if (take) {
                    slot = tr;
                    const uint32_t trial = (uint32_t)(t0 + tr);
                    float z[4];
                    normals4<FAST>(set_lo, trial, set_hi | 0x10000000u, 0u, A.k0, A.k1, z);     // auxiliary normal 0 of the trial
                    const float mu = __builtin_fmaf(Eta, z[0], Nu);                              // :124-125
                    lam1 = (0.25f * (mu * mu)) * inv_D;
                    g1 = mu * (inv_D * 0.318309886183790672f);
                    x1 = mu * inv_D;
                    du = du0; dl = dl0; total = 0.0f; sphere = 0;
                    us.init(set_lo, set_hi, trial);
                    has = setup_sphere();
                }
                const int n_want = (int)__popcll(want);
                next = next + n_want < n_here ? next + n_want : n_here;
            }
            if (!__builtin_amdgcn_ballot_w64(has)) {
                if (next >= n_here) break;
                continue;                                 // (every lane that took a trial ended it at once: start on a boundary)
            }
            // ---- one rejection attempt of every lane that holds a trial (:147-159)
            if (has) {
                const float s2 = us.next(A.k0, A.k1), s1 = us.next(A.k0, A.k1);
                const float nl = rat_neg_log<FAST>(s1);
                const float a = F * nl;
                bool accept = false;
                att++;
                if (att >= RATCLIFF_MAX_ATTEMPTS) accept = true;
                else if (!(a < 0.015625f)) {
                    float tnew = 0.0f, told;
                    int uu = 0;
                    do {
                        told = tnew;
                        uu++;
                        const float k = (float)(2 * uu + 1);
                        const float term = k * rat_exp_neg<FAST>(a * (k * k));
                        tnew = (uu & 1) ? told - term : told + term;
                    } while (tnew != told && uu < RATCLIFF_MAX_TERMS);
                    const float ea = rat_exp_neg<FAST>(a);
                    accept = s2 * ea <= ea + tnew;
                }
                if (accept) {
                    total += nl / lam;                                                   // :161-163
                    if (up ? (du <= dl) : (dl <= du)) {                                  // the nearer boundary is reached (:165-172)
                        staged[slot] = copysignf(total, up ? 1.0f : -1.0f);
                        has = false;
                    } else {
                        const float radius = fminf(du, dl);
                        if (up) { du -= radius; dl += radius; } else { du += radius; dl -= radius; }      // :174-175
                        sphere++;
                        has = setup_sphere();
                    }
                }
            }
        }
        __syncthreads();
        // ---- flush: whole float2 lines + the fused summary (integer sums, decision time in 2^-16 s)
        float2 *out = A.out_trials ? reinterpret_cast<float2 *>(A.out_trials) + set * A.n_total + t0 : nullptr;
        int n_up = 0;
        unsigned long long sk = 0, sk2 = 0, sk_up = 0, sk2_up = 0;
        for (int j = lane; j < n_here; j += WAVE) {
            const float s = staged[j];
            const bool upper = (__float_as_uint(s) >> 31) == 0u;
            const float tot = __builtin_fabsf(s);
            const float rt = Tau + tot;
            if (out) { float2 o; o.x = upper ? rt : -rt; o.y = upper ? 1.0f : 0.0f; out[j] = o; }
            if (A.out_summary) {
                const uint32_t tfix = (uint32_t)__builtin_fmaf(fminf(tot, 1024.0f), 65536.0f, 0.5f);
                const unsigned long long sq = (unsigned long long)tfix * tfix;
                sk += tfix; sk2 += sq;
                if (upper) { n_up++; sk_up += tfix; sk2_up += sq; }
            }
        }
        if (A.out_summary) {
            n_up = wave_sum(n_up);
            sk = wave_sum(sk); sk2 = wave_sum(sk2); sk_up = wave_sum(sk_up); sk2_up = wave_sum(sk2_up);
            if (lane == 0) {
                if (A.partials) {
                    unsigned long long *q = A.partials + vset * 5;
                    q[0] = (unsigned long long)n_up | ((unsigned long long)(n_here - n_up) << 21);
                    q[1] = sk; q[2] = sk2; q[3] = sk_up; q[4] = sk2_up;
                } else {
                    finalize_summary(A.out_summary + set * NDDM_SUMMARY_K, n_up, n_here - n_up, 0, sk, sk2, sk_up, sk2_up, 0, 0, A.n_total,
                                     1.52587890625e-05f, Tau);
                }
            }
        }
        if (A.out_ext && lane == 0 && t0 == 0) {
            float z[4];
            normals4<FAST>(set_lo, 0xffffffffu, set_hi | 0x10000000u, 0u, A.k0, A.k1, z);       // the set's external datum (alpha_not_scaled.py:103-106)
            A.out_ext[set] = __builtin_fmaf(A.ext_sigma, z[0], (A.ext_mode == 0) ? Alpha : 1.0f);
        }
        __syncthreads();
    }
}

}  // namespace nddm

// nddm_kernels.hip -- Euler-Maruyama DDM trial simulators for MI355X (gfx950), and their C ABI.
//
// Replaces, behind include/nddm.h, the numba/NumPy simulators of the reference:
//   basic_ddm_dc.py:85-125, single_trial_alpha_not_scaled.py:107-155 (+ :926-974, :1237-1285,
//   :1471-1519, :1710-1722), imputation_from_stahl_not_scaled.py:120-148, and the
//   alpha_not_scaled.py:52-128 generator recast as an Euler-Maruyama process.
//
// Execution design:
//   * a pre-pass writes one 48-byte HAND-OUT RECORD per parameter set -- everything of a trial hand-out that is
//     floating-point arithmetic on the parameter row -- in PROCESSING order: longest expected trials first (a counting
//     sort by the DDM's closed-form mean first-passage time), so the launch has no slow-set tail and the lanes of a wave
//     work on trials of similar length.  The simulator never reads the parameter rows.
//   * one 64-lane wavefront per workgroup, PERSISTENT grid (as many waves as stay resident: 8 per SIMD); a wave pulls
//     CHUNKS of consecutive records from a device-wide atomic counter and streams through them without draining
//     between chunks; the next tile's record is loaded one tile ahead.
//   * one lane = one trial at a time.  Trial length is heavy-tailed (median 107, p99 2243
//     steps at dt=.001), so lanes are PERSISTENT: a lane whose trial has ended retires it and
//     takes the next unassigned (set, trial) of the wave's stream in order (wave ballot +
//     prefix count), instead of idling until the slowest trial of its set ends.
//   * results are staged in an LDS ring of per-set slots as packed (step index | choice);
//     when the last trial of a set retires the wave FLUSHES the slot: one coalesced float2
//     store sweep to HBM (every line written whole, once; the external datum of the models that have one is recomputed
//     there from the trial's auxiliary stream) plus the fused per-set summary reduction (integer sums reduced across
//     the wave with DPP adds, so summaries are bit-reproducible).
//   * the evidence is carried centred and in noise units, w = (x - a/2) / sigma: one step is
//     w = fma(r, cos|sin, w) + mu with the unit Box-Muller radius, the range test is |w| < h.
//   * the Gaussian stream is counter-based (nddm_rng.h): no RNG state is loaded or stored.
//     Philox's wave-uniform round keys are served from LDS as broadcast reads, because a VALU xor
//     that reads an SGPR operand issues at half the rate of a VGPR-only one on gfx950.
//   * what limits residency is the SGPR file (<= 74 SGPRs for 8 waves per SIMD): hence the kernarg reads at the point of
//     use, the single LDS record per slot, the SMALL template split.
//
// The path is VALU/transcendental-bound: 8 B are written per trial for ~246 Gaussian draws.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <thread>

#include "../../include/nddm.h"
#include "nddm_rng.h"

namespace nddm {

constexpr int WAVE = 64;
constexpr int MAX_REJECT = 64;   // cap of the per-trial latent's rejection loop (P(reject) <= 1/2 per draw)

template <int MODEL> struct ModelTraits;
template <> struct ModelTraits<NDDM_BASIC_DDM_DC>      { static constexpr int P = 5; static constexpr bool HAS_Z = false; static constexpr int TAU = 3; };
template <> struct ModelTraits<NDDM_SINGLE_TRIAL>      { static constexpr int P = 8; static constexpr bool HAS_Z = true;  static constexpr int TAU = 3; };
template <> struct ModelTraits<NDDM_SINGLE_TRIAL_ALT>  { static constexpr int P = 8; static constexpr bool HAS_Z = true;  static constexpr int TAU = 3; };
template <> struct ModelTraits<NDDM_ALPHA_NOT_SCALED>  { static constexpr int P = 6; static constexpr bool HAS_Z = false; static constexpr int TAU = 3; };
template <> struct ModelTraits<NDDM_EXPLICIT_BOUNDARY> { static constexpr int P = 4; static constexpr bool HAS_Z = true;  static constexpr int TAU = 2; };

struct SimArgs {
    const float *params;      // [B, P]
    const float *bounds;      // [B, N] (explicit-boundary model) or null
    float *out_trials;        // [B, N, 2] or null
    float *out_summary;       // [B, K] or null
    float *out_ext;           // [B] or null
    long long B;
    unsigned long long set_offset;
    int n_trials;             // trials per TILE (a set is split into tiles_per_set tiles when it does not fit the LDS ring)
    int n_total;              // trials per set (row stride of out_trials / bounds)
    int tiles_per_set;
    unsigned long long *partials;   // [B * tiles_per_set, partial_words()] integer partial sums of the tiles (summaries requested), else null
    const uint32_t *recs;     // [B, REC] per-set hand-out records in PROCESSING order (longest expected trials first when
                              // the launch is large enough to be sorted, else as given): make_record() / prep_kernel
    int max_k;
    float dt;
    float sqrt_dt;
    float tscale;             // seconds per unit of the packed time field: dt, or dt/256 with the bridge correction
    uint32_t k0, k1;
    int sets_per_chunk;
    int n_chunks;
    unsigned int *chunk_counter;   // device words [0] next chunk to hand out, [1] waves that have left; both are zero
                                   // between launches: the last wave to leave resets them
    int ring;                 // LDS ring slots (power of two)
    int open_ahead;           // tiles staged ahead of the one being handed out (0 when work is scarce, else 1)
    float ext_sigma;
    int ext_mode;
    unsigned long long *dbg;  // optional [8] counters (blocks, refills, memtime, memrealtime, waves); null in production
    int res16;                // results are staged as 16-bit words (step index < 2^14 | code << 14): no bridge, cap < 16384;
                              // 2 = ... and the tile has <= 512 trials (the flush's 32-bit / DPP reduction path)
    int refill_thresh;        // leave the step loop once this many lanes hold a finished trial
    int max_blocks;           // (unused by the kernels: the block limit is 16, switched off by refill_thresh >= 64)
};

// The launch arguments as they sit in the kernarg segment (constant address space: scalar loads).  The rarely executed
// parts of the kernel (opening a tile, flushing a set) read their arguments through this pointer at the point of use,
// behind a compiler barrier, instead of keeping ~30 SGPRs live through the step loop: SGPRs, not VGPRs, limit these
// kernels' residency.
typedef const __attribute__((address_space(4))) SimArgs *ArgsPtr;
__device__ __forceinline__ ArgsPtr fresh_args(ArgsPtr p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// auxiliary normal `a` of (set, trial): stream 1.  One Philox block serves normals 4b..4b+3.  Everything lives in named
// registers (an array indexed by `a & 3` ends up in private scratch): the common hand-out asks for normals 0 and 1 --
// one Box-Muller pair, first_pair() -- and only a rejected draw goes on to normal(a), which evaluates the pair that
// holds normal `a` and selects its cosine or sine half.
template <bool FAST>
struct AuxStream {
    uint32_t trial, set_lo, c3, blk, kbase;
    u32x4 x;
    __device__ __forceinline__ AuxStream(uint32_t kbase_, uint32_t set_lo_, uint32_t set_hi28, uint32_t trial_)
        : trial(trial_), set_lo(set_lo_), c3(set_hi28 | 0x10000000u), blk(0xffffffffu), kbase(kbase_) {}
    __device__ __forceinline__ void block(uint32_t b)
    {
        if (b != blk) { x = philox4x32_10_lds(set_lo, trial, c3, b, kbase); blk = b; }
    }
    // normals 0 and 1
    __device__ __forceinline__ void first_pair(float &z0, float &z1)
    {
        block(0u);
        float r, cs, sn;
        polar_pair<FAST>(x.x, x.y, r, cs, sn);
        r *= noise_unit<FAST>(1.0f);
        z0 = r * cs; z1 = r * sn;
    }
    __device__ __forceinline__ float normal(uint32_t a)
    {
        block(a >> 2);
        const bool second = (a & 2u) != 0u;
        float r, cs, sn;
        polar_pair<FAST>(second ? x.z : x.x, second ? x.w : x.y, r, cs, sn);
        r *= noise_unit<FAST>(1.0f);
        return r * ((a & 1u) ? sn : cs);
    }
};

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ void finalize_summary(float *o, int n_up, int n_lo, int n_miss, unsigned long long sk,
                                                 unsigned long long sk2, unsigned long long sk_up,
                                                 unsigned long long sk2_up, long long sz, long long szz, int n_total,
                                                 float tscale, float tau)
{
    const double dtd = (double)tscale, taud = (double)tau;
    const double n_resp = (double)(n_up + n_lo);
    o[0] = (float)n_up; o[1] = (float)n_lo; o[2] = (float)n_miss;
    if (n_resp > 0) {
        const double mk = (double)sk / n_resp;
        const double vk = (double)sk2 / n_resp - mk * mk;
        o[3] = (float)(taud + dtd * mk);
        o[4] = (float)(dtd * dtd * vk);
    } else { o[3] = __builtin_nanf(""); o[4] = __builtin_nanf(""); }
    if (n_up > 0) {
        const double nu = (double)n_up;
        const double mk = (double)sk_up / nu;
        const double vk = (double)sk2_up / nu - mk * mk;
        o[5] = (float)(taud + dtd * mk);
        o[6] = (float)(dtd * dtd * vk);
    } else { o[5] = __builtin_nanf(""); o[6] = __builtin_nanf(""); }
    const double Nd = (double)n_total;
    const double mz = ((double)sz / 4294967296.0) / Nd;
    const double vz = ((double)szz / 16777216.0) / Nd - mz * mz;
    o[7] = (float)mz;
    o[8] = (float)vz;
    o[9] = (float)(((double)n_up + 0.5 * (double)n_miss) / Nd);
}

// Per-set constants of the trial hand-out.  Everything that is floating-point arithmetic on the parameter row is done
// ONCE per set by a pre-pass (prep_kernel, or the scatter pass of the longest-first sort) into a REC-dword record, stored
// in PROCESSING order so that a wave streams them sequentially; the simulator loads the record of the next tile one tile
// ahead (one 48-byte vector load, consumed a few thousand cycles later) and never reads the parameter rows itself:
//   r[0..3]  A   basic / alpha_ns / explicit: drift*dt/S (alpha_ns: Nu; per-trial drift), 1/S, a/(2S), (a*beta - a/2)/S
//                (explicit: beta, 0 in the last two);  single: drift*dt/S, 1/S, std_alpha, mu_alpha;
//                single_alt (noise scale per trial): drift, alpha, beta, std_dc
//   r[4..7]  B   single: sigma1, gamma, beta;  single_alt: mu_dc, sigma1, gamma;  alpha_ns: Eta
//   r[8]     the set's row index in the caller's arrays, r[9] tau
// S = noise_unit(sqrt(dt) * dc): the state is carried in noise units (nddm_rng.h).
enum { REC = 12, R_A = 0, R_B = 4, R_SET = 8, R_TAU = 9 };

__device__ __forceinline__ void make_record(int model, bool fast, const float *row, float dt, float sqrt_dt, int set,
                                            uint32_t *r)
{
    float drift = 0.0f, a = 0.0f, beta = 0.0f, sig_c = 1.0f, tau;
    float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
    switch (model) {
    case NDDM_BASIC_DDM_DC: drift = row[0]; a = row[1]; beta = row[2]; tau = row[3]; sig_c = row[4]; break;
    case NDDM_SINGLE_TRIAL: drift = row[0]; beta = row[2]; tau = row[3]; sig_c = row[5]; b0 = row[6]; b1 = row[7]; b2 = row[2]; break;
    case NDDM_SINGLE_TRIAL_ALT: tau = row[3]; b0 = row[5]; b1 = row[6]; b2 = row[7]; break;
    case NDDM_ALPHA_NOT_SCALED: a = row[1]; beta = row[2]; tau = row[3]; sig_c = row[5]; b0 = row[4]; break;
    default: drift = row[0]; beta = row[1]; tau = row[2]; sig_c = row[3]; break;
    }
    const float unit = sqrt_dt * sig_c;
    const float inv_s = 1.0f / (fast ? noise_unit<true>(unit) : noise_unit<false>(unit));
    const float hv = 0.5f * a;
    float a0 = (drift * dt) * inv_s, a1 = inv_s, a2 = hv * inv_s, a3 = (a * beta - hv) * inv_s;
    if (model == NDDM_SINGLE_TRIAL) { a2 = row[4]; a3 = row[1]; }
    else if (model == NDDM_SINGLE_TRIAL_ALT) { a0 = row[0]; a1 = row[1]; a2 = row[2]; a3 = row[4]; }
    else if (model == NDDM_ALPHA_NOT_SCALED) { a0 = row[0]; }
    else if (model == NDDM_EXPLICIT_BOUNDARY) { a2 = beta; a3 = 0.0f; }
    r[0] = __float_as_uint(a0); r[1] = __float_as_uint(a1); r[2] = __float_as_uint(a2); r[3] = __float_as_uint(a3);
    r[4] = __float_as_uint(b0); r[5] = __float_as_uint(b1); r[6] = __float_as_uint(b2); r[7] = 0u;
    r[8] = (uint32_t)set; r[9] = __float_as_uint(tau); r[10] = 0u; r[11] = 0u;
}

// The tile's record in LDS, as the hand-out reads it (ds_read_b128 each): dword index into its DV dwords.  The first REC
// dwords are the set's record as loaded (one masked store), the rest is filled in by lane 0 when the tile opens.
enum { D_A = 0, D_B = 4,                             // r[0..3], r[4..7]
       D_SIC = 8, D_TAU = 9, D_TBASE = 10, D_VSET = 11, // in-call set index, tau, first trial of the tile within its set,
                                                     // virtual set (set * tiles_per_set + tile)
       D_CA = 12, D_CB = 13, D_HP1K = 14, D_X1 = 15, // Philox constants of the set (PathSet in nddm_rng.h)
       D_C3 = 16, D_SETLO = 17, D_CNT = 18,          // high set word (28 bits), low set word (auxiliary stream's counter);
                                                     // trials of the tile retired so far
       D_ZSUM = 20,                                  // [20..23] fixed-point sums of z and z^2 (models with a z summary), or
       D_BCA = 20, D_BCB = 21, D_BHP1K = 22, D_BX1 = 23 }; // PathSet of the bridge-uniform stream (stream 3; BRIDGE only)
constexpr int DV = 24;                               // one layout for every model: only two LDS base addresses stay live
static_assert(D_SIC == R_SET && D_TAU == R_TAU, "the LDS record starts with the loaded record");
constexpr int LDS_HEADER_DWORDS = 32;                // key table [0,20) | debug stamps [20,26) | kC kD kE [28,31)

// The per-trial latent of the single-trial family and the external datum that goes with it: a pure function of
// (set, trial) and the set's record, so the hand-out (which needs the latent) and the flush (which writes the datum next
// to the choice-RT, one whole float2 per trial) each evaluate it where they need it; the part a caller does not use is
// dead code there.  Normal 0 of the auxiliary stream is the datum's noise, normals 1, 2, ... the rejection draws.
//   single:     latent = boundary ~ N(mu_alpha, std_alpha) > 0 (single_trial_alpha_not_scaled.py:113-116),  z1 ~ N(gamma * boundary, sigma1) (:134)
//   single_alt: latent = dc       ~ N(mu_dc, std_dc) > 0       (:932-935),                                z1 ~ N(gamma * dc, sigma1)
template <int MODEL, bool FAST>
__device__ __forceinline__ void trial_latent(const uint4 dA, const uint4 dB, uint32_t set_lo, uint32_t c3, uint32_t trial,
                                             uint32_t kbase, float &latent, float &z)
{
    static_assert(MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT, "models with a per-trial latent");
    // single: A = drift*dt/S, 1/S, std_alpha, mu_alpha;  B = sigma1, gamma, beta
    // alt:    A = drift, alpha, beta, std_dc;            B = mu_dc, sigma1, gamma
    const float sd = MODEL == NDDM_SINGLE_TRIAL ? __uint_as_float(dA.z) : __uint_as_float(dA.w);
    const float mean = MODEL == NDDM_SINGLE_TRIAL ? __uint_as_float(dA.w) : __uint_as_float(dB.x);
    const float sigma1 = MODEL == NDDM_SINGLE_TRIAL ? __uint_as_float(dB.x) : __uint_as_float(dB.y);
    const float gamma = MODEL == NDDM_SINGLE_TRIAL ? __uint_as_float(dB.y) : __uint_as_float(dB.z);
    AuxStream<FAST> aux(kbase, set_lo, c3, trial);
    float z0, z1;
    aux.first_pair(z0, z1);
    float v = __builtin_fmaf(sd, z1, mean);
    for (uint32_t ai = 2; !(v > 0.0f) && ai <= MAX_REJECT; ++ai) v = __builtin_fmaf(sd, aux.normal(ai), mean);
    if (!(v > 0.0f)) v = fabsf(v);
    latent = v;
    z = __builtin_fmaf(sigma1, z0, gamma * v);
}

// Fixed-point terms of the external datum's sums: trunc(z * 2^32) and trunc(z^2 * 2^24) with z clamped to +-2^18, as the
// oracle computes them through doubles -- here by integer arithmetic on the float's bits (gfx950 has no f64 -> i64
// conversion: the double route is ~80 instructions per hand-out, this one ~20).  |z| * 2^32 = m * 2^(e-118) with the
// 24-bit significand m and biased exponent e <= 145: (m << 29) >> (147 - e); z^2 * 2^24 = m^2 * 2^(2e-276):
// (m^2 << 15) >> (291 - 2e).  Shift counts are clamped to 63, where the (< 2^63) operand has become 0 as it should.
__device__ __forceinline__ void z_fixed_point(float z, long long &fz, long long &fzz)
{
    z = fminf(fmaxf(z, -262144.0f), 262144.0f);
    const uint32_t b = __float_as_uint(z);
    const uint32_t e = (b >> 23) & 0xffu;
    const uint32_t m = (b & 0x007fffffu) | 0x00800000u;
    const uint32_t s1 = 147u - e, s2 = 291u - 2u * e;
    const unsigned long long mag = ((unsigned long long)m << 29) >> (s1 < 63u ? s1 : 63u);
    fz = (b >> 31) ? -(long long)mag : (long long)mag;
    fzz = (long long)((((unsigned long long)m * m) << 15) >> (s2 < 63u ? s2 : 63u));
}

// Sum of a 32-bit value over the 64 lanes, in the vector ALU's data-parallel-primitive lanes (no LDS traffic, six adds):
// row_shr 1, 2, 4, 8 leave each 16-lane row's sum in its last lane, row_bcast 15 / 31 carry it into the next rows; lanes
// whose DPP source lies outside the row read the `old` operand, 0.  The wave's total ends up in lane 63.  Needs all 64
// lanes active; the caller reads the result in lane 63.  (The shuffle-based wave_sum above costs six ds_bpermute round
// trips per value.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
    v = dpp_add<0x111, 0xf>(v);        // row_shr:1
    v = dpp_add<0x112, 0xf>(v);        // row_shr:2
    v = dpp_add<0x114, 0xf>(v);        // row_shr:4
    v = dpp_add<0x118, 0xf>(v);        // row_shr:8
    v = dpp_add<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return v;                          // lane 63 holds the total (kept in a VGPR: SGPRs are what limits residency)
}

// Integer partial sums of one tile, as the simulator leaves them for combine_partials_kernel (which adds the tiles of a
// set up and finalises the summary row in f64 with one THREAD per set instead of one lane per flush): PW 64-bit words
//   [0] n_upper | n_lower << 21 | n_missing << 42   [1] sum k   [2] sum k^2   [3] sum k (upper)   [4] sum k^2 (upper)
//   [5] sum z (fixed point, 2^-32)   [6] sum z^2 (2^-24)                     -- models with an external datum only
__host__ __device__ constexpr int partial_words(bool has_zsum) { return has_zsum ? 7 : 5; }

// The fused epilogue of one tile (= one parameter set unless the set is split): coalesced float2 (col0, col1) stores --
// 512 B per wave instruction, every line written whole, once -- + summary reduction.  vset = set * tiles_per_set + tile,
// d = the tile's LDS record.  Column 1 of the models with an external datum is NOT staged in LDS (their LDS footprint,
// and with it the occupancy, equals the basic model's): z1 is recomputed here from the trial's auxiliary stream
// (trial_latent), the explicit boundary is re-read from the caller's array.  (Writing it when the trial is handed out
// or retired instead made every 64-byte sector of the output a partial write, twice: WRITE_SIZE 1.93x the output.)
// SMALL: 16-bit staged results and at most 512 trials per tile, the shape of every launch that matters for throughput.
// Then a lane's share of every sum fits 32 bits (<= 8 trials, k < 2^14), the three counters share one word (10 bits
// each), and the seven cross-lane sums are DPP reductions of 32-bit values (the two sums of squares as 16-bit halves).
template <int MODEL, bool FAST, bool SMALL>
__device__ __forceinline__ void flush_set(ArgsPtr Ap, int lane, long long vset, uint32_t *d, const void *res, uint32_t kbase)
{
    using T = ModelTraits<MODEL>;
    constexpr bool ZSUM = MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT;
    const float tau = __uint_as_float(d[D_TAU]);
    [[maybe_unused]] uint4 dA = {0u, 0u, 0u, 0u}, dB = {0u, 0u, 0u, 0u};
    [[maybe_unused]] uint32_t set_lo = 0u, c3 = 0u;
    [[maybe_unused]] long long acc_z = 0, acc_zz = 0;
    if constexpr (ZSUM) {
        dA = *reinterpret_cast<const uint4 *>(d + D_A); dB = *reinterpret_cast<const uint4 *>(d + D_B);
        c3 = d[D_C3]; set_lo = d[D_SETLO];
    }
    const int N = Ap->n_trials;
    const int TPS = Ap->tiles_per_set;
    const long long set_in_call = TPS == 1 ? vset : vset / TPS;
    const int t0 = TPS == 1 ? 0 : (int)(vset - set_in_call * TPS) * N;      // first trial of this tile
    const int n_here = (Ap->n_total - t0) < N ? (Ap->n_total - t0) : N;           // the last tile may be padded
    uint32_t cnt3 = 0, sk32 = 0, sk2_32 = 0, sk_up32 = 0, sk2_up32 = 0;           // SMALL
    int n_up = 0, n_lo = 0, n_miss = 0;                                           // !SMALL
    unsigned long long sk = 0, sk2 = 0, sk_up = 0, sk2_up = 0;
    float2 *out = Ap->out_trials ? reinterpret_cast<float2 *>(Ap->out_trials) + set_in_call * Ap->n_total + t0 : nullptr;
    for (int j = lane; j < n_here; j += WAVE) {
        uint32_t k, code;                                    // time in units of tscale (step index, or 1/256 step);
        if (SMALL || Ap->res16) {                            // code: 0 timeout, 1 upper, 2 lower, 3 invalid trial
            const uint32_t v = static_cast<const uint16_t *>(res)[j];
            k = v & 0x3fffu; code = v >> 14;
        } else {
            const uint32_t v = static_cast<const uint32_t *>(res)[j];
            k = v & 0x3fffffffu; code = v >> 30;
        }
        const float ch = code == 1u ? 1.0f : (code == 2u ? -1.0f : 0.0f);
        const float rt = __builtin_fmaf((float)k, Ap->tscale, tau);
        float2 o;
        if constexpr (MODEL == NDDM_BASIC_DDM_DC) { o.x = rt; o.y = ch; }
        else if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) { o.x = ch * rt; o.y = 0.5f * (ch + 1.0f); }
        else {
            o.x = ch * rt;
            if constexpr (ZSUM) {
                float latent;
                trial_latent<MODEL, FAST>(dA, dB, set_lo, c3, (uint32_t)(t0 + j), kbase, latent, o.y);
                if (Ap->out_summary) { long long fz, fzz; z_fixed_point(o.y, fz, fzz); acc_z += fz; acc_zz += fzz; }
            } else {
                o.y = out ? Ap->bounds[set_in_call * Ap->n_total + t0 + j] : 0.0f;       // the boundary that was given
            }
        }
        if (code == 3u) o.x = __builtin_nanf("");
        if (out) out[j] = o;
        if (Ap->out_summary) {
            if constexpr (SMALL) {
                const uint32_t kk = k * k;                                           // < 2^28
                const bool up = code == 1u, resp = up || code == 2u;
                cnt3 += up ? 1u : (code == 2u ? (1u << 10) : (1u << 20));
                sk32 += resp ? k : 0u; sk2_32 += resp ? kk : 0u;
                sk_up32 += up ? k : 0u; sk2_up32 += up ? kk : 0u;
            } else {
                const unsigned long long kk = (unsigned long long)k * k;
                if (code == 1u) { n_up++; sk += k; sk2 += kk; sk_up += k; sk2_up += kk; }
                else if (code == 2u) { n_lo++; sk += k; sk2 += kk; }
                else n_miss++;
            }
        }
    }
    if (Ap->out_summary) {
        unsigned long long *q = reinterpret_cast<unsigned long long *>(Ap->partials) + vset * partial_words(ZSUM);
        [[maybe_unused]] unsigned long long *zsum = reinterpret_cast<unsigned long long *>(d + D_ZSUM);
        if constexpr (ZSUM) {          // 64-bit sums: no-return LDS adds by every lane; a wave's LDS operations complete in order
            atomicAdd(zsum, (unsigned long long)acc_z);
            atomicAdd(zsum + 1, (unsigned long long)acc_zz);
        }
        if constexpr (SMALL) {
            cnt3 = wave_sum_dpp(cnt3);
            sk32 = wave_sum_dpp(sk32); sk_up32 = wave_sum_dpp(sk_up32);
            const uint32_t a_hi = wave_sum_dpp(sk2_32 >> 16), a_lo = wave_sum_dpp(sk2_32 & 0xffffu);
            const uint32_t u_hi = wave_sum_dpp(sk2_up32 >> 16), u_lo = wave_sum_dpp(sk2_up32 & 0xffffu);
            if (lane == WAVE - 1) {                          // the DPP reductions leave the totals in the last lane
                q[0] = (unsigned long long)(cnt3 & 1023u) | ((unsigned long long)((cnt3 >> 10) & 1023u) << 21) |
                       ((unsigned long long)(cnt3 >> 20) << 42);
                q[1] = sk32; q[2] = ((unsigned long long)a_hi << 16) + a_lo;
                q[3] = sk_up32; q[4] = ((unsigned long long)u_hi << 16) + u_lo;
                if constexpr (ZSUM) { q[5] = zsum[0]; q[6] = zsum[1]; }
            }
        } else {
            n_up = wave_sum(n_up); n_lo = wave_sum(n_lo); n_miss = wave_sum(n_miss);
            sk = wave_sum(sk); sk2 = wave_sum(sk2); sk_up = wave_sum(sk_up); sk2_up = wave_sum(sk2_up);
            if (lane == 0) {
                q[0] = (unsigned long long)n_up | ((unsigned long long)n_lo << 21) | ((unsigned long long)n_miss << 42);
                q[1] = sk; q[2] = sk2; q[3] = sk_up; q[4] = sk2_up;
                if constexpr (ZSUM) { q[5] = zsum[0]; q[6] = zsum[1]; }
            }
        }
    }
    if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) {
        if (Ap->out_ext && lane == 0 && t0 == 0) {
            const unsigned long long gset = Ap->set_offset + (unsigned long long)set_in_call;
            AuxStream<FAST> aux(kbase, (uint32_t)gset, (uint32_t)(gset >> 32) & 0x0fffffffu, 0xffffffffu);
            const float loc = (Ap->ext_mode == 0) ? Ap->params[set_in_call * T::P + 1] : 1.0f;     // Alpha of the set
            Ap->out_ext[set_in_call] = __builtin_fmaf(Ap->ext_sigma, aux.normal(0), loc);
        }
    }
}

// The evidence is carried CENTRED: w = x - a/2, h = a/2, so that (x > 0) && (x < a) is the single compare |w| < h
// (v_cmp_lt_f32 with the |.| source modifier; false for NaN and for h == 0).
__device__ __forceinline__ bool in_range(float w, float h)
{
    return __builtin_fabsf(w) < h;
}


// MODEL: enum nddm_model.  FAST: Gaussian transform.  SMALL: see below.  CAP4: max_steps is a multiple of 4, so the step cap is tested
// once per Philox block instead of once per step.  BRIDGE: Brownian-bridge boundary correction (between two grid
// points inside (0, a) the path still crosses a boundary with probability exp(-2 d0 d1 / (sigma^2 dt))), which removes
// the O(sqrt(dt)) late-detection bias of plain Euler-Maruyama -- used for alpha_not_scaled, whose reference
// generator is an exact first-passage sampler.
// (Tried and dropped: Philox round keys in VGPRs.  A VOP2 xor that reads an SGPR issues at ~4.2 instead of ~2.3
// cycles on gfx950, but the 20 extra VGPRs cut residency from 7 to 5 waves per SIMD and the net was neutral.)
// SMALL: results staged as 16-bit words and at most 512 trials per tile -- the shape of every launch that matters for
// throughput.  A kernel that carries BOTH flush paths (32-bit DPP sums / 64-bit shuffles) needs 79 SGPRs and 60 VGPRs; the
// SMALL one alone 72 and 44, which is what keeps 8 waves per SIMD resident (the SGPR file limits these kernels).
// PACKED: NDDM_GAUSS_PACKED -- 8 Euler-Maruyama steps per Philox block (polar_pair_packed in nddm_rng.h); CAP4 then means
// "max_steps is a multiple of 8".
template <int MODEL, bool FAST, bool CAP4, bool BRIDGE, bool SMALL, bool PACKED>
__global__ __launch_bounds__(WAVE) void sim_kernel(const SimArgs A)
{
    using T = ModelTraits<MODEL>;
    constexpr int P = T::P;
    extern __shared__ uint32_t lds_raw[];

    const ArgsPtr Ak = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x;
    const int N = A.n_trials;
    const int ring = A.ring, ring_mask = A.ring - 1;

    // LDS carve-up.  Header (128 bytes): the ten Philox round-key pairs [0, 80) (philox4x32_10_path), debug stamps
    // [80, 104), the three round keys that fold into the per-trial constants [112, 124).  Then one ring slot per
    // in-flight parameter set ("tile"): its DV-dword record (hand-out constants, counters, z sums), then the packed results
    if (lane < 10) { lds_raw[2 * lane] = A.k0 + (uint32_t)lane * 0x9E3779B9u; lds_raw[2 * lane + 1] = A.k1 + (uint32_t)lane * 0xBB67AE85u; }
    if (lane == 0) { lds_raw[28] = A.k0 + 2u * PHILOX_W0; lds_raw[29] = A.k1 + 2u * PHILOX_W1; lds_raw[30] = A.k0 + 3u * PHILOX_W0; }
    // LDS byte address of the key table in a VGPR (the low 32 bits of a flat LDS address are the LDS offset); the asm
    // keeps it opaque so that every ds_read in the step loop uses this one register + an immediate offset
    uint32_t kbase;
    {
        const uint32_t off = (uint32_t)(size_t)lds_raw;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kbase) : "s"(off));
    }
    uint32_t *dv = lds_raw + LDS_HEADER_DWORDS;                        // [ring][DV], 16-byte aligned
    // staged results: one 32-bit word per trial, or one 16-bit word when the step cap allows (halves the LDS footprint,
    // which is what lets the 7th and 8th wave per SIMD stay resident at 300 trials per set)
    uint32_t *res = dv + ring * DV;
    uint16_t *res_h = reinterpret_cast<uint16_t *>(res);

    // per-lane trial state
    // w: centred evidence, h: boundary / 2, mu_dt: drift per step -- all in NOISE UNITS (divided by noise_unit(sigma))
    float w = 0.0f, h = 0.0f, mu_dt = 0.0f;
    int k = 0;
    uint32_t jit = 0;
    uint32_t ltrial = 0;     // index within the tile (LDS slot position)
    int tile = 0;            // wave-local sequence number of the set this lane works on
    bool invalid = false;
    // which lanes hold a trial / are still stepping: wave-uniform lane masks kept in SGPRs (a per-lane bool that is
    // balloted costs v_cndmask + v_cmp each time; __builtin_amdgcn_inverse_ballot_w64 turns a mask into exec for free)
    unsigned long long has_m = 0ull, act_m = 0ull;
    PathCtr pc = {0u, 0u, 0u, 0u, 0u};
    PathCtr pcb = {0u, 0u, 0u, 0u, 0u};           // bridge-uniform stream (BRIDGE only)

    // wave-uniform bookkeeping.  Tiles (sets) are opened, handed out and flushed strictly in sequence.
    int tile_open = 0;       // tiles whose parameters are staged in LDS
    int flushed = 0;         // tiles already flushed
    int next_tile = 0, next_trial = 0;            // next unassigned trial of the wave's stream
    // (32-bit, compared modulo 2^32: SGPRs are what limits these kernels' residency)
    int to_retire = N;                            // the oldest tile cannot be complete before this many more trials retire
    int chunk_set = 0, chunk_left = 0;            // current chunk: next set index (in-call), sets left; -1: queue exhausted
    // debug counters live in LDS (SGPRs are scarce): dbg_stamp[2] = refill phases << 32 | step-loop blocks
    unsigned long long *dbg_stamp = reinterpret_cast<unsigned long long *>(lds_raw + 20);   // [0], [1]: start clocks
    if (lane == 0) { dbg_stamp[2] = 0; if (Ak->dbg) { dbg_stamp[0] = __builtin_amdgcn_s_memtime(); dbg_stamp[1] = __builtin_amdgcn_s_memrealtime(); } }

    // The record of the NEXT tile of the current chunk, loaded when the tile before it is opened (lanes < REC hold one
    // dword each): by the time it is consumed the load has long completed, so opening a tile never waits on memory.
    // It is valid whenever the chunk has tiles left (the record of chunk position `chunk_set`).
    uint32_t pre = 0u;

    // open tiles (fetch chunk ids from the global queue, stage the hand-out record of each new tile) while ring slots
    // are free -- but LAZILY: only up to `ahead` tiles beyond the one being handed out, so that a wave never hoards
    // sets its neighbours could be working on (with an eager ring fill, 10,000 sets ended up on 2,500 of the
    // 7,168 waves: 6x slower for mid-size batches)
    auto open_tiles = [&]() {
        const ArgsPtr R = fresh_args(Ak);
        while (tile_open < flushed + ring && tile_open <= next_tile + R->open_ahead) {
            const bool have_pre = chunk_left > 0;                        // mid-chunk: this tile's record was prefetched
            if (chunk_left <= 0) {
                if (chunk_left < 0) break;
                unsigned int c = 0;
                if (lane == 0) c = atomicAdd(R->chunk_counter, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= (unsigned int)R->n_chunks) { chunk_left = -1; break; }
                chunk_set = (int)c * R->sets_per_chunk;
                const long long left = R->B - (long long)chunk_set;
                chunk_left = (int)(left < R->sets_per_chunk ? left : R->sets_per_chunk);
            }
            const int slot = tile_open & ring_mask;
            // queue position -> queue row (position / tiles_per_set) -> its record
            const int TPS = R->tiles_per_set;
            const int prow = TPS == 1 ? chunk_set : chunk_set / TPS;
            const int qt = chunk_set - prow * TPS;                       // tile within the set (0 when not tiled)
            uint32_t rec = pre;
            if (!have_pre) rec = lane < REC ? R->recs[(long long)prow * REC + lane] : 0u;
            if (chunk_left > 1) {                                        // prefetch the next tile's record
                const int nrow = TPS == 1 ? chunk_set + 1 : (chunk_set + 1) / TPS;
                if (nrow != prow) pre = lane < REC ? R->recs[(long long)nrow * REC + lane] : 0u;
                else pre = rec;
            }
            const int sic = __builtin_amdgcn_readlane((int)rec, R_SET);  // the set's row in the caller's arrays
            const int vset = sic * TPS + qt;
            uint32_t *d = dv + slot * DV;
            if (lane < REC) d[lane] = rec;
            if (lane == 0) {
                d[D_VSET] = (uint32_t)vset; d[D_CNT] = 0u;
                if constexpr (MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT) { d[D_ZSUM] = 0u; d[D_ZSUM + 1] = 0u; d[D_ZSUM + 2] = 0u; d[D_ZSUM + 3] = 0u; }
                // everything here is wave-uniform: scalar arithmetic
                const unsigned long long gset = R->set_offset + (unsigned long long)sic;
                const uint32_t s_lo = (uint32_t)gset, s_hi = (uint32_t)(gset >> 32) & 0x0fffffffu;
                PathSet ps;
                ps.init(s_lo, s_hi, R->k0, R->k1);              // stream 0: no tag bits in c2
                d[D_CA] = ps.cA; d[D_CB] = ps.cB; d[D_HP1K] = ps.hP1k; d[D_X1] = ps.X1;
                if constexpr (BRIDGE) {
                    PathSet pb;
                    pb.init(s_lo, s_hi | 0x30000000u, R->k0, R->k1);
                    d[D_BCA] = pb.cA; d[D_BCB] = pb.cB; d[D_BHP1K] = pb.hP1k; d[D_BX1] = pb.X1;
                }
                d[D_SETLO] = s_lo;
                d[D_C3] = s_hi;
                d[D_TBASE] = (uint32_t)(qt * N);
            }
            chunk_set++; chunk_left--; tile_open++;
        }
    };
    open_tiles();
    __syncthreads();

    while (true) {
        // ------------------------------------------------------------ retire finished trials
        const unsigned long long fin_mask0 = has_m & ~act_m;
        if (__builtin_amdgcn_inverse_ballot_w64(fin_mask0)) {
            const uint32_t code = invalid ? 3u : (w >= h ? 1u : (w <= -h ? 2u : 0u));
            uint32_t tfix = (uint32_t)k;
            if constexpr (BRIDGE) tfix = ((uint32_t)k << 8) - ((code == 1u || code == 2u) ? jit : 0u);
            const int slot = tile & ring_mask;
            if (SMALL || fresh_args(Ak)->res16) res_h[(size_t)slot * N + ltrial] = (uint16_t)(tfix | (code << 14));
            else res[(size_t)slot * N + ltrial] = tfix | (code << 30);
            atomicAdd(dv + slot * DV + D_CNT, 1u);
        }
        has_m &= ~fin_mask0;
        to_retire -= (int)__popcll(fin_mask0);
        // ------------------------------------------------------------ flush complete sets, in order (rare path:
        // only entered when enough trials have retired for the oldest tile to possibly be complete)
        if (to_retire <= 0) {
            __syncthreads();
            while (flushed < tile_open) {
                const int slot = flushed & ring_mask;
                const int c = __builtin_amdgcn_readfirstlane((int)dv[slot * DV + D_CNT]);
                if (c != N) break;
                const int set_in_call = __builtin_amdgcn_readfirstlane((int)dv[slot * DV + D_VSET]);
                if constexpr (SMALL)
                    flush_set<MODEL, FAST, true>(fresh_args(Ak), lane, (long long)set_in_call, dv + slot * DV, res_h + (size_t)slot * N, kbase);
                else
                    flush_set<MODEL, FAST, false>(fresh_args(Ak), lane, (long long)set_in_call, dv + slot * DV,
                                                  fresh_args(Ak)->res16 ? static_cast<const void *>(res_h + (size_t)slot * N)
                                                                        : static_cast<const void *>(res + (size_t)slot * N), kbase);
                flushed++;
                to_retire += N;
            }
            __syncthreads();
            open_tiles();
            __syncthreads();
        }
        if (flushed == tile_open && chunk_left < 0) break;
        // ------------------------------------------------------------ hand out new trials
        if (tile_open <= next_tile + fresh_args(Ak)->open_ahead && tile_open < flushed + ring && chunk_left >= 0) {
            open_tiles();
            __syncthreads();
        }
        {
            const unsigned long long want_mask = ~has_m;
            int tr = next_trial + (int)lane_rank(want_mask);
            int tl = next_tile;
            while (tr >= N) { tr -= N; tl++; }
            const unsigned long long ok_mask = want_mask & __builtin_amdgcn_ballot_w64(tl < tile_open);
            next_trial += (int)__popcll(ok_mask);
            while (next_trial >= N) { next_trial -= N; next_tile++; }
            has_m |= ok_mask;
            if (__builtin_amdgcn_inverse_ballot_w64(ok_mask)) {
                const ArgsPtr H = fresh_args(Ak);
                tile = tl;
                ltrial = (uint32_t)tr;
                const int slot = tl & ring_mask;
                const uint4 d0 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_A);
                const uint4 d1 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_CA);
                const uint4 d2 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_SIC);      // set index, tau, TBASE
                const float a0 = __uint_as_float(d0.x), a1 = __uint_as_float(d0.y), a2 = __uint_as_float(d0.z),
                            a3 = __uint_as_float(d0.w);                  // the model's A constants (make_record)
                const uint32_t trial = (uint32_t)tr + d2.z;          // index within the set (keys the random stream)
                [[maybe_unused]] uint32_t set_lo = 0u, c3 = 0u;       // the auxiliary stream's set words
                if constexpr (MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT || MODEL == NDDM_ALPHA_NOT_SCALED) {
                    const uint2 sw = *reinterpret_cast<const uint2 *>(dv + slot * DV + D_C3);
                    c3 = sw.x; set_lo = sw.y;
                }
                invalid = false;
                if constexpr (MODEL == NDDM_BASIC_DDM_DC) {
                    mu_dt = a0; h = a2; w = a3;
                } else if constexpr (MODEL == NDDM_SINGLE_TRIAL) {
                    // A = drift*dt/S, 1/S, std_alpha, mu_alpha;  B = sigma1, gamma, beta
                    const uint4 d3 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_B);
                    float a, z_unused;
                    trial_latent<MODEL, FAST>(d0, d3, set_lo, c3, trial, kbase, a, z_unused);      // per-trial boundary
                    const float hv = 0.5f * a;
                    mu_dt = a0;
                    h = hv * a1;
                    w = (a * __uint_as_float(d3.z) - hv) * a1;
                } else if constexpr (MODEL == NDDM_SINGLE_TRIAL_ALT) {
                    // A = drift, alpha, beta, std_dc;  B = mu_dc, sigma1, gamma
                    const uint4 d3 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_B);
                    float sig_c, z_unused;
                    trial_latent<MODEL, FAST>(d0, d3, set_lo, c3, trial, kbase, sig_c, z_unused);  // per-trial noise scale
                    const float inv_t = 1.0f / noise_unit<FAST>(H->sqrt_dt * sig_c);
                    const float hv = 0.5f * a1;
                    mu_dt = (a0 * H->dt) * inv_t;
                    h = hv * inv_t;
                    w = (a1 * a2 - hv) * inv_t;
                } else if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) {
                    // A = Nu, 1/S, a/(2S), w0;  B = Eta
                    AuxStream<FAST> aux(kbase, set_lo, c3, trial);
                    const float eta = __uint_as_float(dv[slot * DV + D_B]);
                    mu_dt = (__builtin_fmaf(eta, aux.normal(0), a0) * H->dt) * a1;
                    h = a2; w = a3;
                } else if constexpr (MODEL == NDDM_EXPLICIT_BOUNDARY) {
                    // A = drift*dt/S, 1/S, beta
                    const float a = trial < (uint32_t)H->n_total ? H->bounds[(long long)d2.x * H->n_total + trial] : 1.0f;   // padded trial of a last tile
                    invalid = !(a >= 0.0f);            // negative or NaN boundary: the reference raises ValueError
                    const float hv = 0.5f * a;
                    mu_dt = a0;
                    h = invalid ? 0.0f : hv * a1;
                    w = (a * a2 - hv) * a1;
                }
                const uint4 kq = *reinterpret_cast<const uint4 *>(lds_raw + 28);       // kC, kD, kE of PathCtr::init
                pc.init(d1.x, d1.y, d1.z, d1.w, trial, kq.x, kq.y, kq.z);
                if constexpr (BRIDGE) {
                    const uint4 d4 = *reinterpret_cast<const uint4 *>(dv + slot * DV + D_BCA);
                    pcb.init(d4.x, d4.y, d4.z, d4.w, trial, kq.x, kq.y, kq.z);
                }
                k = 0;
                jit = 0;
            }
            // fresh compares over all lanes (an invalid trial has h == 0 and is never in range; lanes without a trial
            // are masked by has_m)
            act_m = __builtin_amdgcn_ballot_w64(in_range(w, h)) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
        }
        // ------------------------------------------------------------ step phase
        // leave the loop for a refill once refill_thresh lanes hold a finished trial, or none is stepping, or after
        // MAX_BLOCKS blocks (so that a few finished lanes never wait long for company; a threshold >= 64 -- the lockstep
        // measurement -- switches that exit off)
        constexpr int MAX_BLOCKS = 16;
        int it = 0;
        for (;; ++it) {
            bool active = __builtin_amdgcn_inverse_ballot_w64(act_m);
            // counter word 0 of the path stream = index of the block's first step (a multiple of NS: a lane only starts
            // a block after taking all NS steps of the previous one), so no shift is needed
            constexpr int NS = PACKED ? 8 : 4;
            const uint32_t blk = (uint32_t)k;
            const u32x4 rb = philox4x32_10_path(blk, pc, kbase);
            // noise of the NS steps as (radius, cos | sin) factors: the step is w = fma(r, t, w) + mu_dt, i.e. a
            // v_fmac_f32 + v_add_f32 (2.3 issue cycles each; a three-address v_fma_f32 costs 3.8)
            float rr[NS], tt[NS];
            if constexpr (PACKED) {
                polar_pair_packed<FAST>(rb.x, rr[0], tt[0], tt[1]);
                polar_pair_packed<FAST>(rb.y, rr[2], tt[2], tt[3]);
                polar_pair_packed<FAST>(rb.z, rr[4], tt[4], tt[5]);
                polar_pair_packed<FAST>(rb.w, rr[6], tt[6], tt[7]);
                rr[1] = rr[0]; rr[3] = rr[2]; rr[5] = rr[4]; rr[7] = rr[6];
            } else {
                polar_pair<FAST>(rb.x, rb.y, rr[0], tt[0], tt[1]);
                polar_pair<FAST>(rb.z, rb.w, rr[2], tt[2], tt[3]);
                rr[1] = rr[0]; rr[3] = rr[2];
            }
            uint32_t ub[4] = {0u, 0u, 0u, 0u};
            if constexpr (BRIDGE) {
                const u32x4 u4 = philox4x32_10_path(blk, pcb, kbase);          // stream 3, same constant folding as the path stream
                ub[0] = u4.x; ub[1] = u4.y; ub[2] = u4.z; ub[3] = u4.w;
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (active) {
                    // keep this a real exec-masked region: selects through SGPR masks (v_cndmask_e64, ~4.2 cycles
                    // each on gfx950) cost more VALU issue than the predicated add / count they would replace
                    asm volatile("" ::: "memory");
                    float w1 = __builtin_fmaf(rr[j], tt[j], w) + mu_dt;
                    if constexpr (BRIDGE) {
                        if (in_range(w1, h)) {
                            // P(crossed) = exp(-2 d0 d1 / sigma^2 dt): in noise units the coefficient is a constant
                            // (exact: -2 with e^x; fast: -2 * 2 ln 2 * log2 e = -4 with v_exp_f32's 2^x)
                            constexpr float cb = FAST ? -4.0f : -2.0f;
                            const float eu = cb * ((h - w) * (h - w1));      // distances to the upper boundary
                            const float el = cb * ((h + w) * (h + w1));      // ... and to the lower one
                            float pu, pl;
                            if constexpr (FAST) { pu = __builtin_amdgcn_exp2f(eu); pl = __builtin_amdgcn_exp2f(el); }
                            else { pu = exact_expf_neg(eu); pl = exact_expf_neg(el); }
                            const float uu = (float)(ub[j & 3] >> 8) * 5.9604644775390625e-08f;      // [0, 1), 24 bits
                            if (uu < pu) w1 = h;
                            else if (uu >= 1.0f - pl) w1 = -h;
                        }
                        jit = ub[j & 3] & 0xffu;
                    }
                    w = w1;
                    k++;
                    if (j < NS - 1) {
                        if constexpr (CAP4) active = in_range(w, h);
                        else active = in_range(w, h) && (k < A.max_k);
                    }
                }
            }
            // fresh compares for every lane, combined as SGPR masks: a ballot of a compare is just its SGPR result, a
            // ballot of the loop-carried flag (or of an && of two compares) is rebuilt through v_cndmask + v_cmp
            act_m = __builtin_amdgcn_ballot_w64(in_range(w, h)) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
            if (act_m == 0ull || __popcll(has_m & ~act_m) >= A.refill_thresh) break;
            if (it >= MAX_BLOCKS - 1 && A.refill_thresh < WAVE) break;
        }
        // one refill phase of `it + 1` blocks: a no-return 64-bit LDS add
        if (lane == 0) atomicAdd(dbg_stamp + 2, (1ull << 32) | (unsigned long long)(it + 1));
    }
    // the queue resets itself: every wave has finished pulling chunks before it counts itself out (its pulls returned
    // values it waited for), so when the last one arrives nobody will touch the words again in this launch.  No memset
    // per launch, and a captured launch is kernels only.
    if (lane == 0) {
        unsigned int *const q = fresh_args(Ak)->chunk_counter;
        const unsigned int left = atomicAdd(q + 1, 1u);
        if (left == gridDim.x - 1u) { atomicExch(q, 0u); atomicExch(q + 1, 0u); }
    }
    unsigned long long *const dbg = fresh_args(Ak)->dbg;
    if (dbg && lane == 0) {
        atomicAdd(dbg + 0, dbg_stamp[2] & 0xffffffffull);
        atomicAdd(dbg + 1, dbg_stamp[2] >> 32);
        atomicAdd(dbg + 2, (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_stamp[0]));
        atomicAdd(dbg + 3, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dbg_stamp[1]));
        atomicAdd(dbg + 4, 1ull);
    }
}

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

__global__ void order_scatter_kernel(int model, int fast, const float *params, int P, int B, float dt, float sqrt_dt, int max_k,
                                     int *ws, uint32_t *recs)
{
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
            make_record(model, fast != 0, params + (long long)i * P, dt, sqrt_dt, i, recs + (long long)q * REC);
        }
        __syncthreads();
    }
}

// launches too small to be worth sorting: records in the given order
__global__ void prep_kernel(int model, int fast, const float *params, int P, int B, float dt, float sqrt_dt, uint32_t *recs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) make_record(model, fast != 0, params + (long long)i * P, dt, sqrt_dt, i, recs + (long long)i * REC);
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
                             float gamma, float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    PriorStream s(k0, k1, set_offset + (unsigned long long)i);
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

// ------------------------------------------------------------------------------------------------
// host side
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

static int round_up_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---- process-wide state, all of it behind one mutex -------------------------------------------------------------------
// Entry points are re-entrant: each call takes a SNAPSHOT of the developer knobs at entry and owns the device memory it is
// handed (a LaunchSlot) until the work it enqueued has completed.
struct Tuning { int sets_per_chunk, ring, refill_thresh, max_blocks, grid_waves, tile_trials, no_order; };
static std::mutex g_mu;
static Tuning g_tuning = {0, 0, 0, 0, 0, 0, 0};   // 0 = automatic (nddm_set_tuning overrides; benchmarking aid)
static unsigned long long *g_dbg = nullptr;       // nddm_set_debug_counters (profiling aid)

constexpr int MAX_DEVICES = 64;
struct DeviceInfo { int cus = 0; double clock_hz = 0.0; };
static DeviceInfo g_dev[MAX_DEVICES];

// compute units and clock of a device (cached): every sizing rule below is written in terms of these, not of MI355X's
// 256 CUs / 1024 SIMDs / 2.4 GHz
static bool device_info(int dev, DeviceInfo *out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_dev[dev].cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        g_dev[dev].cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        g_dev[dev].clock_hz = prop.clockRate > 0 ? (double)prop.clockRate * 1e3 : 2.4e9;      // clockRate is in kHz
    }
    *out = g_dev[dev];
    return true;
}

// Device memory a launch needs besides the caller's buffers: the work queue (chunk counter + exit counter; the kernel
// leaves both zero) and up to SLOT_SCRATCH bytes of scratch (longest-first order, integer partial sums).  A slot belongs to
// ONE launch at a time: it is handed out again only to the stream that used it last (launches on a stream are ordered)
// or once the event recorded behind its last launch has completed -- never by launch count.  Launches under stream
// capture do not use slots at all (see graph_alloc).
constexpr size_t SLOT_QUEUE_BYTES = 256, SLOT_SCRATCH = 1u << 20;
constexpr int MAX_SLOTS = 256;
struct LaunchSlot {
    char *base = nullptr;          // [SLOT_QUEUE_BYTES queue words | SLOT_SCRATCH scratch]
    hipEvent_t done = nullptr;     // recorded on `stream` after the slot's last launch
    hipStream_t stream = nullptr;
    bool busy = false;             // between acquire and release (the enqueueing itself)
    bool fresh = true;             // queue words not yet zeroed
};
static LaunchSlot *g_slots[MAX_DEVICES][MAX_SLOTS];
static int g_nslots[MAX_DEVICES];
static int g_slot_limit = MAX_SLOTS;       // nddm_debug_set_slot_limit: lets a test reach the "all slots in flight" path

// hipMalloc with the calling thread's capture mode relaxed: another thread (or this one) may be capturing a graph, and a
// plain hipMalloc is refused then
static hipError_t malloc_relaxed(void **p, size_t bytes)
{
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    const hipError_t e = hipMalloc(p, bytes);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    return e;
}

static LaunchSlot *acquire_slot_once(int dev, hipStream_t st, hipError_t *err);

// every slot can be momentarily busy (other threads in the middle of enqueueing): wait for one of them to finish
static LaunchSlot *acquire_slot(int dev, hipStream_t st, hipError_t *err)
{
    for (int spin = 0; spin < (1 << 22); ++spin) {
        LaunchSlot *s = acquire_slot_once(dev, st, err);
        if (s || *err != hipErrorNotReady) return s;
        std::this_thread::yield();
    }
    return nullptr;
}

static LaunchSlot *acquire_slot_once(int dev, hipStream_t st, hipError_t *err)
{
    *err = hipSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    LaunchSlot *pick = nullptr;
    const int n_use = g_nslots[dev] < g_slot_limit ? g_nslots[dev] : g_slot_limit;      // slots in use (all, unless a test caps them)
    // (a) the slot this stream used last: stream order makes it safe without a query.  hipStreamPerThread is one handle
    //     for a different stream in every thread, so it never qualifies.
    if (st != hipStreamPerThread)
        for (int i = 0; i < n_use && !pick; ++i)
            if (!g_slots[dev][i]->busy && g_slots[dev][i]->stream == st && !g_slots[dev][i]->fresh) pick = g_slots[dev][i];
    // (b) any slot whose last launch has completed
    for (int i = 0; i < n_use && !pick; ++i) {
        LaunchSlot *s = g_slots[dev][i];
        if (!s->busy && (s->fresh || hipEventQuery(s->done) == hipSuccess)) pick = s;
    }
    // (c) a new slot
    if (!pick && g_nslots[dev] < g_slot_limit) {
        LaunchSlot *s = new LaunchSlot();
        hipError_t e = malloc_relaxed(reinterpret_cast<void **>(&s->base), SLOT_QUEUE_BYTES + SLOT_SCRATCH);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
        if (e != hipSuccess) { if (s->base) (void)hipFree(s->base); delete s; *err = e; return nullptr; }
        if (g_nslots[dev] == 0) {
            // keep freed stream-ordered blocks cached instead of handing them back to the OS at every synchronisation
            hipMemPool_t mp;
            if (hipDeviceGetDefaultMemPool(&mp, dev) == hipSuccess) {
                uint64_t keep = UINT64_MAX;
                (void)hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep);
            }
        }
        g_slots[dev][g_nslots[dev]++] = s;
        pick = s;
    }
    // (d) every slot is in flight on other streams: queue behind the one used longest ago
    if (!pick) {
        for (int i = 0; i < n_use && !pick; ++i)
            if (!g_slots[dev][i]->busy) pick = g_slots[dev][i];
        if (!pick) { *err = hipErrorNotReady; return nullptr; }
        const hipError_t e = hipStreamWaitEvent(st, pick->done, 0);
        if (e != hipSuccess) { *err = e; return nullptr; }
        // rotate it to the back so that the next starved launch waits on a different slot
        int at = 0;
        while (g_slots[dev][at] != pick) ++at;
        for (int i = at; i + 1 < n_use; ++i) g_slots[dev][i] = g_slots[dev][i + 1];
        g_slots[dev][n_use - 1] = pick;
    }
    pick->busy = true;
    pick->stream = st;
    return pick;
}

static void release_slot(LaunchSlot *s, hipStream_t st)
{
    (void)hipEventRecord(s->done, st);         // outside the lock: the slot is still marked busy
    std::lock_guard<std::mutex> lock(g_mu);
    s->fresh = false;
    s->busy = false;
}

// Memory of launches captured into a hipGraph: the graph replays with the pointers it captured, at times the library
// cannot see, so such a launch gets an allocation of its own that nothing else ever uses (queue words + ALL of its
// scratch).  It lives until nddm_release_graph_memory().
constexpr size_t GRAPH_SCRATCH_MAX = 64u << 20;
struct GraphAlloc { void *p; GraphAlloc *next; };
static GraphAlloc *g_graph_allocs[MAX_DEVICES];

static char *graph_alloc(int dev, size_t bytes, hipError_t *err)
{
    void *p = nullptr;
    *err = malloc_relaxed(&p, bytes);
    if (*err != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    g_graph_allocs[dev] = new GraphAlloc{p, g_graph_allocs[dev]};
    return static_cast<char *>(p);
}

// zeroes the 64 counting-sort counters / the queue words of a captured launch (a kernel, not hipMemsetAsync: memset NODES
// of a captured launch were observed not to take effect on graph replay with ROCm 7.2)
__global__ void zero_words_kernel(unsigned int *ws) { ws[threadIdx.x] = 0u; }

// resident waves of a kernel instantiation on the current device (persistent grid size)
template <typename K>
static int resident_waves(K kernel, size_t lds_bytes, int cus)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = 8;
    // The occupancy query over-counts on gfx950, for two reasons found with a residency micro-kernel and confirmed by
    // the in-kernel wave-lifetime counters:
    //  * SGPRs: 800 per SIMD, and a wave is charged its SGPRs + 22 (VCC etc. + the trap handler's 16) rounded up to 16:
    //    highest SGPR s70 -> 8 waves per SIMD, s86 -> 7, s94 and up -> 6 (tools/resource_table.py lists every
    //    instantiation); the runtime does not report SGPR counts, so the hardware maximum of 8 is assumed -- a grid slightly
    //    larger than what is resident only adds waves that start late and find the queue empty (measured neutral).
    //  * LDS is allocated in 1280-byte granules (5.3 KB -> 6.4 KB -> 25 workgroups per CU, not 30).
    if (per_cu > 32) per_cu = 32;
    {
        const size_t granules = (lds_bytes + 1279) / 1280;
        const int by_lds = granules ? (int)((160u * 1024u) / (granules * 1280)) : 32;
        if (per_cu > by_lds) per_cu = by_lds;
    }
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel)) == hipSuccess && fa.numRegs > 0) {
        int per_simd = 512 / (((fa.numRegs + 7) / 8) * 8);
        if (per_simd < 1) per_simd = 1;
        if (per_cu > 4 * per_simd) per_cu = 4 * per_simd;
    }
    if (per_cu < 1) per_cu = 1;
    return cus * per_cu;
}

template <int MODEL, bool BRIDGE>
static int launch_model(const SimArgs &A, bool fast, bool packed, size_t lds_bytes, int n_chunks, int cus, int grid_override,
                        bool grid_forced, hipStream_t st)
{
    const bool cap4 = (A.max_k % (packed ? 8 : 4)) == 0;     // the step cap falls on a block boundary
    const dim3 block(WAVE);
#define NDDM_LAUNCH(KERNEL)                                                                    \
    do {                                                                                       \
        int waves = resident_waves(KERNEL, lds_bytes, cus);                                    \
        if (grid_override > 0 && (grid_override < waves || grid_forced)) waves = grid_override;\
        if (waves > n_chunks) waves = n_chunks;                                                \
        hipLaunchKernelGGL(KERNEL, dim3(waves), block, lds_bytes, st, A);                      \
    } while (0)
    const bool small = A.res16 == 2;
    if constexpr (!BRIDGE) {
        if (packed) {               // NDDM_GAUSS_PACKED (the host has checked small)
            if (fast && cap4)       NDDM_LAUNCH((sim_kernel<MODEL, true, true, false, true, true>));
            else if (fast)          NDDM_LAUNCH((sim_kernel<MODEL, true, false, false, true, true>));
            else if (cap4)          NDDM_LAUNCH((sim_kernel<MODEL, false, true, false, true, true>));
            else                    NDDM_LAUNCH((sim_kernel<MODEL, false, false, false, true, true>));
        } else if (small) {
            if (fast && cap4)       NDDM_LAUNCH((sim_kernel<MODEL, true, true, false, true, false>));
            else if (fast)          NDDM_LAUNCH((sim_kernel<MODEL, true, false, false, true, false>));
            else if (cap4)          NDDM_LAUNCH((sim_kernel<MODEL, false, true, false, true, false>));
            else                    NDDM_LAUNCH((sim_kernel<MODEL, false, false, false, true, false>));
        }
    }
    if (BRIDGE || (!small && !packed)) {
        if (fast && cap4)       NDDM_LAUNCH((sim_kernel<MODEL, true, true, BRIDGE, false, false>));
        else if (fast)          NDDM_LAUNCH((sim_kernel<MODEL, true, false, BRIDGE, false, false>));
        else if (cap4)          NDDM_LAUNCH((sim_kernel<MODEL, false, true, BRIDGE, false, false>));
        else                    NDDM_LAUNCH((sim_kernel<MODEL, false, false, BRIDGE, false, false>));
    }
#undef NDDM_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NDDM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

static int simulate(int model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                    int32_t max_steps, uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma,
                    int32_t ext_mode, float *out_trials, float *out_summary, float *out_ext, void *stream)
{
    g_err[0] = 0;
    int P;
    bool has_zsum;
    switch (model) {
    case NDDM_BASIC_DDM_DC: P = 5; has_zsum = false; break;
    case NDDM_SINGLE_TRIAL: P = 8; has_zsum = true; break;
    case NDDM_SINGLE_TRIAL_ALT: P = 8; has_zsum = true; break;
    case NDDM_ALPHA_NOT_SCALED: P = 6; has_zsum = false; break;
    case NDDM_EXPLICIT_BOUNDARY: P = 4; has_zsum = false; break;
    default: return fail(NDDM_ERR_PARAM, "unknown model%s");
    }
    if (B < 0 || n_trials <= 0 || max_steps < 0) return fail(NDDM_ERR_SHAPE, "B < 0, n_trials <= 0 or max_steps < 0%s");
    if (max_steps >= (1 << 30)) return fail(NDDM_ERR_SHAPE, "max_steps must be < 2^30%s");
    if (set_offset >= (1ull << 60) || set_offset + (uint64_t)B > (1ull << 60))
        return fail(NDDM_ERR_SHAPE, "set_offset + B must be <= 2^60 (the random stream is keyed by 60 bits of the set index)%s");
    if (!(dt > 0.0f) || !isfinite(dt)) return fail(NDDM_ERR_PARAM, "dt must be finite and > 0%s");
    if (flags > 7u) return fail(NDDM_ERR_PARAM, "unknown flags%s");
    const bool bridge = (flags & NDDM_BRIDGE) != 0;
    const bool packed = (flags & NDDM_GAUSS_PACKED) != 0;
    if (packed && (bridge || max_steps >= 16384))
        return fail(NDDM_ERR_PARAM, "NDDM_GAUSS_PACKED needs max_steps < 2^14 and cannot be combined with NDDM_BRIDGE%s");
    if (bridge && model != NDDM_ALPHA_NOT_SCALED)
        return fail(NDDM_ERR_PARAM, "NDDM_BRIDGE is only available for NDDM_ALPHA_NOT_SCALED%s");
    if (bridge && max_steps >= (1 << 22)) return fail(NDDM_ERR_SHAPE, "max_steps must be < 2^22 with NDDM_BRIDGE%s");
    if (B == 0) return NDDM_OK;
    if (!params) return fail(NDDM_ERR_NULL, "params is NULL%s");
    if (model == NDDM_EXPLICIT_BOUNDARY && !bounds) return fail(NDDM_ERR_NULL, "bounds is NULL%s");
    if (!out_trials && !out_summary && !out_ext) return fail(NDDM_ERR_NULL, "no output buffer given%s");
    if (B * (((long long)n_trials + 511) / 512) >= (1ll << 31))
        return fail(NDDM_ERR_SHAPE, "B * ceil(n_trials / 512) must be < 2^31 per launch%s");

    // snapshot of the developer knobs, and the device this call runs on
    Tuning tun;
    unsigned long long *dbg;
    { std::lock_guard<std::mutex> lock(g_mu); tun = g_tuning; dbg = g_dbg; }
    int dev = 0;
    {
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return fail(NDDM_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    }
    DeviceInfo di;
    if (!device_info(dev, &di)) return fail(NDDM_ERR_HIP, "hipGetDeviceProperties failed%s");
    const int simds = 4 * di.cus;
    const long long waves7 = 7ll * simds;      // a persistent grid at 7 waves per SIMD (7168 on MI355X): the sizing rules'
    const long long waves6 = 6ll * simds;      // "resident waves"; the launch itself asks resident_waves() per kernel

    SimArgs A;
    memset(&A, 0, sizeof A);
    A.params = params; A.bounds = bounds; A.out_trials = out_trials; A.out_summary = out_summary; A.out_ext = out_ext;
    A.B = B; A.set_offset = set_offset; A.n_trials = n_trials; A.max_k = max_steps; A.dt = dt; A.sqrt_dt = sqrtf(dt);
    A.tscale = bridge ? dt * 0.00390625f : dt;
    A.k0 = (uint32_t)seed; A.k1 = (uint32_t)(seed >> 32);
    A.ext_sigma = ext_sigma; A.ext_mode = ext_mode;
    A.dbg = dbg;

    // tiling: a set whose trials do not fit the LDS ring comfortably is split into equal tiles ("virtual sets");
    // the random stream is keyed by the trial's index within the SET, so results do not depend on the tiling
    // staged result per trial: a packed (time | choice) word of 4 bytes, or 2 when the step index fits 14 bits (no bridge:
    // its time unit is 1/256 step); z columns go straight to HBM
    const bool res16 = (flags & NDDM_BRIDGE) == 0 && max_steps < 16384;
    const size_t per_trial = res16 ? 2 : 4;
    // small and mid-size launches are bound by latency / by the slowest set (a set of 300 slow trials keeps one wave
    // busy for milliseconds): cut the sets into tiles of as few as 64 trials so that there are ~8 tiles per resident
    // wave to balance; from ~30M trials on, one tile of up to 512 trials per set is the efficient shape
    const long long total_trials = B * (long long)n_trials;
    // Grid: as many waves as stay resident -- unless the launch is small.  Then fewer, busier waves win: a wave that gets
    // ~300+ wave-blocks of work (64 lanes x 4 steps each) keeps its lanes balanced and amortises its start-up, and the
    // launch's tail (its longest trial) runs on SIMDs that are less crowded.  Measured (dt=.01, 10k-30k sets x 300 trials,
    // or 100k x 60): 1024-3072 waves are 25-45 % faster than 8192; from ~2.5M wave-blocks on the full grid is best; never
    // below one wave per SIMD.
    long long plan_waves = 8ll * simds;
    if (tun.grid_waves > 0) plan_waves = tun.grid_waves;
    else {
        double est_steps = 0.25 / (double)dt;                            // E[steps] under the reference priors
        if (est_steps > (double)max_steps) est_steps = (double)max_steps;
        if (est_steps < 1.0) est_steps = 1.0;
        const double want = (double)total_trials * est_steps / 256.0 / 300.0;
        if (want < (double)plan_waves) plan_waves = want < (double)simds ? simds : (long long)want;
    }
    const long long waves_for_tiles = plan_waves < waves7 ? plan_waves : waves7;
    int tile_cap = 64;
    while (tile_cap < 512 && (long long)tile_cap * waves_for_tiles * 8 < total_trials) tile_cap <<= 1;
    int tiles = tun.tile_trials > 0 ? (n_trials + tun.tile_trials - 1) / tun.tile_trials
                                    : (n_trials <= tile_cap ? 1 : (n_trials + tile_cap - 1) / tile_cap);
    const int tile_n = (n_trials + tiles - 1) / tiles;
    tiles = (n_trials + tile_n - 1) / tile_n;
    const long long vB = B * (long long)tiles;
    if (vB >= (1ll << 31)) return fail(NDDM_ERR_SHAPE, "B * ceil(n_trials / 512) must be < 2^31 per launch%s");
    if ((long long)tile_n * tiles >= (1ll << 30) || n_trials >= (1 << 30))
        return fail(NDDM_ERR_SHAPE, "n_trials must be < 2^30%s");
    A.n_trials = tile_n; A.n_total = n_trials; A.tiles_per_set = tiles; A.B = vB;
    // chunk = the unit a wave pulls from the global queue: small (tail of the whole launch <= one chunk), but large
    // enough that the queue's atomic counter is touched rarely (~ once per 1200+ trials per wave)
    int spc = tun.sets_per_chunk;
    if (!spc) {
        spc = (1200 + tile_n - 1) / tile_n;
        if (spc < 1) spc = 1;
        if (spc > 64) spc = 64;
        while (spc > 1 && vB / spc < 32 * waves_for_tiles) spc >>= 1;   // keep >= ~32 chunks per wave: the launch's tail is one chunk
        // ... but the queue is ONE atomic word: same-address atomics retire at ~85 M/s on MI355X (measured: 1M chunks
        // take 11.6 ms whatever the work; contention already costs 15 % at 55 M/s), so short-trial workloads (dt = .01,
        // few trials per set) must pull less often: at most ~40 M chunks/s over the shortest time the launch can take --
        // its stepping (E[steps] ~ min(cap, 0.25 / dt) under the reference priors, 265 SIMD cycles per 256 lane-steps)
        // plus one trial that runs to the cap on a full SIMD -- and never fewer than ~8 chunks per wave.  Mid-size
        // launches, whose tail is one chunk, stay below the limit with their fine chunks.
        double est_steps = 0.25 / (double)dt;
        if (est_steps > (double)max_steps) est_steps = (double)max_steps;
        if (est_steps < 1.0) est_steps = 1.0;
        const double t_est = (double)vB * ((double)tile_n * est_steps / 256.0) * 265.0 / ((double)simds * di.clock_hz)
                             + (double)max_steps * 0.25 * 265.0 * 7.0 / di.clock_hz;
        double max_chunks = t_est * 4.0e7;
        if (max_chunks < 8.0 * (double)waves6) max_chunks = 8.0 * (double)waves6;
        if ((double)vB / spc > max_chunks) {
            spc = (int)((double)vB / max_chunks) + 1;
            if (spc > 64) spc = 64;
        }
    }
    // geometry: ring slots.  Sets are flushed in order, so a straggler trial in the oldest set must not stall the lanes
    // that are ahead of it: the wave's window (ring x tile) should span >= ~480 trials (7-8 per lane; measured: 400 costs
    // 7 % of lane efficiency, more than 1024 buys nothing).  But the LDS footprint must leave 7 waves per SIMD resident
    // (28 single-wave workgroups per CU; LDS is allocated in 1280-byte granules, so <= 5120 B each): every wave counts
    // (+3 % from 6 to 7, -6 % at 5, -17 % at 4), which costs more than a short window.
    const auto lds_of = [&](int r) {
        return (size_t)LDS_HEADER_DWORDS * 4 + (size_t)r * (DV * 4) + (size_t)r * tile_n * per_trial;
    };
    int ring = tun.ring;
    if (!ring) {
        ring = round_up_pow2((480 + tile_n - 1) / tile_n);
        if (ring < 4) ring = 4;
        if (ring > 64) ring = 64;
        while (ring > 2 && lds_of(ring) > 5120) ring >>= 1;     // 4 LDS granules of 1280 B: 32 workgroups per CU fit
    }
    if (ring < 2) ring = 2;
    if (ring > 64) ring = 64;
    if (lds_of(ring) > 60 * 1024)
        return fail(NDDM_ERR_SHAPE, "tile too large for the LDS ring (tuning override?)%s");
    A.sets_per_chunk = spc; A.ring = ring;
    // refill threshold: a refill costs ~170 issue cycles whatever the number of lanes it serves, a waiting lane wastes
    // its share of every block; with lambda completions per block the optimum is ~sqrt(61 lambda) finished lanes:
    // 8 when trials last ~64 blocks (dt=.001, cap 4000), ~16-24 when they last ~7 (the reference default dt=.01, cap
    // 400).  The cap is the only hint the host has about trial length.
    // refill when this many lanes hold a finished trial: 8, or 16 where a refill is dearer relative to the stepping
    // between two refills (short trials; models whose hand-out draws per-trial auxiliary normals) -- measured +2..5 %
    const bool aux_handout = model == NDDM_SINGLE_TRIAL || model == NDDM_SINGLE_TRIAL_ALT || model == NDDM_ALPHA_NOT_SCALED;
    A.res16 = res16 ? (tile_n <= 512 ? 2 : 1) : 0;
    if (packed && A.res16 != 2) return fail(NDDM_ERR_PARAM, "NDDM_GAUSS_PACKED needs tiles of <= 512 trials (tuning override?)%s");
    A.refill_thresh = tun.refill_thresh ? tun.refill_thresh : ((max_steps <= 1000 || aux_handout) ? 16 : 8);
    A.max_blocks = tun.max_blocks ? tun.max_blocks : 16;
    const long long n_chunks = (vB + spc - 1) / spc;
    A.n_chunks = (int)n_chunks;
    A.open_ahead = vB >= 4 * waves7 ? 1 : 0;
    const size_t lds = lds_of(ring);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool fast = (flags & NDDM_GAUSS_FAST) != 0;
    // ---- per-launch device memory -----------------------------------------------------------------------------------
    //  * the work queue (chunk counter + exit counter, self-resetting: no memset per launch);
    //  * the longest-first order (64 counters + B indices + the gathered parameter rows) for launches of >= 2048 sets;
    //  * integer partial sums [vB, partial_words()] whenever summaries are requested: the f64 finalisation is done by
    //    combine_partials_kernel with one thread per set instead of by lane 0 of every flush (which also kept ~8 more
    //    VGPRs alive in the simulator kernel).
    // Queue words and up to 1 MB of scratch come from a LaunchSlot (see above: reused on stream order or on event
    // completion), so the small, fixed-shape launches of a training loop allocate nothing.  Larger scratch is
    // stream-ordered (hipMallocAsync / hipFreeAsync).  A launch under stream capture gets a dedicated allocation for
    // everything (graph_alloc) and zeroes its queue words with a captured kernel; it is refused above GRAPH_SCRATCH_MAX.
    const bool want_order = B >= 2048 && tun.no_order == 0;
    const bool want_partials = out_summary != nullptr;
    const int pw = partial_words(has_zsum);
    const size_t order_bytes = (((want_order ? 64 : 0) + (size_t)B * REC) * sizeof(uint32_t) + 255) & ~(size_t)255;   // [counters] | records
    const size_t partial_bytes = want_partials ? (size_t)vB * pw * sizeof(unsigned long long) : 0;
    const size_t scratch_bytes = order_bytes + partial_bytes;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    char *scratch = nullptr;
    bool scratch_async = false;
    LaunchSlot *slot = nullptr;
    if (capturing) {
        if (scratch_bytes > GRAPH_SCRATCH_MAX)
            return fail(NDDM_ERR_PARAM, "this launch needs more than 64 MB of scratch and cannot be captured in a hipGraph%s");
        hipError_t e;
        char *mem = graph_alloc(dev, SLOT_QUEUE_BYTES + scratch_bytes, &e);
        if (!mem) return fail(NDDM_ERR_HIP, "hipMalloc(captured launch): %s", hipGetErrorString(e));
        A.chunk_counter = reinterpret_cast<unsigned int *>(mem);
        scratch = mem + SLOT_QUEUE_BYTES;
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, A.chunk_counter);
    } else {
        hipError_t e;
        slot = acquire_slot(dev, st, &e);
        if (!slot) return fail(NDDM_ERR_HIP, "launch slot: %s", hipGetErrorString(e));
        A.chunk_counter = reinterpret_cast<unsigned int *>(slot->base);
        if (slot->fresh) {
            e = hipMemsetAsync(slot->base, 0, SLOT_QUEUE_BYTES, st);
            if (e != hipSuccess) { release_slot(slot, st); return fail(NDDM_ERR_HIP, "hipMemsetAsync(queue words): %s", hipGetErrorString(e)); }
        }
        if (scratch_bytes && scratch_bytes <= SLOT_SCRATCH) scratch = slot->base + SLOT_QUEUE_BYTES;
        else if (scratch_bytes) {
            e = hipMallocAsync(reinterpret_cast<void **>(&scratch), scratch_bytes, st);
            if (e != hipSuccess) { release_slot(slot, st); return fail(NDDM_ERR_HIP, "hipMallocAsync(scratch): %s", hipGetErrorString(e)); }
            scratch_async = true;
        }
    }
    int rc = NDDM_OK;
    A.partials = want_partials ? reinterpret_cast<unsigned long long *>(scratch + order_bytes) : nullptr;
    {
        // pre-pass: one hand-out record per set (make_record), in processing order -- longest expected trials first when
        // the launch is large enough for the order to matter, else as given
        const int threads = 256;
        long long blocks = (B + threads - 1) / threads;
        hipError_t e = hipSuccess;
        if (want_order) {
            int *order_ws = reinterpret_cast<int *>(scratch);
            uint32_t *recs = reinterpret_cast<uint32_t *>(order_ws + 64);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned int *>(order_ws));
            hipLaunchKernelGGL(order_hist_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, params, P, (int)B, dt,
                               (int)max_steps, order_ws);
            hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, fast ? 1 : 0, params, P,
                               (int)B, dt, A.sqrt_dt, (int)max_steps, order_ws, recs);
            A.recs = recs;
        } else {
            uint32_t *recs = reinterpret_cast<uint32_t *>(scratch);
            hipLaunchKernelGGL(prep_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, fast ? 1 : 0, params, P, (int)B, dt,
                               A.sqrt_dt, recs);
            A.recs = recs;
        }
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(NDDM_ERR_HIP, "pre-pass (hand-out records): %s", hipGetErrorString(e));
    }
    if (rc == NDDM_OK) {
        const int gw = plan_waves < 8ll * simds || tun.grid_waves > 0 ? (int)plan_waves : 0;     // 0 = what is resident
        switch (model) {
        case NDDM_BASIC_DDM_DC: rc = launch_model<NDDM_BASIC_DDM_DC, false>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_SINGLE_TRIAL: rc = launch_model<NDDM_SINGLE_TRIAL, false>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_SINGLE_TRIAL_ALT: rc = launch_model<NDDM_SINGLE_TRIAL_ALT, false>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_ALPHA_NOT_SCALED:
            rc = bridge ? launch_model<NDDM_ALPHA_NOT_SCALED, true>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st)
                        : launch_model<NDDM_ALPHA_NOT_SCALED, false>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st);
            break;
        default: rc = launch_model<NDDM_EXPLICIT_BOUNDARY, false>(A, fast, packed, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        }
    }
    if (rc == NDDM_OK && A.partials) {
        const int tau_idx = model == NDDM_EXPLICIT_BOUNDARY ? 2 : 3;
        const int threads = 256;
        hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned)((B + threads - 1) / threads)), dim3(threads), 0, st,
                           A.partials, pw, params, P, tau_idx, (long long)B, tiles, n_trials, A.tscale, out_summary);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(NDDM_ERR_HIP, "combine kernel launch failed: %s", hipGetErrorString(e));
    }
    if (scratch_async) (void)hipFreeAsync(scratch, st);
    if (slot) release_slot(slot, st);
    return rc;
}

}  // namespace nddm

// ================================================================================================
extern "C" {

int nddm_abi_version(void) { return NDDM_ABI_VERSION; }
const char *nddm_last_error(void) { return nddm::g_err; }
int nddm_summary_k(void) { return NDDM_SUMMARY_K; }

int nddm_model_nparams(int model)
{
    switch (model) {
    case NDDM_BASIC_DDM_DC: return 5;
    case NDDM_SINGLE_TRIAL: return 8;
    case NDDM_SINGLE_TRIAL_ALT: return 8;
    case NDDM_ALPHA_NOT_SCALED: return 6;
    case NDDM_EXPLICIT_BOUNDARY: return 4;
    }
    return -1;
}

int nddm_device_count(int *count)
{
    if (!count) return nddm::fail(NDDM_ERR_NULL, "count is NULL%s");
    const hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return nddm::fail(NDDM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return NDDM_OK;
}

int nddm_set_device(int device)
{
    const hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return nddm::fail(NDDM_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    return NDDM_OK;
}

/* benchmarking aid (not part of the drop-in surface): 0 = automatic */
int nddm_set_tuning(int sets_per_chunk, int ring, int refill_thresh, int max_blocks, int grid_waves, int tile_trials)
{
    if (ring && (ring & (ring - 1))) return nddm::fail(NDDM_ERR_PARAM, "ring must be a power of two%s");
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    const int no_order = nddm::g_tuning.no_order;
    nddm::g_tuning = {sets_per_chunk, ring, refill_thresh, max_blocks, grid_waves, tile_trials, no_order};
    return NDDM_OK;
}

/* benchmarking aid: 1 = process the sets in the given order (no longest-first sort) */
int nddm_set_ordering(int enabled)
{
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_tuning.no_order = enabled ? 0 : 1;
    return NDDM_OK;
}

/* profiling aid: device u64[8], the next launches accumulate (step-loop blocks, refill phases, sum of per-wave
 * s_memtime cycles, sum of per-wave s_memrealtime ticks (100 MHz), waves); NULL switches it off */
int nddm_set_debug_counters(void *dev_u64x8)
{
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_dbg = static_cast<unsigned long long *>(dev_u64x8);
    return NDDM_OK;
}

/* testing aid: cap on the number of launch slots per device that are handed out (1..256) */
int nddm_debug_set_slot_limit(int n)
{
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_slot_limit = n < 1 ? 1 : (n > nddm::MAX_SLOTS ? nddm::MAX_SLOTS : n);
    return NDDM_OK;
}

/* Frees the memory the library allocated for launches that were captured into hipGraphs on the current device.  The
 * caller asserts that every graph that captured a launch of this library has been destroyed (or will not be replayed). */
int nddm_release_graph_memory(void)
{
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= nddm::MAX_DEVICES) return nddm::fail(NDDM_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    nddm::GraphAlloc *list;
    { std::lock_guard<std::mutex> lock(nddm::g_mu); list = nddm::g_graph_allocs[dev]; nddm::g_graph_allocs[dev] = nullptr; }
    while (list) { nddm::GraphAlloc *n = list->next; (void)hipFree(list->p); delete list; list = n; }
    return NDDM_OK;
}

int nddm_basic_ddm_dc_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_BASIC_DDM_DC, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, flags,
                          0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_single_trial_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_SINGLE_TRIAL, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, flags,
                          0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_single_trial_alt_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                                   float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_SINGLE_TRIAL_ALT, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset,
                          flags, 0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_alpha_not_scaled_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma,
                                   int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata,
                                   void *stream)
{
    return nddm::simulate(NDDM_ALPHA_NOT_SCALED, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset,
                          flags, ext_sigma, ext_mode, out_trials, out_summary, out_extdata, stream);
}

int nddm_explicit_boundary_simulate(const float *params, const float *bounds, int64_t B, int32_t n_trials,
                                    float dt, int32_t max_steps, uint64_t seed, uint64_t set_offset,
                                    uint32_t flags, float *out_trials, float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_EXPLICIT_BOUNDARY, params, bounds, B, n_trials, dt, max_steps, seed, set_offset,
                          flags, 0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_simulate(int32_t model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                  int32_t max_steps, uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma, int32_t ext_mode,
                  float *out_trials, float *out_summary, float *out_extdata, void *stream)
{
    return nddm::simulate(model, params, model == NDDM_EXPLICIT_BOUNDARY ? bounds : nullptr, B, n_trials, dt, max_steps,
                          seed, set_offset, flags, ext_sigma, ext_mode, out_trials, out_summary,
                          model == NDDM_ALPHA_NOT_SCALED ? out_extdata : nullptr, stream);
}

int nddm_draw_prior(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, float gamma, float *out_params,
                    void *stream)
{
    nddm::g_err[0] = 0;
    if (model != NDDM_BASIC_DDM_DC && model != NDDM_SINGLE_TRIAL && model != NDDM_SINGLE_TRIAL_ALT)
        return nddm::fail(NDDM_ERR_PARAM, "draw_prior: model has no reference prior%s");
    if (B < 0) return nddm::fail(NDDM_ERR_SHAPE, "B < 0%s");
    if (B == 0) return NDDM_OK;
    if (!out_params) return nddm::fail(NDDM_ERR_NULL, "out_params is NULL%s");
    const int threads = 256;
    const long long blocks = (B + threads - 1) / threads;
    hipLaunchKernelGGL(nddm::prior_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                       reinterpret_cast<hipStream_t>(stream), (int)model, (long long)B, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (unsigned long long)set_offset, gamma, out_params);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nddm::fail(NDDM_ERR_HIP, "prior kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

int nddm_debug_normals(const uint32_t *counters, int64_t n, uint32_t k0, uint32_t k1, uint32_t flags, float *out,
                       void *stream)
{
    nddm::g_err[0] = 0;
    if (n < 0) return nddm::fail(NDDM_ERR_SHAPE, "n < 0%s");
    if (n == 0) return NDDM_OK;
    if (!counters || !out) return nddm::fail(NDDM_ERR_NULL, "null pointer%s");
    const int threads = 256;
    const long long blocks = (n + threads - 1) / threads;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (flags & NDDM_GAUSS_FAST)
        hipLaunchKernelGGL(nddm::debug_normals_kernel<true>, dim3((unsigned)blocks), dim3(threads), 0, st, counters,
                           (long long)n, k0, k1, out);
    else
        hipLaunchKernelGGL(nddm::debug_normals_kernel<false>, dim3((unsigned)blocks), dim3(threads), 0, st, counters,
                           (long long)n, k0, k1, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nddm::fail(NDDM_ERR_HIP, "debug kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

}  // extern "C"

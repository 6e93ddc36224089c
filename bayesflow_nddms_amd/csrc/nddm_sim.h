// nddm_sim.h -- the simulator kernel of the Euler-Maruyama DDM trial simulators (gfx950): launch arguments, the per-set
// hand-out record, the fused flush epilogue and nddm::sim_kernel.  Included by nddm_kernels.hip (one translation unit);
// the execution design is described there and in DESIGN.md section 5.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nddm.h"
#include "nddm_rng.h"

namespace nddm {

constexpr int WAVE = 64;
constexpr int MAX_REJECT = 64;   // cap of the per-trial latent's rejection loop (P(reject) <= 1/2 per draw)

// LDS accessed by byte address held in a VGPR (the step loop's lanes keep addresses, not indices)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32v4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32v4 lds_u32v4;
typedef __attribute__((address_space(3))) u32v2 lds_u32v2;

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
    uint16_t *out_codes;      // [B, N] or null: the trials in the 2-byte wire format (step index | code << 14), models whose second
                              // column is a function of the code (basic, alpha_not_scaled without the bridge), cap < 2^14
    long long B;
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
    int ring;                 // LDS ring slots (any number >= 2: the slot of tile t is t mod ring)
    int open_ahead;           // tiles staged ahead of the one being handed out (0 when work is scarce, else 1)
    float ext_sigma;
    int ext_mode;
    unsigned long long *dbg;  // optional trace buffer (nddm_set_debug_trace): [dbg_waves][8] per-wave records {step-loop blocks,
                              // refill phases, s_memtime cycles, lifetime ticks, start tick, tick at which the wave found
                              // the queue empty, end tick, 1}, indexed by workgroup, then [dbg_chunks] pull ticks of the
                              // chunks; null in production.  Plain stores only: the trace does not perturb the launch
    int dbg_waves, dbg_chunks;
    int res16;                // results are staged as 16-bit words (step index < 2^14 | code << 14): no bridge, cap < 16384;
                              // 2 = ... and the tile has <= 512 trials (the flush's 32-bit / DPP reduction path)
    int refill_thresh;        // leave the step loop once this many lanes hold a finished trial
    uint32_t ring_magic;      // floor(2^32 / ring): t mod ring by a multiply (ring_slot())
    double dt64, sqrt_dt64;   // NDDM_STATE_F64: (double)dt and its correctly rounded double square root (the reference's np.sqrt(dt))
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

// The lane index behind a compiler barrier: a predicate on it (lane == 0, lane < REC) is then recomputed where it is used
// -- one v_cmp in a rare path -- instead of being hoisted into an SGPR pair that stays live through the step loop.
__device__ __forceinline__ int fresh_lane(int lane)
{
    asm volatile("" : "+v"(lane));
    return lane;
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
//   r[10,11] the set's GLOBAL index (low word, high 28 bits): set_offset + *set_offset_dev + row.  It keys the random
//            stream; the pre-pass adds the optional device-resident part of the offset (nddm_simulate_indirect: a
//            replayed hipGraph moves along the stream without new kernel arguments), the simulator only reads it here
// S = noise_unit(sqrt(dt) * dc): the state is carried in noise units (nddm_rng.h).
enum { REC = 12, R_A = 0, R_B = 4, R_SET = 8, R_TAU = 9, R_GLO = 10, R_GHI = 11 };

// f64 (NDDM_STATE_F64, basic and single): A holds the RAW parameters -- drift, dc, boundary | std_alpha, beta | mu_alpha -- and the
// hand-out forms the reference's float64 quantities from them (boundary * beta, drift * dt, sqrt(dt) * dc) in double.
__device__ __forceinline__ void make_record(int model, bool fast, bool f64, const float *row, float dt, float sqrt_dt, int set,
                                            unsigned long long gset, uint32_t *r)
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
    if (f64) {
        a0 = drift; a1 = sig_c;
        if (model == NDDM_BASIC_DDM_DC) { a2 = a; a3 = beta; }      // (single keeps std_alpha, mu_alpha in A.z, A.w and beta in B.z)
    }
    r[0] = __float_as_uint(a0); r[1] = __float_as_uint(a1); r[2] = __float_as_uint(a2); r[3] = __float_as_uint(a3);
    r[4] = __float_as_uint(b0); r[5] = __float_as_uint(b1); r[6] = __float_as_uint(b2); r[7] = 0u;
    r[8] = (uint32_t)set; r[9] = __float_as_uint(tau);
    r[R_GLO] = (uint32_t)gset; r[R_GHI] = (uint32_t)(gset >> 32) & 0x0fffffffu;
}

// The tile's record in LDS, as the hand-out reads it (ds_read_b128 each): dword index into its DV dwords.  The first REC
// dwords are the set's record as loaded (one masked store), the rest is filled in by lane 0 when the tile opens.
enum { D_A = 0, D_B = 4,                             // r[0..3], r[4..7]
       D_SIC = 8, D_TAU = 9, D_TBASE = 10, D_VSET = 11, // in-call set index, tau, first trial of the tile within its set,
                                                     // virtual set (set * tiles_per_set + tile)
       D_CA = 12, D_CB = 13, D_HP1K = 14, D_X1 = 15, // Philox constants of the set (PathSet in nddm_rng.h)
       D_C3 = 16, D_SETLO = 17,                      // high set word (28 bits), low set word (auxiliary stream's counter)
       D_ZSUM = 20,                                  // [20..23] fixed-point sums of z and z^2 (models with a z summary)
       D_SPARE = 20 };                               // [20..23] unused by the models without a z summary
constexpr int DV = 24;                               // one layout for every model: only two LDS base addresses stay live
static_assert(D_SIC == R_SET && D_TAU == R_TAU, "the LDS record starts with the loaded record");
constexpr int LDS_HEADER_DWORDS = 32;                // key table [0,20) | debug stamps [20,28) | kC kD kE [28,31)
// One ring slot = the tile's record followed by its staged results (2 or 4 bytes per trial), 16-byte aligned: a lane
// addresses both from one base (slot * stride), and keeps the LDS byte address of its trial's result word while it steps.
// Models whose trials carry a per-trial latent drawn from the auxiliary stream (single-trial family: boundary / noise
// scale; alpha_not_scaled: drift) keep a 128-entry FIFO of latents behind the header: the latents of the next trials of
// the wave's hand-out sequence are drawn 64 at a time, by all lanes, instead of by the ~12-16 lanes of each hand-out
// (the draw is ~100 VALU instructions whatever the number of lanes it serves).
__host__ __device__ constexpr bool model_has_latent(int model)
{
    return model == NDDM_SINGLE_TRIAL || model == NDDM_SINGLE_TRIAL_ALT || model == NDDM_ALPHA_NOT_SCALED;
}
constexpr int LATENT_FIFO = 128;                     // entries (f32); a power of two >= 2 * WAVE
__host__ __device__ constexpr int lds_header_bytes(int model)
{
    return LDS_HEADER_DWORDS * 4 + (model_has_latent(model) ? LATENT_FIFO * 4 : 0);
}
__host__ __device__ constexpr int slot_stride_bytes(int tile_trials, int bytes_per_result)
{
    return DV * 4 + ((tile_trials * bytes_per_result + 15) & ~15);
}

// The per-trial latent of the single-trial family and the external datum that goes with it: a pure function of
// (set, trial) and the set's record, so the hand-out (which needs the latent) and the flush (which writes the datum next
// to the choice-RT, one whole float2 per trial) each evaluate it where they need it; the part a caller does not use is
// dead code there.  Normal 0 of the auxiliary stream is the datum's noise, normals 1, 2, ... the rejection draws.
//   single:     latent = boundary ~ N(mu_alpha, std_alpha) > 0 (single_trial_alpha_not_scaled.py:113-116),  z1 ~ N(gamma * boundary, sigma1) (:134)
//   single_alt: latent = dc       ~ N(mu_dc, std_dc) > 0       (:932-935),                                z1 ~ N(gamma * dc, sigma1)
template <int MODEL, bool FAST>
__device__ __forceinline__ void trial_latent(const u32v4 dA, const u32v4 dB, uint32_t set_lo, uint32_t c3, uint32_t trial,
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

// NDDM_STATE_F64, single-trial model: the per-trial boundary in the reference's arithmetic,
//     bound_trial = mu_alpha + std_alpha * normal   (single_trial_alpha_not_scaled.py:113-116; float64)
// on the SAME auxiliary normals (1, 2, ...: rejection draws) as trial_latent.  Returns the ACCEPTED normal (a float: what the
// latent FIFO holds); the hand-out rebuilds the double from it with the same two operations.
template <bool FAST>
__device__ __forceinline__ float latent_normal_f64(const u32v4 dA, uint32_t set_lo, uint32_t c3, uint32_t trial, uint32_t kbase)
{
    const double sd = (double)__uint_as_float(dA.z), mean = (double)__uint_as_float(dA.w);
    AuxStream<FAST> aux(kbase, set_lo, c3, trial);
    float z0, z;
    aux.first_pair(z0, z);
    double v = mean + sd * (double)z;
    for (uint32_t ai = 2; !(v > 0.0) && ai <= MAX_REJECT; ++ai) { z = aux.normal(ai); v = mean + sd * (double)z; }
    return z;
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
// Then a lane's share of every sum fits 32 bits (<= 8 trials, k < 2^14), the two counters share one word (10 bits
// each), and the seven cross-lane sums are DPP reductions of 32-bit values (the two sums of squares as 16-bit halves).
template <int MODEL, bool FAST, bool SMALL, bool CODES>
__device__ __forceinline__ void flush_set(ArgsPtr Ap, int lane, long long vset, uint32_t *d, const void *res, uint32_t kbase)
{
    using T = ModelTraits<MODEL>;
    constexpr bool ZSUM = MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT;
    const float tau = __uint_as_float(d[D_TAU]);
    [[maybe_unused]] u32v4 dA = {0u, 0u, 0u, 0u}, dB = {0u, 0u, 0u, 0u};
    [[maybe_unused]] uint32_t set_lo = 0u, c3 = 0u;
    [[maybe_unused]] long long acc_z = 0, acc_zz = 0;
    if constexpr (ZSUM) {
        dA = *reinterpret_cast<const u32v4 *>(d + D_A); dB = *reinterpret_cast<const u32v4 *>(d + D_B);
        c3 = d[D_C3]; set_lo = d[D_SETLO];
    }
    const int N = Ap->n_trials;
    const int TPS = Ap->tiles_per_set;
    const long long set_in_call = TPS == 1 ? vset : vset / TPS;
    const int t0 = TPS == 1 ? 0 : (int)(vset - set_in_call * TPS) * N;      // first trial of this tile
    const int n_here = (Ap->n_total - t0) < N ? (Ap->n_total - t0) : N;           // the last tile may be padded
    uint32_t cnt2 = 0, sk32 = 0, sk2_32 = 0, sk_up32 = 0, sk2_up32 = 0;           // SMALL (cnt2: upper | responded << 10)
    int n_up = 0, n_lo = 0, n_miss = 0;                                           // !SMALL
    unsigned long long sk = 0, sk2 = 0, sk_up = 0, sk2_up = 0;
    float2 *out = Ap->out_trials ? reinterpret_cast<float2 *>(Ap->out_trials) + set_in_call * Ap->n_total + t0 : nullptr;
    // CODES: the launch also (or only) wants the trials in the 2-byte wire format.  A kernel variant of its own: carried by the
    // default kernels the extra pointer and store cost the headline 1.2-1.4 % (A/B on one box: 55 -> 60 SGPRs, 42 -> 44 VGPRs and a
    // different schedule of the step loop), so they do not carry it
    static_assert(!CODES || (SMALL && (MODEL == NDDM_BASIC_DDM_DC || MODEL == NDDM_ALPHA_NOT_SCALED)), "wire-format kernels");
    constexpr bool HAS_CODES = CODES;
    [[maybe_unused]] uint16_t *codes = nullptr;
    if constexpr (HAS_CODES) { if (Ap->out_codes) codes = Ap->out_codes + set_in_call * Ap->n_total + t0; }
    for (uint32_t j = (uint32_t)lane; j < (uint32_t)n_here; j += WAVE) {   // (unsigned: scalar base + 32-bit lane offset)
        uint32_t k, code;                                    // time in units of tscale (step index, or 1/256 step);
        if (SMALL || Ap->res16) {                            // code: 0 timeout, 1 upper, 2 lower, 3 invalid trial
            const uint32_t v = static_cast<const uint16_t *>(res)[j];
            k = v & 0x3fffu; code = v >> 14;
            if constexpr (HAS_CODES) { if (codes) codes[j] = (uint16_t)v; }      // the staged word IS the wire format
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
        if constexpr (MODEL == NDDM_EXPLICIT_BOUNDARY) { if (code == 3u) o.x = __builtin_nanf(""); }   // only this model has invalid trials
        if (out) out[j] = o;
        if (Ap->out_summary) {
            if constexpr (SMALL) {
                // 0 / 1 flags as bit `code` of a constant (v_bfe_u32), sums by 24-bit multiply-adds (k < 2^14): 10
                // instructions where compares and selects took 14.  A code-3 (invalid) trial counts as missing.
                const uint32_t up01 = __builtin_amdgcn_ubfe(0x2u, code, 1u), resp01 = __builtin_amdgcn_ubfe(0x6u, code, 1u);
                const uint32_t kr = __umul24(k, resp01), ku = __umul24(k, up01);
                cnt2 += up01 | (resp01 << 10);
                sk32 += kr; sk_up32 += ku;
                sk2_32 = __umul24(kr, k) + sk2_32; sk2_up32 = __umul24(ku, k) + sk2_up32;
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
            cnt2 = wave_sum_dpp(cnt2);
            sk32 = wave_sum_dpp(sk32); sk_up32 = wave_sum_dpp(sk_up32);
            const uint32_t a_hi = wave_sum_dpp(sk2_32 >> 16), a_lo = wave_sum_dpp(sk2_32 & 0xffffu);
            const uint32_t u_hi = wave_sum_dpp(sk2_up32 >> 16), u_lo = wave_sum_dpp(sk2_up32 & 0xffffu);
            if (fresh_lane(lane) == WAVE - 1) {              // the DPP reductions leave the totals in the last lane
                const uint32_t n_up = cnt2 & 1023u, n_resp = cnt2 >> 10;              // <= 512 each
                q[0] = (unsigned long long)n_up | ((unsigned long long)(n_resp - n_up) << 21) |
                       ((unsigned long long)((uint32_t)n_here - n_resp) << 42);
                q[1] = sk32; q[2] = ((unsigned long long)a_hi << 16) + a_lo;
                q[3] = sk_up32; q[4] = ((unsigned long long)u_hi << 16) + u_lo;
                if constexpr (ZSUM) { q[5] = zsum[0]; q[6] = zsum[1]; }
            }
        } else {
            n_up = wave_sum(n_up); n_lo = wave_sum(n_lo); n_miss = wave_sum(n_miss);
            sk = wave_sum(sk); sk2 = wave_sum(sk2); sk_up = wave_sum(sk_up); sk2_up = wave_sum(sk2_up);
            if (fresh_lane(lane) == 0) {
                q[0] = (unsigned long long)n_up | ((unsigned long long)n_lo << 21) | ((unsigned long long)n_miss << 42);
                q[1] = sk; q[2] = sk2; q[3] = sk_up; q[4] = sk2_up;
                if constexpr (ZSUM) { q[5] = zsum[0]; q[6] = zsum[1]; }
            }
        }
    }
    if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) {
        if (Ap->out_ext && fresh_lane(lane) == 0 && t0 == 0) {
            AuxStream<FAST> aux(kbase, d[D_SETLO], d[D_C3], 0xffffffffu);                           // the set's global index
            const float loc = (Ap->ext_mode == 0) ? Ap->params[set_in_call * T::P + 1] : 1.0f;     // Alpha of the set
            Ap->out_ext[set_in_call] = __builtin_fmaf(Ap->ext_sigma, aux.normal(0), loc);
        }
    }
}

// t mod ring for a wave-uniform tile number (t < 2^31): quotient estimate by the high product with floor(2^32 / ring), one
// correction.  The ring need not be a power of two (six slots of 300 trials fit where eight do not).
__device__ __forceinline__ int ring_slot(int t, int ring, uint32_t magic)
{
    const uint32_t q = (uint32_t)(((unsigned long long)(uint32_t)t * magic) >> 32);
    int r = t - (int)(q * (uint32_t)ring);
    return r >= ring ? r - ring : r;
}

// The evidence is carried CENTRED: w = x - a/2, h = a/2, so that (x > 0) && (x < a) is the single compare |w| < h
// (v_cmp_lt_f32 with the |.| source modifier; false for NaN and for h == 0).
__device__ __forceinline__ bool in_range(float w, float h)
{
    return __builtin_fabsf(w) < h;
}


// MODEL: enum nddm_model.  FAST: Gaussian transform.  SMALL: see below.  CAP4: max_steps is a multiple of 4, so the step cap is tested
// once per Philox block instead of once per step (BRIDGE, PACKED: a multiple of 8, once per 8 steps).  BRIDGE: Brownian-bridge boundary correction (between two grid
// points inside (0, a) the path still crosses a boundary with probability exp(-2 d0 d1 / (sigma^2 dt))), which removes
// the O(sqrt(dt)) late-detection bias of plain Euler-Maruyama -- used for alpha_not_scaled, whose reference
// generator is an exact first-passage sampler.
// (Tried and dropped: Philox round keys in VGPRs.  A VOP2 xor that reads an SGPR issues at ~4.2 instead of ~2.3
// cycles on gfx950, but the 20 extra VGPRs cut residency from 7 to 5 waves per SIMD and the net was neutral.)
// SMALL: results staged as 16-bit words and at most 512 trials per tile -- the shape of every launch that matters for
// throughput.  A kernel that carries BOTH flush paths (32-bit DPP sums / 64-bit shuffles) needs 68-76 SGPRs and 52-66 VGPRs
// (basic 68 / 52), the SMALL one alone 55-71 and 42-60 (basic 55 / 42; tools/resource_table.py prints every instantiation),
// which is what keeps 8 waves per SIMD resident (the SGPR file limits these kernels: <= 74).
// PACKED: NDDM_GAUSS_PACKED -- 8 Euler-Maruyama steps per Philox block (polar_pair_packed in nddm_rng.h); CAP4 then means
// "max_steps is a multiple of 8".
// VKEYS: the Philox round keys of the step loop are held in 13 VGPRs for the whole kernel instead of read from LDS every
// block (nddm_rng.h): no LDS instruction and no wait in the loop.  Same VALU work; measured 1.6 % slower on a full grid
// and 13-29 % faster when a wave has a SIMD (nearly) to itself, so the host picks it for small launches.
// CODES: the variant that also stores the trials as 2-byte codes (flush_set); instantiated for basic / alpha_not_scaled, SMALL only.
// F64: NDDM_STATE_F64 -- the evidence is carried as the REFERENCE carries it (basic_ddm_dc.py:91-103): a float64 in its natural
// units, evidence += drift*dt + sqrt(dt)*dc*normal with every operation a separate IEEE double operation in the reference's
// order, the range test (evidence > 0) && (evidence < boundary) on doubles; the normal is the exact double product of the
// float32 Box-Muller radius and cosine / sine.  Philox, the transform, the hand-out and the flush are the float32 kernel's.
// basic and single only; bit-equal in (step, choice) to oracle_philox_simulate_f64 in exact mode.
template <int MODEL, bool FAST, bool CAP4, bool BRIDGE, bool SMALL, bool PACKED, bool VKEYS, bool CODES = false, bool F64 = false>
__global__ __launch_bounds__(WAVE) void sim_kernel(const SimArgs A)
{
    static_assert(!F64 || ((MODEL == NDDM_BASIC_DDM_DC || MODEL == NDDM_SINGLE_TRIAL) && !BRIDGE && !PACKED && !VKEYS && !CODES),
                  "NDDM_STATE_F64 kernels");
    using T = ModelTraits<MODEL>;
    constexpr int P = T::P;
    extern __shared__ uint32_t lds_raw[];

    const ArgsPtr Ak = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x;
    const int N = A.n_trials;

    // LDS carve-up.  Header (128 bytes): the ten Philox round-key pairs [0, 80) (philox4x32_10_path), debug stamps
    // [80, 104), the three round keys that fold into the per-trial constants [112, 124).  Then one ring slot per
    // in-flight parameter set ("tile"): its DV-dword record (hand-out constants, z sums), then the packed results
    if (lane < 10) { lds_raw[2 * lane] = A.k0 + (uint32_t)lane * 0x9E3779B9u; lds_raw[2 * lane + 1] = A.k1 + (uint32_t)lane * 0xBB67AE85u; }
    if (lane == 0) { lds_raw[28] = A.k0 + 2u * PHILOX_W0; lds_raw[29] = A.k1 + 2u * PHILOX_W1; lds_raw[30] = A.k0 + 3u * PHILOX_W0; }
    // LDS byte address of the key table in a VGPR (the low 32 bits of a flat LDS address are the LDS offset); the asm
    // keeps it opaque so that every ds_read in the step loop uses this one register + an immediate offset
    uint32_t kbase;
    {
        const uint32_t off = (uint32_t)(size_t)lds_raw;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kbase) : "s"(off));
    }
    // ring slots: [DV-dword record | staged results], 16-byte aligned.  Staged results: one 32-bit word per trial, or one
    // 16-bit word when the step cap allows (halves the LDS footprint, which is what lets the 7th and 8th wave per SIMD
    // stay resident at 300 trials per set)
    const int rshift = SMALL ? 1 : (BRIDGE ? 2 : (A.res16 ? 1 : 2));   // log2 bytes per staged result (the bridge's times are 1/256 step: 32 bits)
    constexpr bool LATENT = model_has_latent(MODEL);
    const int stride = DV * 4 + (((N << rshift) + 15) & ~15);          // slot_stride_bytes()
    char *const slots = reinterpret_cast<char *>(lds_raw) + lds_header_bytes(MODEL);
    auto slot_rec = [&](int slot) { return reinterpret_cast<uint32_t *>(slots + slot * stride); };
    auto pack_slots = [&](int t) {                                     // (rare path: kernarg reads, a multiply-high)
        const ArgsPtr R = fresh_args(Ak);
        const int s0 = ring_slot(t, R->ring, R->ring_magic);
        const int o0 = s0 * stride, o01 = (s0 + 1 == R->ring ? 0 : s0 + 1) * stride - o0;
        return (o0 & 0xffff) | (o01 << 16);                            // (slot offsets < 64 KB: the whole LDS footprint is)
    };
    constexpr uint32_t SLOTS_OFF = lds_header_bytes(MODEL);            // byte offset of slot 0 from kbase
    constexpr uint32_t FIFO_OFF = LDS_HEADER_DWORDS * 4;               // ... of the latent FIFO (models with a latent)
    // latent FIFO: fifo_pos = hand-out sequence position of the wave's next trial (entry index = position mod LATENT_FIFO),
    // fifo_avail = latents drawn ahead of it
    [[maybe_unused]] int fifo_pos = 0, fifo_avail = 0;

    // per-lane trial state
    // w: centred evidence, h: boundary / 2, mu_dt: drift per step -- all in NOISE UNITS (divided by noise_unit(sigma))
    float w = 0.0f, h = 0.0f, mu_dt = 0.0f;
    // F64: xe = evidence, ab = boundary, cdt = drift * dt, sdc = sqrt(dt) * dc (fast transform: * sqrt(2 ln 2), its radius unit)
    [[maybe_unused]] double xe = 0.0, ab = 0.0, cdt = 0.0, sdc = 0.0;
    int k = 0;
    // BRIDGE only: ta / tb = h - |w|, the distance to the nearer boundary carried from step to step; jraw = the crossing-uniform
    // word of the trial's current step pair (its jitter bits are taken when the trial retires).  With the bridge, k counts the
    // steps a trial SURVIVED: a trial that crossed at its (k+1)-th step stops with k < max_k, one that ran to the cap with
    // k == max_k -- no crossed-flag is carried
    // (the distance lives alternately in ta and tb: a step reads one and writes the other, so survivors need no copy; at
    // the end of a pass -- an even number of steps -- the current one is ta again.  Both start as h - |w0|: a lane that never
    // stepped has both <= 0, a lane that stepped has the one its last step READ > 0)
    [[maybe_unused]] float ta = 0.0f, tb = 0.0f;
    [[maybe_unused]] uint32_t jraw = 0;
    uint32_t res_addr = 0;   // LDS byte address (relative to the slots) of the trial's staged result
    [[maybe_unused]] int tile = 0;   // !SMALL only: wave-local sequence number of the tile this lane works on
    bool invalid = false;
    // which lanes hold a trial / are still stepping: wave-uniform lane masks kept in SGPRs (a per-lane bool that is
    // balloted costs v_cndmask + v_cmp each time; __builtin_amdgcn_inverse_ballot_w64 turns a mask into exec for free)
    unsigned long long has_m = 0ull, act_m = 0ull;
    PathCtr pc = {0u, 0u, 0u, 0u, 0u};
    [[maybe_unused]] PathKeys pkeys;              // VKEYS: the step loop's Philox round keys
    if constexpr (VKEYS) pkeys.init(A.k0, A.k1);

    // wave-uniform bookkeeping.  Tiles (sets) are opened, handed out and flushed strictly in sequence.
    int tile_open = 0;       // tiles whose parameters are staged in LDS
    int flushed = 0;         // tiles already flushed
    int next_tile = 0, next_trial = 0;            // next unassigned trial of the wave's stream
    // (32-bit, compared modulo 2^32: SGPRs are what limits these kernels' residency)
    int to_retire = N;                            // the oldest tile cannot be complete before this many more trials retire
    int chunk_set = 0, chunk_left = 0;            // current chunk: next set index (in-call), sets left; -1: queue exhausted
    // byte offset of the slot of the tile being handed out (next_tile) in the low half, and -- signed, in the high half -- the
    // step from it to the next tile's slot: recomputed when next_tile moves, unpacked (two scalar ops) at every hand-out
    int slot_pack = pack_slots(0);
    int dirty = 1;                                // the tile counters changed since "open more tiles?" / "all done?" were evaluated
    // trace counters: one wave-uniform 64-bit integer (scalar adds; as an LDS counter bumped by lane 0 they were 7 of the ~36
    // VALU instructions of every refill), and the start / queue-dry stamps in LDS
    unsigned long long dbg_cnt = 0;              // refill phases << 32 | step-loop blocks
    unsigned long long *dbg_stamp = reinterpret_cast<unsigned long long *>(lds_raw + 20);   // [0], [1]: start clocks, [3]: dry
    if (lane == 0) {
        dbg_stamp[3] = 0;
        lds_raw[31] = blockIdx.x;             // read back at exit (kept in LDS: an SGPR held through the kernel costs residency)
        if (Ak->dbg) { dbg_stamp[0] = __builtin_amdgcn_s_memtime(); dbg_stamp[1] = __builtin_amdgcn_s_memrealtime(); }
    }

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
        while (tile_open < flushed + R->ring && tile_open <= next_tile + R->open_ahead) {
            const bool have_pre = chunk_left > 0;                        // mid-chunk: this tile's record was prefetched
            if (chunk_left <= 0) {
                if (chunk_left < 0) break;
                unsigned int c = 0;
                if (fresh_lane(lane) == 0) c = atomicAdd(R->chunk_counter, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= (unsigned int)R->n_chunks) {
                    chunk_left = -1;
                    if (fresh_lane(lane) == 0 && R->dbg) dbg_stamp[3] = __builtin_amdgcn_s_memrealtime();      // the queue ran dry (trace)
                    break;
                }
                if (fresh_lane(lane) == 0 && R->dbg && c < (unsigned int)R->dbg_chunks)                      // trace: when chunk c was pulled
                    R->dbg[8ll * R->dbg_waves + c] = __builtin_amdgcn_s_memrealtime();
                chunk_set = (int)c * R->sets_per_chunk;
                const long long left = R->B - (long long)chunk_set;
                chunk_left = (int)(left < R->sets_per_chunk ? left : R->sets_per_chunk);
            }
            const int slot = ring_slot(tile_open, R->ring, R->ring_magic);
            // queue position -> queue row (position / tiles_per_set) -> its record
            const int TPS = R->tiles_per_set;
            const int prow = TPS == 1 ? chunk_set : chunk_set / TPS;
            const int qt = chunk_set - prow * TPS;                       // tile within the set (0 when not tiled)
            uint32_t rec = pre;
            if (!have_pre) rec = fresh_lane(lane) < REC ? R->recs[(long long)prow * REC + lane] : 0u;
            if (chunk_left > 1) {                                        // prefetch the next tile's record
                const int nrow = TPS == 1 ? chunk_set + 1 : (chunk_set + 1) / TPS;
                if (nrow != prow) pre = fresh_lane(lane) < REC ? R->recs[(long long)nrow * REC + lane] : 0u;
                else pre = rec;
            }
            const int sic = __builtin_amdgcn_readlane((int)rec, R_SET);  // the set's row in the caller's arrays
            const int vset = sic * TPS + qt;
            uint32_t *d = slot_rec(slot);
            if (lane < REC) d[lane] = rec;
            // everything about the set is wave-uniform: scalar arithmetic.  Its global index comes with the record
            const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readlane((int)rec, R_GLO);
            const uint32_t s_hi = (uint32_t)__builtin_amdgcn_readlane((int)rec, R_GHI);
            if (lane == 0) {
                d[D_VSET] = (uint32_t)vset;
                if constexpr (MODEL == NDDM_SINGLE_TRIAL || MODEL == NDDM_SINGLE_TRIAL_ALT) { d[D_ZSUM] = 0u; d[D_ZSUM + 1] = 0u; d[D_ZSUM + 2] = 0u; d[D_ZSUM + 3] = 0u; }
                PathSet ps;
                ps.init(s_lo, s_hi, R->k0, R->k1);              // stream 0: no tag bits in c2
                d[D_CA] = ps.cA; d[D_CB] = ps.cB; d[D_HP1K] = ps.hP1k; d[D_X1] = ps.X1;
                d[D_SETLO] = s_lo;
                d[D_C3] = s_hi;
                d[D_TBASE] = (uint32_t)(qt * N);
            }
            chunk_set++; chunk_left--; tile_open++;
        }
    };

    while (true) {
        // ------------------------------------------------------------ retire finished trials
        const unsigned long long fin_mask0 = has_m & ~act_m;
        if (__builtin_amdgcn_inverse_ballot_w64(fin_mask0)) {
            uint32_t code = invalid ? 3u : (w >= h ? 1u : (w <= -h ? 2u : 0u));
            if constexpr (F64) code = xe >= ab ? 1u : (xe <= 0.0 ? 2u : 0u);        // basic_ddm_dc.py:106-111
            uint32_t tfix = (uint32_t)k;
            if constexpr (BRIDGE) {
                // started: the trial began inside (0, a); stepped: ... and ended by crossing a boundary (at a grid point or, by
                // the bridge test, between two) at step k + 1; otherwise it ran to the cap (k == max_k: timeout) or never moved
                const bool started = fmaxf(ta, tb) > 0.0f;
                const bool stepped = started && k < A.max_k;
                code = (stepped || !started) ? (w >= 0.0f ? 1u : 2u) : 0u;
                // time in 1/256 step: the crossing step's index minus an 8-bit uniform jitter (the crossing happened somewhere
                // inside the step), taken from the bits of the pair's word that the step's own decision did not use: bits 0..7
                // for the pair's first step, 16..23 for its second
                const uint32_t jit = (k & 1) ? ((jraw >> 16) & 0xffu) : (jraw & 0xffu);
                tfix = (((uint32_t)k + (stepped ? 1u : 0u)) << 8) - (stepped ? jit : 0u);
            }
            if (SMALL || rshift == 1) *reinterpret_cast<lds_u16 *>(res_addr + (SLOTS_OFF + DV * 4)) = (uint16_t)(tfix | (code << 14));
            else *reinterpret_cast<lds_u32 *>(res_addr + (SLOTS_OFF + DV * 4)) = tfix | (code << 30);
        }
        has_m &= ~fin_mask0;
        to_retire -= (int)__popcll(fin_mask0);
        // ------------------------------------------------------------ flush complete sets, in order (rare path:
        // only entered when enough trials have retired for the oldest tile to possibly be complete)
        if (to_retire <= 0) {
            __syncthreads();
            while (flushed < tile_open) {
                // the oldest tile is complete when all its trials have been handed out and no lane holds one of them
                // (no per-tile counter: the hand-out is sequential, so these two wave-uniform facts say it all)
                if (next_tile <= flushed) break;
                // (a lane holds a trial of this tile iff its result address lies in the tile's slot: slots are not reused
                // before their tile is flushed)
                const int fslot = ring_slot(flushed, fresh_args(Ak)->ring, fresh_args(Ak)->ring_magic);
                if constexpr (SMALL) {
                    const uint32_t fsb = (uint32_t)(fslot * stride) + kbase;
                    if (__builtin_amdgcn_ballot_w64(res_addr - fsb < (uint32_t)stride) & has_m) break;
                } else {                     // (the general kernels keep the tile number per lane: fewer SGPRs there)
                    if (__builtin_amdgcn_ballot_w64(tile == flushed) & has_m) break;
                }
                uint32_t *const d = slot_rec(fslot);
                const int set_in_call = __builtin_amdgcn_readfirstlane((int)d[D_VSET]);
                flush_set<MODEL, FAST, SMALL, CODES>(fresh_args(Ak), lane, (long long)set_in_call, d, d + DV, kbase);
                flushed++;
                to_retire += N;
                dirty = 1;
            }
            __syncthreads();
        }
        // ------------------------------------------------------------ open tiles (the one place: also the first pass,
        // and the pass that finds the queue empty)
        // (both questions depend only on the tile counters, which change when a tile is flushed, opened or handed out
        // completely: they are asked then, not at every refill -- scalar instructions are not free, tools/ubench_salu)
        if (dirty) {
            dirty = 0;
            if (tile_open <= next_tile + fresh_args(Ak)->open_ahead && tile_open < flushed + fresh_args(Ak)->ring && chunk_left >= 0) {
                open_tiles();
                __syncthreads();
            }
            if (flushed == tile_open && chunk_left < 0) break;
        }
        // ------------------------------------------------------------ hand out new trials
        {
            const unsigned long long want_mask = ~has_m;
            int tr = next_trial + (int)lane_rank(want_mask);
            int tl = next_tile;
            while (tr >= N) { tr -= N; tl++; }
            const unsigned long long ok_mask = want_mask & __builtin_amdgcn_ballot_w64(tl < tile_open);
            // byte offsets of the slots of the (at most two) tiles open for hand-out: next_tile and the one after it
            const int off0 = slot_pack & 0xffff, off01 = slot_pack >> 16;
            const int tile0 = next_tile;
            if constexpr (LATENT) {
                const int n_ok = (int)__popcll(ok_mask);
                if (n_ok > fifo_avail) {
                    // not enough latents drawn ahead: draw those of the next 64 positions of the hand-out sequence (all
                    // lanes; positions whose tile is not open yet are left for the next time -- the open ones are a
                    // prefix, and they include every position this hand-out serves)
                    int btr = next_trial + fifo_avail + lane, btl = next_tile;
                    while (btr >= N) { btr -= N; btl++; }
                    const unsigned long long bmask = __builtin_amdgcn_ballot_w64(btl < tile_open);
                    if (__builtin_amdgcn_inverse_ballot_w64(bmask)) {
                        const uint32_t bsb = (uint32_t)(__mul24(btl - tile0, off01) + off0) + kbase;
                        const lds_u32v4 *const bq = reinterpret_cast<const lds_u32v4 *>(bsb + SLOTS_OFF);
                        const u32v4 dA = bq[D_A / 4];
                        [[maybe_unused]] const u32v4 dB = bq[D_B / 4];
                        const u32v4 dS = bq[D_SIC / 4];                                  // .z = first trial of the tile
                        const u32v2 sw = *reinterpret_cast<const lds_u32v2 *>(bsb + SLOTS_OFF + D_C3 * 4);   // c3, set_lo
                        const uint32_t btrial = (uint32_t)btr + dS.z;
                        float v;
                        if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) {
                            // A = Nu, 1/S, a/(2S), w0;  B = Eta: the trial's drift N(Nu, Eta) per step, in noise units
                            AuxStream<FAST> aux(kbase, sw.y, sw.x, btrial);
                            v = (__builtin_fmaf(__uint_as_float(dB.x), aux.normal(0), __uint_as_float(dA.x)) * fresh_args(Ak)->dt) * __uint_as_float(dA.y);
                        } else if constexpr (F64) {
                            v = latent_normal_f64<FAST>(dA, sw.y, sw.x, btrial, kbase);     // the accepted normal
                        } else {
                            float z_unused;
                            trial_latent<MODEL, FAST>(dA, dB, sw.y, sw.x, btrial, kbase, v, z_unused);
                        }
                        *reinterpret_cast<__attribute__((address_space(3))) float *>(
                            kbase + FIFO_OFF + (((uint32_t)(fifo_pos + fifo_avail + lane) & (LATENT_FIFO - 1)) << 2)) = v;
                    }
                    fifo_avail += (int)__popcll(bmask);
                }
            }
            next_trial += (int)__popcll(ok_mask);
            if (next_trial >= N) {
                // the cursor moves on by one tile (more only when tiles are smaller than a wave): the next tile's slot becomes
                // the current one, the one after it follows -- or wraps to slot 0 at the end of the ring
                const int ring_bytes = fresh_args(Ak)->ring * stride;
                int o0 = slot_pack & 0xffff, o01 = slot_pack >> 16;
                while (next_trial >= N) {
                    next_trial -= N; next_tile++;
                    o0 += o01;
                    o01 = o0 + stride == ring_bytes ? -o0 : stride;
                }
                slot_pack = (o0 & 0xffff) | (o01 << 16);
                dirty = 1;
            }
            has_m |= ok_mask;
            if (__builtin_amdgcn_inverse_ballot_w64(ok_mask)) {
#ifdef NDDM_EXTRA_REFILL_VALU
                // sensitivity experiment (tools/refill_sensitivity.sh): N extra full-rate VALU instructions per hand-out
                { uint32_t dummy = (uint32_t)lane;
#pragma unroll
                  for (int e = 0; e < NDDM_EXTRA_REFILL_VALU; ++e) asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(dummy)); }
#endif
                const ArgsPtr H = fresh_args(Ak);
                if constexpr (!SMALL) tile = tl;
                // LDS byte address of the slot (kbase holds the LDS base: one v_mad_u32_u24), of the trial's result word
                const uint32_t sb = (uint32_t)(__mul24(tl - tile0, off01) + off0) + kbase;
                res_addr = sb + ((uint32_t)tr << rshift);
                const lds_u32v4 *const rq = reinterpret_cast<const lds_u32v4 *>(sb + SLOTS_OFF);
                const lds_u32 *const rw = reinterpret_cast<const lds_u32 *>(sb + SLOTS_OFF);
                const u32v4 d0 = rq[D_A / 4];
                const u32v4 d1 = rq[D_CA / 4];
                const u32v4 d2 = rq[D_SIC / 4];                          // set index, tau, TBASE
                const float a0 = __uint_as_float(d0.x), a1 = __uint_as_float(d0.y), a2 = __uint_as_float(d0.z),
                            a3 = __uint_as_float(d0.w);                  // the model's A constants (make_record)
                const uint32_t trial = (uint32_t)tr + d2.z;          // index within the set (keys the random stream)
                [[maybe_unused]] float latent = 0.0f;                 // the trial's latent, drawn ahead (FIFO)
                if constexpr (LATENT)
                    latent = *reinterpret_cast<const __attribute__((address_space(3))) float *>(
                        kbase + FIFO_OFF + (((uint32_t)fifo_pos + lane_rank(want_mask)) & (LATENT_FIFO - 1)) * 4u);
                invalid = false;
                if constexpr (F64) {
                    // A = drift, dc, boundary | std_alpha, beta | mu_alpha (raw); single: B.z = beta, latent = the accepted normal
                    double beta64;
                    if constexpr (MODEL == NDDM_BASIC_DDM_DC) { ab = (double)a2; beta64 = (double)a3; }
                    else {
                        ab = (double)a3 + (double)a2 * (double)latent;
                        if (!(ab > 0.0)) ab = __builtin_fabs(ab);
                        beta64 = (double)__uint_as_float(rw[D_B + 2]);
                    }
                    xe = ab * beta64;                                               // basic_ddm_dc.py:91
                    cdt = (double)a0 * H->dt64;
                    sdc = H->sqrt_dt64 * (double)a1;
                    if constexpr (FAST) sdc *= 1.1774100225154747;                  // the fast radius is sqrt(-log2 u)
                } else if constexpr (MODEL == NDDM_BASIC_DDM_DC) {
                    mu_dt = a0; h = a2; w = a3;
                } else if constexpr (MODEL == NDDM_SINGLE_TRIAL) {
                    // A = drift*dt/S, 1/S, std_alpha, mu_alpha;  B = sigma1, gamma, beta
                    const float a = latent;                                         // per-trial boundary
                    const float hv = 0.5f * a;
                    mu_dt = a0;
                    h = hv * a1;
                    w = (a * __uint_as_float(rw[D_B + 2]) - hv) * a1;
                } else if constexpr (MODEL == NDDM_SINGLE_TRIAL_ALT) {
                    // A = drift, alpha, beta, std_dc;  B = mu_dc, sigma1, gamma
                    const float sig_c = latent;                                     // per-trial noise scale
                    const float inv_t = 1.0f / noise_unit<FAST>(H->sqrt_dt * sig_c);
                    const float hv = 0.5f * a1;
                    mu_dt = (a0 * H->dt) * inv_t;
                    h = hv * inv_t;
                    w = (a1 * a2 - hv) * inv_t;
                } else if constexpr (MODEL == NDDM_ALPHA_NOT_SCALED) {
                    // A = Nu, 1/S, a/(2S), w0: the per-trial drift N(Nu, Eta) was drawn when the tile opened
                    mu_dt = latent;
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
                k = 0;
                if constexpr (BRIDGE) { ta = h - __builtin_fabsf(w); tb = ta; jraw = 0u; }
            }
            if constexpr (LATENT) { const int n_ok = (int)__popcll(ok_mask); fifo_pos += n_ok; fifo_avail -= n_ok; }
            // fresh compares over all lanes (an invalid trial has h == 0 and is never in range; lanes without a trial
            // are masked by has_m)
            if constexpr (BRIDGE) act_m = __builtin_amdgcn_ballot_w64(ta > 0.0f) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
            else if constexpr (F64) act_m = __builtin_amdgcn_ballot_w64(xe > 0.0) & __builtin_amdgcn_ballot_w64(xe < ab) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
            else act_m = __builtin_amdgcn_ballot_w64(in_range(w, h)) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
        }
        // ------------------------------------------------------------ step phase
        // leave the loop for a refill once refill_thresh lanes hold a finished trial, or none is stepping, or after
        // MAX_BLOCKS blocks (so that a few finished lanes never wait long for company; a threshold >= 64 -- the lockstep
        // measurement -- switches that exit off)
        constexpr int MAX_BLOCKS = 16;
        const int it_limit = A.refill_thresh < WAVE ? MAX_BLOCKS - 1 : 0x7fffffff;      // (one integer, not a lane-mask pair)
        // leave when this many lanes hold a finished trial: the threshold, or all the lanes that hold one at all ("none is
        // stepping" is the same test: the occupied lanes do not change inside the step loop)
        const int occupied = (int)__popcll(has_m);
        const int leave_at = occupied < A.refill_thresh ? occupied : A.refill_thresh;
        int it = 0;
        for (;; ++it) {
            if constexpr (BRIDGE) {
                // ---- Brownian-bridge stepping: 8 steps per pass -- two blocks of the path stream (counters k, k + 4: the
                // stream of the plain kernel) and ONE block of crossing uniforms, whose four words serve two steps each: the first
                // step of a pair takes the whole word as a 32-bit uniform, the second its low 16 bits.
                // A step from w to w1 ends the trial -- boundary side = sign(w1) -- when
                //     u < exp(-2 d0 d1),   d0 = h - |w|,  d1 = h - |w1|    (noise units; 2^(-4 d0 d1) in the fast transform's),
                // the probability that a Brownian bridge between two points on the same side of the interval's centre touched
                // the nearer boundary; for d1 <= 0 (w1 outside: the crossing plain Euler-Maruyama sees) the right-hand side is
                // >= 1 > u, so ONE compare decides both -- and narrows EXEC itself (v_cmpx).  The far boundary's bridge
                // probability, <= exp(-2 h^2), is dropped: < 2^-23 as soon as the boundaries are 5.7 single-step standard
                // deviations apart, below which an Euler-Maruyama path is no approximation of anything.
                // d1 is the next step's d0 (two registers, used alternately); k counts the steps SURVIVED (incremented behind the compare).
                const uint32_t blk = (uint32_t)k;                        // a multiple of 8 in every lane that is stepping
                // the crossing uniforms: the PATH stream's own generator at draw index k | 2^31 (k < 2^22 with the bridge): no second
                // set of per-trial Philox constants to compute at every hand-out and to hold in five VGPRs
                const u32x4 u4 = philox4x32_10_path(blk | 0x80000000u, pc, kbase);
                // (no scheduling barrier here: the compiler interleaves the two generators)
                unsigned long long &live = act_m;        // the lanes still stepping: narrowed in place (SGPR pairs are scarce here)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const u32x4 rb = philox4x32_10_path(blk + 4u * (uint32_t)half, pc, kbase);
                    float r0, r1, c0, c1, c2, c3;
                    polar_pair<FAST, true>(rb.x, rb.y, r0, c0, c1);
                    polar_pair<FAST, true>(rb.z, rb.w, r1, c2, c3);
                    const uint32_t ua = half ? u4.z : u4.x, ub = half ? u4.w : u4.y;
                    if constexpr (FAST && CAP4) {
                        float m, uf;
                        // per step: w1; d1 (into the other distance register); d0*d1; 2^(K - 4 d0 d1) with K = 32 (+ 1 ulp, so that
                        // "outside" beats a uniform that rounds up to 2^32) or 16: the uniform's scale; the uniform's conversion
                        // (it also fills the slot a transcendental's result needs before a VALU instruction may read it on
                        // gfx950); compare -> EXEC; survivors count the step.  The lane mask goes in and comes out in one SGPR pair.
#define NDDM_BSTEP(R, T, D0, D1, CVT, KLIT) "v_fmac_f32 %[w], %[" R "], %[" T "]\n\tv_add_f32 %[w], %[mu], %[w]\n\tv_sub_f32 %[" D1 "], %[h], |%[w]|\n\t" \
    "v_mul_f32 %[m], %[" D0 "], %[" D1 "]\n\tv_fmaak_f32 %[m], -4.0, %[m], " KLIT "\n\tv_exp_f32 %[m], %[m]\n\t" CVT                        \
    "v_cmpx_ge_f32_e64 vcc, %[uf], %[m]\n\tv_add_u32 %[k], 1, %[k]\n\t"
#define NDDM_U32(W) "v_cvt_f32_u32 %[uf], %[" W "]\n\t"
#define NDDM_U16(W) "v_and_b32 %[uf], 0xffff, %[" W "]\n\tv_cvt_f32_u32 %[uf], %[uf]\n\t"
                        asm volatile("s_mov_b64 exec, %[lm]\n\t"
                                     "v_mov_b32 %[jr], %[ua]\n\t"
                                     NDDM_BSTEP("r0", "c0", "ta", "tb", NDDM_U32("ua"), "0x42000001")
                                     NDDM_BSTEP("r0", "c1", "tb", "ta", NDDM_U16("ua"), "0x41800000")
                                     "v_mov_b32 %[jr], %[ub]\n\t"
                                     NDDM_BSTEP("r1", "c2", "ta", "tb", NDDM_U32("ub"), "0x42000001")
                                     NDDM_BSTEP("r1", "c3", "tb", "ta", NDDM_U16("ub"), "0x41800000")
                                     "s_nop 1\n\ts_mov_b64 %[lm], exec\n\ts_mov_b64 exec, -1"
                                     : [w] "+v"(w), [k] "+v"(k), [ta] "+v"(ta), [tb] "+v"(tb), [jr] "+v"(jraw), [lm] "+s"(live), [m] "=&v"(m),
                                       [uf] "=&v"(uf)
                                     : [mu] "v"(mu_dt), [h] "v"(h), [r0] "v"(r0), [r1] "v"(r1), [c0] "v"(c0), [c1] "v"(c1),
                                       [c2] "v"(c2), [c3] "v"(c3), [ua] "v"(ua), [ub] "v"(ub)
                                     : "vcc");
#undef NDDM_BSTEP
#undef NDDM_U32
#undef NDDM_U16
                    } else {
                        // the same steps in C: the exact transform (bit-reproducible on a CPU: the test suite restates it in plain C), and
                        // step caps that are not a multiple of 8 (tested per step)
                        bool active = __builtin_amdgcn_inverse_ballot_w64(live);
                        const float rr4[4] = {r0, r0, r1, r1}, tt4[4] = {c0, c1, c2, c3};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (active) {
                                asm volatile("" ::: "memory");
                                const uint32_t word = j < 2 ? ua : ub;
                                const float w1 = __builtin_fmaf(rr4[j], tt4[j], w) + mu_dt;
                                const float t1 = h - __builtin_fabsf(w1);
                                const float m = ((j & 1) ? tb : ta) * t1;
                                if (j & 1) ta = t1; else tb = t1;           // (written whether or not the trial goes on, as the asm does)
                                if ((j & 1) == 0) jraw = word;
                                const float uf = (j & 1) ? (float)(word & 0xffffu) : (float)word;
                                bool cross;
                                if constexpr (FAST) {
                                    const float kk = (j & 1) ? 16.0f : __uint_as_float(0x42000001u);
                                    cross = !(uf >= __builtin_amdgcn_exp2f(__builtin_fmaf(-4.0f, m, kk)));
                                } else {
                                    const float u = uf * ((j & 1) ? 1.52587890625e-05f : 2.3283064365386963e-10f);   // 2^-16, 2^-32
                                    cross = !(t1 > 0.0f) || u < exact_expf_neg(-2.0f * m);
                                }
                                w = w1;
                                if (cross) active = false;
                                else {
                                    k++;
                                    if constexpr (!CAP4) active = k < A.max_k;
                                }
                            }
                        }
                        live = __builtin_amdgcn_ballot_w64(active);
                    }
                }
                act_m &= __builtin_amdgcn_ballot_w64(k < A.max_k);
                if ((int)__popcll(has_m & ~act_m) >= leave_at) break;
                if (it >= it_limit) break;
                continue;
            }
            [[maybe_unused]] bool active = (CAP4 && !PACKED && !F64) ? false : __builtin_amdgcn_inverse_ballot_w64(act_m);
            // counter word 0 of the path stream = index of the block's first step (a multiple of NS: a lane only starts
            // a block after taking all NS steps of the previous one), so no shift is needed
            constexpr int NS = PACKED ? 8 : 4;
            const uint32_t blk = (uint32_t)k;
            u32x4 rb;
            if constexpr (VKEYS) rb = philox4x32_10_path(blk, pc, pkeys);
            else rb = philox4x32_10_path(blk, pc, kbase);
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
                polar_pair<FAST, true>(rb.x, rb.y, rr[0], tt[0], tt[1]);
                polar_pair<FAST, true>(rb.z, rb.w, rr[2], tt[2], tt[3]);
                rr[1] = rr[0]; rr[3] = rr[2];
            }
            if constexpr (F64) {
                // the reference's recurrence, one IEEE double operation per Python operation (this translation unit is compiled
                // with -ffp-contract=off): t1 = drift*dt, t2 = sqrt(dt)*dc (both per trial), t3 = t2 * normal, t4 = t1 + t3,
                // evidence = evidence + t4; then the loop condition of basic_ddm_dc.py:95
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    if (active) {
                        asm volatile("" ::: "memory");
                        const double zn = (double)rr[j] * (double)tt[j];        // exact: 24 x 24 bits
                        xe = xe + (cdt + sdc * zn);
                        k++;
                        if (j < NS - 1) {
                            if constexpr (CAP4) active = (xe > 0.0) && (xe < ab);
                            else active = (xe > 0.0) && (xe < ab) && (k < A.max_k);
                        }
                    }
                }
                act_m = __builtin_amdgcn_ballot_w64(xe > 0.0) & __builtin_amdgcn_ballot_w64(xe < ab) & __builtin_amdgcn_ballot_w64(k < A.max_k) & has_m;
                if ((int)__popcll(has_m & ~act_m) >= leave_at) break;
                if (it >= it_limit) break;
                continue;
            }
            if constexpr (CAP4 && !PACKED) {
                // The four steps with the execution mask narrowed by the range compare itself (v_cmpx writes EXEC): fmac, add,
                // k + 1, compare per step and not one scalar instruction in between.  As the compiler writes `if (active)` a
                // step carries four (s_and_saveexec, s_cbranch_execz, s_and, s_or), and a SIMD issues one scalar instruction
                // per ~4.2 cycles, only partly in the shadow of the vector ones (tools/ubench_salu).  Measured A/B on one box:
                // +0.5 % at the headline, +1.4 % single_trial, +1.9 % alpha_not_scaled, +3 % at 60 trials per set, 0 at
                // dt=.01 -- and -1.8 % in lockstep, where no lane ever leaves (the compare-to-EXEC dependency costs there
                // what the scalar instructions cost elsewhere), -2 % with the 8-step packed layout, which keeps the C form.
                // The lanes still in range afterwards are EXEC itself.  (EXEC is all ones here -- the step loop is
                // wave-uniform code -- and is left so.)
                unsigned long long still;
#define NDDM_STEP(R, T) "v_fmac_f32 %[w], %[" R "], %[" T "]\n\tv_add_f32 %[w], %[mu], %[w]\n\tv_add_u32 %[k], 1, %[k]\n\tv_cmpx_lt_f32_e64 vcc, |%[w]|, %[h]\n\t"
                asm volatile("s_mov_b64 exec, %[am]\n\t"
                             NDDM_STEP("r0", "t0") NDDM_STEP("r0", "t1") NDDM_STEP("r1", "t2") NDDM_STEP("r1", "t3")
                             "s_nop 1\n\ts_mov_b64 %[st], exec\n\ts_mov_b64 exec, -1"
                             : [w] "+v"(w), [k] "+v"(k), [st] "=s"(still)
                             : [am] "s"(act_m), [mu] "v"(mu_dt), [h] "v"(h), [r0] "v"(rr[0]), [r1] "v"(rr[2]),
                               [t0] "v"(tt[0]), [t1] "v"(tt[1]), [t2] "v"(tt[2]), [t3] "v"(tt[3])
                             : "vcc");
#undef NDDM_STEP
                act_m = still & __builtin_amdgcn_ballot_w64(k < A.max_k);      // (still is a subset of the lanes that stepped)
                if ((int)__popcll(has_m & ~act_m) >= leave_at) break;
                if (it >= it_limit) break;
                continue;
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (active) {
                    // keep this a real exec-masked region: selects through SGPR masks (v_cndmask_e64, ~4.2 cycles
                    // each on gfx950) cost more VALU issue than the predicated add / count they would replace
                    asm volatile("" ::: "memory");
                    w = __builtin_fmaf(rr[j], tt[j], w) + mu_dt;
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
            if ((int)__popcll(has_m & ~act_m) >= leave_at) break;
            if (it >= it_limit) break;
        }
        // one refill phase of `it + 1` blocks
        dbg_cnt += (1ull << 32) | (unsigned long long)(uint32_t)(it + 1);
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
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime(), c_end = __builtin_amdgcn_s_memtime();
        const uint32_t wg = lds_raw[31];
        if (wg < (uint32_t)fresh_args(Ak)->dbg_waves) {
            unsigned long long *const r = dbg + 8ull * wg;
            r[0] = dbg_cnt & 0xffffffffull; r[1] = dbg_cnt >> 32;
            r[2] = c_end - dbg_stamp[0]; r[3] = t_end - dbg_stamp[1];
            r[4] = dbg_stamp[1]; r[5] = dbg_stamp[3]; r[6] = t_end; r[7] = 1ull;
        }
    }
}

}  // namespace nddm

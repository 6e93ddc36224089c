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
//   (then accepted), 4096 spheres (then ended on the nearer boundary).  The FAST mode evaluates the same acceptance function without
//   the loop: three terms of the series or of its Jacobi-dual form, whichever converges (see the attempt below).  Laid out for
//   64-lane waves:
//   * a single-wave workgroup works on a GROUP of consecutive tiles (a tile = one parameter set of <= 512 trials; ~1000 trials per
//     group) with PERSISTENT LANES: the trial loop, the sphere loop and the rejection loop of the reference are flattened into ONE
//     loop whose trip is one rejection attempt, and a lane whose trial ended takes the group's next trial at the top of the next trip
//     (ballot + mbcnt; its set's constants come from a small LDS table), so the three data-dependent loop counts (1-10 spheres, 1-5
//     attempts, 2-18 series terms) do not multiply into idle lanes, and the drain at the end -- lanes idle while the last trials
//     finish -- is paid once per ~1000 trials instead of once per set (measured: 67 trials' worth; 20 % of the time at 300 trials
//     per set, 67 % at the 100 trials per participant of alpha_not_scaled.py:55);
//   * randomness: counter-based like every other stream of the library -- the per-trial drift is auxiliary normal 0 of the trial
//     (stream 1: the very draw the Euler-Maruyama form of the model uses; drawn for 64 slots at a time by all lanes into an LDS FIFO,
//     the way the simulator draws its per-trial latents), the uniforms are stream 3 of the trial, consumed in order;
//   * results are staged in LDS as one float per trial (the decision time with the response as its sign bit) and flushed as whole
//     float2 (y, acc) lines with the fused summary reduction (integer sums of the decision time in 2^-16 s: bit-reproducible).
// The path is VALU-issue bound (per trial ~2.6 attempts, ~1.8 spheres, ~3.2 Philox blocks; the vector pipe is saturated at an exec-mask
// utilisation of 0.70: profiles/r6_ratcliff_summary.md, tools/ratcliff_isa_mix.py); 8 B are written per trial.
#pragma once
#include "nddm_sim.h"

namespace nddm {

constexpr int RATCLIFF_MAX_TERMS = 64, RATCLIFF_MAX_ATTEMPTS = 4096, RATCLIFF_MAX_SPHERES = 4096;
constexpr int RATCLIFF_FIFO = 128;          // entries of the drift-normal FIFO (a power of two >= 2 * WAVE)
constexpr int RATCLIFF_KEYS = 20;           // dwords of the Philox round-key table in LDS

struct RatArgs {
    const float *params;            // [B, 6]: Nu, Alpha, Beta, Tau, Eta, Varsigma
    float *out_trials;              // [B, N, 2] (y, acc) or null
    float *out_summary;             // [B, K] or null (tiles_per_set == 1: written here; else through partials + combine_partials_kernel)
    float *out_ext;                 // [B] or null
    unsigned long long *partials;   // [B * tiles_per_set, 5] or null
    long long n_vsets;              // B * tiles_per_set
    int n_trials;                   // SLOTS per tile: its trials, at least 2 (a one-trial set has a hole)
    int n_total;                    // trials per set
    int tiles_per_set;
    uint32_t k0, k1;
    unsigned long long set_offset;
    float ext_sigma;
    int ext_mode;
    int group;                      // tiles one workgroup works on at a time (their trials are handed out as ONE sequence)
    uint32_t tile_magic;            // ceil(2^32 / n_trials): slot / n_trials by a multiply-high (slots < 2^16)
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

// a / b: IEEE division in the exact mode (the checker's arithmetic), v_rcp_f32 (1 ulp) and a multiply in the fast one -- a sphere has
// five quotients, ~10 instructions each as IEEE divisions
template <bool FAST> __device__ __forceinline__ float rat_div(float a, float b)
{
    if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}

// uniform j of (set, trial): word j & 3 of block j >> 2 of stream 3.  A lane keeps its uniforms in an 8-word LDS ring of its own
// (uniform j sits at ring word j & 7): refill() -- the one place a Philox block of this stream is generated -- runs at the top of a
// trip of the kernel's loop and makes sure at least FOUR uniforms are ahead of the lane, what a trip can consume (one for a sphere's
// direction when the trial was just handed out, two for the attempt, one for the next sphere's direction): ONE block whenever fewer
// than four are left -- also the first block of a trial handed out in this trip, so new and running trials share the instruction
// stream.  (Generating the block inside next() put five inlined Philox bodies into the loop: 106 SGPRs, ten spilled to scratch.)
struct UnifStream {
    uint32_t set_lo, trial, c2, q, gen;                  // q: uniforms consumed, gen: blocks generated
    uint32_t *ring;                                      // this lane's 8 words in LDS
    __device__ __forceinline__ void init(uint32_t set_lo_, uint32_t set_hi28, uint32_t trial_)
    {
        set_lo = set_lo_; trial = trial_; c2 = set_hi28 | 0x30000000u; q = 0u; gen = 0u;
    }
    // KEYS: AllKeys (round keys in 20 VGPRs) or the LDS byte address of the key table -- see the kernel
    template <typename KEYS>
    __device__ __forceinline__ void refill(const KEYS &K)
    {
        if (4u * gen - q < 4u) {                         // (block gen goes where block gen - 2 was: q > 4 (gen - 1), so that one is consumed)
            u32x4 x;
            if constexpr (sizeof(KEYS) == sizeof(uint32_t)) x = philox4x32_10_lds(set_lo, trial, c2, gen, K);
            else x = philox4x32_10_vkeys(set_lo, trial, c2, gen, K);
            *reinterpret_cast<uint4 *>(ring + (gen & 1u) * 4u) = make_uint4(x.x, x.y, x.z, x.w);
            gen++;
        }
    }
    __device__ __forceinline__ float next()
    {
        const uint32_t w = ring[q & 7u];
        q++;
        return uniform01(w);
    }
};

// sum of a 64-bit value < 2^60 over the 64 lanes: three DPP reductions of 20-bit pieces (each total < 2^26); lane 63 holds the total
__device__ __forceinline__ unsigned long long wave_sum_dpp64(unsigned long long v)
{
    const uint32_t s0 = wave_sum_dpp((uint32_t)v & 0xfffffu), s1 = wave_sum_dpp((uint32_t)(v >> 20) & 0xfffffu), s2 = wave_sum_dpp((uint32_t)(v >> 40));
    return (unsigned long long)s0 + ((unsigned long long)s1 << 20) + ((unsigned long long)s2 << 40);
}

// per-tile constants in LDS (RT_WORDS dwords per tile of the group), written by lane l for tile l when a group opens
enum { RT_NU = 0, RT_ETA, RT_INVD, RT_CLAM2, RT_DU0, RT_DL0, RT_TAU, RT_ALPHA, RT_SETLO, RT_SETHI, RT_NHERE, RT_T0, RT_WORDS };

template <bool FAST>
__global__ __launch_bounds__(WAVE) void ratcliff_kernel(const RatArgs A)
{
    extern __shared__ uint32_t lds_raw[];
    const int G = A.group;                                             // tiles per workgroup pass (<= 64)
    // LDS: table | uniform rings | drift FIFO | round keys | staged results.  The table comes FIRST: a field of tile i is then at
    // 48 i + a constant that fits the LDS instructions' offset field (behind the rings it took an address add per read)
    uint32_t *const tbl = lds_raw;                                     // [G][RT_WORDS]
    uint32_t *const rings = tbl + G * RT_WORDS;                        // [WAVE][8]  (16-byte aligned: RT_WORDS is a multiple of 4)
    float *const zfifo = reinterpret_cast<float *>(rings + WAVE * 8);  // [RATCLIFF_FIFO]: the drift normals of the next slots, drawn 64 at a time
    uint32_t *const keys = rings + WAVE * 8 + RATCLIFF_FIFO;           // [RATCLIFF_KEYS]: the ten Philox round-key pairs
    // The round keys are wave-uniform, but an SGPR operand costs VALU issue time on gfx950 (v_xor_b32 with one: 4.1 cycles against 2.4;
    // the pipe is what bounds this kernel): as in the simulator's step loop they are read from LDS as broadcasts, one round ahead, and
    // folded in by v_bitop3_b32 (philox4x32_10_lds) -- 20 three-input XORs per block instead of 20 + 20 two-input ones, half of them
    // with an SGPR: 82 issue cycles per block less.  kbase: the table's LDS byte address in a VGPR (opaque: one register + immediates).
    if (threadIdx.x < 10) { keys[2 * threadIdx.x] = A.k0 + threadIdx.x * PHILOX_W0; keys[2 * threadIdx.x + 1] = A.k1 + threadIdx.x * PHILOX_W1; }
    uint32_t kbase;
    {
        const uint32_t off = (uint32_t)(size_t)keys;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kbase) : "s"(off));
    }
    // ... and the loop's own block (the uniforms) takes them from 20 VGPRs in the fast mode (56 -> 76 VGPRs: six waves per SIMD either
    // way, the SGPRs' bound; 5.06 -> 4.91 ms), from LDS in the exact mode (62 -> 82 VGPRs would cost its sixth wave: 10.35 vs 10.69 ms)
    [[maybe_unused]] AllKeys VK;
    if constexpr (FAST) VK.init(A.k0, A.k1);
    float *const staged = reinterpret_cast<float *>(keys + RATCLIFF_KEYS);        // [G][n_trials]: copysign(decision time, response)
    const int lane = threadIdx.x;
    const long long n_groups = (A.n_vsets + G - 1) / G;
    for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const long long v0 = grp * G;
        const int g_here = (int)((A.n_vsets - v0) < G ? (A.n_vsets - v0) : G);
        const int TPS = A.tiles_per_set;
        __syncthreads();                                               // (the previous group's flush has read the table)
        // ---- per-tile constants: lane l states tile l's
        if (lane < g_here) {
            const long long vset = v0 + lane;
            const long long set = TPS == 1 ? vset : vset / TPS;
            const int t0 = TPS == 1 ? 0 : (int)(vset - set * TPS) * A.n_trials;
            const int n_here = (A.n_total - t0) < A.n_trials ? (A.n_total - t0) : A.n_trials;
            const float *p = A.params + set * 6;
            float Nu = p[0];
            if (Nu < -5.0f || Nu > 5.0f) Nu = Nu > 0.0f ? 5.0f : -5.0f;                // pyhddmjagsutils.py:102-103
            const float Alpha = p[1], Beta = p[2], Tau = p[3], Eta = p[4], Vs = p[5];
            const float D = (Vs * Vs) * 0.5f, inv_D = 1.0f / D;                         // :117
            const float c_lam2 = (0.25f * D) * 9.86960440108935862f;                    // 0.25 D pi^2
            const float zz = Beta * Alpha, du0 = Alpha - zz, dl0 = zz;
            const unsigned long long gset = A.set_offset + (unsigned long long)set;
            uint32_t *t = tbl + lane * RT_WORDS;
            t[RT_NU] = __float_as_uint(Nu); t[RT_ETA] = __float_as_uint(Eta); t[RT_INVD] = __float_as_uint(inv_D);
            t[RT_CLAM2] = __float_as_uint(c_lam2); t[RT_DU0] = __float_as_uint(du0); t[RT_DL0] = __float_as_uint(dl0);
            t[RT_TAU] = __float_as_uint(Tau); t[RT_ALPHA] = __float_as_uint(Alpha);
            t[RT_SETLO] = (uint32_t)gset; t[RT_SETHI] = (uint32_t)(gset >> 32) & 0x0fffffffu;
            t[RT_NHERE] = (uint32_t)n_here; t[RT_T0] = (uint32_t)t0;
        }
        __syncthreads();

        // ---- per-lane trial state
        bool has = false, need_sphere = false;           // need_sphere: the trial's next sphere -- its first, after the hand-out, or the one
                                                          // a step away from the boundary leads to -- is set up at the top of the next trip:
                                                          // ONE site for both (as two inlined sites each ran for a quarter of the lanes)
        int slot = 0;                                     // index of the trial's staged result: tile * n_trials + trial within the tile
        float du = 0.0f, dl = 0.0f, total = 0.0f, lam1 = 0.0f, g1 = 0.0f, x1 = 0.0f, lam = 1.0f, F = 0.0f, c_lam2 = 0.0f;
        bool up = false;
        int sphere = 0, att = 0;
        UnifStream us;
        us.init(0u, 0u, 0u);
        us.ring = rings + lane * 8;
        int next = 0;                                     // wave-uniform: the group's next unassigned slot
        int drawn = 0;                                    // wave-uniform: slots [next, drawn) have their drift normal in the FIFO
        const int n_slots = g_here * A.n_trials;
        // slot -> (tile, trial within the tile); a slot beyond its tile's n_here is a hole
        auto slot_tile = [&](int idx) { return (int)__umulhi((uint32_t)idx, A.tile_magic); };      // (n_trials >= 2: see the launch)

        // the sphere that starts at the current position: its constants and its direction (one uniform) -- or the end of the
        // trial, when the position lies on a boundary (Beta = 0 or 1) or the safety cap is reached; returns "the trial goes on"
        auto setup_sphere = [&]() -> bool {
            // (straight-line but for the store: every level of nested divergence costs the loop ~10 scalar instructions of exec-mask
            //  bookkeeping and copies of the loop-carried state at its edges -- tools/ratcliff_isa_mix.py; a trial that ends computes a
            //  sphere nobody uses)
            float radius;                                 // min(du, dl): ONE v_min_f32 (fminf() quiets its operands first: two more instructions)
            asm("v_min_f32 %0, %1, %2" : "=v"(radius) : "v"(du), "v"(dl));
            const bool dead = !(radius > 0.0f) | (sphere >= RATCLIFF_MAX_SPHERES);
            if (dead) staged[slot] = copysignf(total, du <= dl ? 1.0f : -1.0f);
            lam = lam1 + rat_div<FAST>(c_lam2, radius * radius);                         // :138
            const float Gr = radius * g1;
            F = rat_div<FAST>(1.0f, __builtin_fmaf(Gr, Gr, 1.0f));                       // :140-141, F0^2 / (1 + F0^2) with F0 = 1 / G
            const float x = radius * x1;
            const float e = rat_exp_neg<FAST>(__builtin_fabsf(x));
            const float p_up = rat_div<FAST>((x >= 0.0f) ? 1.0f : e, 1.0f + e);          // :143-144: 1 / (1 + e) or e / (1 + e)
            up = us.next() < p_up;                                                       // :145
            att = 0;
            return !dead;
        };

        while (true) {
            // ---- hand out the group's next trials to the lanes that hold none (the slots of a tile are its trials; a tile
            // shorter than n_trials -- the last tile of a split set -- leaves holes that are skipped)
            const unsigned long long want = __builtin_amdgcn_ballot_w64(!has);
            if (next < n_slots && want) {
                const int n_want = (int)__popcll(want);
                if (next + n_want > drawn && drawn < n_slots) {
                    // the drift normals of the next 64 slots, by ALL lanes (auxiliary normal 0 of each slot's trial: one Philox block
                    // and one Box-Muller pair per trial, ~95 instructions whatever the number of lanes a hand-out serves)
                    // (straight-line: a lane beyond the group's last slot redoes that slot -- the same value to the same entry -- and a
                    //  hole's normal is drawn and never read; nested conditions cost scalar bookkeeping and save nothing here)
                    const int ds = drawn + lane < n_slots ? drawn + lane : n_slots - 1;
                    const int tile = slot_tile(ds);
                    const uint32_t *t = tbl + __mul24(tile, RT_WORDS);
                    const int tr = ds - __mul24(tile, A.n_trials);
                    float z0, z1_unused;
                    AuxStream<FAST> aux(kbase, t[RT_SETLO], t[RT_SETHI], t[RT_T0] + (uint32_t)tr);
                    aux.first_pair(z0, z1_unused);
                    zfifo[ds & (RATCLIFF_FIFO - 1)] = z0;
                    drawn = drawn + WAVE < n_slots ? drawn + WAVE : n_slots;
                    __builtin_amdgcn_wave_barrier();          // (the hand-out below reads other lanes' entries: a wave's LDS operations complete in order)
                }
                const int idx = next + (int)lane_rank(want);
                {
                    // (one condition, evaluated by every lane on a clamped slot: the tile's trial count is read before it is known
                    //  whether this lane takes a trial)
                    const int idc = idx < n_slots ? idx : n_slots - 1;
                    const int tile = slot_tile(idc);
                    const int tr = idc - __mul24(tile, A.n_trials);
                    const uint32_t *t = tbl + __mul24(tile, RT_WORDS);
                    if (!has & (idx < n_slots) & (tr < (int)t[RT_NHERE])) {
                        slot = idx;
                        const uint32_t trial = t[RT_T0] + (uint32_t)tr;
                        const float inv_D = __uint_as_float(t[RT_INVD]);
                        const float mu = __builtin_fmaf(__uint_as_float(t[RT_ETA]), zfifo[idx & (RATCLIFF_FIFO - 1)], __uint_as_float(t[RT_NU]));     // :124-125
                        lam1 = (0.25f * (mu * mu)) * inv_D;
                        g1 = mu * (inv_D * 0.318309886183790672f);
                        x1 = mu * inv_D;
                        c_lam2 = __uint_as_float(t[RT_CLAM2]);
                        du = __uint_as_float(t[RT_DU0]); dl = __uint_as_float(t[RT_DL0]); total = 0.0f; sphere = 0;
                        us.init(t[RT_SETLO], t[RT_SETHI], trial);
                        has = true; need_sphere = true;
                    }
                }
                next = next + n_want < n_slots ? next + n_want : n_slots;
            }
            if (has) { if constexpr (FAST) us.refill(VK); else us.refill(kbase); }
            if (need_sphere) { has = setup_sphere(); need_sphere = false; }
            if (!__builtin_amdgcn_ballot_w64(has)) {
                if (next >= n_slots) break;
                if constexpr (!FAST) continue;            // (every lane that took a trial ended it at once, or met a hole; the fast mode's
            }                                             //  attempt runs for idle lanes anyway)
            // ---- one rejection attempt of every lane that holds a trial (:147-159).  In the fast mode the lanes that hold none run
            // along (no region, no copies of the loop-carried state at its edges): what they compute is never observed -- their state
            // is set when they are handed a trial -- and `accept` is false for them.  The exact mode's series loop must not see their
            // garbage (it could run to its cap), so there the region stays.
            if (FAST || has) {
                const float s2 = us.next(), s1 = us.next();
                const float nl = rat_neg_log<FAST>(s1);
                const float a = F * nl;
                bool accept = false;
                att++;
                if constexpr (FAST) {
                    const float ea = rat_exp_neg<FAST>(a);
                    // The test is s2 e^-a <= theta(a), theta(a) = sum_{k = 1, 3, 5, ..} (-1)^((k-1)/2) k e^{-a k^2} = eta(4 a i / pi)^3.
                    // The reference's series needs up to 18 terms when a is small -- and a wave waits for its smallest a.  The
                    // modular transformation of eta (Jacobi's imaginary transformation) states the same function as
                    //   theta(a) = (pi / 4a)^{3/2} theta(pi^2 / 16a),
                    // a series in e^{-pi^2 k^2 / 16a} that converges the faster the smaller a is; the two meet at a = pi / 4, where the
                    // fourth term of either is 7 e^{-49 pi / 4} ~ 1e-16 of the first: THREE terms of whichever series a calls for
                    // are theta(a) to float32, with no loop and the same instructions for every lane.
                    const bool dual = a < 0.785398163f;
                    const float rs = __builtin_amdgcn_rsqf(a), ra = rs * rs;
                    const float E = dual ? rat_exp_neg<FAST>(0.616850275f * ra) : ea;          // e^{-c}, c = pi^2 / 16a or a
                    const float E2 = E * E, E4 = E2 * E2, E8 = E4 * E4;
                    const float P = __builtin_fmaf(E8, __builtin_fmaf(5.0f * E8, E8, -3.0f), 1.0f);    // 1 - 3 e^{-8c} + 5 e^{-24c}
                    // dual: s2 e^-a <= (pi/4)^{3/2} a^{-3/2} e^{-c} P;  else: s2 e^-a <= e^-a P  (selects, not branches; an a below
                    // 2^-6 -- a = 0 gives NaNs here -- is rejected by the last line, as the exact mode rejects it before its series)
                    const float lhs = s2 * (dual ? ea : 1.0f);
                    const float rhs = P * (dual ? (0.696040999f * (rs * ra)) * E : 1.0f);
                    accept = has & ((att >= RATCLIFF_MAX_ATTEMPTS) | (!(a < 0.015625f) & (lhs <= rhs)));
                } else {
                    if (att >= RATCLIFF_MAX_ATTEMPTS) accept = true;
                    else if (!(a < 0.015625f)) {
                        const float ea = rat_exp_neg<FAST>(a);
                        // (k = 3, 5, 7, .. and the alternating sign are carried as floats: k + 2 and -sgn are exact, and fma(+-1, term, told)
                        //  rounds exactly as told -+ term does -- the same bits as the checker's integer counter and its two branches)
                        float tnew = 0.0f, told, k = 1.0f, sgn = 1.0f;
                        int uu = 0;
                        do {
                            told = tnew;
                            uu++;
                            k += 2.0f;
                            sgn = -sgn;
                            const float term = k * rat_exp_neg<FAST>(a * (k * k));
                            tnew = __builtin_fmaf(sgn, term, told);
                        } while (tnew != told && uu < RATCLIFF_MAX_TERMS);
                        accept = s2 * ea <= ea + tnew;
                    }
                }
                if (accept) {
                    total += rat_div<FAST>(nl, lam);                                     // :161-163
                    // the distances ahead of and behind the step: the nearer boundary is reached when the one ahead is the smaller
                    // (:165-172: du <= dl going up, dl <= du going down) -- and then it is the radius, else the one behind is
                    const float ahead = up ? du : dl, behind = up ? dl : du;
                    const bool hit = ahead <= behind;
                    if (hit) staged[slot] = copysignf(total, up ? 1.0f : -1.0f);
                    // else the position moves by the radius (:174-175); a trial that ended moves too, unobserved
                    const float radius = hit ? ahead : behind;                           // = min(du, dl)
                    const float d = up ? radius : -radius;
                    du -= d; dl += d;
                    sphere++;
                    has = !hit; need_sphere = !hit;
                }
            }
        }
        __syncthreads();
        // ---- flush, tile by tile: whole float2 lines + the fused summary's integer sums (decision time in 2^-16 s), reduced over the
        // wave by DPP (as shuffles the five sums are 54 ds_bpermute round trips per tile; as LDS atomics on one address 200 cycles of
        // the CU's LDS pipe each: 11.5 ms instead of 4.3) and parked in LDS -- lane `tile`'s uniform ring: nobody draws any more --
        // for the group's epilogue.  [0] sum k + (n_upper << 40)  [1] sum k^2  [2] sum k (upper)  [3] sum k^2 (upper);  k < 2^26
        for (int tile = 0; tile < g_here; ++tile) {
            const uint32_t *t = tbl + tile * RT_WORDS;
            const long long vset = v0 + tile;
            const long long set = TPS == 1 ? vset : vset / TPS;
            const int n_here = (int)t[RT_NHERE], t0 = (int)t[RT_T0];
            const float Tau = __uint_as_float(t[RT_TAU]);
            const float *st = staged + tile * A.n_trials;
            float2 *out = A.out_trials ? reinterpret_cast<float2 *>(A.out_trials) + set * A.n_total + t0 : nullptr;
            int n_up = 0;
            unsigned long long sk = 0, sk2 = 0, sk_up = 0, sk2_up = 0;
            for (int j = lane; j < n_here; j += WAVE) {
                const float sv = st[j];
                const bool upper = (__float_as_uint(sv) >> 31) == 0u;
                const float tot = __builtin_fabsf(sv);
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
                n_up = (int)wave_sum_dpp((uint32_t)n_up);
                sk = wave_sum_dpp64(sk); sk2 = wave_sum_dpp64(sk2); sk_up = wave_sum_dpp64(sk_up); sk2_up = wave_sum_dpp64(sk2_up);
                if (lane == WAVE - 1) {                                // (the DPP reductions leave the totals in the last lane)
                    unsigned long long *q = reinterpret_cast<unsigned long long *>(rings + tile * 8);
                    q[0] = sk | ((unsigned long long)n_up << 40);                                  // (sk < 2^36, n_up <= 512)
                    q[1] = sk2; q[2] = sk_up; q[3] = sk2_up;
                }
            }
        }
        // ---- the per-set epilogue of the whole group at once, lane l for tile l: the summary row from the tile's integer sums (seven
        // float64 divisions, ~300 instructions -- by ONE lane after every tile they were a tenth of the kernel's instructions at 300
        // trials per set and a third at 100) and the set's external datum
        __builtin_amdgcn_wave_barrier();
        if (lane < g_here) {
            const uint32_t *t = tbl + lane * RT_WORDS;
            const long long vset = v0 + lane;
            const long long set = TPS == 1 ? vset : vset / TPS;
            const int n_here = (int)t[RT_NHERE];
            if (A.out_summary) {
                const unsigned long long *q = reinterpret_cast<const unsigned long long *>(rings + lane * 8);
                const int n_up = (int)(q[0] >> 40);
                const unsigned long long sk = q[0] & ((1ull << 40) - 1ull);
                if (A.partials) {                                      // a tile of a split set: combine_partials_kernel adds the tiles up
                    unsigned long long *w = A.partials + vset * 5;
                    w[0] = (unsigned long long)n_up | ((unsigned long long)(n_here - n_up) << 21);
                    w[1] = sk; w[2] = q[1]; w[3] = q[2]; w[4] = q[3];
                } else {
                    finalize_summary(A.out_summary + set * NDDM_SUMMARY_K, n_up, n_here - n_up, 0, sk, q[1], q[2], q[3], 0, 0, A.n_total,
                                     1.52587890625e-05f, __uint_as_float(t[RT_TAU]));
                }
            }
            if (A.out_ext && t[RT_T0] == 0u) {
                float z0, z1_unused;
                AuxStream<FAST> aux(kbase, t[RT_SETLO], t[RT_SETHI], 0xffffffffu);                      // the set's external datum (alpha_not_scaled.py:103-106)
                aux.first_pair(z0, z1_unused);
                A.out_ext[set] = __builtin_fmaf(A.ext_sigma, z0, (A.ext_mode == 0) ? __uint_as_float(t[RT_ALPHA]) : 1.0f);
            }
        }
    }
}

}  // namespace nddm

/*
 * nddm.h -- C ABI of libnddm_hip.so: the MI355X (gfx950) Euler-Maruyama drift-diffusion
 * trial simulators that replace the numba/NumPy simulators of mdnunez/bayesflow_nddms.
 *
 * Each entry point is what a ctypes (or cffi / cgo / JNI) binding for the reference's
 * simulator path would bind.  File:line citations are relative to the reference repo.
 *
 * Conventions (all entry points):
 *   - every data pointer is a DEVICE pointer (hipMalloc / torch ROCm tensor .data_ptr());
 *     the caller allocates and frees all buffers; the library keeps no pointer after the
 *     call's work on `stream` has completed.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous with respect to the host: they enqueue work on `stream` and return.
 *     The handle is VALIDATED at entry (ABI 3): a destroyed stream, a stream created by another copy of the HIP runtime
 *     in the process, or an integer that never was a stream is refused with NDDM_ERR_HIP before anything is enqueued,
 *     and a stream of another device than the current one with NDDM_ERR_PARAM.  (HIP itself dereferences such a handle
 *     in hipStreamIsCapturing / hipEventRecord / the launch calls: a SIGSEGV in the caller's thread.)
 *   - parameter rows are row-major float32 [B, P] in the REFERENCE'S parameter order.
 *   - trial output is row-major float32 [B, n_trials, 2]; summary output float32 [B, NDDM_SUMMARY_K].
 *     Either may be NULL (summary-only mode never writes the 8 bytes per trial).
 *   - randomness is a pure function of (seed, set_offset + row, trial, draw): output does not
 *     depend on launch geometry, on how rows are sharded over GPUs, or on call order.
 *   - timeouts are data (choice 0 / choicert 0), never errors.
 *   - return value: NDDM_OK or an nddm_status; nddm_last_error() gives thread-local text.
 *   - re-entrant and thread-safe given distinct output buffers: any number of host threads may call on any streams
 *     concurrently.  The device memory a launch needs besides the caller's buffers (work-queue words, scratch) is owned by
 *     that launch until the work it enqueued has COMPLETED (reuse is keyed on stream order or on a completion event, never
 *     on a launch count); the developer knobs (nddm_set_tuning, nddm_set_debug_trace) are read once, atomically, at entry.
 *   - hipGraph: a call made while `stream` is capturing is recorded as kernels only; the memory such a launch needs is
 *     allocated for that launch alone and belongs to the GRAPH ARENA bound to the capturing thread (nddm_graph_arena_*;
 *     released with its owner, and with nothing else) or, with no arena bound, to an ownerless list that
 *     nddm_release_graph_memory() frees.  Seed and set_offset are baked in;
 *     nddm_simulate_indirect / nddm_draw_prior_indirect add a 64-bit offset read from DEVICE memory when the kernels run,
 *     so a replayed graph moves along the random stream (a captured `*dev += B` between replays).
 *
 * Deliberate deviations from the boundary sketched for this path:
 *   - arithmetic is float32 on the device (state w, the Gaussian transform) with an INTEGER step index, so
 *     rt = k*dt + tau is exact in k; the reference integrates in float64.  Parity with the reference is therefore
 *     distributional (KS < 0.01 on the integer step grid), and bit-for-bit only against the float32 oracle (oracle/).
 *     The deviation as a number: fed the SAME normals, the float32 integrator and the reference's float64 recurrence
 *     (basic_ddm_dc.py:91-103) end a trial on another (step index, choice) in 7.6e-5 of 2 000 000 prior-mixture trials at
 *     dt=.001 / max_steps 4000 (1.5e-4 on the 18 fixed parameter sets) and in 4e-6 at the reference default dt=.01 / 400;
 *     the choice differed in none of them; a differing trial is a grazed boundary one arithmetic counts as crossed and the
 *     other does not, after which it runs on (largest step-index difference seen: 652)
 *     (oracle/ddm_oracle.c: oracle_philox_simulate_f64; tests/test_oracle_golden.py::test_f32_integrator_against_the_reference_f64_recurrence).
 *     Since ABI 4 the deviation is an option rather than a given: with NDDM_STATE_F64 (enum nddm_flags) the basic and single-trial
 *     simulators carry the evidence in float64 exactly as the reference's recurrence does, and their (step, choice) equal that
 *     function's bit for bit.
 *     Parameters are taken as float32 [B, P]; there is no float64 input form.
 *   - there are no `*_cpu` twins in this library: the CPU restatement of the same stream is test infrastructure
 *     (oracle/ddm_oracle.c) and is never linked into, or reachable from, the product.  Without a ROCm device every entry
 *     point fails with NDDM_ERR_HIP / NDDM_ERR_NO_DEVICE.
 */
#ifndef NDDM_H
#define NDDM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the default Gaussian stream takes the angle from the LOW 23 bits of its word (round 2), the bridge-uniform stream
 *    serves 8 steps per block (round 3), nddm_set_debug_counters became nddm_set_debug_trace, and the *_indirect entry
 *    points, nddm_simulate_codes / nddm_decode_codes and nddm_source_hash exist.  The same (seed, set_offset) gives different bits under ABI 1. */
/* 3: memory behind captured launches has an owner (nddm_graph_arena_create / _bind / _info / _release); nddm_release_graph_memory
 *    frees only what was captured with no arena bound (up to ABI 2: every captured launch's memory on the device, whoever's graph
 *    replayed into it); a `stream` the HIP runtime does not know is refused with NDDM_ERR_HIP (validated with hipStreamGetDevice at entry).  The random stream is ABI 2's. */
/* 4: NDDM_STATE_F64 (flag 8), nddm_build_info, nddm_simulratcliff.  The random stream and every ABI-3 entry point are unchanged. */
#define NDDM_ABI_VERSION 4
#define NDDM_SUMMARY_K 10

/* summary_stats[b, :] (single_trial_alpha_not_scaled.py:205-211,
 * simulations/mean_RT_accuracy_effects.py:88-90, simulations/Basic_DDM_simulations.py:137-145) */
enum nddm_summary_col {
    NDDM_S_N_UPPER = 0,       /* trials that hit the upper boundary (choice  1) */
    NDDM_S_N_LOWER = 1,       /* trials that hit the lower boundary (choice -1) */
    NDDM_S_N_MISSING = 2,     /* trials that reached max_steps (choice 0) */
    NDDM_S_MEAN_RT = 3,       /* mean RT (incl. non-decision time) of responded trials; NaN if none */
    NDDM_S_VAR_RT = 4,        /* population variance (ddof=0) of the same */
    NDDM_S_MEAN_RT_UPPER = 5, /* EZ-diffusion MRT: mean RT of upper-boundary trials; NaN if none */
    NDDM_S_VAR_RT_UPPER = 6,  /* EZ-diffusion VRT */
    NDDM_S_MEAN_Z = 7,        /* mean external datum z1 over all trials (0 for models without z1) */
    NDDM_S_VAR_Z = 8,         /* population variance of z1 */
    NDDM_S_CHOICE_MEAN = 9    /* mean(.5 + .5*sign(choicert)) over all trials */
};

enum nddm_model {
    NDDM_BASIC_DDM_DC = 0,     /* basic_ddm_dc.py:85-125; P=5: drift, boundary, beta, tau, dc */
    NDDM_SINGLE_TRIAL = 1,     /* single_trial_alpha_not_scaled.py:107-155 (+_scale :1237-1285, _scale2 :1471-1519);
                                  P=8: drift, mu_alpha, beta, ter, std_alpha, dc, sigma1, gamma (gamma=1 for the base model) */
    NDDM_SINGLE_TRIAL_ALT = 2, /* single_trial_alpha_not_scaled.py:926-974; P=8: drift, alpha, beta, ter, std_dc, mu_dc, sigma1, gamma */
    NDDM_ALPHA_NOT_SCALED = 3, /* alpha_not_scaled.py:52-128 as an Euler-Maruyama process; P=6: Nu, Alpha, Beta, Tau, Eta, Varsigma */
    NDDM_EXPLICIT_BOUNDARY = 4 /* imputation_from_stahl_not_scaled.py:120-148; P=4: drift, beta, ter, dc; bounds[B, n_trials] */
};

enum nddm_status {
    NDDM_OK = 0,
    NDDM_ERR_NULL = 1,        /* a required pointer is NULL */
    NDDM_ERR_SHAPE = 2,       /* B < 0, n_trials <= 0, max_steps < 0, or B * ceil(n_trials / 512) >= 2^31 */
    NDDM_ERR_PARAM = 3,       /* non-finite or non-positive dt; unknown model / flag */
    NDDM_ERR_HIP = 4,         /* a HIP runtime call failed; text in nddm_last_error() */
    NDDM_ERR_NO_DEVICE = 5
};

enum nddm_flags {
    NDDM_GAUSS_EXACT = 0,     /* Box-Muller from IEEE add/mul/fma/sqrt only: bit-reproducible on a CPU (oracle) */
    NDDM_GAUSS_FAST = 1,      /* Box-Muller on v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 */
    NDDM_GAUSS_PACKED = 4,    /* opt-in, with either transform, not with NDDM_BRIDGE, max_steps < 2^14: one 32-bit Philox word per
                                 Box-Muller pair (16-bit radius uniform + 16-bit angle) instead of two, i.e. 8 normals per
                                 Philox block instead of 4 -- ~25 % faster.  The path noise then has |z| <= 5.65 and a 2^-16
                                 grid in the radius; a different (equally reproducible) random stream than the default */
    NDDM_STATE_F64 = 8,       /* (ABI 4; NDDM_BASIC_DDM_DC and NDDM_SINGLE_TRIAL, not with NDDM_BRIDGE / NDDM_GAUSS_PACKED / the wire format)
                                 the REFERENCE'S state arithmetic: the evidence is a float64 in its natural units,
                                 evidence += drift*dt + sqrt(dt)*dc*normal (basic_ddm_dc.py:91-103; single_trial_alpha_not_scaled.py:113-128),
                                 every operation a separate IEEE double operation in the reference's order, the loop condition
                                 (evidence > 0) && (evidence < boundary) on doubles; `normal` is the exact double product of the float32
                                 Box-Muller radius and cosine / sine of the same random stream.  With NDDM_GAUSS_EXACT the (step index,
                                 choice) of every trial equals oracle_philox_simulate_f64's bit for bit (oracle/ddm_oracle.c), i.e. the
                                 float32-state deviation stated at the top of this file becomes a measured option; the single-trial
                                 model's boundary is formed in double from the same auxiliary normals, its external datum z1 stays the
                                 float32 expression.  Costs ~25-30 % of the rate (4 extra double operations per step). */
    NDDM_BRIDGE = 2           /* (alpha_not_scaled only) Brownian-bridge boundary correction: between two grid points
                                 inside (0, a) the path crosses a boundary with probability exp(-2 d0 d1 / (sigma^2 dt));
                                 removes the O(sqrt(dt)) late-detection bias of plain Euler-Maruyama against the exact
                                 first-passage sampler the reference uses for this model (pyhddmjagsutils.py:47-176);
                                 RTs get a uniform sub-step jitter, (k - U) dt, instead of k dt */
};

/* ---- library / device ------------------------------------------------------------- */
int nddm_abi_version(void);
const char *nddm_last_error(void);
int nddm_device_count(int *count);
int nddm_set_device(int device);
int nddm_summary_k(void);
int nddm_model_nparams(int model); /* P of enum nddm_model, -1 if unknown */
/* ---- memory behind launches captured into hipGraphs ---------------------------------------------------------------------
 * A captured launch cannot borrow the library's per-launch memory (the graph replays at times the library cannot see): it gets
 * an allocation of its own (two queue words + its scratch, <= 64 MB).  That allocation is charged to the graph arena that is
 * bound to the CAPTURING THREAD at the time of the call:
 *     uint64_t a;  nddm_graph_arena_create(&a);
 *     nddm_graph_arena_bind(a, &prev);  ... hipStreamBeginCapture / nddm_* calls / hipStreamEndCapture ...  nddm_graph_arena_bind(prev, NULL);
 *     ... replay the graphs ...;  destroy them;  nddm_graph_arena_release(a);
 * The caller of nddm_graph_arena_release asserts that every graph that captured a launch under THIS arena has been destroyed
 * (or will not be replayed); graphs captured under other arenas -- a second trainer (the checkpointed, re-entered training of
 * basic_ddm_dc.py:169-176, 199-207 keeps more than one alive in a notebook), a user's own graph -- are not affected.
 * The binding is per host thread; arena 0 = no owner.  A capture made while a RELEASED arena is still bound is refused
 * (NDDM_ERR_PARAM).  nddm_graph_arena_info: bytes / allocations currently charged to an arena (0 = the ownerless list). */
int nddm_graph_arena_create(uint64_t *arena);
int nddm_graph_arena_bind(uint64_t arena, uint64_t *previous /* may be NULL */);
int nddm_graph_arena_info(uint64_t arena, uint64_t *bytes /* may be NULL */, int32_t *n_allocations /* may be NULL */);
int nddm_graph_arena_release(uint64_t arena);
/* frees the OWNERLESS memory of the current device: what launches captured with no arena bound were given; call only when every
 * such graph has been destroyed.  Never touches an arena's memory. */
int nddm_release_graph_memory(void);
/* testing aid (not part of the drop-in surface): cap the number of launch slots per device, so that a test can drive the
 * library into queueing a launch behind an in-flight one */
int nddm_debug_set_slot_limit(int n);
/* developer aid: geometry of the calling thread's last simulator launch, out[8] = {grid waves, 1 if the kernel variant with
 * the Philox round keys in VGPRs ran, ring slots, trials per tile, tiles per set, sets per chunk, refill threshold, dynamic
 * LDS bytes}.  bench.py uses it to run its lockstep ceiling on the same kernel variant and grid as the timed workload. */
int nddm_debug_last_launch(int32_t *out8);
/* developer knobs of the launch geometry (0 = the library's rule): sets per queue chunk, ring slots (>= 2),
 * refill threshold (lanes holding a finished trial; >= 64 = refill only when no lane is stepping), kernel variant (1 =
 * Philox round keys from LDS, 2 = in VGPRs, taken modulo 3), grid in waves, trials per tile.  Results never depend on
 * them (tests/test_gpu_fuzz.py randomises all six). */
int nddm_set_tuning(int sets_per_chunk, int ring, int refill_thresh, int variant, int grid_waves, int tile_trials);
/* 0 switches the longest-first ordering pre-pass off (sets are processed in the given order) */
int nddm_set_ordering(int enabled);
/* profiling aid: while set, every wave of the simulator kernels stores -- plain stores, no atomics -- one 8-word record
 * at buf[8 w ..], w = workgroup < wave_capacity: {step-loop blocks, refill phases, s_memtime cycles, lifetime, start, "found
 * the queue empty", end in s_memrealtime ticks (100 MHz), 1}; and the tick at which chunk c < chunk_capacity was pulled
 * at buf[8 wave_capacity + c].  buf: device u64 [8 wave_capacity + chunk_capacity], zeroed by the caller; NULL = off. */
int nddm_set_debug_trace(void *dev_u64, int wave_capacity, int chunk_capacity);

/* ---- simulators --------------------------------------------------------------------
 * Common arguments:
 *   params      device f32 [B, P]
 *   B           number of parameter sets (rows)
 *   n_trials    trials per set (the batch-shared N of basic_ddm_dc.py:50-52, 131); any size: sets with more than
 *               512 trials are split into tiles internally, with results independent of the split
 *   dt          Euler-Maruyama step (reference default .01, basic_ddm_dc.py:87; fine study .001, single_trial_alpha_not_scaled.py:1719)
 *   max_steps   step cap (reference default 400; 4000 in the fine study)
 *   seed        64-bit stream key
 *   set_offset  global index of row 0 (makes shards of one logical batch reproducible); set_offset + row must stay
 *               below 2^60 (the random stream is keyed by the low 60 bits of the global set index)
 *   flags       enum nddm_flags
 *   out_trials  device f32 [B, n_trials, 2] or NULL
 *   out_summary device f32 [B, NDDM_SUMMARY_K] or NULL
 */

/* replaces simulate_trials(params, n_trials), basic_ddm_dc.py:114-125 (batched over B sets).
 * out_trials[b, i, :] = (rt, choice); choice in {1, -1, 0 (timeout; the reference's unbound-variable
 * bug at basic_ddm_dc.py:110-111 is resolved to 0)}; rt = n_steps*dt + tau. */
int nddm_basic_ddm_dc_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream);

/* replaces simulate_trials / simulate_trials_fine / simulate_trials_scale / _scale2,
 * single_trial_alpha_not_scaled.py:144-155, :1710-1722, :1274-1285, :1508-1519.
 * out_trials[b, i, :] = (choicert, z1); choicert = +-(ter + rt) or 0; z1 ~ N(gamma*bound_trial, sigma1). */
int nddm_single_trial_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream);

/* replaces simulate_trials_alt, single_trial_alpha_not_scaled.py:963-974 (per-trial diffusion coefficient).
 * out_trials[b, i, :] = (choicert, z1); z1 ~ N(gamma*dc_trial, sigma1). */
int nddm_single_trial_alt_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                                   float *out_summary, void *stream);

/* the data generator of alpha_not_scaled.py:52-128 as an Euler-Maruyama process: per-trial drift
 * ~ N(Nu, Eta) (pyhddmjagsutils.py:124-125), noise Varsigma, one external datum per set
 * out_extdata[b] = (ext_mode == 0 ? Alpha[b] : 1) + ext_sigma * N(0,1)  (alpha_not_scaled.py:103-106).
 * out_trials[b, i, :] = (y, acc): y = +-rt signed by the response (:98-100), acc = (sign(y)+1)/2 (:102).
 * out_extdata: device f32 [B] or NULL. */
int nddm_alpha_not_scaled_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma,
                                   int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata,
                                   void *stream);

/* replaces simulratcliff(N, Alpha, Tau, Nu, Beta, Eta=, Varsigma=), pyhddmjagsutils.py:47-176, as alpha_not_scaled.py:95-108 calls it
 * (rangeTau = rangeBeta = 0): the EXACT first-passage sampler of the diffusion model with per-trial drift ~ N(Nu, Eta) and diffusion
 * coefficient Varsigma (Tuerlinckx et al. 2001: random walk on spheres with rejection) -- the generator config 3's reference actually
 * runs.  No step size: nothing to discretise, no KS caveat.  (ABI 4)
 *   params       device f32 [B, 6] = Nu, Alpha, Beta, Tau, Eta, Varsigma (Nu is clipped to +-5 as :102-103 does)
 *   flags        NDDM_GAUSS_EXACT (every rounding spelled out, the reference's series term by term: equal bit for bit to
 *                oracle/ddm_oracle.c section D) or NDDM_GAUSS_FAST (v_log_f32 / v_exp_f32 / v_rcp_f32, and the SAME acceptance function
 *                from three terms of its series or of the series' Jacobi-dual form, whichever converges: no loop; on 6e6 trials no
 *                response differs from the exact mode's and no response time by more than 1e-6 s); nothing else
 *   out_trials   [B, n_trials, 2] = (y, acc): y = +-(Tau + decision time) signed by the response, acc = (sign + 1) / 2  (:98-102)
 *   out_summary  [B, NDDM_SUMMARY_K] from integer sums of the decision time in 2^-16 s (n_missing is 0: the sampler has no timeout)
 *   out_extdata  [B]: (ext_mode == 0 ? Alpha[b] : 1) + ext_sigma * N(0,1)  (alpha_not_scaled.py:103-106), as nddm_alpha_not_scaled_simulate
 * Randomness: per-trial drift = auxiliary normal 0 of (set, trial) -- the draw the Euler-Maruyama form uses -- and stream 3 of the
 * trial for the sampler's uniforms; a pure function of (seed, set_offset + row, trial).  Sets of more than 512 trials are tiled
 * (same bits); with summaries such a launch cannot be captured into a hipGraph. */
int nddm_simulratcliff(const float *params, int64_t B, int32_t n_trials, uint64_t seed, uint64_t set_offset, uint32_t flags,
                       float ext_sigma, int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata, void *stream);

/* replaces the per-trial loop over diffusion_trial(drift, bound_trial, beta, ter, dc),
 * imputation_from_stahl_not_scaled.py:120-148, :207-213.  bounds: device f32 [B, n_trials].
 * out_trials[b, i, :] = (choicert, bound_trial).  A negative (or NaN) boundary is the reference's ValueError
 * (:124-125): such a trial is written as (NaN, bound) -- validate on the host (Python adapter does) to raise. */
int nddm_explicit_boundary_simulate(const float *params, const float *bounds, int64_t B, int32_t n_trials,
                                    float dt, int32_t max_steps, uint64_t seed, uint64_t set_offset,
                                    uint32_t flags, float *out_trials, float *out_summary, void *stream);

/* generic dispatcher over enum nddm_model (the superset of the arguments above; `bounds` is read only by
 * NDDM_EXPLICIT_BOUNDARY, `ext_sigma` / `ext_mode` / `out_extdata` only by NDDM_ALPHA_NOT_SCALED) */
int nddm_simulate(int32_t model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                  int32_t max_steps, uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma, int32_t ext_mode,
                  float *out_trials, float *out_summary, float *out_extdata, void *stream);

/* nddm_simulate with a device-resident part of the set offset: the global index of row 0 is
 * set_offset + *set_offset_dev, read by the launch's pre-pass kernel when it RUNS on `stream` (set_offset_dev: device u64,
 * NULL = 0).  For hipGraph capture: the graph bakes kernel arguments in, the word in device memory moves between replays
 * (the training loop of basic_ddm_dc.py:199-202 as one graph per iteration).  The sum must stay below 2^60; it is reduced
 * modulo 2^60 on the device, where it cannot be refused. */
int nddm_simulate_indirect(int32_t model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                           int32_t max_steps, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev, uint32_t flags,
                           float ext_sigma, int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata,
                           void *stream);

/* The trials in a 2-byte WIRE FORMAT, for the exchange step (the all-gather that reassembles a training
 * minibatch on every rank).  out_codes[b, i] = step index | code << 14 (code 1 upper, 2 lower, 0 timeout) -- what the kernels
 * stage in LDS anyway -- for the models whose second column is a function of the code: NDDM_BASIC_DDM_DC (rt, choice) and
 * NDDM_ALPHA_NOT_SCALED without NDDM_BRIDGE (y, acc); max_steps < 2^14.  Gathering the float pairs moves 8 bytes per trial over
 * every xGMI link, which at this simulator's rate IS a link's bandwidth; the codes are a quarter of that, and
 * nddm_decode_codes (2 B read + 8 B written per trial, HBM-bound) gives back exactly the floats the simulator writes:
 * rt = fma(float(k), dt, tau) with tau from the (gathered) parameter rows.  out_trials / out_summary may be given as well
 * (NULL = not written); set_offset_dev as in nddm_simulate_indirect (NULL = 0). */
int nddm_simulate_codes(int32_t model, const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                        uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev, uint32_t flags, uint16_t *out_codes,
                        float *out_trials, float *out_summary, void *stream);
int nddm_decode_codes(int32_t model, const uint16_t *codes /* device u16 [B, n_trials] */, const float *params /* device f32 [B, P] */,
                      int64_t B, int32_t n_trials, float dt, float *out_trials /* device f32 [B, n_trials, 2] */, void *stream);

/* ---- prior / context samplers (basic_ddm_dc.py:50-80, single_trial_alpha_not_scaled.py:66-102) -------
 * On-device batched draw_prior(): out device f32 [B, P] in the model's parameter order
 * (P = nddm_model_nparams; gamma column of the single-trial family is filled with `gamma`).
 * Stream: Philox key (seed), counter (draw, 0, row_lo, row_hi | 2<<28). */
int nddm_draw_prior(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, float gamma, float *out_params,
                    void *stream);

/* the same with a device-resident part of the row offset (see nddm_simulate_indirect) */
int nddm_draw_prior_indirect(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev,
                             float gamma, float *out_params, void *stream);

/* sha256 (hex) of the sources the library was built from, as build.py computed it ("unknown" for a hand-made build): the
 * Python binding refuses a library whose sources have changed since */
const char *nddm_source_hash(void);

/* "hipcc=<version>": the compiler that built this library, recorded at build time */
const char *nddm_build_info(void);

/* ---- debugging aid used by the parity tests: the 4 normals of Philox block (c0..c3) under key (k0,k1) --- */
int nddm_debug_normals(const uint32_t *counters /* device u32 [n,4] */, int64_t n, uint32_t k0, uint32_t k1,
                       uint32_t flags, float *out /* device f32 [n,4] */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NDDM_H */

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
//
// This translation unit: nddm_rng.h (random stream) -> nddm_sim.h (sim_kernel) -> nddm_prepass.h (pre-pass, combine, prior)
// -> below: the host side (launch slots, sizing, dispatch) and the extern "C" entry points.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>

#include "../../include/nddm.h"
#include "nddm_prepass.h"
#include "nddm_ratcliff.h"
#include "nddm_rng.h"
#include "nddm_sim.h"

namespace nddm {

// ------------------------------------------------------------------------------------------------
// host side
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

static int round_up_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Every entry point that takes a stream validates the handle FIRST, with the one HIP call that checks instead of dereferencing:
// hipStreamGetDevice answers hipErrorContextIsDestroyed for a destroyed stream, for a stream of another copy of the HIP runtime
// in the process and for an integer that never was a stream, whereas hipStreamIsCapturing, hipStreamQuery, hipStreamSynchronize,
// hipEventRecord (and the launch calls) dereference the handle -- a SIGSEGV in the caller's thread (probed call by call on the HIP
// runtime 7.0.51831 that PyTorch 2.10.0+rocm7.0 loads: tools/probe_stream_validation.py, profiles/r4_stream_validation.txt; bench.py
// records the runtime in `toolchain`).  The stream must also belong to the current device:
// the launch's memory is taken from that device's slots.
static int check_stream(hipStream_t st)
{
    int sdev = -1, cur = -1;
    const hipError_t e = hipStreamGetDevice(st, &sdev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NDDM_ERR_HIP, "`stream` is not a live stream of this process's HIP runtime (destroyed, or created by another copy "
                                  "of the runtime): %s", hipGetErrorString(e));
    }
    if (hipGetDevice(&cur) == hipSuccess && st != nullptr && st != hipStreamPerThread && sdev != cur)
        return fail(NDDM_ERR_PARAM, "`stream` belongs to another device than the current one (nddm_set_device / hipSetDevice)%s");
    return NDDM_OK;
}

// ---- process-wide state, all of it behind one mutex -------------------------------------------------------------------
// Entry points are re-entrant: each call takes a SNAPSHOT of the developer knobs at entry and owns the device memory it is
// handed (a LaunchSlot) until the work it enqueued has completed.
struct Tuning { int sets_per_chunk, ring, refill_thresh, variant, grid_waves, tile_trials, no_order; };
// geometry of the calling thread's last launch (nddm_debug_last_launch)
struct LastLaunch { int grid_waves, vkeys, ring, tile_trials, tiles_per_set, sets_per_chunk, refill_thresh, lds_bytes; };
static thread_local LastLaunch g_last = {0, 0, 0, 0, 0, 0, 0, 0};
static std::mutex g_mu;
static Tuning g_tuning = {0, 0, 0, 0, 0, 0, 0};   // 0 = automatic (nddm_set_tuning overrides; benchmarking aid)
static unsigned long long *g_dbg = nullptr;       // nddm_set_debug_trace (profiling aid)
static int g_dbg_waves = 0, g_dbg_chunks = 0;

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
    // (a) the slot this stream used last: stream order makes it safe -- PROVIDED `st` still is the stream that used it.  A
    //     handle value is recycled once its stream has been destroyed (possibly with the slot's launch still in flight), so
    //     the new launch is also made to wait for the slot's completion event: no wait at all on a genuinely identical
    //     stream (the event is behind it in stream order), the needed one on a recycled handle.  hipStreamPerThread is one
    //     handle for a different stream in every thread, so it never qualifies.
    if (st != hipStreamPerThread)
        for (int i = 0; i < n_use && !pick; ++i)
            if (!g_slots[dev][i]->busy && g_slots[dev][i]->stream == st && !g_slots[dev][i]->fresh) {
                if (hipStreamWaitEvent(st, g_slots[dev][i]->done, 0) != hipSuccess) { (void)hipGetLastError(); continue; }
                pick = g_slots[dev][i];
            }
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
// scratch).  Every such allocation has an OWNER (ABI 3): the graph arena bound to the capturing thread
// (nddm_graph_arena_create / _bind / _release), or -- with no arena bound -- the ownerless list that
// nddm_release_graph_memory() frees.  Releasing one owner never touches what another owner's graphs replay into.
constexpr size_t GRAPH_SCRATCH_MAX = 64u << 20;
struct GraphAlloc { void *p; int dev; size_t bytes; GraphAlloc *next; };
struct GraphArena { uint64_t id; GraphAlloc *list; GraphArena *next; };
static GraphArena g_ownerless = {0, nullptr, nullptr};
static GraphArena *g_arenas = nullptr;                 // live arenas (behind g_mu)
static uint64_t g_next_arena = 1;
static thread_local uint64_t g_cur_arena = 0;          // the arena captured launches of THIS thread are charged to (0 = none)

static GraphArena *find_arena_locked(uint64_t id)
{
    if (id == 0) return &g_ownerless;
    for (GraphArena *a = g_arenas; a; a = a->next)
        if (a->id == id) return a;
    return nullptr;
}

// hipFree with the calling thread's capture mode relaxed (see malloc_relaxed), on the device the block came from
static void free_list(GraphAlloc *list)
{
    if (!list) return;
    // blocks of another device are freed with that device current; the CALLER's current device is put back afterwards (a later
    // nddm_* call or allocation on this thread must not land on the last block's GPU)
    int entry = -1, cur = -1;
    (void)hipGetDevice(&entry);
    cur = entry;
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    while (list) {
        GraphAlloc *n = list->next;
        if (list->dev != cur && hipSetDevice(list->dev) == hipSuccess) cur = list->dev;
        (void)hipFree(list->p);
        delete list;
        list = n;
    }
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (entry >= 0 && cur != entry) (void)hipSetDevice(entry);
}

static char *graph_alloc(int dev, size_t bytes, hipError_t *err)
{
    *err = hipSuccess;
    {   // refuse before allocating when the thread's arena is gone (released while still bound)
        std::lock_guard<std::mutex> lock(g_mu);
        if (!find_arena_locked(g_cur_arena)) { *err = hipErrorInvalidResourceHandle; return nullptr; }
    }
    void *p = nullptr;
    *err = malloc_relaxed(&p, bytes);
    if (*err != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    GraphArena *a = find_arena_locked(g_cur_arena);
    if (!a) a = &g_ownerless;                          // released between the two looks: never leak, never dangle
    a->list = new GraphAlloc{p, dev, bytes, a->list};
    return static_cast<char *>(p);
}

// zeroes the 64 counting-sort counters / the queue words of a captured launch (a kernel, not hipMemsetAsync: a memset NODE of <= 4 KB
// in a captured launch zeroes on the first replay only -- from the second replay on the buffer holds constant garbage (min -2^31, max
// 434269841: the replayed node writes a wrong value), 1 MB nodes are fine -- HIP runtime 7.0.51831, the copy PyTorch 2.10.0+rocm7.0
// loads; the library itself is built by hipcc 7.2.26015.  Probe: tools/probe_graph_memset_node.py, profiles/r6_probe_graph_memset_node.txt)
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

// one (MODEL, BRIDGE, SMALL, PACKED, VKEYS) variant: pick the Gaussian transform and the step-cap form, size the grid
template <int MODEL, bool BRIDGE, bool SMALL, bool PACKED, bool VKEYS, bool CODES = false, bool F64 = false>
static void launch_variant(const SimArgs &A, bool fast, bool cap4, size_t lds_bytes, int n_chunks, int cus, int grid_override,
                           bool grid_forced, hipStream_t st)
{
    const dim3 block(WAVE);
#define NDDM_LAUNCH(KERNEL)                                                                    \
    do {                                                                                       \
        int waves = resident_waves(KERNEL, lds_bytes, cus);                                    \
        if (grid_override > 0 && (grid_override < waves || grid_forced)) waves = grid_override;\
        if (waves > n_chunks) waves = n_chunks;                                                \
        g_last.grid_waves = waves; g_last.vkeys = VKEYS ? 1 : 0;                               \
        hipLaunchKernelGGL(KERNEL, dim3(waves), block, lds_bytes, st, A);                      \
    } while (0)
    if (fast && cap4)       NDDM_LAUNCH((sim_kernel<MODEL, true, true, BRIDGE, SMALL, PACKED, VKEYS, CODES, F64>));
    else if (fast)          NDDM_LAUNCH((sim_kernel<MODEL, true, false, BRIDGE, SMALL, PACKED, VKEYS, CODES, F64>));
    else if (cap4)          NDDM_LAUNCH((sim_kernel<MODEL, false, true, BRIDGE, SMALL, PACKED, VKEYS, CODES, F64>));
    else                    NDDM_LAUNCH((sim_kernel<MODEL, false, false, BRIDGE, SMALL, PACKED, VKEYS, CODES, F64>));
#undef NDDM_LAUNCH
}

// vkeys: the launch is small (its grid was cut, or it has fewer chunks than resident waves) -- the variant that keeps the
// Philox round keys in VGPRs (nddm_sim.h) is 13-29 % faster there and 1.6 % slower on a full grid; it exists for the models
// whose kernels have the registers to spare
template <int MODEL, bool BRIDGE>
static int launch_model(const SimArgs &A, bool fast, bool packed, bool vkeys, bool f64, size_t lds_bytes, int n_chunks, int cus,
                        int grid_override, bool grid_forced, hipStream_t st)
{
    const bool cap4 = (A.max_k % ((packed || BRIDGE) ? 8 : 4)) == 0;     // the step cap falls on a block boundary (bridge, packed: 8 steps per pass)
    const bool small = A.res16 == 2;
    constexpr bool HAS_VKEYS = !BRIDGE && (MODEL == NDDM_BASIC_DDM_DC || MODEL == NDDM_ALPHA_NOT_SCALED || MODEL == NDDM_EXPLICIT_BOUNDARY);
#define NDDM_ARGS A, fast, cap4, lds_bytes, n_chunks, cus, grid_override, grid_forced, st
    constexpr bool HAS_CODES = !BRIDGE && (MODEL == NDDM_BASIC_DDM_DC || MODEL == NDDM_ALPHA_NOT_SCALED);
    constexpr bool HAS_F64 = !BRIDGE && (MODEL == NDDM_BASIC_DDM_DC || MODEL == NDDM_SINGLE_TRIAL);
    if (f64) {                  // NDDM_STATE_F64 (the host has checked: basic / single, no bridge, not packed, no codes); round keys from LDS
        if constexpr (HAS_F64) {
            if (small) launch_variant<MODEL, false, true, false, false, false, true>(NDDM_ARGS);
            else launch_variant<MODEL, false, false, false, false, false, true>(NDDM_ARGS);
        }
    }
    else if (A.out_codes) {     // the wire-format variant (the host has checked: small, not packed, not the bridge); round keys from LDS
        if constexpr (HAS_CODES) launch_variant<MODEL, false, true, false, false, true>(NDDM_ARGS);
    }
    else if constexpr (BRIDGE) launch_variant<MODEL, true, false, false, false>(NDDM_ARGS);
    else if (!small && !packed) launch_variant<MODEL, false, false, false, false>(NDDM_ARGS);
    else if constexpr (HAS_VKEYS) {
        if (packed && vkeys) launch_variant<MODEL, false, true, true, true>(NDDM_ARGS);     // NDDM_GAUSS_PACKED: the host has checked small
        else if (packed) launch_variant<MODEL, false, true, true, false>(NDDM_ARGS);
        else if (vkeys) launch_variant<MODEL, false, true, false, true>(NDDM_ARGS);
        else launch_variant<MODEL, false, true, false, false>(NDDM_ARGS);
    } else {
        if (packed) launch_variant<MODEL, false, true, true, false>(NDDM_ARGS);
        else launch_variant<MODEL, false, true, false, false>(NDDM_ARGS);
    }
#undef NDDM_ARGS
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NDDM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

static int simulate(int model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                    int32_t max_steps, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev, uint32_t flags,
                    float ext_sigma, int32_t ext_mode, float *out_trials, float *out_summary, float *out_ext, void *stream,
                    uint16_t *out_codes = nullptr)
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
    if (flags > 15u) return fail(NDDM_ERR_PARAM, "unknown flags%s");
    const bool bridge = (flags & NDDM_BRIDGE) != 0;
    const bool packed = (flags & NDDM_GAUSS_PACKED) != 0;
    const bool f64 = (flags & NDDM_STATE_F64) != 0;
    if (f64 && ((model != NDDM_BASIC_DDM_DC && model != NDDM_SINGLE_TRIAL) || bridge || packed || out_codes))
        return fail(NDDM_ERR_PARAM, "NDDM_STATE_F64 exists for NDDM_BASIC_DDM_DC and NDDM_SINGLE_TRIAL, without NDDM_BRIDGE, NDDM_GAUSS_PACKED "
                                    "or the 2-byte wire format%s");
    if (packed && (bridge || max_steps >= 16384))
        return fail(NDDM_ERR_PARAM, "NDDM_GAUSS_PACKED needs max_steps < 2^14 and cannot be combined with NDDM_BRIDGE%s");
    if (bridge && model != NDDM_ALPHA_NOT_SCALED)
        return fail(NDDM_ERR_PARAM, "NDDM_BRIDGE is only available for NDDM_ALPHA_NOT_SCALED%s");
    if (bridge && max_steps >= (1 << 22)) return fail(NDDM_ERR_SHAPE, "max_steps must be < 2^22 with NDDM_BRIDGE%s");
    if (B == 0) return NDDM_OK;
    if (!params) return fail(NDDM_ERR_NULL, "params is NULL%s");
    if (model == NDDM_EXPLICIT_BOUNDARY && !bounds) return fail(NDDM_ERR_NULL, "bounds is NULL%s");
    if (!out_trials && !out_summary && !out_ext && !out_codes) return fail(NDDM_ERR_NULL, "no output buffer given%s");
    if (out_codes && ((model != NDDM_BASIC_DDM_DC && model != NDDM_ALPHA_NOT_SCALED) || bridge || packed || max_steps >= 16384))
        return fail(NDDM_ERR_PARAM, "the 2-byte wire format exists for NDDM_BASIC_DDM_DC and NDDM_ALPHA_NOT_SCALED without NDDM_BRIDGE "
                                    "or NDDM_GAUSS_PACKED, max_steps < 2^14%s");
    if (B * (((long long)n_trials + 511) / 512) >= (1ll << 31))
        return fail(NDDM_ERR_SHAPE, "B * ceil(n_trials / 512) must be < 2^31 per launch%s");

    if (const int rc = check_stream(reinterpret_cast<hipStream_t>(stream))) return rc;

    // snapshot of the developer knobs, and the device this call runs on
    Tuning tun;
    unsigned long long *dbg;
    int dbg_waves, dbg_chunks;
    { std::lock_guard<std::mutex> lock(g_mu); tun = g_tuning; dbg = g_dbg; dbg_waves = g_dbg_waves; dbg_chunks = g_dbg_chunks; }
    int dev = 0;
    {
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return fail(NDDM_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    }
    DeviceInfo di;
    if (!device_info(dev, &di)) return fail(NDDM_ERR_HIP, "hipGetDeviceProperties failed%s");
    const int simds = 4 * di.cus;
    // The sizing rules' yardsticks.  They are NOT the grid (the launch asks resident_waves() per kernel: 8 per SIMD for every
    // fast kernel): `waves_sizing` = 7 per SIMD is what tiles, chunks and the open-ahead switch were tuned against, and an
    // A/B of the same rules at 8 per SIMD moved no shape by more than the run-to-run noise (profiles/r3_sizing_ab.txt), so the
    // measured constants stay; `waves_min_chunks` = 6 per SIMD is the floor of the chunk count.
    static const int sizing_per_simd = [] { const char *e = getenv("NDDM_SIZING_WAVES_PER_SIMD"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 7; }();
    const long long waves_sizing = (long long)sizing_per_simd * simds;                 // (the environment variable is the A/B's switch)
    const long long waves_min_chunks = (long long)(sizing_per_simd - 1) * simds;

    SimArgs A;
    memset(&A, 0, sizeof A);
    A.params = params; A.bounds = bounds; A.out_trials = out_trials; A.out_summary = out_summary; A.out_ext = out_ext;
    A.out_codes = out_codes;
    A.B = B; A.n_trials = n_trials; A.max_k = max_steps; A.dt = dt; A.sqrt_dt = sqrtf(dt);
    A.tscale = bridge ? dt * 0.00390625f : dt;
    A.k0 = (uint32_t)seed; A.k1 = (uint32_t)(seed >> 32);
    A.ext_sigma = ext_sigma; A.ext_mode = ext_mode;
    A.dbg = dbg; A.dbg_waves = dbg_waves; A.dbg_chunks = dbg_chunks;
    A.dt64 = (double)dt; A.sqrt_dt64 = sqrt((double)dt);      // (sqrt of a double: correctly rounded, as np.sqrt(dt) is)

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
    // Grid: as many waves as stay resident -- unless the launch is small or mid-size.  Two measured effects (profiles/
    // r2_small_launches.txt):
    //  * amortisation: a wave that gets ~300+ wave-blocks of work (64 lanes x 4 steps each) keeps its lanes balanced and
    //    amortises its start-up (dt=.01, 10k-30k sets x 300 trials, or 100k x 60: 1024-3072 waves are 25-45 % faster than
    //    8192);
    //  * critical path: a trial that runs to the cap is max_steps / 4 DEPENDENT blocks, and a block of one wave takes as
    //    long as the SIMD's waves share it (277 cycles each when it is full).  A launch worth x = blocks / (max_steps / 4)
    //    cap-length trials finishes when its last long trials do, and those run faster on emptier SIMDs: at dt=.001 / cap
    //    4000, 10k / 20k / 50k sets x 300 trials are fastest on 3072 / 4096-5120 / 5120-6144 waves (29 / 45 / 12 % faster
    //    than 8192); from x ~ 50k on the full grid is best.
    // Never below one wave per SIMD.
    double est_steps = 0.25 / (double)dt;                                // E[steps] under the reference priors
    if (est_steps > (double)max_steps) est_steps = (double)max_steps;
    if (est_steps < 1.0) est_steps = 1.0;
    const double est_blocks = (double)total_trials * est_steps / 256.0;
    long long plan_waves = 8ll * simds;
    if (tun.grid_waves > 0) plan_waves = tun.grid_waves;
    else {
        const double cap_blocks = max_steps > 4 ? (double)max_steps * 0.25 : 1.0;
        double want = est_blocks / 300.0;
        const double crit = 3.0 * (double)simds * pow(est_blocks / cap_blocks / (2.8 * (double)simds), 0.4);
        if (crit < want) want = crit;
        if (want < (double)plan_waves) plan_waves = want < (double)simds ? simds : (long long)want;
    }
    const long long waves_for_tiles = plan_waves < waves_sizing ? plan_waves : waves_sizing;
    int tile_cap = 64;
    while (tile_cap < 512 && (long long)tile_cap * waves_for_tiles * 8 < total_trials) tile_cap <<= 1;
    int tiles = tun.tile_trials > 0 ? (n_trials + tun.tile_trials - 1) / tun.tile_trials
                                    : (n_trials <= tile_cap ? 1 : (n_trials + tile_cap - 1) / tile_cap);
    int tile_n = (n_trials + tiles - 1) / tiles;
    tiles = (n_trials + tile_n - 1) / tile_n;
    // Four ring slots must fit the LDS budget (4 granules of 1280 B, below): with two, a straggler in the older tile stalls
    // the hand-out (sets are flushed in order).  Only the kernels that stage 32-bit results get here at 300 trials per set
    // (the bridge: 2 x 300 gave lane efficiency 0.905, 4 x 150 gives 0.93 and +2.4 %): split their sets further.
    if (!tun.ring && !tun.tile_trials)
        while (tile_n > 96 && (size_t)lds_header_bytes(model) + 4u * (size_t)slot_stride_bytes(tile_n, (int)per_trial) > 5120u) {
            tiles++;
            tile_n = (n_trials + tiles - 1) / tiles;
            tiles = (n_trials + tile_n - 1) / tile_n;
        }
    const long long vB = B * (long long)tiles;
    if (vB >= (1ll << 31)) return fail(NDDM_ERR_SHAPE, "B * ceil(n_trials / 512) must be < 2^31 per launch%s");
    if ((long long)tile_n * tiles >= (1ll << 30) || n_trials >= (1 << 30))
        return fail(NDDM_ERR_SHAPE, "n_trials must be < 2^30%s");
    A.n_trials = tile_n; A.n_total = n_trials; A.tiles_per_set = tiles; A.B = vB;
    // chunk = the unit a wave pulls from the global queue: small (tail of the whole launch <= one chunk), but large
    // enough that the queue's atomic counter is touched rarely (~ once per 900+ trials per wave; measured at 1M x 300, dt=.001: 3 sets per chunk 0.6 % faster than 4, 2 the same)
    int spc = tun.sets_per_chunk;
    if (!spc) {
        spc = (900 + tile_n - 1) / tile_n;
        if (spc < 1) spc = 1;
        if (spc > 64) spc = 64;
        while (spc > 1 && vB / spc < 32 * waves_for_tiles) spc >>= 1;   // keep >= ~32 chunks per wave: the launch's tail is one chunk
        // ... but the queue is ONE atomic word: same-address atomics retire at ~85 M/s on MI355X (measured: 1M chunks
        // take 11.6 ms whatever the work; contention already costs 15 % at 55 M/s), so short-trial workloads (dt = .01,
        // few trials per set) must pull less often: at most ~40 M chunks/s over the shortest time the launch can take --
        // its stepping (E[steps] ~ min(cap, 0.25 / dt) under the reference priors, 265 SIMD cycles per 256 lane-steps)
        // plus one trial that runs to the cap on a full SIMD -- and never fewer than ~8 chunks per wave.  Mid-size
        // launches, whose tail is one chunk, stay below the limit with their fine chunks.
        const double t_est = (double)vB * ((double)tile_n * est_steps / 256.0) * 265.0 / ((double)simds * di.clock_hz)
                             + (double)max_steps * 0.25 * 265.0 * 7.0 / di.clock_hz;
        double max_chunks = t_est * 4.0e7;
        if (max_chunks < 8.0 * (double)waves_min_chunks) max_chunks = 8.0 * (double)waves_min_chunks;
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
        return (size_t)lds_header_bytes(model) + (size_t)r * (size_t)slot_stride_bytes(tile_n, (int)per_trial);
    };
    int ring = tun.ring;
    if (!ring) {
        ring = round_up_pow2((480 + tile_n - 1) / tile_n);
        if (ring < 4) ring = 4;
        if (ring > 64) ring = 64;
        while (ring > 2 && lds_of(ring) > 5120) ring >>= 1;     // 4 LDS granules of 1280 B: 32 workgroups per CU fit
        // ... and whatever else fits the same four granules widens the window for free, up to ~2400 trials (the models with
        // long, heavy-tailed trials are window-limited: single_trial 0.927 of its lanes busy with 4 x 300, 0.95 with 8 x 300,
        // which does not fit; seven do).  The ring need not be a power of two (ring_slot() in nddm_sim.h).
        // Only for long step caps: short trials (cap 400) are not window-limited, and at 60 trials per set the 22 slots
        // that fit measured 3.7 % slower than 8.
        if (max_steps > 1000)
            while (ring < 32 && (ring + 1) * tile_n <= 2400 && lds_of(ring + 1) <= 5120) ring++;
    }
    if (ring < 2) ring = 2;
    if (ring > 64) ring = 64;
    // the hand-out carries the byte offset of the current slot and the SIGNED step to the next one as two 16-bit halves of one
    // register (nddm_sim.h: slot_pack): the ring's slots must span < 32 KB.  The library's own geometry stays below 5 KB; only
    // a tuning override gets here.
    if ((size_t)ring * (size_t)slot_stride_bytes(tile_n, (int)per_trial) >= 32768u)
        return fail(NDDM_ERR_SHAPE, "ring x tile too large: the LDS ring's slots must span < 32 KB (tuning override?)%s");
    A.sets_per_chunk = spc; A.ring = ring; A.ring_magic = (uint32_t)(0x100000000ull / (unsigned long long)ring);
    // refill threshold: a refill costs ~27 VALU instructions whatever the number of lanes it serves, a waiting lane
    // wastes its share of every block; with lambda completions per block the optimum is ~sqrt(c lambda) finished lanes:
    // 8 when trials last ~64 blocks (dt=.001, cap 4000), 16 when they last ~7 (the reference default dt=.01, cap 400).
    // The cap is the only hint the host has about trial length.  (The models that draw per-trial latents used to take 16
    // at every cap; since their latents are drawn 64 at a time their refill costs what the basic model's does, and 8 is
    // measured 0.4 / 2 / 3.3 % faster for single_trial / alpha_not_scaled / + bridge at dt=.001.)
    A.res16 = res16 ? (tile_n <= 512 ? 2 : 1) : 0;
    if (packed && A.res16 != 2) return fail(NDDM_ERR_PARAM, "NDDM_GAUSS_PACKED needs tiles of <= 512 trials (tuning override?)%s");
    if (out_codes && A.res16 != 2) return fail(NDDM_ERR_PARAM, "the 2-byte wire format needs tiles of <= 512 trials (tuning override?)%s");
    A.refill_thresh = tun.refill_thresh ? tun.refill_thresh : (max_steps <= 1000 ? 16 : 8);
    const long long n_chunks = (vB + spc - 1) / spc;
    A.n_chunks = (int)n_chunks;
    A.open_ahead = vB >= 4 * waves_sizing ? 1 : 0;
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
        if (!mem && e == hipErrorInvalidResourceHandle)
            return fail(NDDM_ERR_PARAM, "the graph arena bound to this thread has been released (nddm_graph_arena_bind(0) or a live arena)%s");
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
        const int rec_mode = (fast ? 1 : 0) | (f64 ? 2 : 0);
        if (want_order) {
            int *order_ws = reinterpret_cast<int *>(scratch);
            uint32_t *recs = reinterpret_cast<uint32_t *>(order_ws + 64);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned int *>(order_ws));
            hipLaunchKernelGGL(order_hist_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, params, P, (int)B, dt,
                               (int)max_steps, order_ws);
            hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, rec_mode, params, P,
                               (int)B, dt, A.sqrt_dt, (int)max_steps, (unsigned long long)set_offset,
                               reinterpret_cast<const unsigned long long *>(set_offset_dev), order_ws, recs);
            A.recs = recs;
        } else {
            uint32_t *recs = reinterpret_cast<uint32_t *>(scratch);
            hipLaunchKernelGGL(prep_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, model, rec_mode, params, P, (int)B, dt,
                               A.sqrt_dt, (unsigned long long)set_offset, reinterpret_cast<const unsigned long long *>(set_offset_dev), recs);
            A.recs = recs;
        }
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(NDDM_ERR_HIP, "pre-pass (hand-out records): %s", hipGetErrorString(e));
    }
    if (rc == NDDM_OK) {
        const int gw = plan_waves < 8ll * simds || tun.grid_waves > 0 ? (int)plan_waves : 0;     // 0 = what is resident
        // round keys in VGPRs (no LDS round trips in the step loop) whenever part of the launch runs on SIMDs that are not
        // full: a cut grid, fewer chunks than waves, or a launch short enough that its tail matters (measured at dt=.001:
        // 20k sets x 300 trials 32 % faster, 50k 8 %, 100k equal, 300k 2 % slower)
        bool vkeys = plan_waves < 8ll * simds || n_chunks < 8ll * simds || est_blocks < 2.0e4 * (double)simds;
        if (tun.variant % 3 == 1) vkeys = false;                         // developer knob: 1 LDS keys, 2 VGPR keys, else the rule
        else if (tun.variant % 3 == 2) vkeys = true;
        g_last.ring = ring; g_last.tile_trials = tile_n; g_last.tiles_per_set = tiles; g_last.sets_per_chunk = spc;
        g_last.refill_thresh = A.refill_thresh; g_last.lds_bytes = (int)lds;
        switch (model) {
        case NDDM_BASIC_DDM_DC: rc = launch_model<NDDM_BASIC_DDM_DC, false>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_SINGLE_TRIAL: rc = launch_model<NDDM_SINGLE_TRIAL, false>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_SINGLE_TRIAL_ALT: rc = launch_model<NDDM_SINGLE_TRIAL_ALT, false>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
        case NDDM_ALPHA_NOT_SCALED:
            rc = bridge ? launch_model<NDDM_ALPHA_NOT_SCALED, true>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st)
                        : launch_model<NDDM_ALPHA_NOT_SCALED, false>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st);
            break;
        default: rc = launch_model<NDDM_EXPLICIT_BOUNDARY, false>(A, fast, packed, vkeys, f64, lds, (int)n_chunks, di.cus, gw, tun.grid_waves > 0, st); break;
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
int nddm_set_tuning(int sets_per_chunk, int ring, int refill_thresh, int variant, int grid_waves, int tile_trials)
{
    if (ring < 0 || ring == 1) return nddm::fail(NDDM_ERR_PARAM, "ring must be 0 (automatic) or >= 2%s");
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    const int no_order = nddm::g_tuning.no_order;
    nddm::g_tuning = {sets_per_chunk, ring, refill_thresh, variant < 0 ? 0 : variant, grid_waves, tile_trials, no_order};
    return NDDM_OK;
}

/* benchmarking aid: 1 = process the sets in the given order (no longest-first sort) */
int nddm_set_ordering(int enabled)
{
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_tuning.no_order = enabled ? 0 : 1;
    return NDDM_OK;
}

/* profiling aid: the simulator kernels of the next launches record, with plain stores, one 8-word record per workgroup
 * (= wave) w < wave_capacity at buf[8 w ..]: {step-loop blocks, refill phases, s_memtime cycles, lifetime in
 * s_memrealtime ticks (100 MHz), start tick, tick at which the wave found the work queue empty, end tick, 1}, and the
 * tick at which chunk c < chunk_capacity was pulled from the queue at buf[8 wave_capacity + c].  buf = device u64
 * [8 wave_capacity + chunk_capacity], zeroed by the caller; NULL switches the trace off.  (tools/wave_timeline.py) */
int nddm_set_debug_trace(void *dev_u64, int wave_capacity, int chunk_capacity)
{
    if (dev_u64 && (wave_capacity < 0 || chunk_capacity < 0)) return nddm::fail(NDDM_ERR_PARAM, "negative trace capacity%s");
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_dbg = static_cast<unsigned long long *>(dev_u64);
    nddm::g_dbg_waves = dev_u64 ? wave_capacity : 0;
    nddm::g_dbg_chunks = dev_u64 ? chunk_capacity : 0;
    return NDDM_OK;
}

/* developer aid: geometry of the calling thread's last simulator launch: out[8] = {grid waves, 1 if the VGPR-keys kernel
 * variant ran, ring slots, trials per tile, tiles per set, sets per chunk, refill threshold, dynamic LDS bytes} */
int nddm_debug_last_launch(int32_t *out8)
{
    if (!out8) return nddm::fail(NDDM_ERR_NULL, "out8 is NULL%s");
    const nddm::LastLaunch &l = nddm::g_last;
    out8[0] = l.grid_waves; out8[1] = l.vkeys; out8[2] = l.ring; out8[3] = l.tile_trials; out8[4] = l.tiles_per_set;
    out8[5] = l.sets_per_chunk; out8[6] = l.refill_thresh; out8[7] = l.lds_bytes;
    return NDDM_OK;
}

/* testing aid: cap on the number of launch slots per device that are handed out (1..256) */
int nddm_debug_set_slot_limit(int n)
{
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::g_slot_limit = n < 1 ? 1 : (n > nddm::MAX_SLOTS ? nddm::MAX_SLOTS : n);
    return NDDM_OK;
}

/* Frees the OWNERLESS memory behind captured launches of the current device: allocations made for launches that were
 * captured while no graph arena was bound to the capturing thread.  Memory charged to an arena is not touched (ABI 3; up to
 * ABI 2 this call freed every captured launch's memory on the device, whoever's graph it belonged to). */
int nddm_release_graph_memory(void)
{
    nddm::g_err[0] = 0;
    int dev = 0;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= nddm::MAX_DEVICES) return nddm::fail(NDDM_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    nddm::GraphAlloc *mine = nullptr;
    {
        std::lock_guard<std::mutex> lock(nddm::g_mu);
        nddm::GraphAlloc **pp = &nddm::g_ownerless.list;
        while (*pp) {
            nddm::GraphAlloc *a = *pp;
            if (a->dev == dev) { *pp = a->next; a->next = mine; mine = a; }
            else pp = &a->next;
        }
    }
    nddm::free_list(mine);
    return NDDM_OK;
}

/* ---- graph arenas: the owner of the memory behind captured launches ---------------------------------------------------- */
int nddm_graph_arena_create(uint64_t *arena)
{
    nddm::g_err[0] = 0;
    if (!arena) return nddm::fail(NDDM_ERR_NULL, "arena is NULL%s");
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    nddm::GraphArena *a = new nddm::GraphArena{nddm::g_next_arena++, nullptr, nddm::g_arenas};
    nddm::g_arenas = a;
    *arena = a->id;
    return NDDM_OK;
}

int nddm_graph_arena_bind(uint64_t arena, uint64_t *previous)
{
    nddm::g_err[0] = 0;
    {
        std::lock_guard<std::mutex> lock(nddm::g_mu);
        if (!nddm::find_arena_locked(arena)) return nddm::fail(NDDM_ERR_PARAM, "unknown (or released) graph arena%s");
    }
    if (previous) *previous = nddm::g_cur_arena;
    nddm::g_cur_arena = arena;
    return NDDM_OK;
}

int nddm_graph_arena_info(uint64_t arena, uint64_t *bytes, int32_t *n_allocations)
{
    nddm::g_err[0] = 0;
    std::lock_guard<std::mutex> lock(nddm::g_mu);
    const nddm::GraphArena *a = nddm::find_arena_locked(arena);
    if (!a) return nddm::fail(NDDM_ERR_PARAM, "unknown (or released) graph arena%s");
    uint64_t b = 0;
    int32_t n = 0;
    for (const nddm::GraphAlloc *g = a->list; g; g = g->next) { b += g->bytes; ++n; }
    if (bytes) *bytes = b;
    if (n_allocations) *n_allocations = n;
    return NDDM_OK;
}

int nddm_graph_arena_release(uint64_t arena)
{
    nddm::g_err[0] = 0;
    if (arena == 0) return nddm::fail(NDDM_ERR_PARAM, "arena 0 is the ownerless list: nddm_release_graph_memory() frees it%s");
    nddm::GraphArena *a = nullptr;
    {
        std::lock_guard<std::mutex> lock(nddm::g_mu);
        for (nddm::GraphArena **pp = &nddm::g_arenas; *pp; pp = &(*pp)->next)
            if ((*pp)->id == arena) { a = *pp; *pp = a->next; break; }
    }
    if (!a) return nddm::fail(NDDM_ERR_PARAM, "unknown (or already released) graph arena%s");
    if (nddm::g_cur_arena == arena) nddm::g_cur_arena = 0;      // this thread's binding; another thread's is refused at its next capture
    nddm::free_list(a->list);
    delete a;
    return NDDM_OK;
}

int nddm_basic_ddm_dc_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_BASIC_DDM_DC, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, nullptr, flags,
                          0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_single_trial_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                               uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                               float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_SINGLE_TRIAL, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, nullptr, flags,
                          0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_single_trial_alt_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float *out_trials,
                                   float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_SINGLE_TRIAL_ALT, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, nullptr,
                          flags, 0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_alpha_not_scaled_simulate(const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                                   uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma,
                                   int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata,
                                   void *stream)
{
    return nddm::simulate(NDDM_ALPHA_NOT_SCALED, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, nullptr,
                          flags, ext_sigma, ext_mode, out_trials, out_summary, out_extdata, stream);
}

int nddm_explicit_boundary_simulate(const float *params, const float *bounds, int64_t B, int32_t n_trials,
                                    float dt, int32_t max_steps, uint64_t seed, uint64_t set_offset,
                                    uint32_t flags, float *out_trials, float *out_summary, void *stream)
{
    return nddm::simulate(NDDM_EXPLICIT_BOUNDARY, params, bounds, B, n_trials, dt, max_steps, seed, set_offset, nullptr,
                          flags, 0.0f, 0, out_trials, out_summary, nullptr, stream);
}

int nddm_simulate(int32_t model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                  int32_t max_steps, uint64_t seed, uint64_t set_offset, uint32_t flags, float ext_sigma, int32_t ext_mode,
                  float *out_trials, float *out_summary, float *out_extdata, void *stream)
{
    return nddm::simulate(model, params, model == NDDM_EXPLICIT_BOUNDARY ? bounds : nullptr, B, n_trials, dt, max_steps,
                          seed, set_offset, nullptr, flags, ext_sigma, ext_mode, out_trials, out_summary,
                          model == NDDM_ALPHA_NOT_SCALED ? out_extdata : nullptr, stream);
}

int nddm_simulate_codes(int32_t model, const float *params, int64_t B, int32_t n_trials, float dt, int32_t max_steps,
                        uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev, uint32_t flags, uint16_t *out_codes,
                        float *out_trials, float *out_summary, void *stream)
{
    if (!out_codes) return nddm::fail(NDDM_ERR_NULL, "out_codes is NULL%s");
    return nddm::simulate(model, params, nullptr, B, n_trials, dt, max_steps, seed, set_offset, set_offset_dev, flags, 0.0f, 0,
                          out_trials, out_summary, nullptr, stream, out_codes);
}

int nddm_decode_codes(int32_t model, const uint16_t *codes, const float *params, int64_t B, int32_t n_trials, float dt,
                      float *out_trials, void *stream)
{
    nddm::g_err[0] = 0;
    if (model != NDDM_BASIC_DDM_DC && model != NDDM_ALPHA_NOT_SCALED)
        return nddm::fail(NDDM_ERR_PARAM, "the 2-byte wire format exists for NDDM_BASIC_DDM_DC and NDDM_ALPHA_NOT_SCALED%s");
    if (B < 0 || n_trials <= 0) return nddm::fail(NDDM_ERR_SHAPE, "B < 0 or n_trials <= 0%s");
    if (!(dt > 0.0f) || !isfinite(dt)) return nddm::fail(NDDM_ERR_PARAM, "dt must be finite and > 0%s");
    if (B == 0) return NDDM_OK;
    if (!codes || !params || !out_trials) return nddm::fail(NDDM_ERR_NULL, "null pointer%s");
    if (const int rc = nddm::check_stream(reinterpret_cast<hipStream_t>(stream))) return rc;
    const long long n = B * (long long)n_trials;
    const int threads = 256;
    hipLaunchKernelGGL(nddm::decode_codes_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0,
                       reinterpret_cast<hipStream_t>(stream), (int)model, codes, params, nddm_model_nparams(model), 3, (long long)B,
                       (int)n_trials, dt, reinterpret_cast<float2 *>(out_trials));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nddm::fail(NDDM_ERR_HIP, "decode kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

int nddm_simulate_indirect(int32_t model, const float *params, const float *bounds, int64_t B, int32_t n_trials, float dt,
                           int32_t max_steps, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev, uint32_t flags,
                           float ext_sigma, int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata,
                           void *stream)
{
    return nddm::simulate(model, params, model == NDDM_EXPLICIT_BOUNDARY ? bounds : nullptr, B, n_trials, dt, max_steps,
                          seed, set_offset, set_offset_dev, flags, ext_sigma, ext_mode, out_trials, out_summary,
                          model == NDDM_ALPHA_NOT_SCALED ? out_extdata : nullptr, stream);
}

/* simulratcliff (pyhddmjagsutils.py:47-176) as called by alpha_not_scaled.py:95-108: the exact first-passage sampler, no dt.
 * csrc/nddm_ratcliff.h. */
int nddm_simulratcliff(const float *params, int64_t B, int32_t n_trials, uint64_t seed, uint64_t set_offset, uint32_t flags,
                       float ext_sigma, int32_t ext_mode, float *out_trials, float *out_summary, float *out_extdata, void *stream)
{
    using namespace nddm;
    g_err[0] = 0;
    if (B < 0 || n_trials <= 0) return fail(NDDM_ERR_SHAPE, "B < 0 or n_trials <= 0%s");
    if (n_trials >= (1 << 30)) return fail(NDDM_ERR_SHAPE, "n_trials must be < 2^30%s");
    if (set_offset >= (1ull << 60) || set_offset + (uint64_t)B > (1ull << 60))
        return fail(NDDM_ERR_SHAPE, "set_offset + B must be <= 2^60 (the random stream is keyed by 60 bits of the set index)%s");
    if (flags > 1u) return fail(NDDM_ERR_PARAM, "nddm_simulratcliff takes NDDM_GAUSS_EXACT or NDDM_GAUSS_FAST only (there is no step size, "
                                                "hence no bridge, packed layout or float64 state)%s");
    if (B == 0) return NDDM_OK;
    if (!params) return fail(NDDM_ERR_NULL, "params is NULL%s");
    if (!out_trials && !out_summary && !out_extdata) return fail(NDDM_ERR_NULL, "no output buffer given%s");
    const int tile_n_max = 512;
    int tiles = (n_trials + tile_n_max - 1) / tile_n_max;
    const int tile_n = (n_trials + tiles - 1) / tiles;
    tiles = (n_trials + tile_n - 1) / tile_n;
    const long long vB = B * (long long)tiles;
    if (vB >= (1ll << 31)) return fail(NDDM_ERR_SHAPE, "B * ceil(n_trials / 512) must be < 2^31 per launch%s");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (const int rc = check_stream(st)) return rc;
    RatArgs A;
    memset(&A, 0, sizeof A);
    A.params = params; A.out_trials = out_trials; A.out_summary = out_summary; A.out_ext = out_extdata;
    // slots per tile: a tile's trials, but at least 2 -- slot -> tile is a multiply-high by ceil(2^32 / slots), which does not exist for 1;
    // a set of ONE trial is a tile of two slots whose second is a hole (the mechanism of a split set's short last tile)
    const int tile_slots = tile_n < 2 ? 2 : tile_n;
    A.n_vsets = vB; A.n_trials = tile_slots; A.n_total = n_trials; A.tiles_per_set = tiles;
    A.k0 = (uint32_t)seed; A.k1 = (uint32_t)(seed >> 32); A.set_offset = set_offset;
    A.ext_sigma = ext_sigma; A.ext_mode = ext_mode;
    // sets of more than 512 trials are split into tiles whose integer partial sums combine_partials_kernel adds up (stream-ordered
    // scratch; such a launch cannot be captured into a hipGraph)
    unsigned long long *partials = nullptr;
    if (tiles > 1 && out_summary) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(NDDM_ERR_PARAM, "nddm_simulratcliff with summaries of more than 512 trials per set cannot be captured in a hipGraph%s");
        const hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&partials), (size_t)vB * 5 * sizeof(unsigned long long), st);
        if (e != hipSuccess) return fail(NDDM_ERR_HIP, "hipMallocAsync(partial sums): %s", hipGetErrorString(e));
        A.partials = partials;
    }
    // a workgroup works through a GROUP of consecutive tiles at a time: the drain at the end of a group costs ~67 trials' worth of time
    // whatever its size, and the group's staged results (4 B per trial) decide how many workgroups a CU holds (~1000 trials: 6.4 KB of
    // LDS with the rings and the FIFO, six workgroups per SIMD; profiles/r6_ratcliff_shapes.txt).  <= 64 tiles (one lane
    // states one tile's constants).  NDDM_RATCLIFF_GROUP_TRIALS: developer override of the target (A/B runs).
    static const int group_target = [] { const char *e = getenv("NDDM_RATCLIFF_GROUP_TRIALS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 8192 ? v : 1024; }();      // (measured at 300 trials per set: 512 / 1024 / 2048 / 4096 -> 6.3 / 5.4 / 5.8 / 8.5 ms)
    int group = group_target / tile_slots;
    group = group < 1 ? 1 : (group > 64 ? 64 : group);
    if ((long long)group > vB) group = (int)vB;
    A.group = group;
    A.tile_magic = (uint32_t)((0x100000000ull + (unsigned long long)tile_slots - 1ull) / (unsigned long long)tile_slots);
    // uniform rings | drift FIFO | round keys | table | staged results
    const size_t lds = ((size_t)WAVE * 8 + RATCLIFF_FIFO + RATCLIFF_KEYS + (size_t)group * RT_WORDS + (size_t)group * (size_t)tile_slots) * sizeof(float);
    const long long n_groups = (vB + group - 1) / group;
    const dim3 grid((unsigned)n_groups), block(WAVE);
    if (flags & NDDM_GAUSS_FAST) hipLaunchKernelGGL(ratcliff_kernel<true>, grid, block, lds, st, A);
    else hipLaunchKernelGGL(ratcliff_kernel<false>, grid, block, lds, st, A);
    hipError_t e = hipGetLastError();
    int rc = e == hipSuccess ? NDDM_OK : fail(NDDM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (rc == NDDM_OK && partials) {
        const int threads = 256;
        hipLaunchKernelGGL(combine_partials_kernel, dim3((unsigned)((B + threads - 1) / threads)), dim3(threads), 0, st, partials, 5, params, 6,
                           3, (long long)B, tiles, n_trials, 1.52587890625e-05f, out_summary);
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(NDDM_ERR_HIP, "combine kernel launch failed: %s", hipGetErrorString(e));
    }
    if (partials) (void)hipFreeAsync(partials, st);
    return rc;
}

static int draw_prior_impl(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev,
                           float gamma, float *out_params, void *stream)
{
    nddm::g_err[0] = 0;
    if (model != NDDM_BASIC_DDM_DC && model != NDDM_SINGLE_TRIAL && model != NDDM_SINGLE_TRIAL_ALT)
        return nddm::fail(NDDM_ERR_PARAM, "draw_prior: model has no reference prior%s");
    if (B < 0) return nddm::fail(NDDM_ERR_SHAPE, "B < 0%s");
    if (B == 0) return NDDM_OK;
    if (!out_params) return nddm::fail(NDDM_ERR_NULL, "out_params is NULL%s");
    if (const int rc = nddm::check_stream(reinterpret_cast<hipStream_t>(stream))) return rc;
    const int threads = 256;
    const long long blocks = (B + threads - 1) / threads;
    hipLaunchKernelGGL(nddm::prior_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                       reinterpret_cast<hipStream_t>(stream), (int)model, (long long)B, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (unsigned long long)set_offset,
                       reinterpret_cast<const unsigned long long *>(set_offset_dev), gamma, out_params);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nddm::fail(NDDM_ERR_HIP, "prior kernel launch failed: %s", hipGetErrorString(e));
    return NDDM_OK;
}

int nddm_draw_prior(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, float gamma, float *out_params,
                    void *stream)
{
    return draw_prior_impl(model, B, seed, set_offset, nullptr, gamma, out_params, stream);
}

int nddm_draw_prior_indirect(int32_t model, int64_t B, uint64_t seed, uint64_t set_offset, const uint64_t *set_offset_dev,
                             float gamma, float *out_params, void *stream)
{
    return draw_prior_impl(model, B, seed, set_offset, set_offset_dev, gamma, out_params, stream);
}

/* sha256 (hex) of the sources this library was compiled from (build.py passes it; "unknown" for a hand-made build) */
#ifndef NDDM_SOURCE_HASH
#define NDDM_SOURCE_HASH "unknown"
#endif
const char *nddm_source_hash(void)
{
    static const char tag[] = "NDDM_SRC_HASH=" NDDM_SOURCE_HASH;      // the tag makes the hash findable in the file without loading it
    return tag + 14;
}

/* what built this library, as build.py recorded it ("hipcc=<HIP version of the compiler>"; "hipcc=unknown" for a hand-made build):
 * lets a caller report its toolchain without starting the compiler at run time */
#ifndef NDDM_HIPCC_VERSION
#define NDDM_HIPCC_VERSION "unknown"
#endif
const char *nddm_build_info(void) { return "hipcc=" NDDM_HIPCC_VERSION; }

int nddm_debug_normals(const uint32_t *counters, int64_t n, uint32_t k0, uint32_t k1, uint32_t flags, float *out,
                       void *stream)
{
    nddm::g_err[0] = 0;
    if (n < 0) return nddm::fail(NDDM_ERR_SHAPE, "n < 0%s");
    if (n == 0) return NDDM_OK;
    if (!counters || !out) return nddm::fail(NDDM_ERR_NULL, "null pointer%s");
    if (const int rc = nddm::check_stream(reinterpret_cast<hipStream_t>(stream))) return rc;
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

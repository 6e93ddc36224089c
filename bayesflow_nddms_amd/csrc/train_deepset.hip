// train_deepset.hip -- the per-trial part of the amortizer's DeepSet summary network (bayesflow_nddms_amd/amortizer.py::
// InvariantNetwork; the stand-in for bf.networks.InvariantNetwork, basic_ddm_dc.py:163) for the online-training loop
// (basic_ddm_dc.py:199-202): every 3-layer MLP that runs on each trial of each set -- Linear, ReLU, Linear, ReLU, Linear, hidden
// width 64 -- as ONE kernel forward and ONE backward, with what surrounds it fused in:
//   * a per-set context vector (the masked mean over the set's trials of another MLP's output, taken through the context columns
//     of W1) added before the first ReLU: the "equivariant" half of a DeepSet block;
//   * the masked per-set sums of the output: the "invariant" half and the final pooling;
//   * backward: the weight gradients as per-workgroup partial sums (deterministic; one reduction kernel sums them), the input
//     gradient, the context's gradient, the pooled output's gradient spread back over the set's trials.
// In PyTorch each Linear / ReLU / mask / sum / concat / expand is a launch of 4-7 microseconds over [32 sets x <= 300 trials, 64]:
// ~150 launches per training iteration.  Here the [rows, 64] x [64, 64] products are v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered
// fmaf chain) on LDS tiles of 64 rows; a workgroup owns up to `rows_per_wg` consecutive trials of ONE set, so pooling and context
// never cross a workgroup inside a kernel -- they cross KERNELS as [set, workgroup-of-the-set, 64] partial sums.
// gfx950 only.  tests/test_gpu_training.py compares with the PyTorch composition.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nddm_deepset {

constexpr int HS = 64;        // hidden width
constexpr int LD = HS + 4;    // LDS row stride: 16-byte aligned rows, and the operand reads of one wave instruction touch every bank twice
constexpr int TM = 64;        // rows per tile
constexpr int NT = 256;       // threads per workgroup: wave w owns the 32 x 32 tile (w >> 1, w & 1) of every 64 x 64 product
constexpr int DS_MAX = 4;     // a "small" input (the raw trials: rt, choice) goes through layer 1 as plain FMAs
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // (register arrays of HIP's float4 STRUCT were seen kept in memory; of this type not)

#ifdef NDDM_TRAIN_STAMPS      // development only: phase time stamps of one workgroup (tools/train_stamps.py)
__device__ unsigned long long g_stamps[8192];
__device__ int g_nstamps;
#ifndef NDDM_STAMP_BLOCK
#define NDDM_STAMP_BLOCK 0
#endif
#define STAMP(id) do { if (threadIdx.x == 0 && blockIdx.x == NDDM_STAMP_BLOCK) { int k_ = g_nstamps++; if (k_ < 4096) { g_stamps[2 * k_] = (id); g_stamps[2 * k_ + 1] = wall_clock64(); } } } while (0)
#else
#define STAMP(id) do { } while (0)
#endif

struct Mlp { const float *W1; int ldw1; const float *b1, *W2, *b2, *W3, *b3; };
// W1 [64, ldw1]: columns [0, d_in) act on the trial, columns [d_in, d_in + 64) on the set's pooled context (if there is one)

struct Common {
    const float *x; int d_in;               // [B * N, d_in]
    int B, N, S, rows_per_wg;               // S workgroups per set
    const float *mask; int mask_is_count;   // [N]: 1 = a real trial, 0 = padding; null: all real.  mask_is_count: ONE float, the number
                                            // of real trials n (device scalar): trial i is real iff i < n, and 1 / n is taken from it
    const float *inv_n; float inv_n_host;   // 1 / (number of real trials): device scalar, or (null) the host's value
    const float *ctx_part; int S_ctx;       // [B, S_ctx, 64] partial sums of the pooled producer; null: no context
    Mlp P;
    int d_out;                              // rows of W3 / b3 (<= 64): 64 for the per-trial MLPs, the summary width for the last MLP
    const float *x_part; int S_x;           // not null (d_in == 64): row r of x IS inv_n * sum over s of x_part[r][s][.] -- the MLP
};                                          // after the pooling runs on the pooled means, one row per set (launched as ONE "set" of B rows)

// acc += A B over 64 k's.  Lane l holds A[i = l & 31][k] and B[k][j = l & 31] for the 32 k's of its half (l >> 5): the order of
// the sum is free, so step s of the instruction stream takes k = 32 (l >> 5) + s from both.  SA / SB: distance in floats between
// consecutive k's of this lane's operand (1: along a row of the LDS tile -- 16-byte reads; LD: down a column).
template <int SA, int SB>
__device__ __forceinline__ f32x16 mma64(const float *ap, const float *bp, f32x16 acc)
{
    float a[32], b[32];
    if (SA == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(ap + 4 * q);
            a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) a[s] = ap[s * SA];
    }
    if (SB == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(bp + 4 * q);
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) b[s] = bp[s * SB];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}
// register v of lane l of a result tile: row (v & 3) + 8 (v >> 2) + 4 (l >> 5), column l & 31
__device__ __forceinline__ int drow(int v, int kk) { return (v & 3) + 8 * (v >> 2) + 4 * kk; }

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.0f;
    return z;
}

// [rows <= 64] x 64 row-major weights -> LDS [64][LD] (rows beyond: zero), 16-byte loads (ldw: the source's row stride in floats,
// a multiple of 4)
__device__ __forceinline__ void stage64(float (*dst)[LD], const float *src, int ldw, int t, int rows = HS)
{
    float4 v[HS * HS / 4 / NT];
#pragma unroll                                       // all loads first (from a clamped row: no branch), then the stores
    for (int k = 0; k < HS * HS / 4 / NT; ++k) {
        const int p = t + NT * k, r = p >> 4, c4 = p & 15;
        v[k] = *reinterpret_cast<const float4 *>(src + (long long)min(r, rows - 1) * ldw + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < HS * HS / 4 / NT; ++k) {
        const int p = t + NT * k, r = p >> 4, c4 = p & 15;
        *reinterpret_cast<float4 *>(&dst[r][4 * c4]) = r < rows ? v[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// the same in two halves: the loads into registers now, the stores to LDS when the buffer's last readers are done / its first reader
// is about to run -- a kernel that issues EVERY load up front and stores each operand just before the phase that needs it exposes
// one load latency instead of one per operand
__device__ __forceinline__ void fetch64(f32x4 (&v)[HS * HS / 4 / NT], const float *src, int ldw, int t, int rows = HS)
{
#pragma unroll
    for (int k = 0; k < HS * HS / 4 / NT; ++k) {
        const int p = t + NT * k, r = p >> 4, c4 = p & 15;
        v[k] = *reinterpret_cast<const f32x4 *>(src + (long long)min(r, rows - 1) * ldw + 4 * c4);
    }
}
__device__ __forceinline__ void store64(float (*dst)[LD], const f32x4 (&v)[HS * HS / 4 / NT], int t, int rows = HS)
{
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < HS * HS / 4 / NT; ++k) {
        const int p = t + NT * k, r = p >> 4, c4 = p & 15;
        *reinterpret_cast<f32x4 *>(&dst[r][4 * c4]) = r < rows ? v[k] : z;
    }
}

// sum over s = 0 .. S - 1 of p[s * stride], added in that order -- with the loads of eight terms in flight at a time: as a plain loop
// (the trip count is a kernel argument) every term waited for its own round trip to L2, 5 x ~0.6 us at the head of a kernel
__device__ __forceinline__ float sum_in_order(const float *p, int S, long long stride)
{
    float a = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(long long)min(s0 + k, S - 1) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += s0 + k < S ? v[k] : 0.0f;
    }
    return a;
}

// four columns of row `row` of the input of a 64-wide MLP: loaded, or (x_part) the mean of the pooled partial sums
__device__ __forceinline__ float4 x_row4(const Common &C, long long row, int c4, float inv_n)
{
    if (!C.x_part) return *reinterpret_cast<const float4 *>(C.x + row * HS + 4 * c4);
    float4 a = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0; s0 < C.S_x; s0 += 8) {          // (eight partial rows in flight, added in order: see sum_in_order)
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(C.x_part + (row * C.S_x + min(s0 + k, C.S_x - 1)) * HS + 4 * c4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float f = s0 + k < C.S_x ? 1.0f : 0.0f;
            a.x += f * v[k].x; a.y += f * v[k].y; a.z += f * v[k].z; a.w += f * v[k].w;
        }
    }
    a.x *= inv_n; a.y *= inv_n; a.z *= inv_n; a.w *= inv_n;
    return a;
}

__device__ __forceinline__ float inv_count(const Common &C)
{
    return C.mask_is_count ? 1.0f / *C.mask : (C.inv_n ? *C.inv_n : C.inv_n_host);
}
__device__ __forceinline__ float mask_of(const Common &C, int n)
{
    return !C.mask ? 1.0f : (C.mask_is_count ? (n < *C.mask ? 1.0f : 0.0f) : C.mask[n]);
}

// mask_of(n) for a row that may lie beyond the set (-> 0): no branch around the load
__device__ __forceinline__ float mask_or_zero(const Common &C, int n, int n_end)
{
    const float mk = !C.mask ? 1.0f : (C.mask_is_count ? (n < *C.mask ? 1.0f : 0.0f) : C.mask[min(n, n_end - 1)]);
    return n < n_end ? mk : 0.0f;
}

// A tile of rows of a row-major [., 64] global array as a RAW BUFFER whose size is the tile's valid rows: a store (load) to a row beyond
// them is dropped (returns 0) by the hardware's range check -- no `if (row < n_end)` around each of an epilogue's 16 stores, which
// compiled to a branch, an exec save / restore and a 64-bit address computation per store (~1000 cycles of a 4500-cycle phase).
struct RowTile {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ RowTile(const float *tile_base, int n_rows)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile_base), 0, (n_rows > 0 ? n_rows : 0) * HS * 4, 0x00020000)) {}
    __device__ __forceinline__ void put(int row, int col, float v) const
    { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (row * HS + col) * 4, 0, 0); }
    // (dword accesses only: hipcc 7.2.26015 compiles __builtin_amdgcn_raw_buffer_load_b128 to ONE buffer_load_dword -- x is loaded, and y,
    //  z, w come back as COPIES OF x instead of their dwords of the source (got 1 1 1 1 5 5 5 5 for a source 1 2 3 4 5 6 7 8).  Probe:
    //  tools/probe_buffer_load_b128.hip, its MI355X output profiles/r6_probe_buffer_load_b128.txt; the ISA side of the
    //  claim is asserted without a GPU by tests/test_kernel_resources.py::test_toolchain_probes_compile)
    __device__ __forceinline__ float get(int row, int col) const
    { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (row * HS + col) * 4, 0, 0)); }
};

// the set's pooled context: pooled[k] = inv_n * sum over the producer's workgroups; cs[u] = b1[u] + W1[u][d_in + .] . pooled.
// In two halves: every load first -- the partial sums AND this thread's 16 context-column weights (thread (unit t / 4, quarter t % 4)),
// which do not depend on the sums but stood behind the barrier that publishes them: one round trip to L2 instead of two -- then the
// LDS hand-over and the product.
struct CtxLoad { float sum, b1; float w[16]; };
__device__ __forceinline__ void context_issue(const Common &C, int b, int t, CtxLoad &L)
{
    L.sum = 0.0f;
    if (C.ctx_part) {
        const int u = t >> 2, q = t & 3;
        const float *w = C.P.W1 + (long long)u * C.P.ldw1 + C.d_in + 16 * q;
#pragma unroll
        for (int k = 0; k < 16; ++k) L.w[k] = w[k];
        L.b1 = C.P.b1[u];
        if (t < HS) L.sum = sum_in_order(C.ctx_part + (long long)b * C.S_ctx * HS + t, C.S_ctx, HS);
    } else {
        L.b1 = C.P.b1[t & (HS - 1)];
    }
}
__device__ __forceinline__ void context_publish(const Common &C, float *pooled, int t, const CtxLoad &L)
{
    if (C.ctx_part && t < HS) pooled[t] = L.sum * inv_count(C);
}
__device__ __forceinline__ void context_product(const Common &C, const float *pooled, float *cs, int t, const CtxLoad &L)   // (behind a barrier)
{
    if (C.ctx_part) {
        const int u = t >> 2, q = t & 3;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(L.w[k], pooled[16 * q + k], a);
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        if (q == 0) cs[u] = a + L.b1;
    } else if (t < HS) {
        cs[t] = L.b1;
    }
}
__device__ __forceinline__ void context(const Common &C, int b, float *pooled, float *cs, int t)
{
    CtxLoad L;
    context_issue(C, b, t, L);
    context_publish(C, pooled, t, L);
    if (C.ctx_part) __syncthreads();
    context_product(C, pooled, cs, t, L);
}

// ------------------------------------------------------------------------------------------------ forward
struct FwdOut {
    float *h1, *h2;        // [B * N, 64] the two hidden activations (after their ReLU): saved for the backward
    float *y;              // [B * N, ldy] the output (columns [0, d_out)), or null (only its pooled sums are wanted)
    float *pool_part;      // [B, S, 64] masked sums of the output over this workgroup's trials, or null
    int ldy;               // row stride of y (>= d_out + n_extra)
    const float *extra;    // not null: columns [d_out, d_out + n_extra) of y are filled from extra[row * extra_stride + .] -- the
    int n_extra, extra_stride;   // "direct conditions" of the amortizer (log N) appended to the summary without a concatenation
};

template <bool BIG>        // d_in == 64 (layer 1 is an MFMA product) or d_in <= DS_MAX
__global__ __launch_bounds__(NT) void mlp_fwd_kernel(Common C, FwdOut O)
{
    __shared__ __attribute__((aligned(16))) float w1s[BIG ? HS : 1][LD];
    __shared__ __attribute__((aligned(16))) float w2s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w3s[HS][LD];
    __shared__ __attribute__((aligned(16))) float xs[BIG ? TM : 1][LD];
    __shared__ __attribute__((aligned(16))) float h1s[TM][LD];
    __shared__ __attribute__((aligned(16))) float h2s[TM][LD];
    __shared__ float w1small[BIG ? 1 : HS][DS_MAX], xsmall[BIG ? 1 : TM][DS_MAX];
    __shared__ float cs[HS], pooled[HS], red[4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, kk = lane >> 5, rb = wave >> 1, ub = wave & 1;
    const int b = blockIdx.x / C.S, sp = blockIdx.x - b * C.S;
    const int n_begin = sp * C.rows_per_wg, n_end = min(C.N, n_begin + C.rows_per_wg);
    const long long row0 = (long long)b * C.N;
    STAMP(100);
    const float inv_n = inv_count(C);
    auto load_x_tile = [&](const int n0) {           // rows beyond the workgroup's range: zero
        if (BIG) {
#pragma unroll
            for (int k = 0; k < TM * HS / 4 / NT; ++k) {
                const int p = t + NT * k, r = p >> 4, c4 = p & 15, n = n0 + r;
                float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (n < n_end) v = x_row4(C, row0 + n, c4, inv_n);
                *reinterpret_cast<float4 *>(&xs[r][4 * c4]) = v;
            }
        } else {
            for (int p = t; p < TM * C.d_in; p += NT) {
                const int r = p / C.d_in, c = p - r * C.d_in, n = n0 + r;
                xsmall[r][c] = n < n_end ? C.x[(row0 + n) * C.d_in + c] : 0.0f;
            }
        }
    };
    // every global load first: W1 (needed first), W2, W3 into registers; W2 and W3 go to LDS behind layers 1 and 2 (the barriers
    // that are there anyway), so their latency runs beside the input tile's, the context's and layer 1
    // (the loads complete in issue order: the input tile and W1 first, they are stored first)
    f32x4 r1[HS * HS / 4 / NT], r2[HS * HS / 4 / NT], r3[HS * HS / 4 / NT];
    float4 rx[TM * HS / 4 / NT];
    if (BIG) {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            rx[k] = x_row4(C, row0 + min(n_begin + r, n_end - 1), c4, inv_n);       // (clamped: rows beyond the range are zeroed at the store)
        }
    } else load_x_tile(n_begin);
    fetch64(r1, BIG ? C.P.W1 : C.P.W2, BIG ? C.P.ldw1 : HS, t);
    fetch64(r2, C.P.W2, HS, t);
    fetch64(r3, C.P.W3, HS, t, C.d_out);
    if (BIG) {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            *reinterpret_cast<float4 *>(&xs[r][4 * c4]) = n_begin + r < n_end ? rx[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        store64(w1s, r1, t);
    } else
        for (int p = t; p < HS * C.d_in; p += NT) { const int r = p / C.d_in, c = p - r * C.d_in; w1small[r][c] = C.P.W1[(long long)r * C.P.ldw1 + c]; }
    context(C, b, pooled, cs, t);
    __syncthreads();
    const int u = 32 * ub + m;                       // this lane's column of every result tile
    const float bias2 = C.P.b2[u], bias3 = u < C.d_out ? C.P.b3[u] : 0.0f, bias1 = cs[u];
    float pacc = 0.0f;
    STAMP(101);
    for (int n0 = n_begin; n0 < n_end; n0 += TM) {
        if (n0 != n_begin) {
            load_x_tile(n0);
            __syncthreads();
        }
        STAMP(102);
        const RowTile th1(O.h1 + (row0 + n0) * HS, n_end - n0), th2(O.h2 + (row0 + n0) * HS, n_end - n0);
        if (BIG) {                                   // layer 1
            const f32x16 acc = mma64<1, 1>(&xs[32 * rb + m][32 * kk], &w1s[u][32 * kk], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                const float val = fmaxf(acc[v] + bias1, 0.0f);
                h1s[r][u] = val;
                th1.put(r, u, val);
            }
        } else {                                     // thread (unit t & 63, rows 16 (t >> 6) ...)
            const int uu = t & 63, rg = t >> 6;
            float w[DS_MAX];
#pragma unroll
            for (int c = 0; c < DS_MAX; ++c) w[c] = c < C.d_in ? w1small[uu][c] : 0.0f;
            const float c0 = cs[uu];
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int r = 16 * rg + q;
                float a = c0;
#pragma unroll
                for (int c = 0; c < DS_MAX; ++c) if (c < C.d_in) a = fmaf(w[c], xsmall[r][c], a);
                const float val = fmaxf(a, 0.0f);
                h1s[r][uu] = val;
                th1.put(r, uu, val);
            }
        }
        if (n0 == n_begin) store64(w2s, r2, t);      // (its first reader is layer 2, behind the barrier)
        __syncthreads();
        STAMP(103);
        {                                            // layer 2
            const f32x16 acc = mma64<1, 1>(&h1s[32 * rb + m][32 * kk], &w2s[u][32 * kk], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                const float val = fmaxf(acc[v] + bias2, 0.0f);
                h2s[r][u] = val;
                th2.put(r, u, val);
            }
        }
        if (n0 == n_begin) store64(w3s, r3, t, C.d_out);
        __syncthreads();
        STAMP(104);
        {                                            // layer 3 (no activation), and the masked sums of its output
            const f32x16 acc = mma64<1, 1>(&h2s[32 * rb + m][32 * kk], &w3s[u][32 * kk], zero16());
            // (branch-free: y as a raw buffer of the tile's valid rows -- of NO rows if y is not wanted -- whose range check drops the
            //  stores of the other rows; a lane whose column is not written aims past the buffer; the pooled sum takes a zero mask)
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(O.y ? O.y + (row0 + n0) * O.ldy : O.h1, 0,
                                                                                  O.y ? (n_end - n0) * O.ldy * 4 : 0, 0x00020000);
            const bool colv = u < C.d_out, cole = O.extra && !colv && u < C.d_out + O.n_extra;
            const int ue = min(max(u - C.d_out, 0), max(O.n_extra - 1, 0));
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk), n = n0 + r;
                const float val = acc[v] + bias3;
                if (O.pool_part) pacc = fmaf(colv ? mask_or_zero(C, n, n_end) : 0.0f, val, pacc);
                float outv = val;
                if (O.extra) { const float e = O.extra[(row0 + min(n, n_end - 1)) * O.extra_stride + ue]; outv = cole ? e : val; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, outv), ry, (colv || cole) ? (r * O.ldy + u) * 4 : 0x7ffffff0, 0, 0);
            }
        }
        STAMP(105);
        // (the next tile's writes to xs / h1s / h2s are each behind a barrier every reader of this tile has passed)
    }
    if (O.pool_part) {                               // fixed-order sum of the four lanes that share a column
        red[wave][lane] = pacc;
        __syncthreads();
        if (t < HS) {
            const int cb = t >> 5, n = t & 31;
            O.pool_part[((long long)b * C.S + sp) * HS + t] = (red[cb][n] + red[cb][n + 32]) + (red[2 + cb][n] + red[2 + cb][n + 32]);
        }
    }
}

// ---- two MLPs in one launch --------------------------------------------------------------------------------------------------
// An equivariant MLP (A: input x, the set's pooled context through W1's context columns, output y) and the MLP that consumes its
// output row by row (B: the NEXT block's invariant MLP, or the pre-pooling MLP; no context; only the masked per-set sums of its
// output are wanted).  B's rows are A's rows, so nothing crosses a workgroup between them: the y tile stays in LDS as B's input,
// B's three weight matrices are fetched into registers while A computes and stored to LDS when A's readers are done -- one launch
// and one exposed weight-staging latency instead of two of each.  One 64-row tile per workgroup (rows_per_wg == TM).
template <bool BIG>        // of MLP A: d_in == 64 or d_in <= DS_MAX (B's input is A's 64-wide output)
__global__ __launch_bounds__(NT) void mlp2_fwd_kernel(Common C, FwdOut O, Mlp P2, FwdOut O2)
{
    __shared__ __attribute__((aligned(16))) float w1s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w2s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w3s[HS][LD];
    __shared__ __attribute__((aligned(16))) float xs[TM][LD];
    __shared__ __attribute__((aligned(16))) float h1s[TM][LD];
    __shared__ __attribute__((aligned(16))) float h2s[TM][LD];
    __shared__ float w1small[BIG ? 1 : HS][DS_MAX], xsmall[BIG ? 1 : TM][DS_MAX];
    __shared__ float cs[HS], pooled[HS], red[4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, kk = lane >> 5, rb = wave >> 1, ub = wave & 1;
    const int b = blockIdx.x / C.S, sp = blockIdx.x - b * C.S;
    const int n0 = sp * C.rows_per_wg, n_end = min(C.N, n0 + C.rows_per_wg);
    const long long row0 = (long long)b * C.N;
    const float inv_n = inv_count(C);
    // ---- every global load up front, in the order of need: A's input tile and W1 (stored at once), A's W2 and W3 (stored behind
    //      A's layers 1 and 2), B's three matrices (stored when A is done)
    f32x4 r1[HS * HS / 4 / NT], r2[HS * HS / 4 / NT], r3[HS * HS / 4 / NT];
    float4 rx[TM * HS / 4 / NT];
    if (BIG) {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            rx[k] = x_row4(C, row0 + min(n0 + r, n_end - 1), c4, inv_n);
        }
    } else {
        for (int p = t; p < TM * C.d_in; p += NT) {
            const int r = p / C.d_in, c = p - r * C.d_in, n = n0 + r;
            xsmall[r][c] = n < n_end ? C.x[(row0 + n) * C.d_in + c] : 0.0f;
        }
    }
    fetch64(r1, BIG ? C.P.W1 : C.P.W2, BIG ? C.P.ldw1 : HS, t);
    fetch64(r2, C.P.W2, HS, t);
    fetch64(r3, C.P.W3, HS, t);
    f32x4 q1[HS * HS / 4 / NT], q2[HS * HS / 4 / NT], q3[HS * HS / 4 / NT];
    fetch64(q1, P2.W1, HS, t);
    fetch64(q2, P2.W2, HS, t);
    fetch64(q3, P2.W3, HS, t);
    if (BIG) {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            *reinterpret_cast<float4 *>(&xs[r][4 * c4]) = n0 + r < n_end ? rx[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        store64(w1s, r1, t);
    } else
        for (int p = t; p < HS * C.d_in; p += NT) { const int r = p / C.d_in, c = p - r * C.d_in; w1small[r][c] = C.P.W1[(long long)r * C.P.ldw1 + c]; }
    context(C, b, pooled, cs, t);
    __syncthreads();
    const int u = 32 * ub + m;                       // this lane's column of every result tile
    const float bias2 = C.P.b2[u], bias3 = C.P.b3[u], bias1 = cs[u];
    const float c1 = P2.b1[u], c2 = P2.b2[u], c3 = P2.b3[u];
    const int nv = n_end - n0;                       // the tile's valid rows: stores beyond them are dropped by the buffers' range check
    const RowTile th1(O.h1 + (row0 + n0) * HS, nv), th2(O.h2 + (row0 + n0) * HS, nv), ty(O.y + (row0 + n0) * HS, nv);
    const RowTile tg1(O2.h1 + (row0 + n0) * HS, nv), tg2(O2.h2 + (row0 + n0) * HS, nv);
    // ---- MLP A
    if (BIG) {
        const f32x16 acc = mma64<1, 1>(&xs[32 * rb + m][32 * kk], &w1s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            const float val = fmaxf(acc[v] + bias1, 0.0f);
            h1s[r][u] = val;
            th1.put(r, u, val);
        }
    } else {
        const int uu = t & 63, rg = t >> 6;
        float w[DS_MAX];
#pragma unroll
        for (int c = 0; c < DS_MAX; ++c) w[c] = c < C.d_in ? w1small[uu][c] : 0.0f;
        const float c0 = cs[uu];
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int r = 16 * rg + q;
            float a = c0;
#pragma unroll
            for (int c = 0; c < DS_MAX; ++c) if (c < C.d_in) a = fmaf(w[c], xsmall[r][c], a);
            const float val = fmaxf(a, 0.0f);
            h1s[r][uu] = val;
            th1.put(r, uu, val);
        }
    }
    store64(w2s, r2, t);
    __syncthreads();
    {
        const f32x16 acc = mma64<1, 1>(&h1s[32 * rb + m][32 * kk], &w2s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            const float val = fmaxf(acc[v] + bias2, 0.0f);
            h2s[r][u] = val;
            th2.put(r, u, val);
        }
    }
    store64(w3s, r3, t);
    __syncthreads();
    {   // A's output: to global memory (the next equivariant MLP and the backward read it) and into xs as B's input
        const f32x16 acc = mma64<1, 1>(&h2s[32 * rb + m][32 * kk], &w3s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk), n = n0 + r;
            const float val = n < n_end ? acc[v] + bias3 : 0.0f;          // (rows beyond the set: zero, as a loaded tile has them)
            xs[r][u] = val;                                               // (xs: A's layer 1 finished reading it two barriers ago)
            ty.put(r, u, val);
        }
    }
    __syncthreads();                                 // A's readers of w1s / w2s / w3s / h1s / h2s are done; xs is complete
    store64(w1s, q1, t);
    store64(w2s, q2, t);
    store64(w3s, q3, t);
    __syncthreads();
    // ---- MLP B
    {
        const f32x16 acc = mma64<1, 1>(&xs[32 * rb + m][32 * kk], &w1s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            const float val = fmaxf(acc[v] + c1, 0.0f);
            h1s[r][u] = val;
            tg1.put(r, u, val);
        }
    }
    __syncthreads();
    {
        const f32x16 acc = mma64<1, 1>(&h1s[32 * rb + m][32 * kk], &w2s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            const float val = fmaxf(acc[v] + c2, 0.0f);
            h2s[r][u] = val;
            tg2.put(r, u, val);
        }
    }
    __syncthreads();
    float pacc = 0.0f;
    {
        const f32x16 acc = mma64<1, 1>(&h2s[32 * rb + m][32 * kk], &w3s[u][32 * kk], zero16());
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int n = n0 + 32 * rb + drow(v, kk);
            pacc = fmaf(mask_or_zero(C, n, n_end), acc[v] + c3, pacc);
        }
    }
    red[wave][lane] = pacc;                          // fixed-order sum of the four lanes that share a column (as mlp_fwd_kernel)
    __syncthreads();
    if (t < HS) {
        const int cb = t >> 5, n = t & 31;
        O2.pool_part[((long long)b * C.S + sp) * HS + t] = (red[cb][n] + red[cb][n + 32]) + (red[2 + cb][n] + red[2 + cb][n + 32]);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// The gradient of a pooled output per unit of the set, gp[k] = inv_n * sum_i gp_W[i][k] * (sum over the consumer's workgroups of gpool[.][i]),
// in the same two halves as the context: the partial sums and this thread's 16 weights (thread (unit t / 4, quarter t % 4)) in one round trip.
struct GpLoad { float sum; float w[16]; };
__device__ __forceinline__ void gp_issue(const float *gpool, int gp_S, const float *gp_W, int gp_ldw, int b, int t, GpLoad &L)
{
    const int k = t >> 2, q = t & 3;
#pragma unroll
    for (int i = 0; i < 16; ++i) L.w[i] = gp_W[(long long)(16 * q + i) * gp_ldw + k];
    L.sum = t < HS ? sum_in_order(gpool + (long long)b * gp_S * HS + t, gp_S, HS) : 0.0f;
}
__device__ __forceinline__ void gp_product(const float *dsum, float *gp, float inv_n, int t, const GpLoad &L)      // (behind a barrier)
{
    const int k = t >> 2, q = t & 3;
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a = fmaf(L.w[i], dsum[16 * q + i], a);
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (q == 0) gp[k] = a * inv_n;
}

struct BwdIO {
    const float *h1, *h2;            // saved by the forward
    const float *gy; int ldgy;       // [B * N, ldgy] gradient of the output (columns [0, d_out)), or null
    const float *gpool; int gp_S;    // gradient through the pooled output (null: none), one of two forms:
    const float *gp_W; int gp_ldw;   //   gp_W != null: gpool [B, gp_S, 64] = the consumer's per-workgroup sums of d(pre-activation 1),
                                     //     to be taken through the consumer's context columns gp_W [64, gp_ldw] and 1 / n;
                                     //   gp_W == null: gpool [B, 64] = the gradient of the masked MEAN (1 / n is applied here)
    float *gx; int gx_acc;           // [B * N, d_in] gradient of the input, or null; gx_acc: add to what it holds
    float *dctx_part;                // [B, S, 64] this workgroup's sum of d(pre-activation 1) (the context's gradient), or null
    float *wpart; int ld_part;       // row blockIdx.x (stride ld_part floats) takes this workgroup's weight gradients:
                                     // W1 [64, ldw1], b1, W2, b2, W3, b3
};

template <bool BIG>
__global__ __launch_bounds__(NT) void mlp_bwd_kernel(Common C, BwdIO Q)
{
    __shared__ __attribute__((aligned(16))) float w1s[BIG ? HS : 1][LD];
    __shared__ __attribute__((aligned(16))) float w2s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w3s[HS][LD];
    __shared__ __attribute__((aligned(16))) float xs[BIG ? TM : 1][LD];
    __shared__ __attribute__((aligned(16))) float h1s[TM][LD];
    __shared__ __attribute__((aligned(16))) float h2s[TM][LD];
    __shared__ __attribute__((aligned(16))) float gs[TM][LD];        // gradient of the output; later d(pre-activation 1)
    __shared__ __attribute__((aligned(16))) float d2s[TM][LD];       // d(pre-activation 2)
    __shared__ float xsmall[BIG ? 1 : TM][DS_MAX];
    __shared__ float pooled[HS], cs[HS], gp[HS], dsum[HS], red[4][3][HS], redw[BIG ? 1 : 4][HS][DS_MAX], db1s[HS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, kk = lane >> 5, rb = wave >> 1, ub = wave & 1;
    const int b = blockIdx.x / C.S, sp = blockIdx.x - b * C.S;
    const int n_begin = sp * C.rows_per_wg, n_end = min(C.N, n_begin + C.rows_per_wg);
    const long long row0 = (long long)b * C.N;
    const float inv_n = inv_count(C);
    STAMP(200);
    auto load_tile = [&](const int n0) {             // saved activations, input and output gradient (the pooled part: added below)
        // every load unconditional, from a clamped row (rows beyond the set are zeroed at the store), and ALL of them issued before
        // the first store: inside `if (n < n_end)` each of the four passes was a branch and a round trip to L2 of its own
        // (f32x4, the compiler's own vector type: arrays of HIP's float4 struct are kept in memory)
        f32x4 v1[TM * HS / 4 / NT], v2[TM * HS / 4 / NT], vg[TM * HS / 4 / NT], vx[TM * HS / 4 / NT];
        const bool gy_dense = C.d_out == HS && Q.ldgy == HS;
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            const long long row = row0 + min(n0 + r, n_end - 1), o = row * HS + 4 * c4;
            v1[k] = *reinterpret_cast<const f32x4 *>(Q.h1 + o);
            v2[k] = *reinterpret_cast<const f32x4 *>(Q.h2 + o);
            vg[k] = z;
            vx[k] = z;
            if (Q.gy) {
                if (gy_dense) vg[k] = *reinterpret_cast<const f32x4 *>(Q.gy + o);
                else {                               // a narrow output (the summary): columns < d_out, rows ldgy floats apart
                    const float *gr = Q.gy + row * Q.ldgy;
                    const int cl = C.d_out - 1;
                    const float a0 = gr[min(4 * c4, cl)], a1 = gr[min(4 * c4 + 1, cl)], a2 = gr[min(4 * c4 + 2, cl)], a3 = gr[min(4 * c4 + 3, cl)];
                    vg[k] = f32x4{4 * c4 < C.d_out ? a0 : 0.0f, 4 * c4 + 1 < C.d_out ? a1 : 0.0f, 4 * c4 + 2 < C.d_out ? a2 : 0.0f,
                                  4 * c4 + 3 < C.d_out ? a3 : 0.0f};
                }
            }
            if (BIG) { const float4 xv = x_row4(C, row, c4, inv_n); vx[k] = f32x4{xv.x, xv.y, xv.z, xv.w}; }
        }
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            const bool ok = n0 + r < n_end;
            *reinterpret_cast<f32x4 *>(&h1s[r][4 * c4]) = ok ? v1[k] : z;
            *reinterpret_cast<f32x4 *>(&h2s[r][4 * c4]) = ok ? v2[k] : z;
            *reinterpret_cast<f32x4 *>(&gs[r][4 * c4]) = ok ? vg[k] : z;
            if (BIG) *reinterpret_cast<f32x4 *>(&xs[r][4 * c4]) = ok ? vx[k] : z;
        }
        if (!BIG)
            for (int p = t; p < TM * C.d_in; p += NT) {
                const int r = p / C.d_in, c = p - r * C.d_in, n = n0 + r;
                xsmall[r][c] = n < n_end ? C.x[(row0 + n) * C.d_in + c] : 0.0f;
            }
    };
    // the small loads that feed the context and the pooled gradient go FIRST (loads return in issue order): they are back while the
    // tiles and the weights are still on their way
    CtxLoad Lc;
    GpLoad Lg;
    const bool gpw = Q.gpool && Q.gp_W;
    context_issue(C, b, t, Lc);
    if (gpw) gp_issue(Q.gpool, Q.gp_S, Q.gp_W, Q.gp_ldw, b, t, Lg);
    {   // every weight load in flight before the first store (three stage64 calls in a row waited for three load latencies in turn)
        f32x4 r1[HS * HS / 4 / NT], r2[HS * HS / 4 / NT], r3[HS * HS / 4 / NT];
        fetch64(r3, C.P.W3, HS, t, C.d_out);
        fetch64(r2, C.P.W2, HS, t);
        fetch64(r1, BIG ? C.P.W1 : C.P.W2, BIG ? C.P.ldw1 : HS, t);
        load_tile(n_begin);
        store64(w3s, r3, t, C.d_out);
        store64(w2s, r2, t);
        if (BIG && Q.gx) store64(w1s, r1, t);
    }
    context_publish(C, pooled, t, Lc);               // (pooled: also for the context columns' weight gradient)
    if (gpw && t < HS) dsum[t] = Lg.sum;
    __syncthreads();
    context_product(C, pooled, cs, t, Lc);
    if (gpw) gp_product(dsum, gp, inv_n, t, Lg);     // gradient of the pooled output, per unit of this set
    else if (Q.gpool && t < HS) gp[t] = Q.gpool[(long long)b * HS + t] * inv_n;
    __syncthreads();
    const int u = 32 * ub + m;
    f32x16 aW1 = zero16(), aW2 = zero16(), aW3 = zero16();
    float db[3] = {0.0f, 0.0f, 0.0f};                // thread (unit t & 63, rows 16 (t >> 6) ..): column sums of d(pre-activation 1..3)
    float dw1[DS_MAX] = {0.0f, 0.0f, 0.0f, 0.0f};    // (!BIG) row t & 63 of dW1 over the same rows
    const int uu = t & 63, rg = t >> 6;
    for (int n0 = n_begin; n0 < n_end; n0 += TM) {
        if (n0 != n_begin) {
            __syncthreads();                         // the previous tile's readers are done
            load_tile(n0);
        }
        STAMP(201);
        if (Q.gpool) {                               // the pooled output's gradient, spread over the set's real trials
            if (n0 != n_begin) __syncthreads();
#pragma unroll
            for (int k = 0; k < TM * HS / 4 / NT; ++k) {
                const int p = t + NT * k, r = p >> 4, c4 = p & 15, n = n0 + r;
                if (n < n_end) {
                    const float mk = mask_of(C, n);
                    float4 vg = *reinterpret_cast<float4 *>(&gs[r][4 * c4]);
                    vg.x = fmaf(mk, gp[4 * c4], vg.x); vg.y = fmaf(mk, gp[4 * c4 + 1], vg.y);
                    vg.z = fmaf(mk, gp[4 * c4 + 2], vg.z); vg.w = fmaf(mk, gp[4 * c4 + 3], vg.w);
                    *reinterpret_cast<float4 *>(&gs[r][4 * c4]) = vg;
                }
            }
        }
        __syncthreads();
        STAMP(202);
        // layer 3: dW3 [unit out, unit in] += g^T h2 (the k's are the tile's rows); d h2 = g W3 -> d(pre-activation 2)
        aW3 = mma64<LD, LD>(&gs[32 * kk][32 * rb + m], &h2s[32 * kk][u], aW3);
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[2] += gs[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&gs[32 * rb + m][32 * kk], &w3s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                d2s[r][u] = h2s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        __syncthreads();
        STAMP(203);
        // layer 2
        aW2 = mma64<LD, LD>(&d2s[32 * kk][32 * rb + m], &h1s[32 * kk][u], aW2);
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[1] += d2s[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&d2s[32 * rb + m][32 * kk], &w2s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {           // (gs: its readers -- dW3, d h2 -- finished before the barrier above)
                const int r = 32 * rb + drow(v, kk);
                gs[r][u] = h1s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        __syncthreads();
        STAMP(204);
        // layer 1
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[0] += gs[16 * rg + q][uu];
        if (BIG) {
            aW1 = mma64<LD, LD>(&gs[32 * kk][32 * rb + m], &xs[32 * kk][u], aW1);
            if (Q.gx) {
                const f32x16 acc = mma64<1, LD>(&gs[32 * rb + m][32 * kk], &w1s[32 * kk][u], zero16());
                const RowTile tgx(Q.gx + (row0 + n0) * HS, n_end - n0);      // (rows beyond the set: the range check drops them)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = 32 * rb + drow(v, kk);
                    tgx.put(r, u, Q.gx_acc ? tgx.get(r, u) + acc[v] : acc[v]);
                }
            }
        } else {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const float d = gs[16 * rg + q][uu];
#pragma unroll
                for (int c = 0; c < DS_MAX; ++c) if (c < C.d_in) dw1[c] = fmaf(d, xsmall[16 * rg + q][c], dw1[c]);
            }
        }
    }
    STAMP(205);
    // ---- this workgroup's partial sums
    float *wp = Q.wpart + (long long)blockIdx.x * Q.ld_part;
    const int ld1 = C.P.ldw1, oW1 = 0, ob1 = HS * ld1, oW2 = ob1 + HS, ob2 = oW2 + HS * HS, oW3 = ob2 + HS, ob3 = oW3 + C.d_out * HS;
    const RowTile tw3(wp + oW3, C.d_out);
#pragma unroll
    for (int v = 0; v < 16; ++v) {                   // tile (rb, ub): row = unit out, column = unit in
        const int r = 32 * rb + drow(v, kk);
        wp[oW2 + r * HS + u] = aW2[v];
        tw3.put(r, u, aW3[v]);                       // (rows >= d_out: dropped)
        if (BIG) wp[oW1 + r * ld1 + u] = aW1[v];
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) red[rg][l][uu] = db[l];
    if (!BIG) {
#pragma unroll
        for (int c = 0; c < DS_MAX; ++c) redw[rg][uu][c] = dw1[c];
    }
    __syncthreads();
    if (t < 3 * HS) {
        const int l = t >> 6, c = t & 63;
        const float v = (red[0][l][c] + red[1][l][c]) + (red[2][l][c] + red[3][l][c]);
        if (l < 2 || c < C.d_out) wp[l == 0 ? ob1 + c : (l == 1 ? ob2 + c : ob3 + c)] = v;
        if (l == 0) {
            db1s[c] = v;
            if (Q.dctx_part) Q.dctx_part[((long long)b * C.S + sp) * HS + c] = v;
        }
    }
    if (!BIG && t < HS)
        for (int c = 0; c < C.d_in; ++c) wp[oW1 + t * ld1 + c] = (redw[0][t][c] + redw[1][t][c]) + (redw[2][t][c] + redw[3][t][c]);
    if (C.ctx_part) {                                // context columns of W1: (sum of d pre-activation 1) x pooled
        __syncthreads();
        for (int p = t; p < HS * HS; p += NT) {
            const int r = p >> 6, k = p & 63;
            wp[oW1 + r * ld1 + C.d_in + k] = db1s[r] * pooled[k];
        }
    }
    STAMP(206);
}

// ---- two MLPs' backward in one launch --------------------------------------------------------------------------------------------
// The mirror image of mlp2_fwd_kernel: the backward of a POOLING MLP X (the next block's invariant MLP or the pre-pooling MLP: its
// output gradient arrives through the pooled mean only; its input is the output of ...) followed by the backward of the
// EQUIVARIANT MLP Y that produced that input.  X's input gradient -- plus, where the rows also fed an equivariant MLP, that MLP's
// input gradient from the launch before (gx_prev) -- is Y's output gradient, row for row: it stays in LDS.  Y's weights, saved
// activations and input tile are fetched into registers while X computes.  Common describes Y.  One tile per workgroup.
struct Bwd2 {
    const float *x1;                 // [B * N, 64] X's input (= Y's output, saved by the forward)
    Mlp PX;                          // X's weights (no context: W1 [64, 64])
    const float *h1x, *h2x;          // X's saved activations
    const float *gpool; int gp_S;    // the gradient through X's pooled output: the two forms of BwdIO
    const float *gp_W; int gp_ldw;
    const float *gx_prev;            // [B * N, 64] or null: added to X's input gradient
    float *wpart_x;                  // X's weight-gradient partial sums (row blockIdx.x, stride ld_part)
    const float *h1y, *h2y;          // Y's saved activations
    float *gx;                       // [B * N, 64] Y's input gradient, or null (always null for a small input)
    float *dctx_part;                // [B, S, 64] Y's sum of d(pre-activation 1)
    float *wpart_y; int ld_part;
};

template <bool BIG>        // of Y: d_in == 64 or d_in <= DS_MAX
// (143 KB of LDS: one workgroup per CU, i.e. one wave per SIMD -- the whole register file is this wave's; with the default
//  budget the prefetched operands spilled to scratch)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void mlp2_bwd_kernel(Common C, Bwd2 Q)
{
    __shared__ __attribute__((aligned(16))) float w1s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w2s[HS][LD];
    __shared__ __attribute__((aligned(16))) float w3s[HS][LD];
    __shared__ __attribute__((aligned(16))) float xs[TM][LD];
    __shared__ __attribute__((aligned(16))) float h1s[TM][LD];
    __shared__ __attribute__((aligned(16))) float h2s[TM][LD];
    __shared__ __attribute__((aligned(16))) float gs[TM][LD];
    __shared__ __attribute__((aligned(16))) float d2s[TM][LD];
    __shared__ float xsmall[BIG ? 1 : TM][DS_MAX];
    __shared__ float pooled[HS], cs[HS], gp[HS], dsum[HS], red[4][3][HS], redw[BIG ? 1 : 4][HS][DS_MAX], db1s[HS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, kk = lane >> 5, rb = wave >> 1, ub = wave & 1;
    const int b = blockIdx.x / C.S, sp = blockIdx.x - b * C.S;
    const int n0 = sp * C.rows_per_wg, n_end = min(C.N, n0 + C.rows_per_wg);
    const long long row0 = (long long)b * C.N;
    const float inv_n = inv_count(C);
    const int u = 32 * ub + m, uu = t & 63, rg = t >> 6;
    // ---- Y's pooled context and the gradient of X's pooled output, per unit of this set: their loads go out FIRST and are consumed
    //      before the operand fetches below are issued (this kernel has no registers left to hold both) -- behind those fetches,
    //      in issue order, they cost four round trips at the head of the kernel
    {
        CtxLoad Lc;
        GpLoad Lg;
        context_issue(C, b, t, Lc);
        if (Q.gp_W) gp_issue(Q.gpool, Q.gp_S, Q.gp_W, Q.gp_ldw, b, t, Lg);
        context_publish(C, pooled, t, Lc);
        if (Q.gp_W && t < HS) dsum[t] = Lg.sum;
        __syncthreads();
        context_product(C, pooled, cs, t, Lc);
        if (Q.gp_W) gp_product(dsum, gp, inv_n, t, Lg);
        else if (t < HS) gp[t] = Q.gpool[(long long)b * HS + t] * inv_n;
    }
    // ---- every global load up front, in the order of need.  X: h2 and W3 (layer 3 runs first), then h1 and W2, then the input tile
    //      and W1; they are stored to LDS just before the phase that reads them (behind the barriers that are there anyway), so
    //      one load latency is exposed instead of one per operand.  Y's operands follow and stay in registers until X is done.
    f32x4 xh2[TM * HS / 4 / NT], xw3[HS * HS / 4 / NT], xh1[TM * HS / 4 / NT], xw2[HS * HS / 4 / NT], xx[TM * HS / 4 / NT], xw1[HS * HS / 4 / NT];
    auto fetch_tile = [&](f32x4 (&v)[TM * HS / 4 / NT], const float *src) {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            v[k] = *reinterpret_cast<const f32x4 *>(src + (row0 + min(n0 + r, n_end - 1)) * HS + 4 * c4);     // (clamped; zeroed at the store)
        }
    };
    auto store_tile = [&](float (*dst)[LD], const f32x4 (&v)[TM * HS / 4 / NT]) {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) {
            const int p = t + NT * k, r = p >> 4, c4 = p & 15;
            *reinterpret_cast<f32x4 *>(&dst[r][4 * c4]) = n0 + r < n_end ? v[k] : z;
        }
    };
    fetch_tile(xh2, Q.h2x);
    fetch64(xw3, Q.PX.W3, HS, t);
    fetch_tile(xh1, Q.h1x);
    fetch64(xw2, Q.PX.W2, HS, t);
    fetch_tile(xx, Q.x1);
    fetch64(xw1, Q.PX.W1, HS, t);
    // ---- Y's operands -> registers (stored to LDS when X's readers are done)
    f32x4 q1[HS * HS / 4 / NT], q2[HS * HS / 4 / NT], q3[HS * HS / 4 / NT], th1[TM * HS / 4 / NT], th2[TM * HS / 4 / NT], tx[TM * HS / 4 / NT];
    fetch64(q2, C.P.W2, HS, t);
    fetch64(q3, C.P.W3, HS, t);
    fetch64(q1, BIG ? C.P.W1 : C.P.W2, BIG ? C.P.ldw1 : HS, t);          // (unconditional: a guarded fetch is a branch around every load)
    fetch_tile(th1, Q.h1y);
    fetch_tile(th2, Q.h2y);
    if (BIG) fetch_tile(tx, C.x);
    else {
#pragma unroll
        for (int k = 0; k < TM * HS / 4 / NT; ++k) tx[k] = th1[k];
    }
    if (!BIG)
        for (int p = t; p < TM * C.d_in; p += NT) {
            const int r = p / C.d_in, c = p - r * C.d_in, n = n0 + r;
            xsmall[r][c] = n < n_end ? C.x[(row0 + n) * C.d_in + c] : 0.0f;
        }
    // (Y's pooled context and the gradient of X's pooled output were computed at the head of the kernel: see there)
    store_tile(h2s, xh2);
    store64(w3s, xw3, t);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TM * HS / 4 / NT; ++k) {     // the pooled output's gradient, spread over the set's real trials (rows beyond: zero)
        const int p = t + NT * k, r = p >> 4, c4 = p & 15, n = n0 + r;
        const float mk = n < n_end ? mask_of(C, n) : 0.0f;
        *reinterpret_cast<float4 *>(&gs[r][4 * c4]) = make_float4(mk * gp[4 * c4], mk * gp[4 * c4 + 1], mk * gp[4 * c4 + 2], mk * gp[4 * c4 + 3]);
    }
    __syncthreads();
    // ================================================================================ X
    {
        f32x16 aW1, aW2, aW3;
        float db[3] = {0.0f, 0.0f, 0.0f};
        aW3 = mma64<LD, LD>(&gs[32 * kk][32 * rb + m], &h2s[32 * kk][u], zero16());
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[2] += gs[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&gs[32 * rb + m][32 * kk], &w3s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                d2s[r][u] = h2s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        store_tile(h1s, xh1);                        // (layer 2's operands: first read behind the barrier)
        store64(w2s, xw2, t);
        __syncthreads();
        aW2 = mma64<LD, LD>(&d2s[32 * kk][32 * rb + m], &h1s[32 * kk][u], zero16());
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[1] += d2s[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&d2s[32 * rb + m][32 * kk], &w2s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                gs[r][u] = h1s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        store_tile(xs, xx);                          // (layer 1's)
        store64(w1s, xw1, t);
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[0] += gs[16 * rg + q][uu];
        aW1 = mma64<LD, LD>(&gs[32 * kk][32 * rb + m], &xs[32 * kk][u], zero16());
        {   // X's input gradient (+ the other consumer's) = Y's output gradient: into d2s (its readers finished before the barrier above)
            const f32x16 acc = mma64<1, LD>(&gs[32 * rb + m][32 * kk], &w1s[32 * kk][u], zero16());
            const RowTile tprev(Q.gx_prev ? Q.gx_prev + (row0 + n0) * HS : Q.h1y, Q.gx_prev ? n_end - n0 : 0);   // (no such gradient: a buffer of no rows reads zeros)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk), n = n0 + r;
                const float g = tprev.get(r, u) + acc[v];
                d2s[r][u] = n < n_end ? g : 0.0f;
            }
        }
        float *wp = Q.wpart_x + (long long)blockIdx.x * Q.ld_part;
        const int oW1 = 0, ob1 = HS * HS, oW2 = ob1 + HS, ob2 = oW2 + HS * HS, oW3 = ob2 + HS, ob3 = oW3 + HS * HS;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            wp[oW1 + r * HS + u] = aW1[v];
            wp[oW2 + r * HS + u] = aW2[v];
            wp[oW3 + r * HS + u] = aW3[v];
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) red[rg][l][uu] = db[l];
        __syncthreads();                             // (also: every reader of X's tiles and weights is done, d2s is complete)
        if (t < 3 * HS) {
            const int l = t >> 6, c = t & 63;
            wp[l == 0 ? ob1 + c : (l == 1 ? ob2 + c : ob3 + c)] = (red[0][l][c] + red[1][l][c]) + (red[2][l][c] + red[3][l][c]);
        }
    }
    // ================================================================================ Y
    store64(w2s, q2, t);
    store64(w3s, q3, t);
    if (BIG) store64(w1s, q1, t);
    store_tile(h1s, th1);
    store_tile(h2s, th2);
    if (BIG) store_tile(xs, tx);
    __syncthreads();                                 // (and the readers of red[] above are done before Y's epilogue writes it)
    {
        f32x16 aW1 = zero16(), aW2, aW3;
        float db[3] = {0.0f, 0.0f, 0.0f};
        float dw1[DS_MAX] = {0.0f, 0.0f, 0.0f, 0.0f};
        // the output gradient is in d2s; gs takes d(pre-activation 2), then d2s d(pre-activation 1): the single kernel's roles, swapped
        aW3 = mma64<LD, LD>(&d2s[32 * kk][32 * rb + m], &h2s[32 * kk][u], zero16());
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[2] += d2s[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&d2s[32 * rb + m][32 * kk], &w3s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                gs[r][u] = h2s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        __syncthreads();
        aW2 = mma64<LD, LD>(&gs[32 * kk][32 * rb + m], &h1s[32 * kk][u], zero16());
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[1] += gs[16 * rg + q][uu];
        {
            const f32x16 acc = mma64<1, LD>(&gs[32 * rb + m][32 * kk], &w2s[32 * kk][u], zero16());
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * rb + drow(v, kk);
                d2s[r][u] = h1s[r][u] > 0.0f ? acc[v] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 16; ++q) db[0] += d2s[16 * rg + q][uu];
        if (BIG) {
            aW1 = mma64<LD, LD>(&d2s[32 * kk][32 * rb + m], &xs[32 * kk][u], zero16());
            if (Q.gx) {
                const f32x16 acc = mma64<1, LD>(&d2s[32 * rb + m][32 * kk], &w1s[32 * kk][u], zero16());
                const RowTile tgx(Q.gx + (row0 + n0) * HS, n_end - n0);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    tgx.put(32 * rb + drow(v, kk), u, acc[v]);
                }
            }
        } else {
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const float d = d2s[16 * rg + q][uu];
#pragma unroll
                for (int c = 0; c < DS_MAX; ++c) if (c < C.d_in) dw1[c] = fmaf(d, xsmall[16 * rg + q][c], dw1[c]);
            }
        }
        float *wp = Q.wpart_y + (long long)blockIdx.x * Q.ld_part;
        const int ld1 = C.P.ldw1, oW1 = 0, ob1 = HS * ld1, oW2 = ob1 + HS, ob2 = oW2 + HS * HS, oW3 = ob2 + HS, ob3 = oW3 + HS * HS;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * rb + drow(v, kk);
            wp[oW2 + r * HS + u] = aW2[v];
            wp[oW3 + r * HS + u] = aW3[v];
            if (BIG) wp[oW1 + r * ld1 + u] = aW1[v];
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) red[rg][l][uu] = db[l];
        if (!BIG) {
#pragma unroll
            for (int c = 0; c < DS_MAX; ++c) redw[rg][uu][c] = dw1[c];
        }
        __syncthreads();
        if (t < 3 * HS) {
            const int l = t >> 6, c = t & 63;
            const float v = (red[0][l][c] + red[1][l][c]) + (red[2][l][c] + red[3][l][c]);
            wp[l == 0 ? ob1 + c : (l == 1 ? ob2 + c : ob3 + c)] = v;
            if (l == 0) {
                db1s[c] = v;
                Q.dctx_part[((long long)b * C.S + sp) * HS + c] = v;
            }
        }
        if (!BIG && t < HS)
            for (int c = 0; c < C.d_in; ++c) wp[oW1 + t * ld1 + c] = (redw[0][t][c] + redw[1][t][c]) + (redw[2][t][c] + redw[3][t][c]);
        __syncthreads();                             // context columns of W1: (sum of d pre-activation 1) x pooled
        for (int p = t; p < HS * HS; p += NT) {
            const int r = p >> 6, k = p & 63;
            wp[oW1 + r * ld1 + C.d_in + k] = db1s[r] * pooled[k];
        }
    }
}

// out[p] = sum over g of part[g][p], g in fixed order; entries p >= P_main were written by the first G_tail rows only.
// A workgroup takes 64 columns; wave q sums the rows g = q (mod 4) of its columns (eight loads in flight per lane) -- the four sums
// a_q the one-thread-per-column form kept in four accumulators -- and wave 0 adds them as that form did: (a0 + a1) + (a2 + a3), the
// rows past the last whole group of four going to a0.  Same bits, four times the workgroups and a quarter of the dependent chain.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *part, int G, int P, int P_main, int G_tail, float *out)
{
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, p = blockIdx.x * 64 + c, pc = min(p, P - 1);
    if (pc >= P_main) G = G_tail;
    const int G4 = G & ~3;
    const float *src = part + pc;
    float a = 0.0f;
    int g = q;
    for (; g + 28 < G4; g += 32) {                       // eight rows of this wave's residue class: the loads first, then the adds in order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(long long)(g + 4 * k) * P];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += v[k];
    }
    for (; g < G4; g += 4) a += src[(long long)g * P];
    if (q == 0) for (g = G4; g < G; ++g) a += src[(long long)g * P];
    red[q][c] = a;
    __syncthreads();
    if (q == 0 && p < P) out[p] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

}  // namespace nddm_deepset

using namespace nddm_deepset;

extern "C" {

int nddm_deepset_supported(int hidden, int d_in) { return (hidden == HS && (d_in == HS || (d_in >= 1 && d_in <= DS_MAX))) ? 1 : 0; }

static bool common_ok(const Common &C)
{
    return nddm_deepset_supported(HS, C.d_in) && C.B > 0 && C.N > 0 && C.S > 0 && C.rows_per_wg > 0 && (long long)C.S * C.rows_per_wg >= C.N
           && C.P.ldw1 == C.d_in + (C.ctx_part ? HS : 0) && (C.d_in != HS || C.P.ldw1 % 4 == 0) && C.d_out >= 1 && C.d_out <= HS
           && (C.x_part ? (C.d_in == HS && C.S_x >= 1) : C.x != nullptr);
}

/* One per-trial MLP forward.  x [B * N, d_in]; W1 [64, d_in (+ 64 with a context)], W2, W3 [64, 64]; mask [N] or NULL; inv_n: device
 * scalar or NULL (then inv_n_host); ctx_part [B, S_ctx, 64] or NULL.  Writes h1, h2 [B * N, 64], and y [B * N, 64] and / or
 * pool_part [B, S, 64] where not NULL.  S workgroups of up to rows_per_wg trials per set.  d_out: rows of W3 / b3 and width of y
 * (64 but for the last MLP of the network); x_part / S_x: see Common (NULL / 0: x is read).  ldy: row stride of y (0: d_out);
 * extra / n_extra / extra_stride: columns [d_out, d_out + n_extra) of y are copied from extra[row * extra_stride + .] (NULL: none). */
int nddm_deepset_mlp_fwd(const float *x, int d_in, int B, int N, int S, int rows_per_wg, const float *mask, int mask_is_count, const float *inv_n,
                         float inv_n_host, const float *ctx_part, int S_ctx, const float *W1, int ldw1, const float *b1, const float *W2,
                         const float *b2, const float *W3, const float *b3, int d_out, const float *x_part, int S_x, float *h1, float *h2,
                         float *y, float *pool_part, int ldy, const float *extra, int n_extra, int extra_stride, void *stream)
{
    const Common C = {x, d_in, B, N, S, rows_per_wg, mask, mask && mask_is_count, inv_n, inv_n_host, ctx_part, S_ctx, {W1, ldw1, b1, W2, b2, W3, b3}, d_out, x_part, S_x};
    if (ldy <= 0) ldy = d_out;
    if (!common_ok(C) || !h1 || !h2 || (pool_part && d_out != HS) || n_extra < 0 || (extra && (!y || d_out + n_extra > HS)) || ldy < d_out + (extra ? n_extra : 0))
        return 1;
    const FwdOut O = {h1, h2, y, pool_part, ldy, extra, extra ? n_extra : 0, extra_stride};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d_in == HS) hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(B * S), dim3(NT), 0, st, C, O);
    else hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(B * S), dim3(NT), 0, st, C, O);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* The backward of nddm_deepset_mlp_fwd (same first arguments).  gy [B * N, 64] or NULL; gpool / gp_S / gp_W / gp_ldw: see BwdIO;
 * (ldgy: its row stride, 0: d_out); gx [B * N, d_in] or NULL (d_in == 64 only), gx_acc: accumulate; dctx_part [B, S, 64] or NULL; wpart: B * S rows, ld_part floats
 * apart, of 64 ldw1 + 64 + 2 (4096 + 64) weight-gradient partial sums each: reduce with nddm_deepset_reduce. */
int nddm_deepset_mlp_bwd(const float *x, int d_in, int B, int N, int S, int rows_per_wg, const float *mask, int mask_is_count, const float *inv_n,
                         float inv_n_host, const float *ctx_part, int S_ctx, const float *W1, int ldw1, const float *b1, const float *W2,
                         const float *b2, const float *W3, const float *b3, int d_out, const float *x_part, int S_x, const float *h1,
                         const float *h2, const float *gy, int ldgy, const float *gpool, int gp_S, const float *gp_W, int gp_ldw, float *gx,
                         int gx_acc, float *dctx_part, float *wpart, int ld_part, void *stream)
{
    if (ldgy <= 0) ldgy = d_out;
    const Common C = {x, d_in, B, N, S, rows_per_wg, mask, mask && mask_is_count, inv_n, inv_n_host, ctx_part, S_ctx, {W1, ldw1, b1, W2, b2, W3, b3}, d_out, x_part, S_x};
    if (!common_ok(C) || !h1 || !h2 || !wpart || ld_part < HS * ldw1 + HS + HS * HS + HS + d_out * HS + d_out || (gx && d_in != HS)
        || (!gy && !gpool) || (gpool && d_out != HS) || ldgy < d_out)
        return 1;
    const BwdIO Q = {h1, h2, gy, ldgy, gpool, gp_S, gp_W, gp_ldw, gx, gx_acc, dctx_part, wpart, ld_part};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d_in == HS) hipLaunchKernelGGL(mlp_bwd_kernel<true>, dim3(B * S), dim3(NT), 0, st, C, Q);
    else hipLaunchKernelGGL(mlp_bwd_kernel<false>, dim3(B * S), dim3(NT), 0, st, C, Q);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* nddm_deepset_mlp_fwd for an equivariant MLP A (same first arguments; d_out 64, y required) FOLLOWED IN THE SAME LAUNCH by the
 * MLP B that consumes A's output row by row (no context, input width 64; W1b, W2b, W3b [64, 64]): writes B's h1b, h2b [B * N, 64] and
 * the masked sums of B's output pool_part_b [B, S, 64] -- what two calls (A with y, then B on y with pool_part) write, bit for bit.
 * rows_per_wg must be 64 (one tile per workgroup). */
int nddm_deepset_mlp2_fwd(const float *x, int d_in, int B, int N, int S, int rows_per_wg, const float *mask, int mask_is_count, const float *inv_n,
                          float inv_n_host, const float *ctx_part, int S_ctx, const float *W1, int ldw1, const float *b1, const float *W2,
                          const float *b2, const float *W3, const float *b3, int d_out, const float *x_part, int S_x, float *h1, float *h2,
                          float *y, const float *W1b, const float *b1b, const float *W2b, const float *b2b, const float *W3b, const float *b3b,
                          float *h1b, float *h2b, float *pool_part_b, void *stream)
{
    const Common C = {x, d_in, B, N, S, rows_per_wg, mask, mask && mask_is_count, inv_n, inv_n_host, ctx_part, S_ctx, {W1, ldw1, b1, W2, b2, W3, b3}, d_out, x_part, S_x};
    if (!common_ok(C) || rows_per_wg != TM || d_out != HS || x_part || !h1 || !h2 || !y || !W1b || !b1b || !W2b || !b2b || !W3b || !b3b || !h1b || !h2b
        || !pool_part_b)
        return 1;
    const FwdOut O = {h1, h2, y, nullptr, HS, nullptr, 0, 0}, O2 = {h1b, h2b, nullptr, pool_part_b, HS, nullptr, 0, 0};
    const Mlp P2 = {W1b, HS, b1b, W2b, b2b, W3b, b3b};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d_in == HS) hipLaunchKernelGGL(mlp2_fwd_kernel<true>, dim3(B * S), dim3(NT), 0, st, C, O, P2, O2);
    else hipLaunchKernelGGL(mlp2_fwd_kernel<false>, dim3(B * S), dim3(NT), 0, st, C, O, P2, O2);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* The backward of nddm_deepset_mlp2_fwd's pair, in the backward's order: first the pooling MLP X (input x1 [B * N, 64] = Y's output;
 * weights W1x .. b3x, all [64, 64]; saved h1x, h2x; output gradient through its pooled mean: gpool / gp_S / gp_W / gp_ldw as in
 * nddm_deepset_mlp_bwd; gx_prev [B * N, 64] or NULL is added to its input gradient), then the equivariant MLP Y (the first
 * arguments, as nddm_deepset_mlp_bwd: its input x, context ctx_part, weights; saved h1y, h2y) whose output gradient IS that input
 * gradient (it never leaves the chip).  Writes Y's gx [B * N, 64] (or NULL; must be NULL for a small input), dctx_part [B, S, 64] and
 * both MLPs' weight-gradient partial sums (rows of ld_part floats, one per workgroup) -- what the two separate calls write, bit for
 * bit.  rows_per_wg must be 64. */
int nddm_deepset_mlp2_bwd(const float *x, int d_in, int B, int N, int S, int rows_per_wg, const float *mask, int mask_is_count, const float *inv_n,
                          float inv_n_host, const float *ctx_part, int S_ctx, const float *W1, int ldw1, const float *b1, const float *W2,
                          const float *b2, const float *W3, const float *b3, int d_out, const float *x_part, int S_x, const float *h1y,
                          const float *h2y, float *gx, float *dctx_part, float *wpart_y, const float *x1, const float *W1x, const float *b1x,
                          const float *W2x, const float *b2x, const float *W3x, const float *b3x, const float *h1x, const float *h2x,
                          const float *gpool, int gp_S, const float *gp_W, int gp_ldw, const float *gx_prev, float *wpart_x, int ld_part,
                          void *stream)
{
    const Common C = {x, d_in, B, N, S, rows_per_wg, mask, mask && mask_is_count, inv_n, inv_n_host, ctx_part, S_ctx, {W1, ldw1, b1, W2, b2, W3, b3}, d_out, x_part, S_x};
    if (!common_ok(C) || rows_per_wg != TM || d_out != HS || x_part || !ctx_part || !h1y || !h2y || !dctx_part || !wpart_y || !x1 || !W1x || !W2x || !W3x
        || !h1x || !h2x || !gpool || !wpart_x || (gx && d_in != HS) || ld_part < HS * ldw1 + HS + 2 * (HS * HS + HS))
        return 1;
    const Bwd2 Q = {x1, {W1x, HS, b1x, W2x, b2x, W3x, b3x}, h1x, h2x, gpool, gp_S, gp_W, gp_ldw, gx_prev, wpart_x, h1y, h2y, gx, dctx_part, wpart_y, ld_part};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d_in == HS) hipLaunchKernelGGL(mlp2_bwd_kernel<true>, dim3(B * S), dim3(NT), 0, st, C, Q);
    else hipLaunchKernelGGL(mlp2_bwd_kernel<false>, dim3(B * S), dim3(NT), 0, st, C, Q);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int nddm_deepset_reduce(const float *part, int G, int P, int P_main, int G_tail, float *out, void *stream)
{
    if (G <= 0 || P <= 0 || P_main < 0 || P_main > P || G_tail < 0 || G_tail > G) return 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((P + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), part, G, P, P_main,
                       G_tail, out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"

#ifdef NDDM_TRAIN_STAMPS
extern "C" int nddm_deepset_read_stamps(unsigned long long *out, int cap)
{
    int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(nddm_deepset::g_nstamps), sizeof(int));
    if (n > cap) n = cap;
    if (n > 4096) n = 4096;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(nddm_deepset::g_stamps), sizeof(unsigned long long) * 2 * n);
    int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(nddm_deepset::g_nstamps), &zero, sizeof(int));
    return n;
}
#endif

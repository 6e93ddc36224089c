// train_kernels.hip -- the amortizer's conditional normalising flow (bayesflow_nddms_amd/amortizer.py::InvertibleNetwork; the
// stand-in for bf.networks.InvertibleNetwork, basic_ddm_dc.py:163-165) as ONE kernel forward and TWO backward, for the batch
// sizes of the online-training loop (32 sets per rank: basic_ddm_dc.py:199-202): per layer an ActNorm, a fixed permutation and
// two conditional affine-coupling half-layers, and (optionally) the maximum-likelihood loss on top.
//
// At batch 32 the flow was ~400 of the ~650 kernels of a graph-replayed training iteration, each a 3-5 microsecond launch that
// touches a few kilobytes: concatenate, three GEMMs of 32 rows, two ELUs, the soft clamp, exp, multiply-add, and twice that
// backward -- 33 launches per half-layer, twelve half-layers.  Here a half-layer is a loop body, the kernels walk the layers
// (every row is independent but for the weight gradients), and everything between the layers (slices, concatenations,
// ActNorm, permutation) is index arithmetic:
//     in = [x_h | cond]  ->  h1 = elu(W1 in + b1)  ->  h2 = elu(W2 h1 + b2)  ->  (o_s | o_t) = W3 h2 + b3
//     s = clamp * tanh(o_s / clamp),   y = x_tr * exp(s) + o_t                      (returns y and s; log|det| = sum of s)
// What is bought is launches and latency, not arithmetic: the kernels are chains of small dependent phases on one or a few
// CUs, so weights are staged in LDS with coalesced loads once per half-layer and the 32 x 128 x 128 products are f32 MFMAs
// (v_mfma_f32_16x16x4_f32 / 32x32x2: exact f32) on LDS operands -- as register-blocked FMAs their LDS reads were the time.
// Hidden width 128 (the networks' default), at most 32 inputs and 8 transformed columns; everything else takes the PyTorch path.
// gfx950 only.  tests/test_gpu_training.py compares with the PyTorch composition.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nddm_train {

constexpr int H = 128;        // hidden width
constexpr int TR = 32;        // rows per tile
constexpr int DI_MAX = 32;    // inputs of the sub-network (x_h columns + condition columns)
constexpr int M_MAX = 16;     // outputs (2 * transformed columns)

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

__device__ __forceinline__ float elu(float v) { return v > 0.0f ? v : expm1f(v); }
// ELU' from the saved PRE-activation: exp(a) itself.  (Until round 5 the forward saved the OUTPUT h and the backward took h + 1: an
// absolute 2^-24 on a factor that can be far below 1 -- PyTorch keeps the input -- which left the activation-gradient chain at ~2 x
// PyTorch's error against float64, tools/fused_accuracy.py.  The saved tensors h_all now hold a = W in + b; the weight-gradient kernel
// re-derives h = elu(a) when it stages its tiles.)
__device__ __forceinline__ float elu_grad_from_pre(float a) { return a > 0.0f ? 1.0f : expf(a); }   // alpha = 1

// ------------------------------------------------------------------------------------------------ forward
// grid = ceil(R / 16) workgroups of 512 threads over tiles of 16 rows (two workgroups at batch 32: every product then fills the
// 16 rows of an MFMA tile, and a SIMD holds two waves -- one wave's LDS reads and epilogue run under the other's MFMAs; four
// workgroups of 8 rows x 256 threads were 11 % slower, one of 32 x 1024 more), each taking its rows through every layer.  What a
// row carries from half-layer to half-layer (the permuted ActNorm output z, the layer's output, the condition) stays in LDS;
// global memory only receives what the backward needs.  All three layers are v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered
// fmaf chain) on LDS operands: layer 2, the 128 x 128 product, with 16-byte reads (one 16 x 16 tile per wave); layer 1 (K = the
// <= 32 inputs, zero-padded) and layer 3 (K = the 128 units split over the waves, partial tiles summed by the affine phase) with
// dword reads.  Which k a lane group takes in which MFMA is free (the instruction sums over the four groups) and is chosen per
// operand so that the lanes an LDS instruction serves together fall on different banks.  The NEXT half-layer's weights are
// fetched into registers at the top of a half-layer and written to LDS once their buffers' last readers are done: a half-layer
// is a chain of short phases, and a cold load at the head of each was 3.3 of its 8.4 microseconds (now 5.2 in all).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDR = H + 4;      // row stride of the [row][unit] tiles and of W2 in LDS: 16-byte aligned rows whose operand reads
                                // (16 rows x 4 column groups per wave instruction) spread evenly over the banks
// A barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store the thread has in flight
// -- which is exactly what the forward's weight prefetch must not do: its loads are issued a half-layer ahead and nothing in
// this kernel reads global memory that the kernel wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The rows [r0, r0 + rows) of a row-major global array as a RAW BUFFER of just their size: a store to a row beyond them (the rows of
// a tile past R) is dropped by the hardware's range check -- the `if (r0 + r < R)` around every one of a phase's stores compiled
// to a branch, an exec save / restore and a 64-bit address each.
struct RowBuf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ RowBuf(float *first_row, int rows, int ld)
        : r(__builtin_amdgcn_make_buffer_rsrc(first_row, 0, (rows > 0 ? rows : 0) * ld * 4, 0x00020000)) {}
    __device__ __forceinline__ void put(int off_floats, float v) const
    { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off_floats * 4, 0, 0); }
};

constexpr int L_MAX = 8, D_MAX = 8;
struct HalfP { const float *W1, *b1, *W2, *b2, *W3, *b3; };
struct LayerP { const float *scale, *bias; HalfP a, b; };
struct FlowP { LayerP layer[L_MAX]; unsigned char perm[L_MAX][D_MAX]; };
struct FlowDims { int L, R, D, d1, C; float clamp; };
struct FlowSaved { float *z_all, *out_all, *s_all, *h_all; };    // [L, R, D] permuted ActNorm outputs, layer outputs, clamped
                                                                  // log-scales; [L, 4, R, H] activations of the sub-networks
template <int TF, int NT>        // rows per workgroup, threads: 8 x 256 (four CUs at batch 32) or 16 x 512 (two CUs, full 16-row MFMA tiles)
__global__ __launch_bounds__(NT) void flow_fwd_kernel(FlowDims Q, FlowP P, const float *theta, const float *cond, FlowSaved S, float *ld)
{
    constexpr int UW = H / (NT / 64);          // units per wave in layer 2: 32 (two 16 x 16 tiles) or 16 (one)
    static_assert(TF % 4 == 0 && TF <= 16 && (UW == 32 || UW == 16) && M_MAX == 16 && TF * D_MAX <= NT && DI_MAX == 32, "tile shape");
    __shared__ float in_s[16 * (DI_MAX + 2)];   // the sub-network's input rows, an MFMA operand read a dword at a time down 16 rows: row stride 33 or 34 by
                                                // the parity of DI (see layer 1); columns >= DI zero; rows TF.. stay zero
    __shared__ __attribute__((aligned(16))) float h1r[16][LDR];        // rows TF..15 stay zero (the MFMA tile has 16 rows)
    __shared__ __attribute__((aligned(16))) float h2r[TF][LDR];
    __shared__ float o_s[TF][M_MAX];
    __shared__ float w3s[16][H + 2];          // W3 [output][unit], rows >= M zero: an MFMA operand read a dword at a time down its 16 rows (+ 2: banks 2 n + k)
    __shared__ float part3[NT / 64][16][16];  // layer 3's partial tiles, one per wave
    __shared__ __attribute__((aligned(16))) float w2s[H][LDR];
    __shared__ float xs[2][TF][D_MAX];       // the layer's input / output rows (alternating), zs: its permuted ActNorm output
    __shared__ float zs[TF][D_MAX], cs[TF][DI_MAX];
    __shared__ float w1f[H * DI_MAX + DI_MAX];  // W1 as it lies in memory, [unit][DI]: the other MFMA operand of layer 1 (+ DI_MAX: the last unit's reads past its row)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int r0 = blockIdx.x * TF, D = Q.D, d1 = Q.d1, d2 = D - d1;
    const long long RD = (long long)Q.R * D, RH = (long long)Q.R * H;
    float ld_acc = 0.0f;                                     // threads t < TF: log|det| of row r0 + t, summed as the layers go
    STAMP(0);
    // one thread's share of the next half-layer's weights (plain local arrays indexed by unrolled constants: registers)
    f32x4 nw2[H * H / 4 / NT];           // (the compiler's own vector type: HIP's float4 struct kept the array in memory)
    float nw1[H * DI_MAX / NT], nw3[M_MAX * H / NT], nb10, nb11, nb20, nb21, nb3s, nb3t, nsc = 0.0f, nbi = 0.0f;
#define NDDM_FETCH_WEIGHTS(Wn, DIn, Mn) do {                                                                                  \
        const f32x4 *src_ = reinterpret_cast<const f32x4 *>((Wn).W2);                                                         \
        _Pragma("unroll") for (int k = 0; k < H * H / 4 / NT; ++k) nw2[k] = src_[t + NT * k];                                \
        /* (coalesced; every load unconditional, from a clamped address: a guarded load is a branch and a block of its own) */     \
        _Pragma("unroll") for (int k = 0; k < H * DI_MAX / NT; ++k) nw1[k] = (Wn).W1[min(t + NT * k, H * (DIn) - 1)];          \
        _Pragma("unroll") for (int k = 0; k < M_MAX * H / NT; ++k) {                                                          \
            const int p_ = t + NT * k; const float v_ = (Wn).W3[min(p_, (Mn) * H - 1)]; nw3[k] = p_ < (Mn) * H ? v_ : 0.0f; } \
        nb10 = (Wn).b1[UW * w + (lane & 15)]; nb11 = (Wn).b1[min(UW * w + 16, H - 16) + (lane & 15)];                                \
        nb20 = (Wn).b2[UW * w + (lane & 15)]; nb21 = (Wn).b2[min(UW * w + 16, H - 16) + (lane & 15)];                     \
        {   const int Dt_ = (Mn) / 2, d_ = t % Dt_; nb3s = (Wn).b3[d_]; nb3t = (Wn).b3[Dt_ + d_]; }    /* thread (row t / Dt, column t % Dt) of the affine phase */ \
    } while (0)
    NDDM_FETCH_WEIGHTS(P.layer[0].a, d1 + Q.C, 2 * d2);
    if (t < TF * D) { const int p = P.perm[0][t % D]; nsc = P.layer[0].scale[p]; nbi = P.layer[0].bias[p]; }
    for (int p = t; p < (16 - TF) * H; p += NT) h1r[TF + (p >> 7)][p & (H - 1)] = 0.0f;
    for (int p = t; p < 16 * (DI_MAX + 2); p += NT) in_s[p] = 0.0f;
    for (int p = H * DI_MAX + t; p < H * DI_MAX + DI_MAX; p += NT) w1f[p] = 0.0f;
    if (t < TF * D) { const int r = t / D, c = t - r * D; xs[0][r][c] = r0 + r < Q.R ? theta[(long long)(r0 + r) * D + c] : 0.0f; }
    if (t < TF * Q.C) { const int r = t / Q.C, c = t - r * Q.C; cs[r][c] = r0 + r < Q.R ? cond[(long long)(r0 + r) * Q.C + c] : 0.0f; }
    for (int hl = 0; hl < 2 * Q.L; ++hl) {
        const int l = hl >> 1;
        const bool second = hl & 1;
        const LayerP &Y = P.layer[l];
        const HalfP &W = second ? Y.b : Y.a;
        const int Dh = second ? d2 : d1, Dt = second ? d1 : d2, DI = Dh + Q.C, M = 2 * Dt;
        float *z_g = S.z_all + l * RD, *out_g = S.out_all + l * RD, *sl = S.s_all + l * RD + (second ? d2 : 0), *h = S.h_all + 4 * l * RH;
        float *h1_out = h + (second ? 2 : 0) * RH, *h2_out = h + (second ? 3 : 1) * RH;
        float (*xin)[D_MAX] = xs[l & 1], (*xout)[D_MAX] = xs[(l & 1) ^ 1];
        const int nvr = Q.R - r0;                            // the tile's valid rows: stores beyond them are dropped by the buffers' range check
        const RowBuf bh1(h1_out + (long long)r0 * H, nvr, H), bh2(h2_out + (long long)r0 * H, nvr, H), bz(z_g + (long long)r0 * D, nvr, D);
        // this half-layer's weights: out of the registers (fetched a half-layer ago) into LDS / this thread's W1 row
        lds_barrier();                                     // (the previous half-layer's readers of w2s, w3s, xs are done)
#pragma unroll
        for (int k = 0; k < H * H / 4 / NT; ++k) { const int p4 = t + NT * k; *reinterpret_cast<f32x4 *>(&w2s[p4 >> 5][4 * (p4 & 31)]) = nw2[k]; }
#pragma unroll
        for (int k = 0; k < M_MAX * H / NT; ++k) { const int p = t + NT * k; w3s[p >> 7][p & (H - 1)] = nw3[k]; }
#pragma unroll
        for (int k = 0; k < H * DI_MAX / NT; ++k) w1f[t + NT * k] = nw1[k];    // (entries beyond H * DI: copies of the last one, finite)
        const float b10 = nb10, b11 = nb11, b20 = nb20, b21 = nb21, b3s = nb3s, b3t = nb3t, scv = nsc, biv = nbi;
        STAMP(7);
        if (hl + 1 < 2 * Q.L) {                              // ... and the next one's into the registers
            const HalfP &Wn = second ? P.layer[l + 1].a : Y.b;
            NDDM_FETCH_WEIGHTS(Wn, (second ? d1 : d2) + Q.C, 2 * (second ? d2 : d1));
            if (second && t < TF * D) { const int p = P.perm[l + 1][t % D]; nsc = P.layer[l + 1].scale[p]; nbi = P.layer[l + 1].bias[p]; }
        }
        STAMP(8);
        if (!second) {
            if (t < TF * D) {                               // ActNorm, then the permutation: z[:, c] = u[:, perm[c]]
                const int r = t / D, c = t - r * D, p = P.perm[l][c];
                const float v = fmaf(xin[r][p], expf(scv), biv);
                zs[r][c] = v;
                bz.put(r * D + c, v);
            }
            if (t < TF) {
#pragma unroll
                for (int d = 0; d < D_MAX; ++d) if (d < D) ld_acc += Y.scale[d];
            }
            lds_barrier();
        }
        STAMP(9);
        // first:  conditioned on z[:, :d1], transforms z[:, d1:] -> out[:, d1:], log-scales -> s[:, :d2]
        // second: conditioned on out[:, d1:], transforms z[:, :d1] -> out[:, :d1], log-scales -> s[:, d2:]
        const int ild = DI_MAX + 1 + (~DI & 1);           // in_s row stride: 33 (DI odd) or 34 (DI even)
        for (int p = t; p < TF * DI_MAX; p += NT) {        // (every column: those >= DI are zeros -- they meet the next unit's weights in layer 1)
            const int r = p >> 5, c = p & (DI_MAX - 1);
            in_s[r * ild + c] = c < Dh ? (second ? xout[r][d1 + c] : zs[r][c]) : c < DI ? cs[r][c - Dh] : 0.0f;
        }
        lds_barrier();
        STAMP(1);
        {   // layer 1, [16 rows x 32 inputs] x [32 x H] as v_mfma_f32_16x16x4_f32 on LDS operands read a dword at a time; W1 stays as
            // it lies in memory (row stride DI: a linear copy) and the inputs >= DI are zeros.  Which input a lane group takes in
            // which of the eight MFMAs is free, and it decides the banks (lanes 0-31 = groups 0 and 1 are served together, rows n = 0 ..
            // 15 each): DI odd -- group kk takes 8 kb + i (kb = 0, 2, 1, 3): W1 banks DI n + 16 kk + i, inputs (stride 33) n + 16 kk + i, all
            // 32 distinct; DI even -- group kk takes 4 i + kk: W1 banks DI n + kk + 4 i, inputs (stride 34) 2 n + kk + 4 i, all distinct
            const int n = lane & 15, kk = lane >> 4, odd = DI & 1;
            const int k0 = odd ? 8 * (((kk & 1) << 1) | (kk >> 1)) : kk;
            const float *ap = &in_s[n * ild + k0], *bp0 = &w1f[(UW * w + n) * DI + k0], *bp1 = &w1f[(min(UW * w + 16, H - 16) + n) * DI + k0];
            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
            if (odd) {                                       // (two copies: the reads' offsets are then immediates)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = ap[i];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp0[i], acc0, 0, 0, 0);
                    if constexpr (UW == 32) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp1[i], acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = ap[4 * i];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp0[4 * i], acc0, 0, 0, 0);
                    if constexpr (UW == 32) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp1[4 * i], acc1, 0, 0, 0);
                }
            }
            if (kk < TF / 4) {                               // D: unit = tile base + (l & 15), row = 4 (l >> 4) + register
                const int u0 = UW * w + n, u1 = u0 + 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * kk + q;
                    const float a0 = acc0[q] + b10, v0 = elu(a0);
                    h1r[r][u0] = v0;
                    bh1.put(r * H + u0, a0);
                    if constexpr (UW == 32) {
                        const float a1 = acc1[q] + b11, v1 = elu(a1);
                        h1r[r][u1] = v1;
                        bh1.put(r * H + u1, a1);
                    }
                }
            }
        }
        lds_barrier();
        STAMP(2);
        {   // layer 2: wave w owns units [32 w, 32 w + 32) as two 16 x 16 tiles (two independent accumulators);
            // lane l: A[row l & 15][k] = h1[row][k], B[k][unit l & 15] = W2[unit][k] for the k's 32 (l >> 4) + s, s = 0 .. 31
            // (which 32 k's a lane group takes is free -- the MFMA sums over all four groups -- and it decides the LDS banks: ds_read_b128
            //  serves lanes {0-3, 12-15, 20-27} together, i.e. rows n of group 0 with rows n' of group 1, on 64 banks = 16 slots of 16
            //  bytes; with a row stride of 33 slots, groups 0 and 1 must start a multiple of 16 slots = 64 floats apart, not 32, or
            //  every read is a 2-way conflict: groups 0, 1, 2, 3 take the k blocks 0, 2, 1, 3)
            const int n = lane & 15, kk = lane >> 4, kb = ((kk & 1) << 1) | (kk >> 1);
            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
            const float4 *ap = reinterpret_cast<const float4 *>(&h1r[n][32 * kb]);
            const float4 *bp0 = reinterpret_cast<const float4 *>(&w2s[UW * w + n][32 * kb]);
            const float4 *bp1 = reinterpret_cast<const float4 *>(&w2s[min(UW * w + 16, H - 16) + n][32 * kb]);
            if constexpr (UW == 32) {
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 a = ap[q4], b0 = bp0[q4], b1 = bp1[q4];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc1, 0, 0, 0);
                }
            } else {                                         // one tile per wave: the even and the odd groups of four k's in two accumulators
#pragma unroll
                for (int q4 = 0; q4 < 8; q4 += 2) {
                    const float4 a0 = ap[q4], b0 = bp0[q4], a1 = ap[q4 + 1], b1 = bp0[q4 + 1];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
                }
                acc0 += acc1;
            }
            // D: unit = tile base + (l & 15), row = 4 (l >> 4) + register: rows < TF live in lanes 0..31
            if (kk < TF / 4) {
                const int u0 = UW * w + n, u1 = u0 + 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * kk + q;
                    const float a0 = acc0[q] + b20, v0 = elu(a0);
                    h2r[r][u0] = v0;
                    bh2.put(r * H + u0, a0);
                    if constexpr (UW == 32) {
                        const float a1 = acc1[q] + b21, v1 = elu(a1);
                        h2r[r][u1] = v1;
                        bh2.put(r * H + u1, a1);
                    }
                }
            }
        }
        lds_barrier();
        STAMP(3);
        {   // layer 3, [16 rows x H] x [H x 16 outputs] as MFMAs with the H units split over the waves (16 each: four MFMAs, lane group
            // kk takes the units 16 w + 4 i + kk); the waves' partial tiles are summed, in fixed order, by the affine phase's threads
            constexpr int KW = H / (NT / 64);
            const int n = lane & 15, kk = lane >> 4;
            const float *ap = &h2r[n][KW * w + kk], *bp = &w3s[n][KW * w + kk];
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < KW / 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * i], bp[4 * i], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) part3[w][4 * kk + q][n] = acc[q];      // D: output = l & 15, row = 4 (l >> 4) + register
        }
        lds_barrier();
        STAMP(4);
        if (t < TF * Dt) {                                  // the outputs' sums, soft clamp + affine transform
            const int r = t / Dt, d = t - r * Dt, row = r0 + r;
            float os = b3s, ot = b3t;
#pragma unroll
            for (int v = 0; v < NT / 64; ++v) { os += part3[v][r][d]; ot += part3[v][r][Dt + d]; }
            const float sv = Q.clamp * tanhf(os / Q.clamp);
            const float yv = fmaf(second ? zs[r][d] : zs[r][d1 + d], expf(sv), ot);
            xout[r][second ? d : d1 + d] = yv;
            o_s[r][d] = sv;                                  // (for the rows' log|det| below)
            if (row < Q.R) {
                sl[(long long)row * D + d] = sv;
                out_g[(long long)row * D + (second ? d : d1 + d)] = yv;
            }
        }
        lds_barrier();
        if (t < TF) {
#pragma unroll
            for (int d = 0; d < M_MAX / 2; ++d) if (d < Dt) ld_acc += o_s[t][d];
        }
        STAMP(5);
    }
    if (t < TF && r0 + t < Q.R) ld[r0 + t] = ld_acc;        // log|det| of the row: every log-scale, and the ActNorms'
    STAMP(6);
}

// ------------------------------------------------------------------------------------------------ backward
// Two kernels.  (1) The chain of ACTIVATION gradients, half-layers last to first -- the critical path: every row is independent,
// so workgroup w takes rows [16 w, 16 w + 16) through all of it (512 threads, two workgroups at batch 32: as one workgroup of
// 32 rows x 1024 threads the 128 x 128 product ran four waves per SIMD deep on ONE CU's matrix pipe and the kernel took 72.6 us
// against 57; every product -- d(h2) = d(out) W3, d(h1) = d(a2) W2, d(in) = d(a1) W1 -- is v_mfma_f32_16x16x4_f32 on LDS
// operands) and leaves d(pre-activation 1, 2) and d(output) of every half-layer in global memory.  (2) The WEIGHT gradients of all
// half-layers at once, one workgroup per half-layer (dW2 = d(a2)^T h1 as v_mfma_f32_32x32x2_f32, a 32 x 32 tile per wave): they
// are off the critical path, and as one kernel with (1) they were 60 % of a half-layer's 19 microseconds.
// Every weight is staged in LDS once per half-layer (64 + 4 + 17 + 8 KB of gfx950's 160 KB): read from global memory where they
// are used, their latency was the kernel's time (columns of W2: 10 us, W1 in the input gradient: 13 us of the original 33).
constexpr int NTB = 1024;
constexpr int LDT = H + 4;       // row stride of W2 TRANSPOSED in LDS, w2t[in unit][out unit]: d(h1) = d(a2) W2 sums over the OUT units, and
                                 // with the out units along a row both MFMA operands are read 16 bytes at a time (as in the forward:
                                 // the order of the sum is free, lane group k takes the k's [32 k, 32 k + 32)) instead of a dword at a
                                 // time down a column -- 16 LDS reads per lane instead of 64 in the longest phase of a half-layer

struct HalfD {                   // one half-layer of kernel (1), resolved by the host
    const float *xtr, *W1, *W2, *W3, *s, *h1, *h2;       // xtr, s: [R, .] slices with row stride D (saved by the forward)
    float *da2, *da1, *dos;      // da2, da1 [R, H], dos [R, M_MAX]: for kernel (2)
    int Dh, Dt;
};
struct NormD { const float *scale; float *gz; unsigned char perm[D_MAX]; };      // gz [R, D]: d(permuted ActNorm output), for kernel (2)
struct FlowD { HalfD half[2 * L_MAX]; NormD norm[L_MAX]; };
struct FlowBwdBuf {
    const float *g_z, *g_ld;       // [R, D], [R]: gradients of the flow's output and log|det| (g_nll null)
    float *gx;                     // [R, D]: the gradient of theta
    float *gcond;                  // [R, C]
    const float *g_nll;            // not null: the gradient of the mean negative log-likelihood (a scalar): the gradients of z and
    float *w_gld;                  // log|det| are then z g / R and -g / R, from the forward's z (w_gld [R] is written for kernel (2))
    const float *z_last;
};

// What a row carries from half-layer to half-layer (the gradients of the layer's output, of its permuted ActNorm output, of the
// condition) stays in LDS, the next half-layer's weights and saved activations are fetched into registers a half-layer ahead, and
// the barriers order LDS traffic only (nothing here reads global memory that the kernel wrote): as in the forward.
template <int TRD, int NTD>      // rows per workgroup, threads: 32 x 1024 (one CU takes all of batch 32) or 16 x 512 (two CUs: full 16-row MFMA tiles each)
__global__ __launch_bounds__(NTD) void flow_dgrad_kernel(FlowDims Q, FlowD T, FlowBwdBuf U)
{
    constexpr int G = NTD / H;        // thread t = (unit j = t % H, group g = t / H): a group takes rows 4 g .. 4 g + 3 in the VALU phases
    static_assert(TRD == 4 * G && TRD % 16 == 0 && NTD / 64 == 8 * (TRD / 16), "one wave per 16 x 16 tile of the [TRD, H] MFMA phase");
    __shared__ __attribute__((aligned(16))) float h1r[TRD][LDR];        // [row][unit]
    __shared__ __attribute__((aligned(16))) float h2r[TRD][LDR];        // later: d(pre-activation 1)
    __shared__ __attribute__((aligned(16))) float da2r[TRD][LDR];       // d(pre-activation 2); later: partial tiles of the input gradient
    __shared__ float do_s[TRD][M_MAX + 1];      // d(outputs): an MFMA operand read a dword at a time down the rows (+ 1: banks)
    __shared__ __attribute__((aligned(16))) float w2t[H][LDT];
    __shared__ float w1f[H * DI_MAX + DI_MAX];  // W1 as it lies in memory, [unit][DI] (+ DI_MAX: the last units' reads past their rows)
    __shared__ float w3s[M_MAX][H + 16];        // W3 [output][unit], rows >= M zero (+ 16: lane groups 0 and 1 read rows m, m + 1: banks 16 apart)
    __shared__ float gout_s[TRD][D_MAX], gz_s[TRD][D_MAX], gy2_s[TRD][D_MAX], gcond_s[TRD][DI_MAX];
    const int t = threadIdx.x, j = t & (H - 1), g = t >> 7, lane = t & 63, wave = t >> 6, D = Q.D, d1 = Q.d1, d2 = D - d1;
    const int r0 = blockIdx.x * TRD;
    // this thread's (row, column) of the affine phase: rows 0 .. TRD - 1 x up to M_MAX / 2 columns
    float gld_sc = 0.0f;                                     // nll form: the uniform gradient of log|det|
    if (U.g_nll) gld_sc = -*U.g_nll / (float)Q.R;
    if (t < TRD * D) {
        const int r = t / D, c = t - r * D, row = r0 + r;
        float v = 0.0f;
        if (row < Q.R) v = U.g_nll ? U.z_last[(long long)row * D + c] * (-gld_sc) : U.g_z[(long long)row * D + c];
        gout_s[r][c] = v;
    }
    if (U.g_nll && t < TRD && r0 + t < Q.R) U.w_gld[r0 + t] = gld_sc;
    for (int p = t; p < TRD * DI_MAX; p += NTD) (&gcond_s[0][0])[p] = 0.0f;
    for (int p = t; p < TRD * (M_MAX + 1); p += NTD) (&do_s[0][0])[p] = 0.0f;      // (columns >= M are never written: they meet zero rows of w3s)
    // one thread's share of the next half-layer's operands
    f32x4 nw2[H * H / 4 / NTD];       // thread (in unit j, g): out units 4 (g + G k4) .. + 3 of column j
    float nw1[H * DI_MAX / NTD], nw3[M_MAX * H / NTD], nh1[TRD * H / NTD], nh2[TRD * H / NTD], ns = 0.0f, nx = 0.0f, ngl = 0.0f;
#define NDDM_FETCH_HALF(X) do {                                                                                               \
        const int DIn_ = (X).Dh + Q.C, Mn_ = 2 * (X).Dt;                                                                       \
        _Pragma("unroll") for (int k4 = 0; k4 < H * H / 4 / NTD; ++k4) {          /* (a wave: 64 consecutive floats of a row) */      \
            const float *w_ = (X).W2 + (long long)(4 * (g + G * k4)) * H + j;                                                    \
            nw2[k4] = f32x4{w_[0], w_[H], w_[2 * H], w_[3 * H]}; }                                                              \
        _Pragma("unroll") for (int k = 0; k < H * DI_MAX / NTD; ++k) nw1[k] = (X).W1[min(t + NTD * k, H * DIn_ - 1)];          \
        _Pragma("unroll") for (int k = 0; k < M_MAX * H / NTD; ++k) {                                                          \
            const int p_ = t + NTD * k; const float v_ = (X).W3[min(p_, Mn_ * H - 1)]; nw3[k] = p_ < Mn_ * H ? v_ : 0.0f; }      \
        _Pragma("unroll") for (int k = 0; k < TRD * H / NTD; ++k) {                                                             \
            const long long o_ = (long long)min(r0 + g + G * k, Q.R - 1) * H + j;                                              \
            nh1[k] = (X).h1[o_]; nh2[k] = (X).h2[o_];                                                                          \
        }                                                                                                                      \
        {   const int r_ = t / (X).Dt, d_ = t - r_ * (X).Dt; const long long o_ = (long long)min(r0 + min(r_, TRD - 1), Q.R - 1) * D + d_; \
            ns = (X).s[o_]; nx = (X).xtr[o_]; ngl = U.g_nll ? gld_sc : U.g_ld[min(r0 + min(r_, TRD - 1), Q.R - 1)]; }           \
    } while (0)
    NDDM_FETCH_HALF(T.half[2 * Q.L - 1]);
    for (int hl = 2 * Q.L - 1; hl >= 0; --hl) {
        const HalfD &X = T.half[hl];
        const bool second = hl & 1;
        const int DI = X.Dh + Q.C, M = 2 * X.Dt, Dt = X.Dt;
        const RowBuf bda2(X.da2 + (long long)r0 * H, Q.R - r0, H), bda1(X.da1 + (long long)r0 * H, Q.R - r0, H);
        lds_barrier();                    // the previous half-layer's readers are done
        STAMP(10);
#pragma unroll
        for (int k4 = 0; k4 < H * H / 4 / NTD; ++k4) *reinterpret_cast<f32x4 *>(&w2t[j][4 * (g + G * k4)]) = nw2[k4];
#pragma unroll
        for (int k = 0; k < H * DI_MAX / NTD; ++k) w1f[t + NTD * k] = nw1[k];      // (a linear copy; entries beyond H * DI: copies of the last one)
#pragma unroll
        for (int k = 0; k < M_MAX * H / NTD; ++k) { const int p = t + NTD * k; w3s[p >> 7][p & (H - 1)] = nw3[k]; }
#pragma unroll
        for (int k = 0; k < TRD * H / NTD; ++k) {         // saved activations (rows beyond R: zero)
            const int r = g + G * k;
            const bool ok = r0 + r < Q.R;
            h1r[r][j] = ok ? nh1[k] : 0.0f;
            h2r[r][j] = ok ? nh2[k] : 0.0f;
        }
        STAMP(15);
        if (t < TRD * Dt) {                               // through the affine transform and the soft clamp
            const int r = t / Dt, d = t - r * Dt, row = r0 + r;
            // second: the gradient of out[:, :d1]; first: of out[:, d1:] plus what came through net 2's conditioning input
            const float gg = second ? gout_s[r][d] : gout_s[r][d1 + d] + gy2_s[r][d];
            // d s / d o = 1 - tanh^2 = (clamp - s)(clamp + s) / clamp^2: the DIFFERENCE clamp - s is exact where it matters (a log-scale
            // near the clamp's bound: Sterbenz), whereas 1 - (s / clamp)^2 rounds the quotient and its square first -- three times
            // PyTorch's error on saturated units (it keeps tanh's own output), and a sharply trained flow sits at the bound
            const float es = expf(ns), dclamp = (Q.clamp - ns) * (Q.clamp + ns) / (Q.clamp * Q.clamp);
            const float d_os = row < Q.R ? fmaf(gg * nx, es, ngl) * dclamp : 0.0f, d_t = row < Q.R ? gg : 0.0f;
            gz_s[r][second ? d : d1 + d] = gg * es;      // d z of the transformed half (the other half: the input gradient below)
            do_s[r][d] = d_os;
            do_s[r][Dt + d] = d_t;
            if (row < Q.R) {
                X.dos[(long long)row * M_MAX + d] = d_os;
                X.dos[(long long)row * M_MAX + Dt + d] = d_t;
            }
        }
        STAMP(16);
        if (hl > 0) NDDM_FETCH_HALF(T.half[hl - 1]);     // (the loads' results are not waited for before the next half-layer)
        STAMP(17);
        lds_barrier();
        STAMP(11);
        {   // d h2 = d(outputs) W3 -> d(pre-activation 2), [TRD x 16] x [16 x H] as MFMAs (outputs >= M: zero rows of w3s): wave ->
            // (row block, 16 units), lane group kk takes the outputs 4 i + kk
            const int n = lane & 15, kk = lane >> 4, rb = wave >> 3, ib = wave & 7;
            const float *ap = &do_s[16 * rb + n][kk], *bp = &w3s[kk][16 * ib + n];
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < M_MAX / 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * i], bp[4 * i * (H + 16)], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // D: unit 16 ib + (l & 15), row 16 rb + 4 (l >> 4) + register
                const int r = 16 * rb + 4 * kk + q, i = 16 * ib + n;
                const float v = acc[q] * elu_grad_from_pre(h2r[r][i]);
                da2r[r][i] = v;
                bda2.put(r * H + i, v);
            }
        }
        lds_barrier();
        STAMP(12);
        {   // d h1 = da2 W2 -> d(pre-activation 1).  Lane l: A[row l & 15][k] = da2[row][k], B[k][i = l & 15] = W2[k][i] = w2t[i][k] for the
            // out units k = 32 (l >> 4) + s, s = 0 .. 31 (two accumulators: the even and the odd groups of four, a shorter dependent chain)
            const int n = lane & 15, kk = lane >> 4, rb = wave >> 3, ib = wave & 7;
            const int kb = ((kk & 1) << 1) | (kk >> 1);      // lane groups 0, 1, 2, 3 take the k blocks 0, 2, 1, 3: see the forward's layer 2
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc;
            const float4 *ap = reinterpret_cast<const float4 *>(&da2r[16 * rb + n][32 * kb]);
            const float4 *bp = reinterpret_cast<const float4 *>(&w2t[16 * ib + n][32 * kb]);
#pragma unroll
            for (int q4 = 0; q4 < 8; q4 += 2) {
                const float4 a0 = ap[q4], b0 = bp[q4], a1 = ap[q4 + 1], b1 = bp[q4 + 1];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
            }
            acc += acc1;
            // D: unit 16 ib + (l & 15), row 16 rb + 4 (l >> 4) + register  (h2r: its readers finished before the barrier above)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * rb + 4 * kk + q, i = 16 * ib + n;
                const float v = acc[q] * elu_grad_from_pre(h1r[r][i]);
                h2r[r][i] = v;
                bda1.put(r * H + i, v);
            }
        }
        lds_barrier();
        STAMP(13);
        {   // d in = d(a1) W1, [TRD x DI] with k over the H units: wave -> (row block, column block, a quarter of the k's), 8 MFMAs;
            // the four partial tiles are summed through LDS in fixed order.  Lane l: A[row l & 15][k], B[k][column l & 15] for
            // k = 32 q + 8 (l >> 4) + s, s = 0 .. 7.  (Columns >= DI read the next unit's weights: they only reach columns >= DI of the result, which nobody reads.)
            constexpr int RB = TRD / 16;
            const int n = lane & 15, kk = lane >> 4, rb = wave % RB, cb = (wave / RB) & 1, q = wave / (2 * RB);
            float (*part)[TRD][DI_MAX + 1] = reinterpret_cast<float (*)[TRD][DI_MAX + 1]>(&da2r[0][0]);    // (da2r's readers are done)
            if (16 * cb < DI) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                const float *ap = &h2r[16 * rb + n][32 * q + 8 * kk], *bp = &w1f[(32 * q + 8 * kk) * DI + 16 * cb + n];
                const float4 a0 = *reinterpret_cast<const float4 *>(ap), a1 = *reinterpret_cast<const float4 *>(ap + 4);
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], bp[s_ * DI], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) part[q][16 * rb + 4 * kk + v][16 * cb + n] = acc[v];
            }
            lds_barrier();
            if (t < TRD * DI) {
                const int r = t / DI, c = t - r * DI;
                const float acc = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
                if (c >= X.Dh) gcond_s[r][c - X.Dh] += acc;
                else if (second) gy2_s[r][c] = acc;      // net 2 is conditioned on out[:, d1:]
                else gz_s[r][c] += acc;                  // net 1 on z[:, :d1], which net 2 also transformed
            }
        }
        STAMP(14);
        if (second) continue;
        lds_barrier();
        // back through the permutation and the ActNorm (the parameters' gradients: kernel (2))
        const NormD &N = T.norm[hl >> 1];
        if (t < TRD * D) {
            const int r = t / D, c = t - r * D, row = r0 + r, p = N.perm[c];
            const float gv = gz_s[r][c];
            gout_s[r][p] = gv * expf(N.scale[p]);
            if (row < Q.R) N.gz[(long long)row * D + c] = gv;
        }
    }
#undef NDDM_FETCH_HALF
    lds_barrier();
    if (t < TRD * D) { const int r = t / D, c = t - r * D; if (r0 + r < Q.R) U.gx[(long long)(r0 + r) * D + c] = gout_s[r][c]; }
    for (int p = t; p < TRD * Q.C; p += NTD) {
        const int r = p / Q.C, c = p - r * Q.C;
        if (r0 + r < Q.R) U.gcond[(long long)(r0 + r) * Q.C + c] = gcond_s[r][c];
    }
}

struct HalfW {                   // one half-layer of kernel (2)
    const float *xh, *h1, *h2, *da2, *da1, *dos;
    float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3;
    int Dh, Dt, norm, pad_;
};
struct NormW { const float *x, *scale, *gz; float *gscale, *gbias; unsigned char perm[D_MAX]; };
struct FlowW { HalfW half[2 * L_MAX]; NormW norm[L_MAX]; };

//   dW2: wave w owns the 32 x 32 tile (rows 32 (w / 4), columns 32 (w % 4)): 16 accumulator registers
//   dW1: thread (unit j = t % 128, g = t / 128): columns c with c % 8 == g      <= 4 registers
//   dW3: column j of rows m with m % 8 == g                                     <= 2
__global__ __launch_bounds__(NTB) void flow_wgrad_kernel(FlowDims Q, FlowW T, const float *cond, const float *g_ld)
{
    __shared__ float in_s[TR][DI_MAX];
    __shared__ float h1r[TR][LDR];
    __shared__ float h2r[TR][LDR];
    __shared__ float da2r[TR][LDR];
    __shared__ float da1r[TR][LDR];
    __shared__ float do_s[TR][M_MAX];
    const HalfW &X = T.half[blockIdx.x];
    const int t = threadIdx.x, j = t & (H - 1), g = t >> 7, lane = t & 63, wave = t >> 6, D = Q.D;
    const int DI = X.Dh + Q.C, M = 2 * X.Dt;
    f32x16 dW2;
#pragma unroll
    for (int q = 0; q < 16; ++q) dW2[q] = 0.0f;
    // The sums one thread carries through ALL rows in sequence -- the three bias gradients, dW1, dW3 (and the ActNorm's below) -- are
    // accumulated in float64: a sequential float32 sum of R same-signed terms loses ~R/2 ulps where PyTorch's tree reduction loses
    // log2 R, measured as 3 .. 12 x PyTorch's error against a float64 evaluation on the bias gradients (tools/fused_accuracy.py,
    // profiles/r5_fused_accuracy.txt).  A handful of f64 adds per row tile in a kernel whose time is load latency; the products stay f32
    // (exact in f64).  dW2 is the MFMA chain -- the same K-long f32 accumulation a GEMM makes: at PyTorch's error already.
    double dW1[DI_MAX / 8], dW3[M_MAX / 8];
    double db2 = 0.0, db1 = 0.0, db3 = 0.0;
#pragma unroll
    for (int q = 0; q < DI_MAX / 8; ++q) dW1[q] = 0.0;
#pragma unroll
    for (int q = 0; q < M_MAX / 8; ++q) dW3[q] = 0.0;
    for (int r0 = 0; r0 < Q.R; r0 += TR) {
        if (r0) __syncthreads();          // the previous tile's readers are done
        // every load unconditional, from a clamped row, ALL of them issued before the first is used (a guarded load -- `ok ? p[o] : 0`
        // -- is a branch and a wait of its own: the 18 loads of this prologue were ten round trips in a row, most of the kernel)
        float v1[TR * H / NTB], v2[TR * H / NTB], v3[TR * H / NTB], v4[TR * H / NTB];
#pragma unroll
        for (int k = 0; k < TR * H / NTB; ++k) {
            const long long o = (long long)min(r0 + g + 8 * k, Q.R - 1) * H + j;
            v1[k] = X.h1[o]; v2[k] = X.h2[o]; v3[k] = X.da2[o]; v4[k] = X.da1[o];
        }
        float vin = 0.0f, vdo = 0.0f;
        {
            const int r = t / DI, c = t - r * DI, row = min(r0 + min(r, TR - 1), Q.R - 1);     // (TR * DI <= NTB: one element per thread)
            const float a = X.xh[(long long)row * D + min(c, X.Dh - 1)], b = cond[(long long)row * Q.C + min(max(c - X.Dh, 0), Q.C - 1)];
            vin = c < X.Dh ? a : b;
            const int r2 = t / M_MAX, m2 = t - r2 * M_MAX;
            vdo = X.dos[(long long)min(r0 + min(r2, TR - 1), Q.R - 1) * M_MAX + m2];
        }
        if (t < TR * DI) { const int r = t / DI, c = t - r * DI; in_s[r][c] = r0 + r < Q.R ? vin : 0.0f; }
#pragma unroll
        for (int k = 0; k < TR * H / NTB; ++k) {         // (rows beyond R: zero)
            const int r = g + 8 * k;
            const bool ok = r0 + r < Q.R;
            h1r[r][j] = ok ? elu(v1[k]) : 0.0f;          // (the saved tensors hold the pre-activations)
            h2r[r][j] = ok ? elu(v2[k]) : 0.0f;
            da2r[r][j] = ok ? v3[k] : 0.0f;
            da1r[r][j] = ok ? v4[k] : 0.0f;
        }
        if (t < TR * M_MAX) { const int r = t / M_MAX, m = t - r * M_MAX; do_s[r][m] = (r0 + r < Q.R && m < M) ? vdo : 0.0f; }
        __syncthreads();
        // layer 3: thread (column j, rows m = g, g + 8)
#pragma unroll
        for (int q = 0; q < M_MAX / 8; ++q) {
            const int m = 8 * q + g;
            if (m < M) {
                double acc = dW3[q];
#pragma unroll 8
                for (int r = 0; r < TR; ++r) acc = fma((double)do_s[r][m], (double)h2r[r][j], acc);
                dW3[q] = acc;
            }
        }
        if (t < M) {
#pragma unroll 4
            for (int r = 0; r < TR; ++r) db3 += (double)do_s[r][t];
        }
        {   // layer 2, dW2[jo][i] += sum_r da2[r][jo] h1[r][i].  Lane l: A[jo = l & 31][k = l >> 5], B[k][i = l & 31];
            // the sum's order is free, so step s takes rows r = 16 k + s (A and B alike)
            const int m = lane & 31, kk = lane >> 5;
            const float *ap = &da2r[16 * kk][32 * (wave >> 2) + m], *bp = &h1r[16 * kk][32 * (wave & 3) + m];
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) dW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s_ * LDR], bp[s_ * LDR], dW2, 0, 0, 0);
        }
        if (t < H) {
#pragma unroll 8
            for (int r = 0; r < TR; ++r) db2 += (double)da2r[r][t];
        }
        // layer 1: thread (row j, columns c = g, g + 8, ...)
#pragma unroll
        for (int q = 0; q < DI_MAX / 8; ++q) {
            const int c = 8 * q + g;
            if (c < DI) {
                double acc = dW1[q];
#pragma unroll 8
                for (int r = 0; r < TR; ++r) acc = fma((double)da1r[r][j], (double)in_s[r][c], acc);
                dW1[q] = acc;
            }
        }
        if (g == 0) {
#pragma unroll 8
            for (int r = 0; r < TR; ++r) db1 += (double)da1r[r][j];
        }
    }
    {   // dW2's tile: register v of lane l is row (v & 3) + 8 (v >> 2) + 4 (l >> 5), column l & 31
        const int m = lane & 31, kk = lane >> 5;
        float *o = X.gW2 + (long long)(32 * (wave >> 2) + 4 * kk) * H + 32 * (wave & 3) + m;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[((v & 3) + 8 * (v >> 2)) * H] = dW2[v];
    }
#pragma unroll
    for (int q = 0; q < DI_MAX / 8; ++q) { const int c = 8 * q + g; if (c < DI) X.gW1[j * DI + c] = (float)dW1[q]; }
#pragma unroll
    for (int q = 0; q < M_MAX / 8; ++q) { const int m = 8 * q + g; if (m < M) X.gW3[(long long)m * H + j] = (float)dW3[q]; }
    if (t < H) X.gb2[t] = (float)db2;
    if (g == 0) X.gb1[j] = (float)db1;
    if (t < M) X.gb3[t] = (float)db3;
    if (X.norm < 0) return;
    // the ActNorm's parameters: 32 threads per column, fixed order
    const NormW &N = T.norm[X.norm];
    if (t < 32 * D) {
        const int c = t >> 5, i = t & 31, p = N.perm[c];
        const float ex = expf(N.scale[p]);
        double sb = 0.0, ss = 0.0, sd = 0.0;                 // (f64 sums: as above)
        for (int row = i; row < Q.R; row += 32) {
            const float gv = N.gz[(long long)row * D + c];
            sb += (double)gv;
            ss = fma((double)(gv * ex), (double)N.x[(long long)row * D + p], ss);
            sd += (double)g_ld[row];
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
            sb += __shfl_xor(sb, m, 32);
            ss += __shfl_xor(ss, m, 32);
            sd += __shfl_xor(sd, m, 32);
        }
        if (i == 0) { N.gbias[p] = (float)sb; N.gscale[p] = (float)(ss + sd); }
    }
}

// mean over the rows of |z|^2 / 2 - log|det| (fixed summation order; the running sums in float64: one workgroup, latency-bound)
__global__ __launch_bounds__(256) void nll_kernel(const float *z, const float *ld, int R, int D, float *out)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int r = threadIdx.x; r < R; r += 256) {
        double q = 0.0;
        for (int c = 0; c < D; ++c) { const double v = (double)z[(long long)r * D + c]; q = fma(v, v, q); }
        a += 0.5 * q - (double)ld[r];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if (threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] / (double)R);
}

}  // namespace nddm_train

using namespace nddm_train;

extern "C" {

// 1 for the shapes the fused flow covers (the caller takes the PyTorch path otherwise)
int nddm_train_flow_supported(int hidden, int L, int D, int d1, int C)
{
    const int d2 = D - d1;
    return (hidden == H && L >= 1 && L <= L_MAX && D >= 2 && D <= D_MAX && d1 >= 1 && d2 >= 1 && d1 + C <= DI_MAX && d2 + C <= DI_MAX
            && C >= 0 && 2 * d1 <= M_MAX && 2 * d2 <= M_MAX) ? 1 : 0;
}

/* params: L x 14 device pointers per layer -- ActNorm log-scale [D] and bias [D], then W1 [H, Dh + C], b1 [H], W2 [H, H], b2 [H],
 * W3 [2 Dt, H], b3 [2 Dt] of sub-network 1 (conditioned on the first d1 columns) and of sub-network 2; perm: L x D host ints. */
static void fill(FlowP &P, int L, int D, const void *const *params, const int *perm)
{
    for (int l = 0; l < L; ++l) {
        const float *const *q = reinterpret_cast<const float *const *>(params) + 14 * l;
        P.layer[l] = {q[0], q[1], {q[2], q[3], q[4], q[5], q[6], q[7]}, {q[8], q[9], q[10], q[11], q[12], q[13]}};
        for (int d = 0; d < D; ++d) P.perm[l][d] = (unsigned char)perm[l * D + d];
    }
}

/* -> z = out_all[L - 1] and ld [R] (log|det|, ActNorm terms included); z_all / out_all / s_all [L, R, D] and h_all [L, 4, R, H] (the PRE-activations of the coupling nets' two hidden layers)
 * are what the backward needs.  nll (may be NULL): the maximum-likelihood loss, mean over the rows of |z|^2 / 2 - log|det|. */
int nddm_train_flow_fwd(int L, int R, int D, int d1, int C, float clamp, const void *const *params, const int *perm,
                        const float *theta, const float *cond, float *z_all, float *out_all, float *s_all, float *h_all, float *ld,
                        float *nll, void *stream)
{
    if (!nddm_train_flow_supported(H, L, D, d1, C) || R <= 0) return 1;
    for (int i = 0; i < L * D; ++i) if (perm[i] < 0 || perm[i] >= D) return 1;
    FlowP P = {};
    fill(P, L, D, params, perm);
    const FlowDims Q = {L, R, D, d1, C, clamp};
    const FlowSaved S = {z_all, out_all, s_all, h_all};
    hipLaunchKernelGGL((flow_fwd_kernel<16, 512>), dim3((R + 15) / 16), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), Q, P, theta, cond, S, ld);
    if (nll)
        hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out_all + (long long)(L - 1) * R * D,
                           ld, R, D, nll);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* grads: L x 14 device pointers, the layout of params.  g_z [R, D], g_ld [R]: gradients of the forward's two results -- or, with
 * g_nll (the gradient of the forward's nll, a device scalar) not NULL, g_z unused (may be NULL) and g_ld [R] scratch that the
 * kernel fills.
 * Scratch: gz_all [L, R, D], work [2 L, R, 2 * 128 + 16].  gx [R, D] ends as the gradient of theta, gcond [R, C]
 * as that of the condition. */
int nddm_train_flow_bwd(int L, int R, int D, int d1, int C, float clamp, const void *const *params, const int *perm,
                        void *const *grads, const float *theta, const float *cond, float *z_all, float *out_all, float *s_all,
                        float *h_all, const float *g_z, float *g_ld, const float *g_nll, float *gz_all, float *gx, float *gcond,
                        float *work, void *stream)
{
    if (!nddm_train_flow_supported(H, L, D, d1, C) || R <= 0) return 1;
    for (int i = 0; i < L * D; ++i) if (perm[i] < 0 || perm[i] >= D) return 1;
    const int d2 = D - d1;
    const long long RD = (long long)R * D, RH = (long long)R * H, RW = (long long)R * (2 * H + M_MAX);
    FlowD TD = {};
    FlowW TW = {};
    for (int l = 0; l < L; ++l) {
        const float *const *q = reinterpret_cast<const float *const *>(params) + 14 * l;
        float *const *g = reinterpret_cast<float *const *>(grads) + 14 * l;
        const float *x = l ? out_all + (l - 1) * RD : theta;
        const float *z = z_all + l * RD, *out = out_all + l * RD, *sl = s_all + l * RD, *h = h_all + 4 * l * RH;
        float *gz = gz_all + l * RD, *wa = work + (2 * l) * RW, *wb = work + (2 * l + 1) * RW;
        // sub-network 2 (runs first): conditioned on out[:, d1:], transformed z[:, :d1]
        TD.half[2 * l + 1] = {z, q[8], q[10], q[12], sl + d2, h + 2 * RH, h + 3 * RH, wb, wb + RH, wb + 2 * RH, d2, d1};
        TW.half[2 * l + 1] = {out + d1, h + 2 * RH, h + 3 * RH, wb, wb + RH, wb + 2 * RH, g[8], g[9], g[10], g[11], g[12], g[13], d2, d1, -1, 0};
        // sub-network 1: conditioned on z[:, :d1], transformed z[:, d1:]
        TD.half[2 * l] = {z + d1, q[2], q[4], q[6], sl, h, h + RH, wa, wa + RH, wa + 2 * RH, d1, d2};
        TW.half[2 * l] = {z, h, h + RH, wa, wa + RH, wa + 2 * RH, g[2], g[3], g[4], g[5], g[6], g[7], d1, d2, l, 0};
        TD.norm[l] = {q[0], gz, {}};
        TW.norm[l] = {x, q[0], gz, g[0], g[1], {}};
        for (int d = 0; d < D; ++d) TD.norm[l].perm[d] = TW.norm[l].perm[d] = (unsigned char)perm[l * D + d];
    }
    const FlowDims Q = {L, R, D, d1, C, clamp};
    const FlowBwdBuf U = {g_z, g_ld, gx, gcond, g_nll, g_ld, out_all + (long long)(L - 1) * RD};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((flow_dgrad_kernel<16, 512>), dim3((R + 15) / 16), dim3(512), 0, st, Q, TD, U);
    hipLaunchKernelGGL(flow_wgrad_kernel, dim3(2 * L), dim3(NTB), 0, st, Q, TW, cond, static_cast<const float *>(g_ld));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"

#ifdef NDDM_TRAIN_STAMPS
extern "C" int nddm_train_read_stamps(unsigned long long *out, int cap)
{
    int n = 0;
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(nddm_train::g_nstamps), sizeof(int));
    if (n > cap) n = cap;
    if (n > 4096) n = 4096;
    hipMemcpyFromSymbol(out, HIP_SYMBOL(nddm_train::g_stamps), sizeof(unsigned long long) * 2 * n);
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(nddm_train::g_nstamps), &zero, sizeof(int));
    return n;
}
#endif

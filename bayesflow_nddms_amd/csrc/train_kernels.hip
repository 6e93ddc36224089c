// train_kernels.hip -- fused forward / backward of ONE conditional affine-coupling half-layer of the amortizer's flow
// (bayesflow_nddms_amd/amortizer.py::_AffineCoupling; the stand-in for bf.networks.InvertibleNetwork, basic_ddm_dc.py:163-165),
// for the batch sizes of the online-training loop (32 sets per rank: basic_ddm_dc.py:199-202).
//
// At batch 32 the flow is ~400 of the ~650 kernels of a graph-replayed training iteration, each a 3-5 microsecond launch that
// touches a few kilobytes: concatenate, three GEMMs of 32 rows, two ELUs, the soft clamp, exp, multiply-add, and twice that
// backward -- 33 launches per half-layer, twelve half-layers.  Here a half-layer is ONE launch each way:
//     in = [x_h | cond]  ->  h1 = elu(W1 in + b1)  ->  h2 = elu(W2 h1 + b2)  ->  (o_s | o_t) = W3 h2 + b3
//     s = clamp * tanh(o_s / clamp),   y = x_tr * exp(s) + o_t                      (returns y and s; log|det| = sum of s)
// Not MFMA work: 32 x 128 x 128 multiply-adds per GEMM is a microsecond of plain FMAs; what is bought is launches.
// Hidden width 128 (the networks' default), at most 32 inputs and 8 transformed columns; everything else takes the PyTorch path.
// gfx950 only.  Test infrastructure compares both paths (tests/test_gpu_training.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nddm_train {

constexpr int H = 128;        // hidden width
constexpr int TR = 32;        // rows per tile
constexpr int DI_MAX = 32;    // inputs of the sub-network (x_h columns + condition columns)
constexpr int M_MAX = 16;     // outputs (2 * transformed columns)

__device__ __forceinline__ float elu(float v) { return v > 0.0f ? v : expm1f(v); }
__device__ __forceinline__ float elu_grad_from_out(float o) { return o > 0.0f ? 1.0f : o + 1.0f; }   // alpha = 1

struct Args {
    const float *xh; int ldh, Dh;         // [R, Dh] with row stride ldh: the conditioning half
    const float *cond; int C;             // [R, C]
    const float *xtr; int ldt, Dt;        // [R, Dt] with row stride ldt: the transformed half
    const float *W1, *b1, *W2, *b2, *W3, *b3;   // [H, Dh+C], [H], [H, H], [H], [2 Dt, H], [2 Dt]
    float clamp;
    int R;
};

// ------------------------------------------------------------------------------------------------ forward
// grid = ceil(R / TRF) workgroups of 256 threads over tiles of TRF = 8 rows (four workgroups at batch 32: the work is latency,
// not arithmetic, so it is spread).  Thread t: hidden unit j = t % 128, row half rh = t / 128 (TRF / 2 rows each).
constexpr int TRF = 8, RPT = TRF / 2;
constexpr int TRFP = TRF + 4;   // row stride of the transposed activations in LDS: lanes index the UNIT, and a stride of 8 (or 32)
                                // floats puts a whole wave on two banks; + 4 keeps 16-byte alignment and leaves 4-way conflicts
__global__ __launch_bounds__(256) void coupling_fwd_kernel(Args A, float *y, int ldy, float *s_out, int lds, float *h1_out, float *h2_out)
{
    __shared__ float in_s[TRF][DI_MAX];
    __shared__ __attribute__((aligned(16))) float h1t[H][TRFP];        // transposed: [unit][row]
    __shared__ __attribute__((aligned(16))) float h2t[H][TRFP];
    __shared__ float o_s[TRF][M_MAX];
    const int t = threadIdx.x, j = t & (H - 1), rh = t >> 7;
    const int r0 = blockIdx.x * TRF, DI = A.Dh + A.C, M = 2 * A.Dt;
    for (int p = t; p < TRF * DI; p += 256) {
        const int r = p / DI, c = p - r * DI, row = r0 + r;
        float v = 0.0f;
        if (row < A.R) v = c < A.Dh ? A.xh[(long long)row * A.ldh + c] : A.cond[(long long)row * A.C + (c - A.Dh)];
        in_s[r][c] = v;
    }
    __syncthreads();
    {   // layer 1 (register arrays are only ever indexed by unrolled constants: no private scratch)
        float acc[RPT];
        const float b = A.b1[j];
#pragma unroll
        for (int q = 0; q < RPT; ++q) acc[q] = b;
#pragma unroll 2
        for (int c = 0; c < DI; ++c) {
            const float wv = A.W1[j * DI + c];
#pragma unroll
            for (int q = 0; q < RPT; ++q) acc[q] = fmaf(wv, in_s[rh * RPT + q][c], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = rh * RPT + q;
            const float v = elu(acc[q]);
            h1t[j][r] = v;
            if (r0 + r < A.R) h1_out[(long long)(r0 + r) * H + j] = v;
        }
    }
    __syncthreads();
    {   // layer 2: thread j streams its own row of W2
        float acc[RPT];
        const float b = A.b2[j];
#pragma unroll
        for (int q = 0; q < RPT; ++q) acc[q] = b;
        const float4 *wrow = reinterpret_cast<const float4 *>(A.W2 + (long long)j * H);
#pragma unroll 2
        for (int i4 = 0; i4 < H / 4; ++i4) {
            const float4 w4 = wrow[i4];
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *hr = &h1t[4 * i4 + u][rh * RPT];
#pragma unroll
                for (int q = 0; q < RPT; ++q) acc[q] = fmaf(wv[u], hr[q], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = rh * RPT + q;
            const float v = elu(acc[q]);
            h2t[j][r] = v;
            if (r0 + r < A.R) h2_out[(long long)(r0 + r) * H + j] = v;
        }
    }
    __syncthreads();
    for (int p = t; p < TRF * M; p += 256) {          // layer 3
        const int r = p / M, m = p - r * M;
        float acc = A.b3[m];
        const float *w = A.W3 + (long long)m * H;
#pragma unroll 4
        for (int i = 0; i < H; ++i) acc = fmaf(w[i], h2t[i][r], acc);
        o_s[r][m] = acc;
    }
    __syncthreads();
    for (int p = t; p < TRF * A.Dt; p += 256) {       // soft clamp + affine transform
        const int r = p / A.Dt, d = p - r * A.Dt, row = r0 + r;
        if (row >= A.R) continue;
        const float s = A.clamp * tanhf(o_s[r][d] / A.clamp);
        s_out[(long long)row * lds + d] = s;
        y[(long long)row * ldy + d] = fmaf(A.xtr[(long long)row * A.ldt + d], expf(s), o_s[r][A.Dt + d]);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// ONE workgroup of 1024 threads walks the row tiles and keeps the weight gradients in registers (deterministic sums, no
// atomics).  Thread t: unit j = t % 128, group g = t / 128 (0..7):
//   dW2: row j, columns [16 g, 16 g + 16)                 16 registers
//   dW1: row j, columns c with c % 8 == g                 <= 4
//   dW3: column j of rows m with m % 8 == g               <= 2
//   activations' gradients: unit j, rows [4 g, 4 g + 4) of the tile
struct BwdOut {
    float *gxh; int ldgh;      // [R, Dh]
    float *gcond;              // [R, C]
    float *gxtr; int ldgt;     // [R, Dt]
    float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3;
    int acc_gxh, acc_gcond;    // add to what gxh / gcond hold instead of overwriting (the second half-layer of a coupling layer)
};
struct BwdIn {
    const float *s; int lds;       // [R, Dt] clamped log-scales saved by the forward
    const float *gy; int ldgy;     // [R, Dt] gradient of the transformed half
    const float *gy2; int ldgy2;   // optional second contribution to it (null: none)
    const float *gs; int ldgs;     // [R, Dt] gradient of the log-scales
};
constexpr int NTB = 1024;
constexpr int TRP = TR + 4;      // (as TRFP: unpadded, the transposing stores and every per-unit read were 32-way bank conflicts)

__global__ __launch_bounds__(NTB) void coupling_bwd_kernel(Args A, BwdIn I, const float *h1_in, const float *h2_in, BwdOut O)
{
    __shared__ float in_s[TR][DI_MAX];
    __shared__ __attribute__((aligned(16))) float h1t[H][TRP];
    __shared__ __attribute__((aligned(16))) float h2t[H][TRP];         // later: d(pre-activation 1), transposed
    __shared__ __attribute__((aligned(16))) float da2t[H][TRP];
    __shared__ float do_s[TR][M_MAX];
    __shared__ float w2s[H][H];           // W2 staged once (64 KB of gfx950's 160 KB): its column reads below were the kernel's latency
    const int t = threadIdx.x, j = t & (H - 1), g = t >> 7;
    for (int p = t; p < H * H; p += NTB) w2s[p >> 7][p & (H - 1)] = A.W2[p];
    const int DI = A.Dh + A.C, M = 2 * A.Dt;
    float dW2[16], dW1[DI_MAX / 8], dW3[M_MAX / 8];
    float db2 = 0.0f, db1 = 0.0f, db3 = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) dW2[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < DI_MAX / 8; ++q) dW1[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < M_MAX / 8; ++q) dW3[q] = 0.0f;

    for (int r0 = 0; r0 < A.R; r0 += TR) {
        __syncthreads();                  // the previous tile's readers are done
        for (int p = t; p < TR * DI; p += NTB) {
            const int r = p / DI, c = p - r * DI, row = r0 + r;
            float v = 0.0f;
            if (row < A.R) v = c < A.Dh ? A.xh[(long long)row * A.ldh + c] : A.cond[(long long)row * A.C + (c - A.Dh)];
            in_s[r][c] = v;
        }
        for (int p = t; p < TR * H; p += NTB) {          // saved activations, transposed into LDS (rows beyond R: zero)
            const int r = p >> 7, i = p & (H - 1), row = r0 + r;
            h1t[i][r] = row < A.R ? h1_in[(long long)row * H + i] : 0.0f;
            h2t[i][r] = row < A.R ? h2_in[(long long)row * H + i] : 0.0f;
        }
        for (int p = t; p < TR * A.Dt; p += NTB) {       // through the affine transform and the soft clamp
            const int r = p / A.Dt, d = p - r * A.Dt, row = r0 + r;
            float d_os = 0.0f, d_t = 0.0f;
            if (row < A.R) {
                const float s = I.s[(long long)row * I.lds + d], es = expf(s);
                float gg = I.gy[(long long)row * I.ldgy + d];
                if (I.gy2) gg += I.gy2[(long long)row * I.ldgy2 + d];
                const float d_sc = fmaf(gg * A.xtr[(long long)row * A.ldt + d], es, I.gs[(long long)row * I.ldgs + d]);
                const float u = s / A.clamp;
                d_os = d_sc * (1.0f - u * u);
                d_t = gg;
                O.gxtr[(long long)row * O.ldgt + d] = gg * es;
            }
            do_s[r][d] = d_os;
            do_s[r][A.Dt + d] = d_t;
        }
        __syncthreads();
        // layer 3 weight gradient: thread (column j, rows m = g, g + 8)
#pragma unroll
        for (int q = 0; q < M_MAX / 8; ++q) {
            const int m = 8 * q + g;
            if (m < M) {
                float acc = dW3[q];
#pragma unroll 4
                for (int r = 0; r < TR; ++r) acc = fmaf(do_s[r][m], h2t[j][r], acc);
                dW3[q] = acc;
            }
        }
        if (t < M) {
#pragma unroll 4
            for (int r = 0; r < TR; ++r) db3 += do_s[r][t];
        }
        // d h2 -> d(pre-activation 2): thread (unit j, rows 4 g .. 4 g + 3)
        {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
            for (int m = 0; m < M; ++m) {
                const float w = A.W3[(long long)m * H + j];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(w, do_s[4 * g + q][m], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) da2t[j][4 * g + q] = acc[q] * elu_grad_from_out(h2t[j][4 * g + q]);
        }
        __syncthreads();
        // layer 2 weight gradient: thread (row j of dW2, columns 16 g .. 16 g + 15); rows outside, columns unrolled (the
        // accumulators are the only register array)
#pragma unroll 2
        for (int rr = 0; rr < TR / 4; ++rr) {          // four rows per 16-byte LDS read: a quarter of the LDS instructions
            const float4 av = *reinterpret_cast<const float4 *>(&da2t[j][4 * rr]);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 hv = *reinterpret_cast<const float4 *>(&h1t[16 * g + q][4 * rr]);
                dW2[q] = fmaf(av.w, hv.w, fmaf(av.z, hv.z, fmaf(av.y, hv.y, fmaf(av.x, hv.x, dW2[q]))));
            }
            if (g == 0) db2 += (av.x + av.y) + (av.z + av.w);
        }
        // d h1 -> d(pre-activation 1): thread (unit i = j, rows 4 g .. 4 g + 3); W2 by columns, from LDS
        float da1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
        for (int jj = 0; jj < H; ++jj) {
            const float w = w2s[jj][j];
            const float *ar = &da2t[jj][4 * g];
#pragma unroll
            for (int q = 0; q < 4; ++q) da1[q] = fmaf(w, ar[q], da1[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) da1[q] *= elu_grad_from_out(h1t[j][4 * g + q]);
        __syncthreads();                  // h2t's readers (dW3, d h2) are done: it now holds d(pre-activation 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) h2t[j][4 * g + q] = da1[q];
        __syncthreads();
        // layer 1 weight gradient: thread (row j, columns c = g, g + 8, ...)
#pragma unroll
        for (int q = 0; q < DI_MAX / 8; ++q) {
            const int c = 8 * q + g;
            if (c < DI) {
                float acc = dW1[q];
#pragma unroll 4
                for (int r = 0; r < TR; ++r) acc = fmaf(h2t[j][r], in_s[r][c], acc);
                dW1[q] = acc;
            }
        }
        if (g == 0) {
#pragma unroll 4
            for (int r = 0; r < TR; ++r) db1 += h2t[j][r];
        }
        // d in: (row, column) pairs
        for (int p = t; p < TR * DI; p += NTB) {
            const int r = p / DI, c = p - r * DI, row = r0 + r;
            if (row >= A.R) continue;
            float acc = 0.0f;
#pragma unroll 4
            for (int jj = 0; jj < H; ++jj) acc = fmaf(A.W1[jj * DI + c], h2t[jj][r], acc);
            if (c < A.Dh) {
                float *o = &O.gxh[(long long)row * O.ldgh + c];
                *o = O.acc_gxh ? *o + acc : acc;
            } else {
                float *o = &O.gcond[(long long)row * A.C + (c - A.Dh)];
                *o = O.acc_gcond ? *o + acc : acc;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) O.gW2[(long long)j * H + 16 * g + q] = dW2[q];
#pragma unroll
    for (int q = 0; q < DI_MAX / 8; ++q) { const int c = 8 * q + g; if (c < DI) O.gW1[j * DI + c] = dW1[q]; }
#pragma unroll
    for (int q = 0; q < M_MAX / 8; ++q) { const int m = 8 * q + g; if (m < M) O.gW3[(long long)m * H + j] = dW3[q]; }
    if (g == 0) { O.gb2[j] = db2; O.gb1[j] = db1; }
    if (t < M) O.gb3[t] = db3;
}

}  // namespace nddm_train

using namespace nddm_train;

extern "C" {

// returns 0, or 1 for shapes the fused path does not cover (the caller then takes the PyTorch path)
int nddm_train_coupling_supported(int hidden, int Dh, int C, int Dt)
{
    return (hidden == H && Dh >= 1 && C >= 0 && Dh + C <= DI_MAX && Dt >= 1 && 2 * Dt <= M_MAX) ? 1 : 0;
}

int nddm_train_coupling_fwd(const float *xh, int ldh, int Dh, const float *cond, int C, const float *xtr, int ldt, int Dt,
                            const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                            float clamp, int R, float *y, int ldy, float *s, int lds, float *h1, float *h2, void *stream)
{
    if (!nddm_train_coupling_supported(H, Dh, C, Dt) || R <= 0) return 1;
    Args A = {xh, ldh, Dh, cond, C, xtr, ldt, Dt, W1, b1, W2, b2, W3, b3, clamp, R};
    hipLaunchKernelGGL(coupling_fwd_kernel, dim3((R + TRF - 1) / TRF), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, y, ldy,
                       s, lds, h1, h2);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* gy2 (optional, may be NULL): a second contribution to the gradient of the transformed half, added in the kernel.
 * acc_gxh / acc_gcond: add to the values gxh / gcond already hold (the other half-layer's contribution) instead of overwriting. */
int nddm_train_coupling_bwd(const float *xh, int ldh, int Dh, const float *cond, int C, const float *xtr, int ldt, int Dt,
                            const float *W1, const float *W2, const float *W3, float clamp, int R, const float *s, int lds,
                            const float *h1, const float *h2, const float *gy, int ldgy, const float *gy2, int ldgy2,
                            const float *gs, int ldgs, float *gxh, int ldgh, int acc_gxh, float *gcond, int acc_gcond, float *gxtr,
                            int ldgt, float *gW1, float *gb1, float *gW2, float *gb2, float *gW3, float *gb3, void *stream)
{
    if (!nddm_train_coupling_supported(H, Dh, C, Dt) || R <= 0) return 1;
    Args A = {xh, ldh, Dh, cond, C, xtr, ldt, Dt, W1, nullptr, W2, nullptr, W3, nullptr, clamp, R};
    BwdIn I = {s, lds, gy, ldgy, gy2, ldgy2, gs, ldgs};
    BwdOut O = {gxh, ldgh, gcond, gxtr, ldgt, gW1, gb1, gW2, gb2, gW3, gb3, acc_gxh, acc_gcond};
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(1), dim3(NTB), 0, reinterpret_cast<hipStream_t>(stream), A, I, h1, h2, O);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"

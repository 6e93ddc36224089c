// train_update.hip -- the optimizer step of the graph-replayed training iteration (bayesflow_nddms_amd/graph_trainer.py; the loop
// of basic_ddm_dc.py:199-202 with BayesFlow's defaults: Adam, global-norm clipping, cosine learning-rate decay) on FLAT buffers:
// every parameter is a view of one array, and so are the gradients and Adam's two moments.  In PyTorch the step is ~20 launches
// (learning rate: 5, clip_grad_norm_: 6, the fused multi-tensor Adam: 4 of 7-23 microseconds over 130 tensors, counters and the
// loss history: 4); here it is two: squared-norm partial sums, and the update (whose last workgroup advances the counters).  Everything is deterministic (fixed
// summation order).  gfx950 only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace nddm_update {

constexpr int NB = 256, NT = 256;      // blocks of the partial sums = values the update kernel reduces again

__global__ __launch_bounds__(NT) void sqnorm_partial_kernel(const float4 *g, long long n4, float *partial)
{
    __shared__ float red[NT];
    float a = 0.0f;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)NB * NT) {
        const float4 v = g[i];
        a = fmaf(v.x, v.x, a); a = fmaf(v.y, v.y, a); a = fmaf(v.z, v.z, a); a = fmaf(v.w, v.w, a);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct Hyper { float grad_scale, clip, lr0, total_steps, beta1, beta2, eps; };

// learning rate of the step that is about to run: 0.5 lr0 (1 + cos(pi min(step, total) / total)) -- past `total` steps the rate
// HOLDS the schedule's final value, as Keras' CosineDecay (BayesFlow's default schedule) does; it does not climb back
__device__ __forceinline__ float cosine_lr(const Hyper &H, float step_f)
{
    const float total = fmaxf(1.0f, H.total_steps);
    return 0.5f * H.lr0 * (1.0f + cosf(fminf(step_f, total) * (float)(M_PI / (double)total)));
}

// beta^n for an integer step count by repeated squaring (~2 log2 n double multiplications: the library's pow() is several hundred
// instructions, twice, on one thread of every workgroup while the other 255 wait)
__device__ __forceinline__ double ipow(double b, long long n)
{
    double r = 1.0;
    for (; n > 0; n >>= 1) { if (n & 1) r *= b; b *= b; }
    return r;
}

// (the counters of the step -- loss into the history, the rate that was used, step_i / step_f -- are advanced by the LAST workgroup
// to finish: every workgroup has read them by then; `ticket` is a device word that is zero between launches)
__global__ __launch_bounds__(NT) void adam_kernel(float4 *p, const float4 *g, float4 *m, float4 *v, long long n4, const float *partial,
                                                  Hyper H, long long *step_i, float *step_f, float *lr_out, float *loss_buf, int loss_cap,
                                                  float *loss_slot, unsigned int *ticket)
{
    __shared__ float red[NB];
    __shared__ float sh[4];
    red[threadIdx.x] = partial[threadIdx.x];
    __syncthreads();
    for (int s = NB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float gn = H.grad_scale * sqrtf(red[0]);                      // the norm of the (rank-averaged) gradient
        const float coef = fminf(H.clip / (gn + 1e-6f), 1.0f);              // torch.nn.utils.clip_grad_norm_
        const double bc1 = 1.0 - ipow((double)H.beta1, *step_i + 1), bc2 = 1.0 - ipow((double)H.beta2, *step_i + 1);
        sh[0] = H.grad_scale * coef;
        sh[1] = (float)((double)cosine_lr(H, *step_f) / bc1);               // step size
        sh[2] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float gmul = sh[0], step_size = sh[1], bc2s = sh[2], b1 = H.beta1, b2 = H.beta2, eps = H.eps;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        float4 pv = p[i], mv = m[i], vv = v[i];
        const float4 gv = g[i];
        const float gg[4] = {gv.x * gmul, gv.y * gmul, gv.z * gmul, gv.w * gmul};
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, ww[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = fmaf(gg[k] - mm[k], 1.0f - b1, mm[k]);                  // lerp(m, g, 1 - beta1)
            ww[k] = fmaf(gg[k] * gg[k], 1.0f - b2, b2 * ww[k]);
            pp[k] -= step_size * mm[k] / (sqrtf(ww[k]) / bc2s + eps);
        }
        p[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        m[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        v[i] = make_float4(ww[0], ww[1], ww[2], ww[3]);
    }
    // every thread of this workgroup has read step_i / step_f (through thread 0, before the barrier above); the last workgroup to
    // get here advances them -- after all the others have passed the same point
    __syncthreads();
    if (threadIdx.x == 0) {
        // (relaxed: nothing this workgroup wrote is read inside the kernel, and its reads of the counters completed before the
        //  barrier; an acquire-release here is an L2 write-back and invalidate per workgroup -- measured: + 10 us on this kernel)
        const unsigned int done = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            const long long s = *step_i;
            *loss_slot *= H.grad_scale;
            if (loss_cap > 0) loss_buf[s % loss_cap] = *loss_slot;      // a RING: the host drains it before it wraps (GraphTrainer._drain_losses)
            *lr_out = cosine_lr(H, *step_f);
            *step_i = s + 1;
            *step_f += 1.0f;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// dst[0] = a, dst[1] = b: the batch-shared number of trials and its logarithm (basic_ddm_dc.py:151-155) in ONE launch
__global__ void set2_kernel(float *dst, float a, float b) { if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; } }

// The head of a pipelined training iteration in ONE launch: the produced batch (parameter rows, trials) into the tensors the
// training graph reads, and the batch's N and log N into their device scalars.  n_a, n_b: counts of float4.
__global__ __launch_bounds__(256) void stage_kernel(float4 *dst_a, const float4 *src_a, long long n_a, float4 *dst_b, const float4 *src_b,
                                                    long long n_b, float *n2, float nv, float lognv)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_b) dst_b[i] = src_b[i];
    if (i < n_a) dst_a[i] = src_a[i];
    if (i == 0 && n2) { n2[0] = nv; n2[1] = lognv; }
}

}  // namespace nddm_update

using namespace nddm_update;

/* One optimizer step on flat buffers of n floats (n a multiple of 4, 16-byte aligned): p -= Adam(clip(grad_scale * g)) with the
 * cosine learning rate of step min(*step_f, total_steps); then loss_buf[*step_i mod loss_cap] = grad_scale * *loss_slot (written
 * back to the slot too; the history is a ring the caller reads out before it wraps), *lr_out = the rate used, and both counters advance.  partial: 256 floats of scratch + one word, partial[256], that must be ZERO at the first launch (the kernel leaves it zero).  Adam's step count is *step_i + 1. */
extern "C" int nddm_train_adam_step(float *p, const float *g, float *m, float *v, long long n, float *partial, float grad_scale, float clip,
                                    float lr0, float total_steps, float beta1, float beta2, float eps, long long *step_i, float *step_f,
                                    float *lr_out, float *loss_buf, int loss_cap, float *loss_slot, void *stream)
{
    // partial: 256 floats of scratch + ONE word (index 256) that is zero between launches: the workgroups' completion ticket
    if (n <= 0 || (n & 3) || !p || !g || !m || !v || !partial || !step_i || !step_f || !lr_out || !loss_slot) return 1;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
        return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Hyper H = {grad_scale, clip, lr0, total_steps, beta1, beta2, eps};
    const long long n4 = n / 4;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(NB), dim3(NT), 0, st, reinterpret_cast<const float4 *>(g), n4, partial);
    const int blocks = (int)((n4 + NT - 1) / NT < 1024 ? (n4 + NT - 1) / NT : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(NT), 0, st, reinterpret_cast<float4 *>(p), reinterpret_cast<const float4 *>(g),
                       reinterpret_cast<float4 *>(m), reinterpret_cast<float4 *>(v), n4, partial, H, step_i, step_f, lr_out, loss_buf, loss_cap,
                       loss_slot, reinterpret_cast<unsigned int *>(partial + NB));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int nddm_train_set2(float *dst, float a, float b, void *stream)
{
    if (!dst) return 1;
    hipLaunchKernelGGL(set2_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dst, a, b);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

/* dst_a[0 .. n_a) = src_a, dst_b[0 .. n_b) = src_b (floats; counts multiples of 4, pointers 16-byte aligned), n2[0] = nv, n2[1] = lognv
 * (n2 NULL: not written): one launch. */
extern "C" int nddm_train_stage(float *dst_a, const float *src_a, long long n_a, float *dst_b, const float *src_b, long long n_b, float *n2,
                                float nv, float lognv, void *stream)
{
    if (n_a < 0 || n_b < 0 || (n_a & 3) || (n_b & 3) || (n_a && (!dst_a || !src_a)) || (n_b && (!dst_b || !src_b))) return 1;
    if ((reinterpret_cast<uintptr_t>(dst_a) | reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(dst_b) | reinterpret_cast<uintptr_t>(src_b)) & 15)
        return 1;
    const long long n = (n_a > n_b ? n_a : n_b) / 4;
    hipLaunchKernelGGL(stage_kernel, dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<float4 *>(dst_a), reinterpret_cast<const float4 *>(src_a), n_a / 4, reinterpret_cast<float4 *>(dst_b),
                       reinterpret_cast<const float4 *>(src_b), n_b / 4, n2, nv, lognv);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

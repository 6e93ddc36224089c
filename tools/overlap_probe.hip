// overlap_probe.hip -- stand-in for a collective's kernel on ONE GPU: `wgs` workgroups of `threads` threads that stream
// `bytes_per_wg` bytes each from src to dst over and over until `hold_ticks` (100 MHz s_memrealtime ticks) have passed, the way
// an all-gather kernel holds its channels' workgroups for as long as the links need (the copy rate of the real thing is set by
// xGMI, ~30 ms for 16.8 GB; here by the hold time).  Workgroup 0 records its start and end ticks.  Used by tools/overlap_probe.py
// to measure (a) whether such a kernel gets placed while the simulator's persistent grid is resident and (b) what it costs the
// simulator.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void hold_kernel(const float4 *src, float4 *dst, long long n4_per_wg, unsigned long long hold_ticks,
                            unsigned long long *stamps)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const float4 *s = src + (long long)blockIdx.x * n4_per_wg;
    float4 *d = dst + (long long)blockIdx.x * n4_per_wg;
    unsigned long long now = t0;
    do {
        for (long long i = threadIdx.x; i < n4_per_wg; i += blockDim.x) d[i] = s[i];
        now = __builtin_amdgcn_s_memrealtime();
    } while (now - t0 < hold_ticks);
    if (threadIdx.x == 0) { atomicMin(stamps, t0); atomicMax(stamps + 1, now); }
}

extern "C" int overlap_probe_launch(const void *src, void *dst, long long bytes_per_wg, int wgs, int threads,
                                    unsigned long long hold_ticks, void *stamps, void *stream)
{
    hipLaunchKernelGGL(hold_kernel, dim3(wgs), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const float4 *>(src), static_cast<float4 *>(dst), bytes_per_wg / 16, hold_ticks,
                       static_cast<unsigned long long *>(stamps));
    return (int)hipGetLastError();
}

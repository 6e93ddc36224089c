// ubench_salu.hip -- does scalar-ALU work compete with the VALU issue of the DDM step loop?  The loop carries 53 SALU
// instructions per 65 VALU (exec-mask handling, the refill test), the refill path is mostly scalar.  Measures, per SIMD and
// at 1/2/4/8 waves per SIMD: the issue cost of a SALU instruction alone, and of 16 full-rate VALU instructions with 0 / 8 /
// 16 / 32 SALU instructions interleaved (independent chains).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_salu tools/ubench_salu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 16384;

#define V16 \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(c)); \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(c));
#define S8 \
    asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) :: "scc"); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s1) :: "scc"); \
    asm volatile("s_add_u32 %0, %0, 7" : "+s"(s2) :: "scc"); asm volatile("s_add_u32 %0, %0, 9" : "+s"(s3) :: "scc"); \
    asm volatile("s_add_u32 %0, %0, 3" : "+s"(s4) :: "scc"); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s5) :: "scc"); \
    asm volatile("s_add_u32 %0, %0, 7" : "+s"(s6) :: "scc"); asm volatile("s_add_u32 %0, %0, 9" : "+s"(s7) :: "scc");

#define DEF(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, uint64_t *clk)                                \
    {                                                                                                     \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = 1.0001f; \
        uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7; \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();               \
        for (int i = 0; i < ITERS; ++i) { BODY }                                                          \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7); \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                  \
    }
DEF(k_s16, S8 S8)
DEF(k_v16, V16)
DEF(k_v16_s8, V16 S8)
DEF(k_v16_s16, V16 S8 S8)
DEF(k_v16_s32, V16 S8 S8 S8 S8)

struct Entry { const char *name; void (*k)(float *, uint64_t *); int per_trip; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: CUs=%d\n", cus);
    float *out; uint64_t *clk;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(float)));
    CHECK(hipMalloc(&clk, 16));
    Entry es[] = {{"16 s_add_u32", k_s16, 16}, {"16 v_add_f32", k_v16, 16}, {"16 v_add_f32 + 8 s_add_u32", k_v16_s8, 16},
                  {"16 v_add_f32 + 16 s_add_u32", k_v16_s16, 16}, {"16 v_add_f32 + 32 s_add_u32", k_v16_s32, 16}};
    printf("%-32s      1       2       4       8   waves/SIMD: SIMD cycles per loop trip / 16\n", "loop body");
    const int wpc_list[] = {4, 8, 16, 32};
    for (auto &e : es) {
        printf("%-32s", e.name);
        for (int wpc : wpc_list) {
            const int blocks = cus * wpc / 4;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / (double)h[1] * 0.1;
            const double per = (ms * 1e-3) * ghz * 1e9 / ((double)(wpc / 4) * ITERS * e.per_trip);
            printf("  %6.2f", per);
        }
        printf("\n");
    }
    return 0;
}

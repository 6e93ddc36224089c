// ubench_bank.hip -- does the VGPR bank (register index mod 4) of the source operands change the issue cost of the
// three-operand instructions of the Philox rounds (v_bitop3_b32, v_mad_u64_u32) on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_bank tools/ubench_bank.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 16384;

// 8 independent chains v20..v27; sources from v1..v12.  S1/S2: source register numbers of chain 0 (chain i uses +i)
#define X3(D, A, B) "v_bitop3_b32 v" #D ", v" #D ", v" #A ", v" #B " bitop3:0x96\n"
#define BODY_DISTINCT X3(20, 1, 2) X3(21, 2, 3) X3(22, 3, 4) X3(23, 4, 5) X3(24, 5, 6) X3(25, 6, 7) X3(26, 7, 8) X3(27, 8, 9)        /* banks d, d+1, d+2 */
#define BODY_TWO_SAME X3(20, 4, 1) X3(21, 5, 2) X3(22, 6, 3) X3(23, 7, 4) X3(24, 8, 1) X3(25, 9, 2) X3(26, 10, 3) X3(27, 11, 4)     /* dst bank == src1 bank */
#define BODY_ALL_SAME X3(20, 4, 8) X3(21, 5, 9) X3(22, 6, 10) X3(23, 7, 11) X3(24, 8, 12) X3(25, 9, 1) X3(26, 10, 2) X3(27, 11, 3)   /* all three in one bank (first 5) */
#define SRC_SAME X3(20, 1, 5) X3(21, 2, 6) X3(22, 3, 7) X3(23, 4, 8) X3(24, 5, 9) X3(25, 6, 10) X3(26, 7, 11) X3(27, 8, 12)          /* src1 bank == src2 bank != dst */
#define M64(D, A) "v_mad_u64_u32 v[" #D ":" #D "+1], vcc, v" #A ", v13, 0\n"
#define MAD_DIFF "v_mad_u64_u32 v[20:21], vcc, v2, v13, 0\nv_mad_u64_u32 v[22:23], vcc, v3, v13, 0\nv_mad_u64_u32 v[24:25], vcc, v4, v13, 0\nv_mad_u64_u32 v[26:27], vcc, v6, v13, 0\n" \
                 "v_mad_u64_u32 v[28:29], vcc, v7, v13, 0\nv_mad_u64_u32 v[30:31], vcc, v8, v13, 0\nv_mad_u64_u32 v[32:33], vcc, v10, v13, 0\nv_mad_u64_u32 v[34:35], vcc, v11, v13, 0\n"   /* v13 bank 1; a in banks 2,3,0 */
#define MAD_SAME "v_mad_u64_u32 v[20:21], vcc, v1, v13, 0\nv_mad_u64_u32 v[22:23], vcc, v5, v13, 0\nv_mad_u64_u32 v[24:25], vcc, v9, v13, 0\nv_mad_u64_u32 v[26:27], vcc, v1, v13, 0\n" \
                 "v_mad_u64_u32 v[28:29], vcc, v5, v13, 0\nv_mad_u64_u32 v[30:31], vcc, v9, v13, 0\nv_mad_u64_u32 v[32:33], vcc, v1, v13, 0\nv_mad_u64_u32 v[34:35], vcc, v5, v13, 0\n"     /* a in bank 1 like v13 */
#define CLOB "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","vcc"

#define DEF(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, uint64_t *clk)                                \
    {                                                                                                     \
        asm volatile("v_mov_b32 v1, 1\nv_mov_b32 v2, 2\nv_mov_b32 v3, 3\nv_mov_b32 v4, 4\nv_mov_b32 v5, 5\nv_mov_b32 v6, 6\nv_mov_b32 v7, 7\n" \
                     "v_mov_b32 v8, 8\nv_mov_b32 v9, 9\nv_mov_b32 v10, 10\nv_mov_b32 v11, 11\nv_mov_b32 v12, 12\nv_mov_b32 v13, 0x12345\n"        \
                     "v_mov_b32 v20, 1\nv_mov_b32 v21, 1\nv_mov_b32 v22, 1\nv_mov_b32 v23, 1\nv_mov_b32 v24, 1\nv_mov_b32 v25, 1\nv_mov_b32 v26, 1\nv_mov_b32 v27, 1\n" ::: CLOB); \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                       \
        for (int i = 0; i < ITERS; ++i) asm volatile(BODY BODY ::: CLOB);                                 \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                       \
        uint32_t r; asm volatile("v_mov_b32 %0, v20" : "=v"(r) :: CLOB);                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)r;                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                        \
    }
DEF(k_distinct, BODY_DISTINCT)
DEF(k_two_same, BODY_TWO_SAME)
DEF(k_all_same, BODY_ALL_SAME)
DEF(k_src_same, SRC_SAME)
DEF(k_mad_diff, MAD_DIFF)
DEF(k_mad_same, MAD_SAME)

int main()
{
    float *out; uint64_t *clk, h;
    CHECK(hipMalloc(&out, 256 * 8192 * sizeof(float))); CHECK(hipMalloc(&clk, 16));
    struct { const char *name; void (*k)(float *, uint64_t *); } ks[] = {
        {"v_bitop3: dst/src0, src1, src2 in three banks", k_distinct}, {"v_bitop3: src1 in the bank of dst/src0", k_two_same},
        {"v_bitop3: all operands in one bank", k_all_same}, {"v_bitop3: src1 and src2 share a bank", k_src_same},
        {"v_mad_u64_u32: a and b in different banks", k_mad_diff}, {"v_mad_u64_u32: a and b in one bank", k_mad_same}};
    printf("instr / operand banks                              cycles (at 2.4 GHz) per wave64 instruction per SIMD, 8 waves/SIMD\n");
    for (auto &e : ks) {
        const int blocks = 256 * 8;                     // 8 workgroups of 4 waves per CU = 8 waves per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, clk);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        // every SIMD holds 8 waves, each issuing ITERS * 16 instructions: wall time x nominal clock / instructions per SIMD
        printf("%-50s %.2f   (wall %.3f ms; one wave's own s_memtime span: %.2f ticks per instruction)\n", e.name,
               ms * 1e-3 * 2.4e9 / (ITERS * 16.0 * 8.0), ms, (double)h / (ITERS * 16.0));
    }
    return 0;
}

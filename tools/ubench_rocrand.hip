// ubench_rocrand.hip -- an OUTSIDE yardstick for the simulator's VALU ceiling (measurement only: never linked into the product).
// north_star names "hiprandStatePhilox per lane"; this is the vendor's device API doing exactly that on this chip:
// one rocrand_state_philox4x32_10 per lane (hipRAND's hiprandStatePhilox4_32_10_t is this type on ROCm), in a loop
//   mode 0  rocrand4()          raw Philox4x32-10, 4 x u32 per call
//   mode 1  rocrand_normal4()   Philox4x32-10 + the vendor's Box-Muller, 4 normals per call
//   mode 2  rocrand_normal4() + one Euler-Maruyama step per normal (fma, add, |w| < h compare-and-count): the work of the
//           simulator's step loop with the vendor's generator in it
// at 1 / 2 / 4 / 8 waves per SIMD (256-thread workgroups, W per CU).  Prints G values per second (u32 words or normals) for the
// whole chip.  Compare: the simulator's step loop sustains 2.5e12 normals-and-steps per second in lockstep (bench.py's
// roofline_valu.ceiling) and 2.37e12 on the headline workload with refills, flushes and idle lanes included.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_rocrand tools/ubench_rocrand.hip   (add -ffast-math for the second table)
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void gen(unsigned int *sink, int iters, unsigned long long seed, float mu, float h)
{
    const unsigned int gid = blockIdx.x * blockDim.x + threadIdx.x;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, gid, 0, &st);
    unsigned int acc = 0;
    float w = 0.0f;
    int k = 0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
            const uint4 u = rocrand4(&st);
            acc ^= u.x ^ u.y ^ u.z ^ u.w;
        } else {
            const float4 z = rocrand_normal4(&st);
            if constexpr (MODE == 1) {
                acc ^= __float_as_uint(z.x) ^ __float_as_uint(z.y) ^ __float_as_uint(z.z) ^ __float_as_uint(z.w);
            } else {
                // evidence += drift*dt + sqrt(dt)*dc*N(0,1); count the steps inside (0, boundary)  (basic_ddm_dc.py:95-101)
                w = fmaf(z.x, 0.03f, w) + mu; k += fabsf(w) < h ? 1 : 0;
                w = fmaf(z.y, 0.03f, w) + mu; k += fabsf(w) < h ? 1 : 0;
                w = fmaf(z.z, 0.03f, w) + mu; k += fabsf(w) < h ? 1 : 0;
                w = fmaf(z.w, 0.03f, w) + mu; k += fabsf(w) < h ? 1 : 0;
            }
        }
    }
    if constexpr (MODE == 2) acc = (unsigned int)k ^ __float_as_uint(w);
    if (acc == 0x12345678u) sink[0] = acc;          // keeps the loop alive; practically never true
}

template <int MODE>
static int run(const char *what, int cus, unsigned int *sink)
{
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(gen<MODE>)));
    printf("%-58s (%d VGPRs)\n", what, fa.numRegs);
    for (int W : {1, 2, 4, 8}) {
        const int iters = 16384 / W * (MODE == 0 ? 2 : 1);
        const int blocks = W * cus;                       // 256 threads = one wave per SIMD of a CU; W workgroups per CU
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(gen<MODE>, dim3(blocks), dim3(256), 0, 0, sink, iters, 2023ull + rep, 1e-6f, 1e30f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;          // first launch: code upload
        }
        const double values = (double)blocks * 256.0 * iters * 4.0;
        printf("  %d wave%s per SIMD: %8.3f ms  %9.1f G %s/s  = %6.1f SIMD cycles per wave per 4 values at 2.4 GHz\n", W, W > 1 ? "s" : " ",
               best, values / (best * 1e-3) / 1e9, MODE == 0 ? "u32" : "normals",
               best * 1e-3 * 2.4e9 * (4.0 * cus) / ((double)blocks * 4.0 * iters));
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s  CUs=%d  clock=%d MHz  rocRAND %d\n", prop.name, cus, prop.clockRate / 1000, ROCRAND_VERSION);
#ifdef __FAST_MATH__
    printf("compiled with -ffast-math\n");
#else
    printf("compiled without -ffast-math\n");
#endif
    unsigned int *sink;
    CHECK(hipMalloc(&sink, 64));
    if (run<0>("rocrand4: raw Philox4x32-10 words", cus, sink)) return 1;
    if (run<1>("rocrand_normal4: Philox4x32-10 + vendor Box-Muller", cus, sink)) return 1;
    if (run<2>("rocrand_normal4 + fma/add/compare-and-count per normal", cus, sink)) return 1;
    CHECK(hipFree(sink));
    return 0;
}

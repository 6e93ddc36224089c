// Probe behind the workaround at csrc/train_deepset.hip (struct RowTile: "dword accesses only"): what does
// __builtin_amdgcn_raw_buffer_load_b128 compile to, and what does it return?
//
// Finding (hipcc 7.2.26015, clang 22.0.0git roc-7.2.0, --offload-arch=gfx950): the builtin is compiled to ONE buffer_load_dword instead of a
// buffer_load_dwordx4: only the x component is loaded; y / z / w come back as COPIES OF x instead of their dwords of the source (MI355X:
// got 1 1 1 1 5 5 5 5 for a source 1 2 3 4 5 6 7 8 -- the probe prints the first values it got beside the source).  Checkable WITHOUT a GPU from the ISA:
//     hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o - tools/probe_buffer_load_b128.hip | grep buffer_load
// (tests/test_kernel_resources.py::test_toolchain_probes_compile asserts exactly that, so a toolchain that fixes it makes the test --
// and with it the workaround -- stand out), and at run time on the MI355X:
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/probe_b128 tools/probe_buffer_load_b128.hip && /tmp/probe_b128
// prints, per tile height, how many of the 4096 floats copied through a b128 load differ from the source (profiles/r6_probe_buffer_load_b128.txt).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// a [rows, 64] float tile as a raw buffer (range-checked: rows beyond `rows` read as zero), copied with 16-byte loads
__global__ void copy_b128(const float *src, float *dst, int rows)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, rows * 64 * 4, 0x00020000);
    for (int kk = 0; kk < 4; ++kk) {
        const int p = threadIdx.x + 256 * kk, row = p >> 4, c4 = p & 15;
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (row * 64 + 4 * c4) * 4, 0, 0);
        float *o = dst + row * 64 + 4 * c4;
        o[0] = __builtin_bit_cast(float, v.x); o[1] = __builtin_bit_cast(float, v.y);
        o[2] = __builtin_bit_cast(float, v.z); o[3] = __builtin_bit_cast(float, v.w);
    }
}

// the same copy with dword loads: what the product uses
__global__ void copy_b32(const float *src, float *dst, int rows)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, rows * 64 * 4, 0x00020000);
    for (int kk = 0; kk < 16; ++kk) {
        const int p = threadIdx.x + 256 * kk;
        dst[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, p * 4, 0, 0));
    }
}

int main()
{
    static float h[64 * 64], g[64 * 64];
    float *s, *d;
    int rt = 0;
    (void)hipRuntimeGetVersion(&rt);
    printf("# HIP runtime %d.%d.%d, compiled by %s (HIP_VERSION %d.%d.%d)\n", rt / 10000000, rt / 100000 % 100, rt % 100000,
           __clang_version__, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH);
    if (hipMalloc(&s, sizeof h) != hipSuccess || hipMalloc(&d, sizeof h) != hipSuccess) { printf("no device memory\n"); return 2; }
    for (int i = 0; i < 64 * 64; ++i) h[i] = (float)(i + 1);
    (void)hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
    int worst = 0;
    for (int form = 0; form < 2; ++form)
        for (int rows : {64, 44}) {
            (void)hipMemset(d, 0xff, sizeof h);
            if (form == 0) hipLaunchKernelGGL(copy_b128, dim3(1), dim3(256), 0, 0, s, d, rows);
            else hipLaunchKernelGGL(copy_b32, dim3(1), dim3(256), 0, 0, s, d, rows);
            (void)hipMemcpy(g, d, sizeof h, hipMemcpyDeviceToHost);
            int bad = 0, first = -1;
            for (int i = 0; i < 64 * 64; ++i) {
                const float want = i / 64 < rows ? h[i] : 0.0f;
                if (g[i] != want) { if (first < 0) first = i; ++bad; }
            }
            printf("%s, %d valid rows of 64: %d of 4096 floats differ from the source", form == 0 ? "raw_buffer_load_b128" : "raw_buffer_load_b32 ", rows, bad);
            if (first >= 0) {
                printf(" (first at %d: got %g, source %g)\n    first 8 values got   :", first, g[first], first / 64 < rows ? h[first] : 0.0f);
                for (int i = 0; i < 8; ++i) printf(" %g", g[i]);
                printf("\n    first 8 of the source:");
                for (int i = 0; i < 8; ++i) printf(" %g", h[i]);
            }
            printf("\n");
            if (form == 0 && bad > worst) worst = bad;
        }
    printf(worst ? "=> the b128 builtin does NOT load 16 bytes with this toolchain: dword loads in the product (csrc/train_deepset.hip RowTile)\n"
                 : "=> the b128 builtin loads 16 bytes: the workaround in csrc/train_deepset.hip RowTile can go\n");
    return 0;
}

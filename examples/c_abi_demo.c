/*
 * c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no PyTorch, only include/nddm.h and the HIP
 * runtime API for memory.  It simulates B parameter sets x N trials of basic_ddm_dc (the path of
 * basic_ddm_dc.py:85-125 in the reference) and writes params, trials and summaries as raw float32 so that
 * tests/test_gpu_c_abi.py can compare them with the CPU oracle bit for bit.
 *
 * Build (done by __graft_entry__.build()):
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo \
 *       -Lbayesflow_nddms_amd -lnddm_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../bayesflow_nddms_amd' -Wl,-rpath,/opt/rocm/lib
 * Run:  examples/c_abi_demo B N dt max_steps seed flags out.bin
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "nddm.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define NDDM_CHECK(x) do { int rc_ = (x); if (rc_ != NDDM_OK) { fprintf(stderr, "nddm status %d: %s\n", rc_, nddm_last_error()); return 3; } } while (0)

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s B N dt max_steps seed flags out.bin\n", argv[0]); return 1; }
    const int64_t B = atoll(argv[1]);
    const int32_t N = atoi(argv[2]);
    const float dt = (float)atof(argv[3]);
    const int32_t max_steps = atoi(argv[4]);
    const uint64_t seed = strtoull(argv[5], NULL, 10);
    const uint32_t flags = (uint32_t)strtoul(argv[6], NULL, 10);
    if (nddm_abi_version() != NDDM_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    int n_dev = 0;
    if (nddm_device_count(&n_dev) != NDDM_OK || n_dev < 1) { fprintf(stderr, "no device: %s\n", nddm_last_error()); return 1; }
    NDDM_CHECK(nddm_set_device(0));

    /* parameter rows in the reference's order (basic_ddm_dc.py:118): drift, boundary, beta, tau, dc */
    const size_t np = (size_t)B * 5, nt = (size_t)B * N * 2, ns = (size_t)B * NDDM_SUMMARY_K;
    float *p = (float *)malloc(np * sizeof(float));
    for (int64_t i = 0; i < B; i++) {
        p[5 * i + 0] = -2.0f + 0.25f * (float)(i % 17);
        p[5 * i + 1] = 0.8f + 0.2f * (float)(i % 7);
        p[5 * i + 2] = 0.3f + 0.1f * (float)(i % 5);
        p[5 * i + 3] = 0.2f + 0.05f * (float)(i % 3);
        p[5 * i + 4] = 0.8f + 0.1f * (float)(i % 4);
    }
    float *d_p, *d_t, *d_s;
    HIP_OK(hipMalloc((void **)&d_p, np * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_t, nt * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_s, ns * sizeof(float)));
    HIP_OK(hipMemcpy(d_p, p, np * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    NDDM_CHECK(nddm_basic_ddm_dc_simulate(d_p, B, N, dt, max_steps, seed, 0, flags, d_t, d_s, (void *)st));   /* warm-up */
    HIP_OK(hipStreamSynchronize(st));
    const double t0 = now_s();
    NDDM_CHECK(nddm_basic_ddm_dc_simulate(d_p, B, N, dt, max_steps, seed, 0, flags, d_t, d_s, (void *)st));
    HIP_OK(hipStreamSynchronize(st));
    const double t1 = now_s();

    float *t = (float *)malloc(nt * sizeof(float)), *s = (float *)malloc(ns * sizeof(float));
    HIP_OK(hipMemcpy(t, d_t, nt * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(s, d_s, ns * sizeof(float), hipMemcpyDeviceToHost));
    FILE *f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 1; }
    fwrite(p, sizeof(float), np, f); fwrite(t, sizeof(float), nt, f); fwrite(s, sizeof(float), ns, f);
    fclose(f);

    /* error convention: a bad shape is a status code + message, not a crash */
    int rc = nddm_basic_ddm_dc_simulate(d_p, B, 0, dt, max_steps, seed, 0, flags, d_t, d_s, (void *)st);

    /* ABI 3 from C: the same launch captured into a hipGraph under a GRAPH ARENA (the owner of the memory the library pins
     * behind a captured launch), replayed into zeroed outputs, compared with the direct launch; then the graph is destroyed and
     * the arena -- and only the arena -- released. */
    uint64_t arena = 0, prev = 0, bytes = 0;
    int32_t n_alloc = 0;
    hipGraph_t graph;
    hipGraphExec_t exec;
    NDDM_CHECK(nddm_graph_arena_create(&arena));
    NDDM_CHECK(nddm_graph_arena_bind(arena, &prev));
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    NDDM_CHECK(nddm_basic_ddm_dc_simulate(d_p, B, N, dt, max_steps, seed, 0, flags, d_t, d_s, (void *)st));
    HIP_OK(hipStreamEndCapture(st, &graph));
    NDDM_CHECK(nddm_graph_arena_bind(prev, NULL));
    NDDM_CHECK(nddm_graph_arena_info(arena, &bytes, &n_alloc));
    HIP_OK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
    int replay_equal = 1;
    float *t2 = (float *)malloc(nt * sizeof(float));
    for (int rep = 0; rep < 3; rep++) {
        HIP_OK(hipMemsetAsync(d_t, 0, nt * sizeof(float), st));
        HIP_OK(hipGraphLaunch(exec, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(t2, d_t, nt * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nt; i++) if (((uint32_t *)t2)[i] != ((uint32_t *)t)[i]) { replay_equal = 0; break; }
    }
    free(t2);
    HIP_OK(hipGraphExecDestroy(exec));
    HIP_OK(hipGraphDestroy(graph));
    NDDM_CHECK(nddm_graph_arena_release(arena));
    const int released_twice = nddm_graph_arena_release(arena);           /* NDDM_ERR_PARAM: the handle is gone */

    /* a stream handle the runtime does not know is refused (HIP itself would dereference it) */
    hipStream_t dead;
    HIP_OK(hipStreamCreate(&dead));
    HIP_OK(hipStreamDestroy(dead));
    const int dead_stream_status = nddm_basic_ddm_dc_simulate(d_p, B, N, dt, max_steps, seed, 0, flags, d_t, d_s, (void *)dead);

    /* ABI 4 from C: (i) the same launch with NDDM_STATE_F64 -- the reference's float64 state arithmetic; (ii) the reference's exact
     * first-passage sampler, nddm_simulratcliff (no dt): parameter rows Nu, Alpha, Beta, Tau, Eta, Varsigma.  Both are appended to
     * the output file (trials of (i), then parameter rows, trials and summaries of (ii)) for the test to compare with the oracle. */
    NDDM_CHECK(nddm_basic_ddm_dc_simulate(d_p, B, N, dt, max_steps, seed, 0, flags | NDDM_STATE_F64, d_t, d_s, (void *)st));
    HIP_OK(hipStreamSynchronize(st));
    float *t64 = (float *)malloc(nt * sizeof(float));
    HIP_OK(hipMemcpy(t64, d_t, nt * sizeof(float), hipMemcpyDeviceToHost));
    const size_t nq = (size_t)B * 6;
    float *q = (float *)malloc(nq * sizeof(float)), *d_q;
    for (int64_t i = 0; i < B; i++) {
        q[6 * i + 0] = -4.0f + 0.5f * (float)(i % 17);     /* Nu */
        q[6 * i + 1] = 0.8f + 0.1f * (float)(i % 7);       /* Alpha */
        q[6 * i + 2] = 0.3f + 0.1f * (float)(i % 5);       /* Beta */
        q[6 * i + 3] = 0.15f + 0.05f * (float)(i % 9);     /* Tau */
        q[6 * i + 4] = 0.25f * (float)(i % 8);             /* Eta */
        q[6 * i + 5] = 0.8f + 0.1f * (float)(i % 6);       /* Varsigma */
    }
    HIP_OK(hipMalloc((void **)&d_q, nq * sizeof(float)));
    HIP_OK(hipMemcpy(d_q, q, nq * sizeof(float), hipMemcpyHostToDevice));
    NDDM_CHECK(nddm_simulratcliff(d_q, B, N, seed, 0, flags & NDDM_GAUSS_FAST, 0.0f, 0, d_t, d_s, NULL, (void *)st));
    HIP_OK(hipStreamSynchronize(st));
    float *tr = (float *)malloc(nt * sizeof(float)), *sr = (float *)malloc(ns * sizeof(float));
    HIP_OK(hipMemcpy(tr, d_t, nt * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(sr, d_s, ns * sizeof(float), hipMemcpyDeviceToHost));
    const int bad_flag_status = nddm_simulratcliff(d_q, B, N, seed, 0, NDDM_BRIDGE, 0.0f, 0, d_t, d_s, NULL, (void *)st);   /* no dt, no bridge */
    f = fopen(argv[7], "ab");
    if (!f) { perror(argv[7]); return 1; }
    fwrite(t64, sizeof(float), nt, f); fwrite(q, sizeof(float), nq, f); fwrite(tr, sizeof(float), nt, f); fwrite(sr, sizeof(float), ns, f);
    fclose(f);
    printf("{\"build\": \"%s\", \"ratcliff_bad_flag_status\": %d}\n", nddm_build_info(), bad_flag_status);
    HIP_OK(hipFree(d_q)); free(t64); free(q); free(tr); free(sr);

    printf("{\"sets\": %lld, \"n_trials\": %d, \"seconds\": %.6f, \"trials_per_s\": %.4e, \"first_rt\": %.6f, \"first_choice\": %.0f, "
           "\"bad_shape_status\": %d, \"graph_replay_equal\": %d, \"arena_bytes\": %llu, \"arena_allocations\": %d, "
           "\"arena_released_twice_status\": %d, \"dead_stream_status\": %d}\n", (long long)B, N, t1 - t0, (double)B * N / (t1 - t0), t[0], t[1], rc,
           replay_equal, (unsigned long long)bytes, (int)n_alloc, released_twice, dead_stream_status);
    HIP_OK(hipFree(d_p)); HIP_OK(hipFree(d_t)); HIP_OK(hipFree(d_s)); HIP_OK(hipStreamDestroy(st));
    free(p); free(t); free(s);
    return 0;
}

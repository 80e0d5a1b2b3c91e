// tools/mfma_peak2.hip -- fp32 MFMA rate vs waves per SIMD, accumulators per wave, s_setprio and interleaved LDS reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int PRIO, int LDSREADS>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ a, const float* __restrict__ b, float* out, int iters)
{
    __shared__ float4 lds[1024];
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (LDSREADS) { for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(a[i], a[i + 1], 0.f, 1.f); __syncthreads(); }
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = a[(size_t)tid * 8 + i]; bv[i] = b[(size_t)tid * 8 + i]; }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    float4 nxt = make_float4(0, 0, 0, 0);
    int ldsoff = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) {      // software-pipelined LDS reads: issue now, consume next iteration
            float4 cur = nxt;
#pragma unroll
            for (int r = 0; r < LDSREADS; ++r) { nxt = lds[(ldsoff + r * 37) & 1023]; }
            ldsoff += 3;
            av[0] += cur.z;     // cur.z == 0: keeps the dependency without changing the data
        }
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[(k + j) & 7], acc[j], 0, 0, 0);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    }
    float s = nxt.x;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 123.456f) out[tid] = s;
}

template <int NACC, int PRIO, int LDSREADS>
static void run(const char* tag, int waves_per_simd, const float* da, const float* db, float* dout)
{
    const int blocks = 256 * waves_per_simd;
    const int iters = 16000 / (NACC * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, PRIO, LDSREADS>), dim3(blocks), dim3(256), 0, 0, da, db, dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("%-14s waves/SIMD %d  acc %d  prio %d  ldsreads %d : %8.3f ms %7.1f TFLOP/s (%.1f%%)\n", tag, waves_per_simd, NACC, PRIO, LDSREADS,
           best, flop / best / 1e9, flop / best / 1e9 / 1.573);
}

int main()
{
    const int n = 256 * 8 * 256 * 8;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; h[i] = r > 0 ? r : 0.f; }
    float *da, *db, *dout;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dout, 256 * 8 * 256 * 4);
    (void)hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    (void)hipMemcpy(db, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) run<4, 0, 0>("relu-like", w, da, db, dout);
    for (int w = 1; w <= 4; ++w) run<2, 0, 0>("relu-like", w, da, db, dout);
    for (int w = 1; w <= 2; ++w) run<1, 0, 0>("relu-like", w, da, db, dout);
    for (int w = 1; w <= 2; ++w) run<8, 0, 0>("relu-like", w, da, db, dout);
    for (int w = 2; w <= 4; w += 2) run<4, 1, 0>("static prio1", w, da, db, dout);
    for (int w = 2; w <= 4; w += 2) run<4, 2, 0>("prio around", w, da, db, dout);
    run<4, 0, 2>("lds interleave", 1, da, db, dout);
    run<4, 0, 8>("lds interleave", 1, da, db, dout);
    run<4, 0, 8>("lds interleave", 2, da, db, dout);
    run<3, 0, 6>("lds interleave", 1, da, db, dout);
    (void)hipMemset(da, 0, n * 4);
    for (int w = 1; w <= 4; ++w) run<4, 0, 0>("zeros", w, da, db, dout);
    return 0;
}

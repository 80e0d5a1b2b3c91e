// tools/mfma_peak3.hip -- fp32 MFMA rate for the accumulator counts / register classes used by the conv kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MINW>
__global__ __launch_bounds__(256, MINW) void mfma_loop(const float* __restrict__ a, const float* __restrict__ b, float* out, int iters)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = a[(size_t)tid * 8 + i]; bv[i] = b[(size_t)tid * 8 + i]; }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[(k + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 123.456f) out[tid] = s;
}

template <int NACC, int MINW>
static void run(int waves_per_simd, int rounds, const float* da, const float* db, float* dout)
{
    const int blocks = 256 * waves_per_simd * rounds;
    const int iters = 24000 / (NACC * waves_per_simd * rounds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, MINW>), dim3(blocks), dim3(256), waves_per_simd == 1 ? 90 * 1024 : (waves_per_simd == 2 ? 60 * 1024 : 0), 0, da, db, dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("acc %d  launch_bounds(256,%d)  resident waves/SIMD %d  rounds %d : %8.3f ms %7.1f TFLOP/s (%.1f%%)\n", NACC, MINW, waves_per_simd, rounds,
           best, flop / best / 1e9, flop / best / 1e9 / 1.573);
}

int main()
{
    const int n = 256 * 16 * 256 * 8;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; h[i] = r > 0 ? r : 0.f; }
    float *da, *db, *dout;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dout, 256 * 16 * 256 * 4);
    (void)hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    (void)hipMemcpy(db, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)mfma_loop<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mfma_loop<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mfma_loop<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mfma_loop<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run<4, 2>(1, 1, da, db, dout);   // warm-up
    for (int w = 1; w <= 2; ++w) for (int r = 1; r <= 4; r += 3) run<3, 2>(w, r, da, db, dout);
    for (int w = 1; w <= 2; ++w) run<3, 1>(w, 1, da, db, dout);
    for (int w = 1; w <= 2; ++w) run<4, 2>(w, 1, da, db, dout);
    for (int w = 1; w <= 2; ++w) run<6, 2>(w, 1, da, db, dout);
    return 0;
}

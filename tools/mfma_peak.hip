// tools/mfma_peak.hip -- sustainable v_mfma_f32_32x32x2_f32 rate on this chip as a function of operand data
// (DVFS: the fp32 matrix pipe is power-managed; zero operands clock higher than dense random ones).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ a, const float* __restrict__ b, float* out, int iters)
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

int main()
{
    const int blocks = 256 * 4, threads = 256, n = blocks * threads * 8, iters = 4000;
    std::vector<float> h(n);
    float *da, *db, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"zeros", "ones", "uniform[-1,1)", "relu-like (half zeros)", "small-int"};
    for (int pat = 0; pat < 5; ++pat) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            float r = (float)rand() / RAND_MAX * 2.f - 1.f;
            h[i] = pat == 0 ? 0.f : pat == 1 ? 1.f : pat == 2 ? r : pat == 3 ? (r > 0 ? r : 0.f) : (float)(rand() % 5 - 2);
        }
        hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < n; ++i) { float r = (float)rand() / RAND_MAX * 2.f - 1.f; if (pat == 2 || pat == 3) h[i] = r * 0.1f; }
        hipMemcpy(db, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(threads), 0, 0, da, db, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 8 * 4 * 4096.0;   // waves * iters * k * NACC * flop per mfma
            if (rep) printf("%-24s 4 waves/SIMD x 4 acc: %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", names[pat], ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
        }
    }
    // one wave per SIMD
    hipMemset(da, 0, n * 4);
    for (int pat = 0; pat < 2; ++pat) {
        if (pat) { srand(2); for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f; hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, h.data(), n * 4, hipMemcpyHostToDevice); }
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(256), dim3(threads), 0, 0, da, db, dout, iters * 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)256 * 4 * iters * 4 * 8 * 4 * 4096.0;
        printf("%-24s 1 wave/SIMD x 4 acc:  %8.3f ms  %7.1f TFLOP/s\n", pat ? "uniform" : "zeros", ms, flop / ms / 1e9);
    }
    return 0;
}

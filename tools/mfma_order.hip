// Does v_mfma_f32_16x16x4_f32 add its four k's in order (k0, k1, k2, k3), like two v_mfma_f32_32x32x2_f32 (k0,k1),(k2,k3)?
// Computes C[i][j] = sum_k A[i][k] * B[k][j] for K = 64 three ways: host sequential fmaf chain, 32x32x2 MFMAs, 16x16x4 MFMAs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;

__global__ void k32(const float* A, const float* B, float* C)   // A [32][K], B [K][32], C [32][32]
{
    const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + kh], B[(k + kh) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = acc[r];
}
__global__ void k16(const float* A, const float* B, float* C)   // four 16x16 tiles of the same 32x32 product
{
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(ti * 16 + i) * K + k + kq], B[(k + kq) * 32 + tj * 16 + i], acc, 0, 0, 0);
            for (int r = 0; r < 4; ++r) C[(ti * 16 + 4 * kq + r) * 32 + tj * 16 + i] = acc[r];
        }
}
int main()
{
    std::vector<float> A(32 * K), B(K * 32), C0(1024), C1(1024), C2(1024);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : B) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) { float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 32 + j], s); C0[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost);
    int d01 = 0, d02 = 0, d12 = 0;
    for (int n = 0; n < 1024; ++n) { d01 += memcmp(&C0[n], &C1[n], 4) != 0; d02 += memcmp(&C0[n], &C2[n], 4) != 0; d12 += memcmp(&C1[n], &C2[n], 4) != 0; }
    printf("differing elements: host-fma vs 32x32x2 = %d, host-fma vs 16x16x4 = %d, 32x32x2 vs 16x16x4 = %d (of 1024)\n", d01, d02, d12);
    return 0;
}

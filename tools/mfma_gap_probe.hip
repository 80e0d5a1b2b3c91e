// What does an instruction between two MFMAs cost the matrix pipe?  One wave per SIMD (256 threads, 160 KB of LDS per block), chains of
// dependent v_mfma_f32_32x32x2_f32 (16 per accumulator tile, as conv_wino_kernel issues them) with a filler pattern in the gaps:
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_gap_probe tools/mfma_gap_probe.hip && tools/_build/mfma_gap_probe
// prints ns per MFMA for every pattern and the cost per filler instruction relative to the bare chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { BARE, PK1_EVERY, PK2_EVERY, PK2_EVERY2, PK4_EVERY4, PK8_EVERY8, LDS1_EVERY, LDS1_EVERY2, LDS2_EVERY2, VMEM_EVERY4, VMEM_SADD_EVERY4, MIX_KERNEL, NOP_EVERY, SALU_EVERY, NMODES };
static const char* NAMES[NMODES] = {"bare chain", "1 v_pk_add in every gap", "2 v_pk_add in every gap", "2 v_pk_add in every 2nd gap", "4 v_pk_add in every 4th gap",
                                    "8 v_pk_add in every 8th gap", "1 ds_read_b128 in every gap", "1 ds_read_b128 in every 2nd gap", "2 ds_read_b128 in every 2nd gap",
                                    "1 buffer_load_b128 in every 4th gap", "s_add + buffer_load_b128 in every 4th gap",
                                    "per 4 MFMAs: load, ds_read, 2 pk, 2 pk (the kernel's mix)", "1 s_nop 0 in every gap", "1 s_add in every gap"};
static const int FILLERS_PER_128[NMODES] = {0, 128, 256, 128, 128, 128, 128, 64, 128, 32, 32, 32 + 32 + 128, 128, 128};

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const float* g, float* out, int iters)
{
    extern __shared__ float4 smem[];
    const int tid = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    float a = g[tid], b = g[tid + 256];
    f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b};
    f32x4 l0 = {0, 0, 0, 0}, l1 = l0, w0 = l0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1 << 20, 0x00020000);
    smem[tid] = make_float4(a, b, a, b);
    __syncthreads();
    const unsigned laddr = (unsigned)(tid * 16), voff = (unsigned)(tid * 16);
    unsigned so = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 128; ++m) {
            acc[m >> 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m >> 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            auto pk = [&](int n) {
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    if (i & 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(p3));
                    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
                }
            };
            if (MODE == PK1_EVERY) pk(1);
            if (MODE == PK2_EVERY) pk(2);
            if (MODE == PK2_EVERY2 && (m & 1) == 0) pk(2);
            if (MODE == PK4_EVERY4 && (m & 3) == 0) pk(4);
            if (MODE == PK8_EVERY8 && (m & 7) == 0) pk(8);
            // (the loads are written as asm so that nothing waits for them inside the chain -- the kernel consumes its loads 2 - 8 steps
            //  later; the counters are drained once per 128 MFMAs)
            auto lds = [&](f32x4& d, int k) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(laddr), "n"(0) ); (void)k; };
            auto vmem = [&](unsigned soff) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(w0) : "v"(voff), "s"(rs), "s"(soff)); };
            if (MODE == LDS1_EVERY || (MODE == LDS1_EVERY2 && (m & 1) == 0)) lds(l0, m);
            if (MODE == LDS2_EVERY2 && (m & 1) == 0) { lds(l0, m); lds(l1, m + 1); }
            if (MODE == VMEM_EVERY4 && (m & 3) == 0) vmem((unsigned)(m >> 2) * 4096u);
            if (MODE == VMEM_SADD_EVERY4 && (m & 3) == 0) { asm volatile("s_add_u32 %0, %0, 4096\n s_and_b32 %0, %0, 0xffff" : "+s"(so)); vmem(so); }
            if (MODE == MIX_KERNEL) {
                const int e = m & 3;
                if (e == 0) vmem((unsigned)(m >> 2) * 4096u);
                else if (e == 1) lds(l0, m);
                else pk(2);
            }
            if (MODE == NOP_EVERY) asm volatile("s_nop 0");
            if (MODE == SALU_EVERY) asm volatile("s_add_u32 %0, %0, 4" : "+s"(so));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = p0[0] + p2[1] + (float)so + l0[0] + l1[1] + w0[2];
#pragma unroll
    for (int f = 0; f < 8; ++f) s += acc[f][0] + acc[f][15];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static double run(const float* g, float* out, int iters)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 160 * 1024 - 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), lds, 0, g, out, iters);        // warm-up (clocks)
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), lds, 0, g, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), lds, 0, g, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / ((double)iters * 128);        // ns per MFMA (of one wave)
}

template <int MODE>
static void all(const float* g, float* out, int iters, std::vector<double>& t)
{
    t.push_back(run<MODE>(g, out, iters));
    if constexpr (MODE + 1 < NMODES) all<MODE + 1>(g, out, iters, t);
}

int main()
{
    float *g, *out;
    hipMalloc(&g, 1 << 20); hipMalloc(&out, 256 * 256 * 4);
    hipMemset(g, 0, 1 << 20);
    const int iters = 4000;
    std::vector<double> t;
    all<0>(g, out, iters, t);
    std::vector<double> t2;
    all<0>(g, out, iters, t2);                    // second sweep: warm clocks
    printf("{\n \"ns_per_mfma\": {\n");
    for (int m = 0; m < NMODES; ++m) {
        const double d = t2[m] - t2[0];
        printf("  \"%s\": {\"ns_per_mfma\": %.3f, \"vs_bare\": %.4f, \"extra_ns_per_filler\": %.3f}%s\n", NAMES[m], t2[m], t2[m] / t2[0],
               FILLERS_PER_128[m] ? d * 128.0 / FILLERS_PER_128[m] : 0.0, m + 1 < NMODES ? "," : "");
    }
    printf(" },\n \"note\": \"one wave per SIMD, 256 CUs busy; a 32x32x2 fp32 MFMA is 64 cycles = 26.7 ns at 2.4 GHz\"\n}\n");
    return 0;
}

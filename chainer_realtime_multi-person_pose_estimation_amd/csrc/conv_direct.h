// conv_direct.h -- what the direct fp32-MFMA kernels (conv_mfma.hip) share with the opt-in bf16x3 kernels (conv_bf16x3.hip, not part of
// the default build): vector types, the block configuration and the common epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include "pmx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS, int TH, int TW, int BN, int CK, int WM, int WN>
struct ConvCfg {
    static constexpr int T = KS * KS;
    static constexpr int PADK = KS / 2;
    static constexpr int M = TH * TW;                 // real pixels per tile
    static constexpr int MTILES = (M + 31) / 32;      // 32-row MFMA tiles (rows >= M are masked)
    static constexpr bool MASK_M = (M % 32) != 0;
    static constexpr int NTILES = BN / 32;
    static constexpr int MT = MTILES / WM;   // 32-row tiles per wave
    static constexpr int NT = NTILES / WN;   // 32-col tiles per wave
    static constexpr int HALO_H = TH + KS - 1;
    static constexpr int HALO_W = TW + KS - 1;
    static constexpr int LDP = CK + 4;       // padded LDS row (floats): breaks the power-of-two stride
    static constexpr int IN_ELEMS = HALO_H * HALO_W * LDP;
    static constexpr int W_ELEMS = BN * LDP;
    static constexpr int LDS_BYTES = (IN_ELEMS + 2 * W_ELEMS) * 4;
    static constexpr int WREGS = (BN * CK / 4 + 255) / 256;   // float4 per thread per weight panel (1 or 2)
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BN % 32 == 0, "BN must be a multiple of the 32x32 MFMA");
    static_assert(M % 4 == 0, "whole 2x2 windows");
    static_assert(MTILES % WM == 0 && NTILES % WN == 0, "wave grid must divide the tile grid");
    static_assert(TH % 2 == 0 && TW % 2 == 0, "2x2 window mapping");
    static_assert(CK % 8 == 0, "k8 steps");
    static_assert((BN * CK / 4) % 256 == 0 && WREGS <= 2, "weight panel must be 1 or 2 float4 per thread");
};


// ---- shared epilogue: bias + ReLU (+ 2x2 max-pool) + masked NHWC store ----------------------------------------
// C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// `biasv` is loaded in the kernel prologue: a load inside the guarded store blocks makes the compiler drain the
// memory queue (s_waitcnt vmcnt(0)) in front of every one of the 16*MT*NT stores, serialising the store latencies.
template <typename C, int TW>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[C::MT][C::NT], const float (&biasv)[C::NT], const ConvArgs& a,
                                              float* gout, int cout, int bimg, int y0, int x0, int n0, int wm, int wn, int li, int kh)
{
    const int H = a.H, W = a.W;
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
#pragma unroll
        for (int u = 0; u < C::NT; ++u) {
            const int n = n0 + (wn * C::NT + u) * 32 + li;
            const bool nok = n < cout;
            const float bias = biasv[u];
            if (!a.pool) {
                float* out_b = gout + (size_t)bimg * H * W * a.ldc + n;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    const int m = (wm * C::MT + t) * 32 + row;
                    const int q = m >> 2, r = m & 3;
                    const int wy = q / (TW / 2), wx = q % (TW / 2);
                    const int gy = y0 + 2 * wy + (r >> 1), gx = x0 + 2 * wx + (r & 1);
                    float v = acc[t][u][reg] + bias;
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (nok && (!C::MASK_M || m < C::M) && gy < H && gx < W) out_b[(size_t)(gy * W + gx) * a.ldc] = v;
                }
            } else {
                const int Hp = H >> 1, Wp = W >> 1;
                float* out_b = gout + (size_t)bimg * Hp * Wp * a.ldc + n;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v = fmaxf(fmaxf(acc[t][u][4 * g4 + 0], acc[t][u][4 * g4 + 1]),
                                    fmaxf(acc[t][u][4 * g4 + 2], acc[t][u][4 * g4 + 3]));
                    v += bias;
                    if (a.relu) v = fmaxf(v, 0.f);
                    const int q = (wm * C::MT + t) * 8 + 2 * g4 + kh;
                    const int wy = q / (TW / 2), wx = q % (TW / 2);
                    const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
                    if (nok && (!C::MASK_M || q < C::M / 4) && oy < Hp && ox < Wp) out_b[(size_t)(oy * Wp + ox) * a.ldc] = v;
                }
            }
        }
    }
}

// bias of this lane's output channels, fetched up front and pinned in registers (see conv_epilogue)
template <typename C>
__device__ __forceinline__ void conv_load_bias(float (&biasv)[C::NT], const float* gbias, int n0, int wn, int li)
{
#pragma unroll
    for (int u = 0; u < C::NT; ++u) {
        biasv[u] = gbias[n0 + (wn * C::NT + u) * 32 + li];     // bias is padded to cout_pad
        asm volatile("" : "+v"(biasv[u]));                      // materialise now, not at the first use
    }
}


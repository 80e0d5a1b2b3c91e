// conv_mfma.hip -- im2col-free implicit-GEMM convolution on the fp32 matrix cores of gfx950.
//
// Replaces every `L.Convolution2D` (+ `F.relu`, + `F.max_pooling_2d(2,2)`) call of the reference network
// (models/CocoPoseNet.py:26-129 layer table, :136-260 dataflow): stride 1, zero pad ksize/2, ksize in {1,3,7}.
//
//   GEMM view        M = pixels of one (TH x TW) spatial tile, N = BN output channels, K = ks*ks*Cin
//   matrix core      v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 cycles / SIMD, 157.3 TFLOP/s chip peak)
//   activations      NHWC fp32, channel stride lda/ldc (lets a layer read/write a channel slice of a wider
//                    buffer, which is how F.concat (CocoPoseNet.py:168) is eliminated)
//   input panel      (TH+ks-1) x (TW+ks-1) halo tile x CK channels staged once per channel chunk in LDS
//                    (zero-filled outside the image = the conv's zero padding), re-used by all ks*ks taps
//   weight panel     per (tap, chunk): BN x CK floats, pre-packed contiguous [tap][chunk][cout_pad][CK]
//   kernel families  v1  conv_mfma_kernel     weight panel double-buffered in LDS, one barrier per tap (1x1 layers; simple
//                                             reference implementation of the 3x3 / 7x7 layers)
//                    v5  conv_mfma_v5_kernel  weights L2 -> registers, software-pipelined, taps unrolled, <= 2 blocks per CU
//                                             (general 3x3 / 7x7 kernel)
//                    v6  conv_mfma_v6_kernel  one block per CU, 17 (or 9) row tiles of consecutive pixels per wave
//                                             (maps a multiple of 46 wide whose blocks fill whole rounds of the CUs)
//                    c3  conv3x3_c3_kernel    conv1_1 (3 input channels, K packed to 14 k-pairs)
//                    all walk K in the same order (chunk -> tap -> half -> k) => bit-identical outputs, equal to
//                    oracle/conv_fma_ref.c (a plain-C fmaf chain)
//   K ordering       one ds_read_b128 gives a lane 4 consecutive channels; lanes 0-31 hold k-half 0 (channels
//                    c..c+3), lanes 32-63 k-half 1 (c+4..c+7); MFMA step e pairs channel c+e with c+4+e.  The
//                    same permutation is applied to A (pixels) and B (weights), so the sum over K is unchanged.
//   M ordering       row m of a 32-row MFMA tile = pixel (2*wy + (m>>1&1), 2*wx + (m&1)) of 2x2 window m>>2, so the
//                    4 accumulator registers (reg&3) of a lane are one pooling window: the 2x2 max-pool is done
//                    in registers in the epilogue.
//   epilogue         + bias, ReLU, optional 2x2 max-pool, masked NHWC store (lanes run along output channels:
//                    128 B contiguous per half-wave).
//   grid             x = tiles * batch, y = cout_pad / BN, z = group (the PAF and heat-map branches of a stage
//                    run as the two groups of one launch).
#include <hip/hip_ext.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "pmx_common.h"

#include <map>
#include <string.h>

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

template <int KS, int TH, int TW, int BN, int CK, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a)
{
    using C = ConvCfg<KS, TH, TW, BN, CK, WM, WN>;
    extern __shared__ float4 smem4[];
    float* const s_in = reinterpret_cast<float*>(smem4);
    float* const s_w = s_in + C::IN_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31;      // row (A) / column (B, C/D) index inside a 32x32 MFMA tile
    const int kh = lane >> 5;      // k-half

    // select the group's arguments field by field (a dynamic index into the by-value kernarg struct would be
    // copied to scratch)
    const bool g1 = blockIdx.z != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = g1 ? a.g[1].out : a.g[0].out;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    const int H = a.H, W = a.W;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bimg = blockIdx.x / tiles_per_img;
    const int trem = blockIdx.x - bimg * tiles_per_img;
    const int y0 = (trem / a.tiles_x) * TH;
    const int x0 = (trem % a.tiles_x) * TW;
    const int n0 = blockIdx.y * BN;

    const float* in_b = G.in + (size_t)bimg * H * W * a.lda;
    float biasv[C::NT];
    conv_load_bias<C>(biasv, G.bias, n0, wn, li);

    // per-lane LDS offsets of the A (pixel) and B (weight) fragments
    int a_base[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        int m = (wm * C::MT + t) * 32 + li;
        if (C::MASK_M && m >= C::M) m = C::M - 1;     // padded rows recompute the last pixel; never stored
        const int q = m >> 2, r = m & 3;
        const int wy = q / (TW / 2), wx = q % (TW / 2);
        const int py = 2 * wy + (r >> 1), px = 2 * wx + (r & 1);
        a_base[t] = (py * C::HALO_W + px) * C::LDP + kh * 4;
    }
    int b_base[C::NT];
#pragma unroll
    for (int u = 0; u < C::NT; ++u) b_base[u] = ((wn * C::NT + u) * 32 + li) * C::LDP + kh * 4;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int u = 0; u < C::NT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    // weight panel (tap, chunk) for this block's BN columns: contiguous BN*CK floats
    const size_t w_panel_stride = (size_t)a.cout_pad * CK;     // floats between consecutive (tap, chunk) panels
    const float* const w_blk = G.w + (size_t)n0 * CK;

    for (int ch = 0; ch < a.nch; ++ch) {
        __syncthreads();   // previous chunk's MFMAs are done with s_in / s_w
        // ---- stage the input halo tile for this channel chunk (zero fill = conv zero padding) ----
        for (int f = tid; f < C::HALO_H * C::HALO_W * (CK / 4); f += 256) {
            const int hp = f / (CK / 4), c4 = f % (CK / 4);
            const int hy = hp / C::HALO_W, hx = hp - hy * C::HALO_W;
            const int gy = y0 + hy - C::PADK, gx = x0 + hx - C::PADK;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v = *reinterpret_cast<const float4*>(in_b + ((size_t)gy * W + gx) * a.lda + ch * CK + c4 * 4);
            *reinterpret_cast<float4*>(&s_in[hp * C::LDP + c4 * 4]) = v;
        }
        // ---- stage tap 0's weight panel ----
        {
            const float* wp = w_blk + (size_t)ch * w_panel_stride;     // tap 0
#pragma unroll
            for (int r = 0; r < C::WREGS; ++r) {
                const int f = tid + r * 256;
                const int n = f / (CK / 4), c4 = f % (CK / 4);
                const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)f * 4);
                *reinterpret_cast<float4*>(&s_w[n * C::LDP + c4 * 4]) = v;
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int tap = 0; tap < C::T; ++tap) {
            // prefetch the next tap's weight panel into registers (lands under this tap's MFMAs)
            float4 wr0, wr1;      // named registers (a small array here ends up in scratch)
            {
                // Unconditional loads, store and barrier (the last tap re-reads its own panel into the idle
                // buffer): with a conditional use the compiler sinks the loads below the MFMAs and the L2 latency
                // is exposed every tap.
                const int tn = (tap + 1 < C::T) ? tap + 1 : tap;
                const float* wp = w_blk + ((size_t)tn * a.nch + ch) * w_panel_stride;
                wr0 = *reinterpret_cast<const float4*>(wp + (size_t)tid * 4);
                if constexpr (C::WREGS > 1) wr1 = *reinterpret_cast<const float4*>(wp + (size_t)(tid + 256) * 4);
                else wr1 = wr0;
                __builtin_amdgcn_sched_barrier(0);
            }
            const float* sw = s_w + (tap & 1) * C::W_ELEMS;
            const int ky = tap / KS, kx = tap - ky * KS;
            const int tapoff = (ky * C::HALO_W + kx) * C::LDP;
#pragma unroll
            for (int s = 0; s < CK / 8; ++s) {
                float4 av[C::MT], bv[C::NT];
#pragma unroll
                for (int t = 0; t < C::MT; ++t)
                    av[t] = *reinterpret_cast<const float4*>(&s_in[a_base[t] + tapoff + s * 8]);
#pragma unroll
                for (int u = 0; u < C::NT; ++u)
                    bv[u] = *reinterpret_cast<const float4*>(&sw[b_base[u] + s * 8]);
#pragma unroll
                for (int t = 0; t < C::MT; ++t)
#pragma unroll
                    for (int u = 0; u < C::NT; ++u) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, bv[u].x, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, bv[u].y, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].z, bv[u].z, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].w, bv[u].w, acc[t][u], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the wait for the prefetched panel behind this tap's MFMAs
            {
                float* swn = s_w + ((tap + 1) & 1) * C::W_ELEMS;
                {
                    const int n = tid / (CK / 4), c4 = tid % (CK / 4);
                    *reinterpret_cast<float4*>(&swn[n * C::LDP + c4 * 4]) = wr0;
                }
                if constexpr (C::WREGS > 1) {
                    const int f = tid + 256;
                    const int n = f / (CK / 4), c4 = f % (CK / 4);
                    *reinterpret_cast<float4*>(&swn[n * C::LDP + c4 * 4]) = wr1;
                }
                __syncthreads();   // next panel visible; everyone done reading the buffer it will replace next
            }
        }
    }

    // ---- epilogue: bias + ReLU (+ 2x2 max-pool) + masked NHWC store ----
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
conv_epilogue<C, TW>(acc, biasv, a, G.out, G.cout, bimg, y0, x0, n0, wm, wn, li, kh);
}

// =============================================================================================================
// =============================================================================================================
// v5: the general 3x3 / 7x7 kernel (two blocks per CU)
//
// History (profiles/r01_conv_*; the intermediate generations v2 - v4 are in the git history):
//   * with the waves of a block split along N (WN = 4) a wave's weight fragment is private, so it streams L2 -> registers
//     one tap ahead instead of going through LDS: no barrier inside a channel chunk, LDS holds only the halo tile (v2);
//   * the fp32 MFMA pipe loses ~20 % with 3-4 waves per SIMD, so latency is hidden inside the wave (A fragments one k-step
//     ahead in registers, next chunk's halo prefetched into registers and written to a second LDS buffer: one barrier per
//     chunk) and an LDS floor keeps at most two blocks per CU (v3);
//   * a wave issues in order: memory instructions issued in a clump leave the matrix pipe idle, so exactly one memory
//     instruction goes into the gap after each of the first MFMAs of a k-step, and two weight register sets ping-pong (v4);
//   * every VALU / SALU instruction between two MFMAs costs matrix-pipe time: all KS*KS taps of a chunk are unrolled so
//     the LDS tap offsets are ds_read immediates, and the weights come through a buffer resource (descriptor and per-tap
//     panel offset in SGPRs), leaving only MFMAs, ds_reads, buffer_loads and s_waitcnts in the loop (v5: +6 % over v4).
// The block index is remapped so that consecutive tiles (which share halo rows) run on the same XCD and hit in its L2
// (blocks are dispatched round-robin over the 8 XCDs).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename C, int KS>
struct TapBody {
    // one tap = two k8 steps.  bc: this tap's weight fragments; bn: filled with the next tap's (prefetch).
    // av0: this tap's step-0 A fragments (prefetched); on return holds the next tap's step-0 fragments.
    // Weights come through a buffer resource (SGPR descriptor of the group's packed weights) with
    // a per-lane 32-bit byte offset (VGPR, loop invariant) and the next tap's panel offset in an SGPR (`wnext`), so the
    // loop has no address VALU for them; the taps are fully unrolled by the caller, which makes tapoff / tapoff_next
    // compile-time constants that fold into the ds_read immediate offsets.
    static __device__ __forceinline__ void run_u(f32x16 (&acc)[C::MT][C::NT], f32x4 (&av0)[C::MT], f32x4 (&av1)[C::MT],
                                                 const f32x4 (&bc)[C::NT][2], f32x4 (&bn)[C::NT][2], __amdgpu_buffer_rsrc_t wrsrc,
                                                 unsigned wnext, const unsigned (&b_off)[C::NT], const float* cur,
                                                 const int (&a_base)[C::MT], int tapoff, int tapoff_next)
    {
        constexpr int NM = 4 * C::MT * C::NT;
        constexpr int NB = 2 * C::NT;
        static_assert(NM >= NB + C::MT, "not enough MFMA gaps for the memory instructions of a k-step");
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int e = i / (C::MT * C::NT), t = (i / C::NT) % C::MT, u = i % C::NT;
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t][e], bc[u][0][e], acc[t][u], 0, 0, 0);
            if (i < NB) {
                bn[i >> 1][i & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i >> 1] + (i & 1) * 32, wnext, 0));
                __builtin_amdgcn_sched_barrier(0);
            } else if (i < NB + C::MT) {
                av1[i - NB] = *reinterpret_cast<const f32x4*>(&cur[a_base[i - NB] + tapoff + 8]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int e = i / (C::MT * C::NT), t = (i / C::NT) % C::MT, u = i % C::NT;
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t][e], bc[u][1][e], acc[t][u], 0, 0, 0);
            if (i < C::MT) {
                av0[i] = *reinterpret_cast<const f32x4*>(&cur[a_base[i] + tapoff_next]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
};

template <int KS, int TH, int TW, int BN, int CK, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_mfma_v5_kernel(const ConvArgs a)
{
    using C = ConvCfg<KS, TH, TW, BN, CK, WM, WN>;
    static_assert(CK == 16 && KS > 1 && (KS * KS) % 2 == 1, "v5: 16-channel chunks, odd number of taps");
    constexpr int NHF = (C::HALO_H * C::HALO_W * (CK / 4) + 255) / 256;
    extern __shared__ float4 smem4[];
    float* const s_in = reinterpret_cast<float*>(smem4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31;
    const int kh = lane >> 5;

    // blockIdx.z = K slice * ngroups + group (ksplit == 1: the group index)
    const int kslice = a.ksplit > 1 ? (a.ngroups > 1 ? (int)blockIdx.z >> 1 : (int)blockIdx.z) : 0;
    const int zgrp = a.ksplit > 1 ? (a.ngroups > 1 ? (int)blockIdx.z & 1 : 0) : (int)blockIdx.z;
    const bool g1 = zgrp != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = (g1 ? a.g[1].out : a.g[0].out) + (size_t)kslice * a.slab_stride;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    const int H = a.H, W = a.W;
    // this slice's chunk range [c0, c1): packed boundaries (a dynamic index into a kernarg array would go through scratch)
    const int c0 = a.ksplit > 1 ? (int)((a.kbounds >> (8 * kslice)) & 0xffull) : 0;
    const int c1 = (a.ksplit > 1 && kslice + 1 < a.ksplit) ? (int)((a.kbounds >> (8 * (kslice + 1))) & 0xffull) : a.nch;

    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bimg = tile / tiles_per_img;
    const int trem = tile - bimg * tiles_per_img;
    const int y0 = (trem / a.tiles_x) * TH;
    const int x0 = (trem % a.tiles_x) * TW;
    const int n0 = blockIdx.y * BN;
    const float* in_b = G.in + (size_t)bimg * H * W * a.lda + c0 * CK;
    float biasv[C::NT];
    conv_load_bias<C>(biasv, G.bias, n0, wn, li);

    int a_base[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        int m = (wm * C::MT + t) * 32 + li;
        if (C::MASK_M && m >= C::M) m = C::M - 1;
        const int q = m >> 2, r = m & 3;
        const int wy = q / (TW / 2), wx = q % (TW / 2);
        const int py = 2 * wy + (r >> 1), px = 2 * wx + (r & 1);
        a_base[t] = (py * C::HALO_W + px) * C::LDP + kh * 4;
    }
    const size_t w_panel_stride = (size_t)a.cout_pad * CK;
    // buffer resource over the group's packed weights (raw buffer, 32-bit byte offsets) + per-lane byte offsets
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    unsigned b_off[C::NT];
#pragma unroll
    for (int u = 0; u < C::NT; ++u) b_off[u] = (unsigned)(((n0 + (wn * C::NT + u) * 32 + li) * CK + kh * 4) * 4);

    int h_lds[NHF], h_goff[NHF];
    bool h_ok[NHF];
#pragma unroll
    for (int r = 0; r < NHF; ++r) {
        const int f = tid + r * 256;
        const bool slot = f < C::HALO_H * C::HALO_W * (CK / 4);
        const int hp = slot ? f / (CK / 4) : 0, c4 = f % (CK / 4);
        const int hy = hp / C::HALO_W, hx = hp - hy * C::HALO_W;
        const int gy = y0 + hy - C::PADK, gx = x0 + hx - C::PADK;
        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        h_lds[r] = slot ? hp * C::LDP + c4 * 4 : -1;
        h_goff[r] = (cy * W + cx) * a.lda + c4 * 4;
        h_ok[r] = inb;
    }

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int u = 0; u < C::NT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

    f32x4 bA[C::NT][2], bB[C::NT][2];      // ping-pong weight fragment sets
    const unsigned first_b = (unsigned)((size_t)c0 * w_panel_stride * 4);      // tap 0 of the slice's first chunk
#pragma unroll
    for (int u = 0; u < C::NT; ++u) {
        bA[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[u], first_b, 0));
        bA[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[u] + 32, first_b, 0));
    }
#pragma unroll
    for (int r = 0; r < NHF; ++r) {
        float4 v = *reinterpret_cast<const float4*>(in_b + h_goff[r]);
        if (!h_ok[r]) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h_lds[r] >= 0) *reinterpret_cast<float4*>(&s_in[h_lds[r]]) = v;
    }
    __syncthreads();
    f32x4 av0[C::MT], av1[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        av0[t] = *reinterpret_cast<const f32x4*>(&s_in[a_base[t]]);
    }

    for (int ch = c0; ch < c1; ++ch) {
        const float* cur = s_in + ((ch - c0) & 1) * C::IN_ELEMS;
        float* nxt = s_in + ((ch - c0 + 1) & 1) * C::IN_ELEMS;
        const bool more_ch = ch + 1 < c1;
        float4 hreg[NHF];
        {
            // next chunk's halo: global -> registers now, registers -> LDS after the last tap (in_b points at chunk c0)
            const int cn = (more_ch ? ch + 1 : ch) - c0;
#pragma unroll
            for (int r = 0; r < NHF; ++r) hreg[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r] + cn * CK);
        }
        // all taps unrolled (LDS offsets become immediates); the two weight register sets alternate at compile time
        const unsigned chunk_b = (unsigned)((size_t)ch * w_panel_stride * 4);
        unsigned tap_b = (unsigned)((size_t)a.nch * w_panel_stride * 4);
        asm volatile("" : "+s"(tap_b));          // keep the KS*KS panel offsets from being hoisted out of the chunk loop
        unsigned soff = chunk_b;                 // byte offset of the current tap's panel
#pragma unroll
        for (int tap = 0; tap < C::T; ++tap) {
            const int toff = ((tap / KS) * C::HALO_W + tap % KS) * C::LDP;
            const int tnx = tap + 1 < C::T ? tap + 1 : tap;
            const int toff_n = ((tnx / KS) * C::HALO_W + tnx % KS) * C::LDP;
            // next tap's panel; the last tap prefetches tap 0 of the next chunk (or re-reads its own panel at the very end)
            unsigned wnext;
            if (tap + 1 < C::T) { soff += tap_b; wnext = soff; }
            else wnext = more_ch ? chunk_b + (unsigned)(w_panel_stride * 4) : soff;
            if (tap & 1) TapBody<C, KS>::run_u(acc, av0, av1, bB, bA, wrsrc, wnext, b_off, cur, a_base, toff, toff_n);
            else TapBody<C, KS>::run_u(acc, av0, av1, bA, bB, wrsrc, wnext, b_off, cur, a_base, toff, toff_n);
        }
#pragma unroll
        for (int u = 0; u < C::NT; ++u) { bA[u][0] = bB[u][0]; bA[u][1] = bB[u][1]; }     // T is odd
        if (more_ch) {
#pragma unroll
            for (int r = 0; r < NHF; ++r) {
                float4 v = hreg[r];
                if (!h_ok[r]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (h_lds[r] >= 0) *reinterpret_cast<float4*>(&nxt[h_lds[r]]) = v;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < C::MT; ++t) av0[t] = *reinterpret_cast<const f32x4*>(&nxt[a_base[t]]);
        }
    }

    // ---- epilogue (shared with v1) ----
    conv_epilogue<C, TW>(acc, biasv, a, G.out, G.cout, bimg, y0, x0, n0, wm, wn, li, kh);
}

// ---- v6: one block per CU, 17 MFMA row tiles of consecutive pixels per wave -------------------------------------------
// For maps whose launch would otherwise quantise badly (46x46 at batch 32: 1472 strip blocks on 512 slots = 2.875 rounds,
// 92 -> 96 padded rows): the map is cut into vertical SLABS of 46 columns (46 / 92 / 184 / 368-wide maps = 1 / 2 / 4 / 8
// slabs; the left / right halo columns come from the neighbouring slab or are zero at the image border) and a block owns
// MT*32 CONSECUTIVE pixels of one slab (MT = 17: 544 px, 4 blocks per 46x46 map = 2.8 % padding; batch 32 x 2 branch
// groups x 4 = 256 blocks = one per CU) and all 128 output channels of the group (wave w = channels 32w..32w+31, all MT
// row tiles: 272 accumulator registers, 1 wave per SIMD).  Pixel order inside a slab: row-major, or (POOL) row pairs
// interleaved - index = (y / 2) * 92 + 2 * x + (y & 1) - so that four consecutive MFMA rows are one 2x2 pooling window.
// Same 16-channel chunks, same tap / k8-step / k order as every other generation -> bit-identical results.
//   LDS: the SPAN+KS-1 input rows the pixel run touches, full width + padding columns, 16 channels, double-buffered
//        (7x7: 2 x 19 x 52 x 20 floats = 158 080 B -> exactly one block per CU).
//   Inner loop (one kernel row = KS taps x 2 k8-steps x MT tiles, fully unrolled; rows are a run-time loop): per "unit"
//        (tap, step, tile) 4 MFMAs + one ds_read_b128 that refills a RING-deep A-fragment ring DIST units ahead (immediate
//        offsets, also across the row boundary); one buffer_load_dwordx4 of weights per step, one step ahead.
template <int KS, int MT, int POOL>
struct V6Cfg {
    static constexpr int W = 46;                                  // slab width
    static constexpr int PADK = KS / 2, T = KS * KS, CK = 16, LDP = 20, M = MT * 32;
    // image rows a run of M consecutive slab pixels can touch (POOL: whole row pairs)
    static constexpr int SPAN = POOL ? 2 * ((2 * W - 1 + M - 1) / (2 * W) + 1) : (W - 1 + M - 1) / W + 1;
    static constexpr int HALO_H = SPAN + KS - 1, HALO_W = W + KS - 1;
    static constexpr int IN_ELEMS = HALO_H * HALO_W * LDP;
    static constexpr int LDS_BYTES = 2 * IN_ELEMS * 4;
    static constexpr int NHF = (HALO_H * HALO_W * (CK / 4) + 255) / 256;
    static constexpr int UNITS_ROW = KS * 2 * MT;
    static constexpr int RING = KS == 7 ? 7 : 6, DIST = RING - 1;
    static_assert(UNITS_ROW % RING == 0, "the A ring must close over one kernel row");
    static_assert(LDS_BYTES <= 160 * 1024, "halo double buffer exceeds the LDS");
    static_assert(NHF <= 32, "halo slot mask is 32 bits");
};

// One of the MT accumulator tiles must live in VGPRs: the compiler's MFMAs take C/D from AGPRs and 17 tiles need 272 > 256
// of them (it would shuttle one tile through v_accvgpr moves around every use).  VGPR-form MFMA, same instruction.
__device__ __forceinline__ void mfma_32x32x2_vgpr(f32x16& acc, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int KS, int MT, int POOL>
__global__ __launch_bounds__(256, 1) void conv_mfma_v6_kernel(const ConvArgs a)
{
    using C = V6Cfg<KS, MT, POOL>;
    constexpr int SW = C::W;
    constexpr int CK = C::CK;
    constexpr int MTA = MT > 16 ? 16 : MT;                  // tiles accumulated in AGPRs (compiler MFMAs); the rest in VGPRs
    static_assert(MT <= 17, "at most one VGPR-resident accumulator tile");
    extern __shared__ float4 smem4[];
    float* const s_in = reinterpret_cast<float*>(smem4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;

    const bool g1 = blockIdx.z != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = g1 ? a.g[1].out : a.g[0].out;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    const int H = a.H, W = a.W;
    const int SP = H * SW;                                  // pixels of one slab

    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // tile -> (image, slab, block of the slab); a.tiles_x = blocks per slab, a.tiles_y = slabs per image
    const int bimg = tile / (a.tiles_x * a.tiles_y);
    const int trem = tile - bimg * a.tiles_x * a.tiles_y;
    const int slab = trem / a.tiles_x;
    const int sx0 = slab * SW;                              // first image column of the slab
    const int p0 = (trem - slab * a.tiles_x) * C::M;        // first slab pixel (in slab order) of this block
    const int y0 = POOL ? 2 * (p0 / (2 * SW)) : p0 / SW;    // first image row it touches
    const int n0 = blockIdx.y * 128;
    const int n = n0 + wave * 32 + li;                      // this lane's output channel
    const float* in_b = G.in + (size_t)bimg * H * W * a.lda;
    float bias = G.bias[n];                                 // padded to cout_pad (pinned to a register below, once the first halo loads are issued)

    // LDS element offsets of this lane's pixel in each row tile (kernel row 0, tap column 0, buffer 0); advanced per
    // kernel row and rewound / switched to the other buffer per chunk
    int a_cur[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int p = p0 + t * 32 + li;
        if (p >= SP) p = SP - 1;                            // padded rows recompute the last pixel; never stored
        int y, x;
        if (POOL) { const int rp = p / (2 * SW), q = p - rp * 2 * SW; y = 2 * rp + (q & 1); x = q >> 1; }
        else { y = p / SW; x = p - y * SW; }
        a_cur[t] = ((y - y0) * C::HALO_W + x) * C::LDP + kh * 4;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    const unsigned b_off = (unsigned)((n * CK + kh * 4) * 4);
    const unsigned panel_b = (unsigned)((size_t)a.cout_pad * CK * 4);     // bytes between (tap, chunk) panels
    const unsigned tap_b0 = panel_b * (unsigned)a.nch;                    // bytes between taps

    // halo staging slots of this thread (same for every chunk): clamped global offset + in-bounds bit; the LDS offset is
    // recomputed at the write (hp * LDP + c4 * 4) to save registers
    int h_goff[C::NHF];
    unsigned h_ok = 0;
#pragma unroll
    for (int r = 0; r < C::NHF; ++r) {
        const int f = tid + r * 256;
        const bool slot = f < C::HALO_H * C::HALO_W * (CK / 4);
        const int hp = slot ? f / (CK / 4) : 0, c4 = f % (CK / 4);
        const int hy = hp / C::HALO_W, hx = hp - hy * C::HALO_W;
        const int gy = y0 + hy - C::PADK, gx = sx0 + hx - C::PADK;
        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        h_goff[r] = (cy * W + cx) * a.lda + c4 * 4;
        h_ok |= (slot && inb) ? (1u << r) : 0u;
    }
    auto halo_store = [&](float* buf, const float4 (&hv)[C::NHF]) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) {
            const int f = tid + r * 256;
            float4 v = hv[r];
            if (!((h_ok >> r) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < C::HALO_H * C::HALO_W * (CK / 4)) *reinterpret_cast<float4*>(&buf[(f >> 2) * C::LDP + (f & 3) * 4]) = v;
        }
    };

    f32x16 acc[MTA];
    f32x16 accv;                                            // tile MT-1 when MT == 17
#pragma unroll
    for (int i = 0; i < 16; ++i) accv[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MTA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // chunk 0 halo -> buffer 0; first weight fragment (tap 0, step 0)
    f32x4 bw[2];
    bw[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off, 0u, 0));
    {
        float4 hv[C::NHF];
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) hv[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r]);
        halo_store(s_in, hv);
    }
    asm volatile("" : "+v"(bias));       // (pinned right after its load, the block waited a memory round trip before issuing anything else)
    __syncthreads();

    f32x4 av[C::RING];
    for (int ch = 0; ch < a.nch; ++ch) {
        float* nxt = s_in + ((ch + 1) & 1) * C::IN_ELEMS;
        const bool more_ch = ch + 1 < a.nch;
        float4 hreg[C::NHF];
        // prime the A ring: units 0 .. DIST-1 of kernel row 0 (tap 0, step 0, tiles 0 ..)
#pragma unroll
        for (int d = 0; d < C::DIST; ++d) {
            const int kxn = d / (2 * MT), stepn = (d / MT) % 2, tn = d % MT;
            av[d] = *reinterpret_cast<const f32x4*>(&s_in[a_cur[tn] + kxn * C::LDP + stepn * 8]);
        }
        const unsigned chunk_b = (unsigned)ch * panel_b;
        unsigned tap_b = tap_b0;
        asm volatile("" : "+s"(tap_b));          // keep the per-row panel offsets inside the chunk loop
        unsigned soff_row = chunk_b;             // panel of the first tap of the current kernel row
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
            // weights that follow this row's last step: next row's first tap, or tap 0 of the next chunk, or (at the very
            // end) this row's last panel again
            const unsigned wnext_end = ky + 1 < KS ? soff_row + KS * tap_b
                                                   : (more_ch ? chunk_b + panel_b : soff_row + (KS - 1) * tap_b);
#pragma unroll
            for (int q = 0; q < 2 * KS; ++q) {              // q = 2 * kx + step within the row
                {                                           // start of a k8 step: fetch the next step's weight fragment
                    const int qn = q + 1;
                    const unsigned so = qn < 2 * KS ? soff_row + (unsigned)(qn / 2) * tap_b : wnext_end;
                    bw[qn & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (qn & 1) * 32, so, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int u = q * MT + t;
                    if (u == 1 && ky == 0) {
                        // next chunk's halo: global -> registers now, registers -> LDS after the last row (issued behind
                        // the weight load so that the in-order vmcnt wait of the next step covers them for free)
                        const int cn = more_ch ? ch + 1 : ch;
#pragma unroll
                        for (int r = 0; r < C::NHF; ++r) hreg[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r] + cn * CK);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (t < MTA) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[t < MTA ? t : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u % C::RING][e], bw[q & 1][e], acc[t < MTA ? t : 0], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) mfma_32x32x2_vgpr(accv, av[u % C::RING][e], bw[q & 1][e]);
                    }
                    {
                        int un = u + C::DIST, rowadd = 0;
                        if (un >= C::UNITS_ROW) { un -= C::UNITS_ROW; rowadd = C::HALO_W * C::LDP; }
                        const int kxn = un / (2 * MT), stepn = (un / MT) % 2, tn = un % MT;
                        av[(u + C::DIST) % C::RING] = *reinterpret_cast<const f32x4*>(&s_in[a_cur[tn] + rowadd + kxn * C::LDP + stepn * 8]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) a_cur[t] += C::HALO_W * C::LDP;
            soff_row += KS * tap_b;
        }
        {
            // rewind to kernel row 0 and switch to the other halo buffer
            const int delta = ((ch & 1) ? -C::IN_ELEMS : C::IN_ELEMS) - KS * C::HALO_W * C::LDP;
#pragma unroll
            for (int t = 0; t < MT; ++t) a_cur[t] += delta;
        }
        if (more_ch) {
            halo_store(nxt, hreg);
            __syncthreads();
        }
    }
    // the VGPR-form MFMAs are opaque to the compiler's hazard recogniser: let the last one retire before VALU reads accv
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");

    // ---- epilogue: bias + ReLU (+ 2x2 max-pool) + masked NHWC store
    // (C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * kh)
    const bool nok = n < G.cout;
    if (!POOL) {
        float* out_b = G.out + (size_t)bimg * H * W * a.ldc + n;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int p = p0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                const int y = p / SW, x = p - y * SW;
                float v = (t < MTA ? acc[t < MTA ? t : 0][reg] : accv[reg]) + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                if (nok && p < SP) out_b[((size_t)y * W + sx0 + x) * a.ldc] = v;
            }
        }
    } else {
        // four consecutive slab pixels (index % 4 == 0) are one window: pooled pixel (p / (2 * SW), (p % (2 * SW)) / 4)
        const int Hp = H >> 1, Wp = W >> 1;
        float* out_b = G.out + (size_t)bimg * Hp * Wp * a.ldc + n;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v;
                if (t < MTA) {
                    const f32x16& A = acc[t < MTA ? t : 0];
                    v = fmaxf(fmaxf(A[4 * g4 + 0], A[4 * g4 + 1]), fmaxf(A[4 * g4 + 2], A[4 * g4 + 3]));
                } else {
                    v = fmaxf(fmaxf(accv[4 * g4 + 0], accv[4 * g4 + 1]), fmaxf(accv[4 * g4 + 2], accv[4 * g4 + 3]));
                }
                v += bias;
                if (a.relu) v = fmaxf(v, 0.f);
                const int p = p0 + t * 32 + 8 * g4 + 4 * kh;
                const int rp = p / (2 * SW), ox = (p - rp * 2 * SW) >> 2;
                if (nok && p < SP) out_b[((size_t)rp * Wp + (sx0 >> 1) + ox) * a.ldc] = v;
            }
        }
    }
}

// ---- v7 (opt-in, option "precision" = 1): fp32-grade convolution on the BF16 matrix cores ------------------------------------
// The fp32 MFMA runs at 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md).  Every fp32 value is split into three bf16 terms,
// x = hi + mid + lo (each the round-to-nearest bf16 of the remaining residual: 3 x 8 mantissa bits, |x - hi - mid - lo| <= 2^-27 |x|),
// and x * w is accumulated in fp32 from the six bf16 products that matter -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid (each
// exact in fp32; the three dropped ones are <= 2^-24 relative) -- on v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 32x32x16 block
// instead of 8 x 64 with the fp32 MFMA = 2.67x the matrix rate at fp32-like accuracy (it is NOT the fp32 FMA chain of the
// other kernels: results differ from them by summation-order-sized noise, not bit for bit; reported as its own dtype).
// Geometry = v6 (one block per CU, MT x 32 consecutive pixels of a 46-column slab x 128 channels, wave = 32 channels x all
// row tiles, 16 tiles in AGPRs + 1 in VGPRs).  Activations stay fp32 in HBM; the split happens while the halo of a 16-channel
// chunk is staged: LDS holds [pixel][plane][16 ch] bf16 at a 112-byte pitch (28 dwords: conflict-free ds_read_b128 over the
// 16-lane groups).  Weights are split once on the host: [tap][chunk][plane][cout_pad][16] bf16.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KS, int MT, int POOL>
struct V7Cfg {
    static constexpr int W = 46;
    static constexpr int PADK = KS / 2, T = KS * KS, CK = 16, M = MT * 32;
    static constexpr int SPAN = POOL ? 2 * ((2 * W - 1 + M - 1) / (2 * W) + 1) : (W - 1 + M - 1) / W + 1;
    static constexpr int HALO_H = SPAN + KS - 1, HALO_W = W + KS - 1;
    static constexpr int PITCH = 112;                                   // bytes per halo pixel
    static constexpr int IN_BYTES = HALO_H * HALO_W * PITCH;
    static constexpr int LDS_BYTES = IN_BYTES + HALO_W * PITCH;         // + one row: the A prefetch runs one kernel row ahead
    static constexpr int NHF = (HALO_H * HALO_W * (CK / 4) + 255) / 256;
    static_assert(LDS_BYTES <= 160 * 1024, "halo exceeds the LDS");
};

__device__ __forceinline__ void mfma_bf16_vgpr(f32x16& acc, const f32x4& a, const f32x4& b)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// x -> (hi, mid, lo) bf16, each the round-to-nearest of what is left; four channels at a time, packed for ds_write_b64
__device__ __forceinline__ void split3_store(char* dst, const float4& v)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    __bf16 h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (__bf16)x[i];
        const float r1 = x[i] - (float)h[i];
        m[i] = (__bf16)r1;
        const float r2 = r1 - (float)m[i];
        l[i] = (__bf16)r2;
    }
    const bf16x2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]}, m01 = {m[0], m[1]}, m23 = {m[2], m[3]}, l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
    *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    *reinterpret_cast<uint2*>(dst + 32) = make_uint2(__builtin_bit_cast(unsigned, m01), __builtin_bit_cast(unsigned, m23));
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

template <int KS, int MT, int POOL>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_kernel(const ConvArgs a)
{
    using C = V7Cfg<KS, MT, POOL>;
    constexpr int SW = C::W, CK = C::CK, PITCH = C::PITCH;
    constexpr int MTA = MT > 16 ? 16 : MT;
    static_assert(MT <= 17, "at most one VGPR-resident accumulator tile");
    extern __shared__ float4 smem4[];
    char* const s_in = reinterpret_cast<char*>(smem4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const bool g1 = blockIdx.z != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;              // bf16x3 pack of the layer (the host passes it in place of the fp32 pack)
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = g1 ? a.g[1].out : a.g[0].out;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    const int H = a.H, W = a.W;
    const int SP = H * SW;

    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bimg = tile / (a.tiles_x * a.tiles_y);
    const int trem = tile - bimg * a.tiles_x * a.tiles_y;
    const int slab = trem / a.tiles_x;
    const int sx0 = slab * SW;
    const int p0 = (trem - slab * a.tiles_x) * C::M;
    const int y0 = POOL ? 2 * (p0 / (2 * SW)) : p0 / SW;
    const int n0 = blockIdx.y * 128;
    const int n = n0 + wave * 32 + li;
    const float* in_b = G.in + (size_t)bimg * H * W * a.lda;
    float bias = G.bias[n];                       // (pinned to a register below, once the first halo loads are issued)

    // LDS byte offsets of this lane's pixel in each row tile (kernel row 0, tap column 0, plane 0)
    int a_cur[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int p = p0 + t * 32 + li;
        if (p >= SP) p = SP - 1;
        int y, x;
        if (POOL) { const int rp = p / (2 * SW), q = p - rp * 2 * SW; y = 2 * rp + (q & 1); x = q >> 1; }
        else { y = p / SW; x = p - y * SW; }
        a_cur[t] = ((y - y0) * C::HALO_W + x) * PITCH + kh * 16;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    const unsigned b_off = (unsigned)(n * 32 + kh * 16);                    // bytes inside one [cout_pad][16] bf16 plane
    const unsigned plane_b = (unsigned)a.cout_pad * 32u;                    // bytes between planes
    const unsigned panel_b = 3u * plane_b;                                  // bytes between (tap, chunk) panels
    const unsigned tap_b0 = panel_b * (unsigned)a.nch;                      // bytes between taps

    int h_goff[C::NHF];
    unsigned h_ok = 0;
#pragma unroll
    for (int r = 0; r < C::NHF; ++r) {
        const int f = tid + r * 256;
        const bool slot = f < C::HALO_H * C::HALO_W * (CK / 4);
        const int hp = slot ? f / (CK / 4) : 0, c4 = f % (CK / 4);
        const int hy = hp / C::HALO_W, hx = hp - hy * C::HALO_W;
        const int gy = y0 + hy - C::PADK, gx = sx0 + hx - C::PADK;
        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        h_goff[r] = (cy * W + cx) * a.lda + c4 * 4;
        h_ok |= (slot && inb) ? (1u << r) : 0u;
    }
    auto halo_store = [&](const float4 (&hv)[C::NHF]) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) {
            const int f = tid + r * 256;
            float4 v = hv[r];
            if (!((h_ok >> r) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < C::HALO_H * C::HALO_W * (CK / 4)) split3_store(s_in + (f >> 2) * PITCH + (f & 3) * 8, v);
        }
    };

    f32x16 acc[MTA];
    f32x16 accv;
#pragma unroll
    for (int i = 0; i < 16; ++i) accv[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MTA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // weight fragments of one tap: [plane] (hi, mid, lo); bc = current tap, bn = next tap
    f32x4 bc[3], bn[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bc[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off, pl * plane_b, 0));
    {
        float4 hv[C::NHF];
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) hv[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r]);
        halo_store(hv);
    }
    asm volatile("" : "+v"(bias));
    __syncthreads();

    // The row tiles are processed in PAIRS (t, t + 1): the six products of the two tiles are interleaved, so consecutive MFMAs
    // are independent (a dependent 8-pass MFMA cannot issue until its predecessor has left the pipe), and the A fragments of the
    // next pair (6 ds_read_b128) are fetched one pair = 12 MFMAs = 384 cycles ahead.  NP pairs per tap (the last one is a
    // single tile when MT is odd); the pair ring has two slots and the row has KS * NP pairs.
    constexpr int NP = (MT + 1) / 2;
    f32x4 ar[2][2][3];                            // [ring slot][tile of the pair][plane hi | mid | lo]
    auto mfma_pair = [&](const f32x4 (&A0)[3], const f32x4 (&A1)[3], int t0, bool two) {
        // smallest terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !two) continue;
                const int t = t0 + j;
                const f32x4& av = j ? A1[PA[q]] : A0[PA[q]];
                if (t < MTA) {
                    f32x16& c = acc[t < MTA ? t : 0];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bc[PB[q]]), c, 0, 0, 0);
                } else {
                    mfma_bf16_vgpr(accv, av, bc[PB[q]]);
                }
            }
        }
    };
    for (int ch = 0; ch < a.nch; ++ch) {
        const bool more_ch = ch + 1 < a.nch;
        float4 hreg[C::NHF];
        {
            const int cn = more_ch ? ch + 1 : ch;                   // next chunk's halo: global -> registers under this chunk's MFMAs
#pragma unroll
            for (int r = 0; r < C::NHF; ++r) hreg[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r] + cn * CK);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ar[0][j][pl] = *reinterpret_cast<const f32x4*>(s_in + a_cur[j < MT ? j : 0] + pl * 32);
        const unsigned chunk_b = (unsigned)ch * panel_b;
        unsigned tap_b = tap_b0;
        asm volatile("" : "+s"(tap_b));
        unsigned soff = chunk_b;                                    // panel of the current tap
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                {   // next tap's weights (the last tap of the chunk fetches tap 0 of the next chunk, or itself at the very end)
                    const bool last_tap = (kx == KS - 1) && (ky == KS - 1);
                    const unsigned so = !last_tap ? soff + tap_b : (more_ch ? chunk_b + panel_b : soff);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bn[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off, so + pl * plane_b, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int u = kx * NP + pp;
                    {   // A fragments of the next pair: next tiles of this tap, or tiles 0 / 1 of the next tap / next kernel row
                        int pn = pp + 1, kxn = kx, rowadd = 0;
                        if (pn == NP) { pn = 0; kxn = kx + 1; if (kxn == KS) { kxn = 0; rowadd = C::HALO_W * PITCH; } }
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int tn = 2 * pn + j;
                            if (tn >= MT) continue;
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                ar[(u + 1) & 1][j][pl] = *reinterpret_cast<const f32x4*>(s_in + a_cur[tn < MT ? tn : 0] + rowadd + kxn * PITCH + pl * 32);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mfma_pair(ar[u & 1][0], ar[u & 1][1], 2 * pp, 2 * pp + 1 < MT);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bc[pl] = bn[pl];
                soff += tap_b;
            }
            if ((KS * NP) & 1) {        // odd number of pairs per kernel row: the prefetched pair sits in the other slot
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ar[0][j][pl] = ar[1][j][pl];
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) a_cur[t] += C::HALO_W * PITCH;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) a_cur[t] -= KS * C::HALO_W * PITCH;
        if (more_ch) {
            __syncthreads();                    // every wave is done reading this chunk's halo
            halo_store(hreg);
            __syncthreads();
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");

    // ---- epilogue (as v6): bias + ReLU (+ 2x2 max-pool) + masked NHWC store
    const bool nok = n < G.cout;
    if (!POOL) {
        float* out_b = G.out + (size_t)bimg * H * W * a.ldc + n;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int p = p0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                const int y = p / SW, x = p - y * SW;
                float v = (t < MTA ? acc[t < MTA ? t : 0][reg] : accv[reg]) + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                if (nok && p < SP) out_b[((size_t)y * W + sx0 + x) * a.ldc] = v;
            }
        }
    } else {
        const int Hp = H >> 1, Wp = W >> 1;
        float* out_b = G.out + (size_t)bimg * Hp * Wp * a.ldc + n;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v;
                if (t < MTA) {
                    const f32x16& A = acc[t < MTA ? t : 0];
                    v = fmaxf(fmaxf(A[4 * g4 + 0], A[4 * g4 + 1]), fmaxf(A[4 * g4 + 2], A[4 * g4 + 3]));
                } else {
                    v = fmaxf(fmaxf(accv[4 * g4 + 0], accv[4 * g4 + 1]), fmaxf(accv[4 * g4 + 2], accv[4 * g4 + 3]));
                }
                v += bias;
                if (a.relu) v = fmaxf(v, 0.f);
                const int p = p0 + t * 32 + 8 * g4 + 4 * kh;
                const int rp = p / (2 * SW), ox = (p - rp * 2 * SW) >> 2;
                if (nok && p < SP) out_b[((size_t)rp * Wp + (sx0 >> 1) + ox) * a.ldc] = v;
            }
        }
    }
}

// ---- v8: the bf16x3 arithmetic of v7 on the small tiles of v5 (single images / small batches; split-K capable) ------------------
// Block = 8 x 8 pixels x 64 channels, 2 x 2 waves of one 32 x 32 tile each (the v5 "small" geometry, 2 blocks per CU), halo
// (8 + KS - 1)^2 pixels x [3 planes x 16 ch bf16] at the 112-byte pitch, double-buffered, converted from fp32 while it is staged.
// With one tile per wave a tap is only 6 MFMAs = 192 cycles, less than an L2 round trip: the weight fragments of a tap (3 planes)
// are fetched RB taps ahead into a register ring, the A fragments one tap ahead.  All KS * KS taps of a chunk are unrolled.
// K slices / slabs / combine kernel exactly as in the v5 kernels (ConvArgs::ksplit, kbounds, slab_stride).
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_small_kernel(const ConvArgs a)
{
    using C = ConvCfg<KS, 8, 8, 64, 16, 2, 2>;
    constexpr int TW = 8, CK = 16, PITCH = 112, T = KS * KS, RB = 8, RA = 3;      // weight ring: RB - 1 taps ahead (L2); A ring: RA - 1 taps ahead (LDS)
    constexpr int IN_BYTES = C::HALO_H * C::HALO_W * PITCH;
    constexpr int NHF = (C::HALO_H * C::HALO_W * (CK / 4) + 255) / 256;
    extern __shared__ float4 smem4[];
    char* const s_in = reinterpret_cast<char*>(smem4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, kh = lane >> 5;
    const int kslice = a.ksplit > 1 ? (a.ngroups > 1 ? (int)blockIdx.z >> 1 : (int)blockIdx.z) : 0;
    const int zgrp = a.ksplit > 1 ? (a.ngroups > 1 ? (int)blockIdx.z & 1 : 0) : (int)blockIdx.z;
    const bool g1 = zgrp != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;              // bf16x3 pack
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = (g1 ? a.g[1].out : a.g[0].out) + (size_t)kslice * a.slab_stride;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    const int H = a.H, W = a.W;
    const int c0 = a.ksplit > 1 ? (int)((a.kbounds >> (8 * kslice)) & 0xffull) : 0;
    const int c1 = (a.ksplit > 1 && kslice + 1 < a.ksplit) ? (int)((a.kbounds >> (8 * (kslice + 1))) & 0xffull) : a.nch;

    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bimg = tile / tiles_per_img;
    const int trem = tile - bimg * tiles_per_img;
    const int y0 = (trem / a.tiles_x) * 8, x0 = (trem % a.tiles_x) * 8;
    const int n0 = blockIdx.y * 64;
    const float* in_b = G.in + (size_t)bimg * H * W * a.lda + c0 * CK;
    float biasv[1];
    conv_load_bias<C>(biasv, G.bias, n0, wn, li);

    // this lane's pixel (MFMA row m <-> pixel of 2x2 window m >> 2, as in every other kernel: the pool happens in registers)
    int a_base;
    {
        const int m = wm * 32 + li, q = m >> 2, r = m & 3;
        const int py = 2 * (q / (TW / 2)) + (r >> 1), px = 2 * (q % (TW / 2)) + (r & 1);
        a_base = (py * C::HALO_W + px) * PITCH + kh * 16;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    const unsigned b_off = (unsigned)((n0 + wn * 32 + li) * 32 + kh * 16);
    const unsigned plane_b = (unsigned)a.cout_pad * 32u, panel_b = 3u * plane_b, tap_b = panel_b * (unsigned)a.nch;

    int h_goff[NHF], h_lds[NHF];
    unsigned h_ok = 0;
#pragma unroll
    for (int r = 0; r < NHF; ++r) {
        const int f = tid + r * 256;
        const bool slot = f < C::HALO_H * C::HALO_W * (CK / 4);
        const int hp = slot ? f / (CK / 4) : 0, c4 = f % (CK / 4);
        const int hy = hp / C::HALO_W, hx = hp - hy * C::HALO_W;
        const int gy = y0 + hy - C::PADK, gx = x0 + hx - C::PADK;
        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        h_goff[r] = (cy * W + cx) * a.lda + c4 * 4;
        h_lds[r] = slot ? hp * PITCH + c4 * 8 : -1;
        h_ok |= (slot && inb) ? (1u << r) : 0u;
    }
    auto halo_store = [&](char* buf, const float4 (&hv)[NHF]) {
#pragma unroll
        for (int r = 0; r < NHF; ++r) {
            float4 v = hv[r];
            if (!((h_ok >> r) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h_lds[r] >= 0) split3_store(buf + h_lds[r], v);
        }
    };

    f32x16 acc[1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][0][i] = 0.f;

    // weight ring: tap t of the running tap sequence sits in slot t % RB
    f32x4 bw[RB][3];
    auto load_b = [&](f32x4 (&dst)[3], unsigned soff) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off, soff + pl * plane_b, 0));
    };
    const unsigned first_b = (unsigned)c0 * panel_b;
#pragma unroll
    for (int i = 0; i < RB - 1; ++i) load_b(bw[i], first_b + (unsigned)i * tap_b);      // taps 0 .. RB-2 of the first chunk (T >= RB - 1)
    {
        float4 hv[NHF];
#pragma unroll
        for (int r = 0; r < NHF; ++r) hv[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r]);
        halo_store(s_in, hv);
    }
    __syncthreads();

    f32x4 ar[RA][3];
    for (int ch = c0; ch < c1; ++ch) {
        const char* cur = s_in + ((ch - c0) & 1) * IN_BYTES;
        char* nxt = s_in + ((ch - c0 + 1) & 1) * IN_BYTES;
        const bool more_ch = ch + 1 < c1;
        float4 hreg[NHF];
        {
            const int cn = (more_ch ? ch + 1 : ch) - c0;
#pragma unroll
            for (int r = 0; r < NHF; ++r) hreg[r] = *reinterpret_cast<const float4*>(in_b + h_goff[r] + cn * CK);
        }
#pragma unroll
        for (int i = 0; i < RA - 1; ++i) {          // taps 0 .. RA-2 of this chunk
            const int toff = ((i / KS) * C::HALO_W + i % KS) * PITCH;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ar[i][pl] = *reinterpret_cast<const f32x4*>(cur + a_base + toff + pl * 32);
        }
        const unsigned chunk_b = (unsigned)ch * panel_b;
        // T % RB taps shift the ring position from chunk to chunk; the ring index is kept compile-time by rotating the
        // registers at the chunk end (3 * (T % RB) moves per chunk)
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            {   // weights RB - 1 taps ahead: a later tap of this chunk, or an early tap of the next chunk (or harmlessly this one again)
                const int tn = tap + RB - 1;
                unsigned so;
                if (tn < T) so = chunk_b + (unsigned)tn * tap_b;
                else so = (more_ch ? chunk_b + panel_b : chunk_b) + (unsigned)(tn - T) * tap_b;
                load_b(bw[tn % RB], so);
            }
            {   // A fragments RA - 1 taps ahead
                const int tx = tap + RA - 1 < T ? tap + RA - 1 : T - 1;
                const int toff = ((tx / KS) * C::HALO_W + tx % KS) * PITCH;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ar[(tap + RA - 1) % RA][pl] = *reinterpret_cast<const f32x4*>(cur + a_base + toff + pl * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 (&A)[3] = ar[tap % RA];
            const f32x4 (&Bf)[3] = bw[tap % RB];
            f32x16& c = acc[0][0];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[1]), __builtin_bit_cast(bf16x8, Bf[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2]), __builtin_bit_cast(bf16x8, Bf[0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, Bf[2]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[1]), __builtin_bit_cast(bf16x8, Bf[0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, Bf[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, Bf[0]), c, 0, 0, 0);
        }
        // rotate the weight ring so that the next chunk's tap 0 is in slot 0 again: slot (T + i) % RB -> slot i
        if (T % RB) {
            f32x4 tmp[RB][3];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) tmp[i][pl] = bw[(T + i) % RB][pl];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bw[i][pl] = tmp[i][pl];
        }
        if (more_ch) {
            halo_store(nxt, hreg);
            __syncthreads();
        }
    }
    conv_epilogue<C, TW>(acc, biasv, a, G.out, G.cout, bimg, y0, x0, n0, wm, wn, li, kh);
}

// ---- Winograd F(2x2, 3x3) (fp32, option "conv_algo") ------------------------------------------------------------------
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A: a 2 x 2 output tile from a 4 x 4 input window costs 16 multiplies per channel pair instead of
// 36 -> 2.25x less matrix work; the transforms are additions only (B^T, A^T) or done once on the host (G g G^T, in double, rounded
// to fp32).  Block = 32 Winograd tiles (4 rows x 8 columns of 2 x 2 = an 8 x 16 pixel output tile) x 128 output channels; wave w owns
// 32 channels and all 16 "frequencies": 16 accumulator tiles of 32 (Winograd tiles) x 32 (channels) = 256 AGPRs, one block per CU.
// Everything runs as PHASES of 8 planes x 4 k8-steps x 4 MFMAs per wave: while a phase multiplies the 8 planes in one half of the
// U[plane][tile][channel] LDS buffer, every thread transforms its (tile, 4 channels) item of the raw halo for the next phase into the
// other half, one LDS / VALU instruction per slot between two MFMAs; the transformed weights stream from L2 in a register ring 32 MFMAs
// ahead (pinned with sched_barrier: left alone the compiler sinks the loads to one step ahead and the single wave per SIMD stalls on
// L2); one s_barrier per phase (eight MFMAs before its end: see p1_step).  Epilogue: A^T M A per lane (the 16 frequencies of a (tile, channel) sit in one lane's registers),
// bias, ReLU, 2x2 max-pool = max over the tile's four outputs.
// KS = 7 (the 7x7 layers of stages 2-6), 100 instead of 196 products per tile and channel pair: pass 1 -- the taps (0..5, 0..5) are four
// 3x3 sub-kernels, each a Winograd product on its own shifted window, all four accumulated in the SAME frequency-domain accumulators
// (the output transform is linear: 4 x 16); pass 2a -- row 6 as two 1x3 sub-kernels, 1-D F(2,3) along x (2 x 8), and tap (6, 6) direct
// (4); pass 2b -- column 6 as two 3x1 sub-kernels along y (2 x 8).  The raw halo of a 32-channel chunk is staged once per pass, round-robin.
// UNIT = 1 (single images): a block runs one unit (pass 1 over a chunk range / row 6 / column 6 / tap (6, 6)) and writes its share of y to
// a slab; conv_splitk_reduce_kernel adds the slabs in unit order.
// The arithmetic is DEFINED -- transform additions in a fixed order, one sequential FMA chain per plane over (chunk, sub-kernel, k8-step,
// k), output transforms in a fixed order, units added in order -- and oracle/conv_fma_ref.c::conv_wino_ref restates it bit for bit; it
// is not the direct kernels' chain (results agree to fp32 rounding, ~1e-6 of the map scale).
// float4 add / subtract as two packed-fp32 instructions (v_pk_add_f32, the subtrahend negated by the source modifier: same rounding as
// v_sub_f32); the scheduler-pinned one-op-per-slot transform code otherwise compiles to four scalar VALU instructions per float4
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_add2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// Diagnostic build only (tools/block_timing.py compiles this file with -DPMX_BLOCK_TIMING into its own library; the product library
// never defines it): thread 0 of the first 8192 blocks of a Winograd launch stamps the 100 MHz wall clock at entry / pipeline primed /
// before the stores / exit, and the CU it runs on.  (The stamps perturb the register allocation of the loops -- a 7x7 block runs 1.4x
// slower in that build -- so only the prologue, the epilogue and the hand-over gap between two blocks on a CU are read off it.)
#ifdef PMX_BLOCK_TIMING
__device__ unsigned long long g_blk_t[8192 * 8];
extern "C" int pmx_debug_block_times(unsigned long long* out, size_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_t), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#define PMX_T(k) do { if (threadIdx.x == 0) { const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
                                              if (lin_ < 8192) g_blk_t[lin_ * 8 + (k)] = (k) == 7 ? (unsigned long long)__smid() : wall_clock64(); } } while (0)
#else
#define PMX_T(k)
#endif

// GEOM 0: a block = 4 x 8 Winograd tiles = an 8 x 16 pixel output rectangle (any map size).
// GEOM 1 ("runs", 46-pixel-wide maps = the 46 x 46 maps of a 368 x 368 input): a block = 32 CONSECUTIVE Winograd tiles in row-major order
// of the 23-tile-wide grid of one image (at most 3 tile rows): a 46 x 46 map is 529 tiles = 16 full blocks + 17 tiles instead of 18
// rectangles (8.9 % padding), and the 16 full blocks of 32 images x 2 branch groups are exactly 4 rounds of 256 CUs; the part-filled last
// block of every image runs in unit mode (pmx_api.hip::run_conv).  Raw halo = the 6 + KS - 1 input rows the three tile rows touch x all
// 46 + KS - 1 columns (7x7: 12 x 52 pixels x 32 channels = 90 KB next to the 74 KB of U: 256 bytes short of the 160 KB LDS).
// GEOM 3 ("merged tails", unit mode only): the part-filled last blocks of ALL images of the launch as one stream -- image b's tail tiles
// [32 nfull, ntiles) (they lie in one tile row; nt of them, 16 <= nt <= 23) are the stream positions [b nt, (b + 1) nt), and block j owns the
// positions [32 j, 32 j + 32): up to three images' segments, every MFMA row a real tile (46 x 46: 17 tiles per image -- one image per block
// filled 17 of the 32 rows).  Raw halo = one tile row, the segments side by side, each with its own KS - 1 columns of overlap.
#ifndef PMX_WINO_LDR3
#define PMX_WINO_LDR3 48
#endif
// how far ahead of their MFMAs the transformed weights are requested: pass 1 in steps of 4 MFMAs (ring of 16), pass 2 in steps of 8 (ring
// of 8).  8 / 4 = ~2000 cycles; 10 / 5, 12 / 6 and 15 / 7 measured 0.5 - 3 % slower on the 7x7 layers (tools/kernel_variants.py): the
// weight stream is not what the matrix pipe waits for
#ifndef PMX_WINO_WLEAD1
#define PMX_WINO_WLEAD1 8
#endif
#ifndef PMX_WINO_WLEAD2
#define PMX_WINO_WLEAD2 4
#endif
// Diagnostic builds only (tools/kernel_variants.py; the product library is built with 0): leave out parts of the phases' side work to see
// what the matrix pipe waits for -- 1: the transform slots (LDS reads of the raw halo, B^T d B, U stores, halo staging), 2: the weight
// loads, 4: the A-fragment LDS reads, 8: the barriers inside the phases; pass 1 only: 16: the raw-halo LDS reads of the transform, 32: its
// VALU work, 64: its U stores, 128 / 256: the halo staging's LDS stores / global loads.  The results are wrong; only the launch time is read.
#ifndef PMX_ABLATE
#define PMX_ABLATE 0
#endif
#ifndef PMX_WINO_HOFF3
#define PMX_WINO_HOFF3 0
#endif
#ifndef PMX_WINO_SOFF
#define PMX_WINO_SOFF 1
#endif
// The transform's packed adds in ONE gap per group (1) instead of two per gap (0, the default).  tools/mfma_gap_probe.hip
// (profiles/r04_mfma_gap_probe.json): LDS reads, buffer loads and scalar instructions between two MFMAs of the single wave on a SIMD are
// free, but a v_pk_add_f32 is not -- a gap that holds VALU work costs ~3.2 ns of matrix-pipe time once plus ~2.2 ns per instruction (2 in
// every 2nd gap: 3.3 ns each, 8 in every 8th: 2.6 ns each).  In the kernel the clustered schedule (bit-identical, 60 Winograd tests) measured
// +0.5 % on the 7x7 layers and -1.3 % on conv4_2 (profiles/r04_wino_ablation.json "vcluster"): the cluster waits for all twelve raw-halo
// reads at once where the spread schedule waits for two -- not adopted
#ifndef PMX_WINO_VCLUSTER
#define PMX_WINO_VCLUSTER 0
#endif
static_assert(PMX_WINO_WLEAD1 >= 4 && PMX_WINO_WLEAD1 <= 15 && PMX_WINO_WLEAD2 >= 4 && PMX_WINO_WLEAD2 <= 7, "weight ring lead");
template <int KS, int GEOM>
struct WinoCfg {
    static constexpr int TH = 8, TW = 16, PADK = KS / 2, CKW = 32, LDU = CKW + 4;
    // raw-halo pixel pitch (floats).  36 (7x7: all the LDS allows): a transform read of 16 lanes covers two tiles 2 pixels = 72 floats
    // apart -> their 128-byte rows overlap in 24 of 64 banks (PMC: 25 % of the LDS cycles are bank conflicts).  3x3: the halo is small
    // enough for a pitch of 48 -> 2 pixels = 96 floats = 32 banks apart, no overlap
    // (GEOM 3, 7x7: 8 x 82 pixels only fit with a pitch of 32 -- two tiles of a transform read then share their banks: 2-way conflicts,
    //  on a launch that is 3 % of a layer)
    static constexpr int LDR = (KS == 3 && PMX_WINO_LDR3 > 0) ? PMX_WINO_LDR3 : GEOM == 3 ? CKW : CKW + 4;
    static constexpr int RUN_TX = PMX_WINO_RUN_TX, RUN_W = 2 * RUN_TX;
    // GEOM 3 ("merged tails"): one tile row of up to three images side by side: 32 tiles + three times the KS - 1 columns of overlap
    static constexpr int HH = GEOM == 3 ? 2 + KS - 1 : GEOM ? 6 + KS - 1 : TH + KS - 1;
    static constexpr int HW = GEOM == 3 ? 2 * PMX_WINO_RUN_TILES + 3 * (KS - 1) : GEOM ? RUN_W + KS - 1 : TW + KS - 1, NPX = HH * HW;
    static constexpr int NSUB = KS == 3 ? 1 : 4;                       // 3x3 sub-kernels done as Winograd products
    static constexpr int NDIR = KS == 3 ? 0 : 13;                      // taps outside the 3x3 sub-kernels (KS = 7: row 6, column 6 -> pass 2)
    static constexpr int RAW_ELEMS = NPX * LDR, U_ELEMS = 16 * 32 * LDU;
    static constexpr int LDS_BYTES = (RAW_ELEMS + U_ELEMS) * 4;
    static constexpr int NHF = (NPX * (CKW / 4) + 255) / 256;
    static_assert(KS == 3 || KS == 7, "Winograd kernel: 3x3 or 7x7");
    static_assert(LDS_BYTES <= 160 * 1024, "Winograd kernel: LDS");
};

template <int KS, int POOL, int UNIT, int GEOM>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a)
{
    using C = WinoCfg<KS, GEOM>;
    static_assert(!(POOL && KS != 3), "pooling only with the 3x3 variant");
    static_assert(!(UNIT && POOL), "unit mode: the combine kernel pools");
    static_assert(GEOM != 3 || UNIT, "merged tails run in unit mode");
    constexpr bool MERGE = GEOM == 3;
    // the transform's packed adds clustered into one gap per group (PMX_WINO_VCLUSTER): the run-geometry and merged-tail forms only -- on
    // the rectangle and multi-slab forms the register allocator answers the clusters with 3.4 KB of scratch per lane
    constexpr bool VCL = PMX_WINO_VCLUSTER && (GEOM == 1 || GEOM == 3);
    // UNIT (single images: 36 blocks of a 46x46 7x7 layer cannot fill 256 CUs): blockIdx.z = unit * groups + group, and a block runs
    // ONE unit of the work -- unit u < nu1: pass 1 over the chunks [u g, u g + g) (g = a.kbounds); 7x7: unit nu1: row 6 (pass 2a without
    // tap (6, 6)); unit nu1 + 1: column 6 (pass 2b); unit nu1 + 2: tap (6, 6) -- and writes its untransformed share of y (no bias / ReLU) to slab `unit`; conv_splitk_reduce_kernel adds the slabs in unit order
    extern __shared__ float4 smem4[];
    PMX_T(0); PMX_T(7);
    float* const s_raw = reinterpret_cast<float*>(smem4);
    float* const s_u = s_raw + C::RAW_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int unit = UNIT ? (int)blockIdx.z / a.ngroups : 0;
    const bool g1 = (UNIT ? (int)blockIdx.z % a.ngroups : (int)blockIdx.z) != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;              // transformed weights [plane][chunk32][k8-step][cout_pad][8] (pmx_api.hip::pack_wino)
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = g1 ? a.g[1].out : a.g[0].out;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    if (UNIT) G.out += (size_t)unit * (size_t)a.slab_stride;
    const int H = a.H, W = a.W;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // GEOM 1: the map is cut into vertical slabs of 46 columns (23 tile columns; 46 / 92 / 184-wide maps = 1 / 2 / 4 slabs); block trem of
    // this launch in (image, slab) bslab = the 32 consecutive Winograd tiles [t0, t0 + 32) of that slab (row-major), tile rows r0 .. r0 + 2;
    // raw halo row 0 / column 0 = image row 2 r0 - PADK / column 46 slab - PADK (halo columns inside the map come from the neighbour slab)
    const int tiles_per_img = MERGE ? 1 : GEOM ? a.run_nb : a.tiles_x * a.tiles_y;
    const int bslab = tile / tiles_per_img;
    const int trem = tile - bslab * tiles_per_img;
    // (GEOM 1 = a single slab, the 46-wide maps of the 7x7 layers: the slab arithmetic is compiled out -- its extra scalar registers
    //  pushed the 7x7 kernel's transition code into 30 more spill reloads per block, +3 %; GEOM 2 = any number of slabs)
    constexpr bool SLABS = GEOM == 2;
    const int bimg = SLABS ? bslab / a.run_nslab : bslab;
    const int sx0 = SLABS ? (bslab - bimg * a.run_nslab) * C::RUN_W : 0;
    const int t0 = MERGE ? a.run_j0 * PMX_WINO_RUN_TILES : GEOM ? (a.run_j0 + trem) * PMX_WINO_RUN_TILES : 0;      // (MERGE: first tail tile of an image)
    const int r0 = GEOM ? t0 / C::RUN_TX : 0;
    const int ntiles = C::RUN_TX * ((a.H + 1) >> 1);
    const int y0 = GEOM ? 2 * r0 : (trem / a.tiles_x) * C::TH, x0 = GEOM ? sx0 : (trem % a.tiles_x) * C::TW;
    // MERGE: block `tile` = stream positions [32 tile, 32 tile + 32) = segment s (s = 0, 1, 2) of image mg_img0 + s: mg_n0 / mg_n1 / the
    // rest tiles from tail tile mg_tt0 (s = 0) / 0 on, halo columns from 0 / mg_cb1 / mg_cb2 on (2 n + KS - 1 of them)
    const int mg_nt = ntiles - t0, mg_tx0 = t0 - r0 * C::RUN_TX;
    const int mg_p0 = tile * PMX_WINO_RUN_TILES, mg_img0 = MERGE ? mg_p0 / mg_nt : 0, mg_tt0 = mg_p0 - mg_img0 * mg_nt;
    const int mg_n0 = min(mg_nt - mg_tt0, PMX_WINO_RUN_TILES), mg_n1 = min(mg_nt, PMX_WINO_RUN_TILES - mg_n0);
    const int mg_cb1 = 2 * mg_n0 + KS - 1, mg_cb2 = mg_cb1 + 2 * mg_n1 + KS - 1;
    const int n0 = blockIdx.y * 128;
    const int n = n0 + wave * 32 + li;
    const float* in_b = G.in + (MERGE ? (size_t)0 : (size_t)bimg * H * W * a.lda);
    float bias = G.bias[n];                       // (pinned to a register further down, once the first halo loads are on their way:
                                                  //  pinned here the block waited a full memory round trip before issuing anything else)
    const int nch = a.nch;                        // chunks of 32 input channels
    const int ug = UNIT ? (int)a.kbounds : nch;   // chunks per pass-1 unit
    const int nu1 = UNIT ? (nch + ug - 1) / ug : 1;
    const int c0 = UNIT ? min(unit, nu1 - 1) * ug : 0;                     // pass-1 chunk range of this block
    const int c1 = UNIT ? min(nch, c0 + ug) : nch;
    const bool do_p1 = !UNIT || unit < nu1, do_p2a = !UNIT || unit == nu1, do_p2b = !UNIT || unit == nu1 + 1;
    const bool do_pd = UNIT && unit == nu1 + 2;   // unit mode: tap (6, 6) is a unit of its own (in pass 2a otherwise)

    // raw halo staging: slot r of a thread = pixel (tid >> 3) + 32 r of the halo, channels 4 (tid & 7) .. + 3 of the chunk.
    // GEOM 0: global offsets and an in-image mask per slot in registers (the LDS offset is recomputed at the write).
    // GEOM 1 (20 slots for 7x7): nothing per slot is kept -- the loads go through a buffer resource that spans exactly this image, so
    // rows above / below the map fall out of its range and return 0; columns left / right of it get an out-of-range offset
    int h_goff[GEOM ? 1 : C::NHF];
    unsigned h_ok = 0;
    if (!GEOM) {
#pragma unroll
        for (int r = 0; r < (GEOM ? 0 : C::NHF); ++r) {
            const int f = tid + r * 256;
            const bool slot = f < C::NPX * (C::CKW / 4);
            const int hp = slot ? f / (C::CKW / 4) : 0, c4 = f % (C::CKW / 4);
            const int hy = hp / C::HW, hx = hp - hy * C::HW;
            const int gy = y0 + hy - C::PADK, gx = x0 + hx - C::PADK;
            const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            h_goff[r] = (cy * W + cx) * a.lda + c4 * 4;
            h_ok |= (slot && inb) ? (1u << r) : 0u;
        }
    }
    // (MERGE: the resource spans the whole batch -- rows outside an image would land in its neighbour, so they are masked like the columns)
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0,
                                                                           MERGE ? (unsigned)(a.B * H * W * a.lda) * 4u : GEOM ? (unsigned)(H * W * a.lda) * 4u : 0u, 0x00020000);
    const int hbase_b = (((y0 - C::PADK) * W + x0 - C::PADK) * a.lda + (tid & 7) * 4) * 4;     // byte offset of halo pixel (0, 0), may be negative
    const int hrow_skip = SLABS ? W - C::HW : -(KS - 1);                                     // image pixels between the end of a halo row and the next
    const int lda_b = a.lda * 4;
    // GEOM 1 / 2: byte offset of slot r's pixel in the image for chunk 0, computed ONCE with the first halo load (0x80000000 = the column is
    // outside the map: stays out of the buffer's range whatever chunk offset is added) and kept in registers -- recomputed per use, the
    // compiler hoisted a second copy of this arithmetic (20 slots x (mul_hi, mul_lo, mad, cmp)) to right in front of the first MFMA
    // (7x7 only: on the 3x3 instantiations the kept offsets measured slower -- conv3_3 +6 % -- than the compiler's own placement)
    constexpr bool HOFF = (GEOM != 0 && (KS == 7 || PMX_WINO_HOFF3)) || GEOM == 3;     // (merged tails: the per-slot segment arithmetic is never repeated)
    int h_off[HOFF ? C::NHF : 1];
    auto halo_off_calc = [&](int r) -> int {
        const unsigned hp = (unsigned)(tid >> 3) + 32u * r;
        const unsigned hy = hp / (unsigned)C::HW, hx = hp - hy * (unsigned)C::HW;
        if constexpr (MERGE) {
            const int sg = (int)hx >= mg_cb2 ? 2 : (int)hx >= mg_cb1 ? 1 : 0;
            const int lx = (int)hx - (sg == 2 ? mg_cb2 : sg == 1 ? mg_cb1 : 0);
            const int ns = sg == 2 ? PMX_WINO_RUN_TILES - mg_n0 - mg_n1 : sg == 1 ? mg_n1 : mg_n0;
            const int img = mg_img0 + sg;
            const int gy = y0 - C::PADK + (int)hy, gx = 2 * (mg_tx0 + (sg == 0 ? mg_tt0 : 0)) - C::PADK + lx;
            const bool ok = lx < 2 * ns + KS - 1 && img < a.B && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && hy < (unsigned)C::HH;
            return ok ? (((img * H + gy) * W + gx) * a.lda + (tid & 7) * 4) * 4 : (int)0x80000000;
        }
        // halo pixel (hy, hx) = image pixel (y0 - PADK + hy, x0 - PADK + hx): hp + (W - HW) hy pixels after halo pixel (0, 0) in the image
        int off = hbase_b + ((int)hp + (int)hy * hrow_skip) * lda_b;
        if (SLABS ? (unsigned)(x0 - C::PADK) + hx >= (unsigned)W          // (left of the map the sum wraps around: also out)
                  : hx - (unsigned)C::PADK >= (unsigned)C::RUN_W) off = (int)0x80000000;
        return off;
    };
    auto halo_off_init = [&](int r) { if constexpr (HOFF) h_off[r] = halo_off_calc(r); };
    auto halo_load_slot = [&](float4 (&hv)[C::NHF], int chunk, int r) {       // r is a compile-time constant at every call
        if constexpr (GEOM != 0) {
            const int off0 = HOFF ? h_off[HOFF ? r : 0] : halo_off_calc(r);
            // (the chunk's byte offset goes into the scalar offset, which the range check ignores: a pixel outside the image stays out of
            //  range, a pixel inside it stays inside its own channel row -- one VALU add less per load)
            if (PMX_WINO_SOFF) hv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off0, chunk * (C::CKW * 4), 0));
            else hv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off0 + chunk * (C::CKW * 4), 0, 0));
        } else {
            hv[r] = *reinterpret_cast<const float4*>(in_b + h_goff[GEOM ? 0 : r] + chunk * C::CKW);
        }
    };
    float* const s_raw_t = s_raw + (tid >> 3) * C::LDR + (tid & 7) * 4;       // slot r of this thread: + r * 32 * LDR floats (an immediate offset)
    auto halo_store_slot = [&](const float4 (&hv)[C::NHF], int r) {
        const int f = tid + r * 256;
        float4 v = hv[r];
        if (!GEOM && !((h_ok >> r) & 1)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        // (only the last slot can fall behind the halo; spelled out because the compiler does not bound tid by the block size)
        if (r * 256 + 255 < C::NPX * (C::CKW / 4) || f < C::NPX * (C::CKW / 4)) *reinterpret_cast<float4*>(s_raw_t + r * (32 * C::LDR)) = v;
    };
    auto halo_load = [&](float4 (&hv)[C::NHF], int chunk) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) halo_load_slot(hv, chunk, r);
    };
    auto halo_store = [&](const float4 (&hv)[C::NHF]) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) halo_store_slot(hv, r);
    };
    // position of Winograd tile m of the block in the raw halo (top-left pixel of sub-kernel 0's 4 x 4 window), in pixels.
    // GEOM 0: 4 x 8 grid; GEOM 1: tile t0 + m of the row-major run (tiles past the end of the map repeat the last one; never stored)
    auto tile_px = [&](int m) -> int {
        if (MERGE) {         // (positions past the end of the stream repeat the last tile of the last image; never read back)
            const int mc = min(m, a.B * mg_nt - 1 - mg_p0);
            const int sg = mc >= mg_n0 + mg_n1 ? 2 : mc >= mg_n0 ? 1 : 0;
            return (sg == 2 ? mg_cb2 - 2 * (mg_n0 + mg_n1) : sg == 1 ? mg_cb1 - 2 * mg_n0 : 0) + 2 * mc;
        }
        if (GEOM) {
            const int t = min(t0 + m, ntiles - 1);
            const int ty = t / C::RUN_TX;
            return (2 * (ty - r0)) * C::HW + 2 * (t - ty * C::RUN_TX);
        }
        return (2 * (m >> 3)) * C::HW + 2 * (m & 7);
    };
    // transform item of this thread: Winograd tile tt, channels 4 * tc .. + 3 of the chunk
    const int tt = tid >> 3, tc = tid & 7;
    const int t_raw = tile_px(tt) * C::LDR + tc * 4;
    const int t_u = tt * C::LDU + tc * 4;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    // weight panels are [plane][chunk32][k8-step 4][cout_pad][8]: the 64 lanes of one fragment load (32 channels x 2 halves x 16 B)
    // read 1 KB of contiguous, fully used cache lines (with the channel-major [cout_pad][32] layout each load touched 32 lines and used a
    // quarter of each, relying on the 32 KB L1 to keep them for the next three k8-steps -- it did not: weight loads cost 7.5 %)
    // (PMX_WINO_WLAYOUT 1: [plane][chunk32][cout_pad / 32][k8-step 4][32][8] -- the k8-steps of this wave's 32 channels are 1 KB apart: a
    //  constant on the vector offset = the load's immediate offset, no scalar add per load)
    const unsigned b_off = PMX_WINO_WLAYOUT ? (unsigned)(((n >> 5) * 1024 + (n & 31) * 8 + kh * 4) * 4) : (unsigned)((n * 8 + kh * 4) * 4);
    const unsigned st_b = PMX_WINO_WLAYOUT ? 0u : (unsigned)a.cout_pad * 8u * 4u;      // bytes between the k8-steps of a panel (in the scalar offset)
    constexpr unsigned st_v = PMX_WINO_WLAYOUT ? 1024u : 0u;                               // ... (in the vector offset)
    const unsigned panel_b = (unsigned)a.cout_pad * C::CKW * 4u;          // bytes of one (plane, chunk) panel
    const unsigned freq_b = panel_b * (unsigned)nch;                       // bytes between planes (sub-kernel * 16 + frequency)

    f32x16 acc[16];
    // the 256 accumulator registers are zeroed in the shadow of the first halo / weight loads (left to the compiler the 256
    // v_accvgpr_write sat right in front of the first MFMA, after both prologue barriers: ~0.5 us per block, fully exposed)
    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < 16; ++f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
            asm volatile("" : "+a"(acc[f]));
        }
    };

    // ---- pass 1, software pipeline.  A (chunk, sub-kernel) step is two PHASES of 8 frequencies x 4 k8-steps x 4 MFMAs: phase 0 runs
    // the frequencies 0..7 (rows 0, 1 of V, U half 0) while the threads transform rows 2, 3 of the same window into U half 1; phase 1
    // runs the frequencies 8..15 while they transform rows 0, 1 of the NEXT step's window into U half 0 (when that step starts a new
    // chunk, the raw halo is replaced first: registers -> LDS, one extra barrier).  Every LDS / VALU instruction of the transform
    // sits in a fixed slot between two MFMAs (36 of the 64 slots of a phase), so the matrix pipe never waits for it; one barrier per
    // phase.  Weight fragments: ring of 16 steps, loaded 8 steps (32 MFMAs, ~2000 cycles) ahead across phase, sub-kernel and chunk
    // boundaries (left alone, the compiler sinks the loads to one step ahead and the single wave per SIMD stalls on L2).
    float4 hreg[C::NHF];
    const int a_off = li * C::LDU + kh * 4;
    f32x4 bw[16];
    if (do_p1) {
#pragma unroll
    for (int r = 0; r < C::NHF; ++r) {               // first halo: offsets computed and loads issued slot by slot
        if (GEOM) halo_off_init(r);
        halo_load_slot(hreg, c0, r);
    }
#pragma unroll
    for (int st8 = 0; st8 < PMX_WINO_WLEAD1; ++st8)   // the first steps of the first phase: frequencies 0, 1, .. (x 4 k8-steps) of plane 0
        bw[st8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)(st8 & 3) * st_v, (unsigned)c0 * panel_b + (unsigned)(st8 >> 2) * freq_b + (unsigned)(st8 & 3) * st_b, 0));
    __builtin_amdgcn_sched_barrier(0);
    zero_acc();
    __builtin_amdgcn_sched_barrier(0);
    halo_store(hreg);
    asm volatile("" : "+v"(bias));
    __syncthreads();
    if (c1 - c0 > 1) {
        halo_load(hreg, c0 + 1);
    }
    {   // rows 0, 1 of the first window (not overlapped)
        f32x4 wv4[2][4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (0 * C::HW + jx) * C::LDR]);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (1 * C::HW + jx) * C::LDR]);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (2 * C::HW + jx) * C::LDR]);
            wv4[0][jx] = d0 - d2;
            wv4[1][jx] = d1 + d2;
        }
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 0) * 32 * C::LDU + t_u]) = wv4[il][0] - wv4[il][2];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 1) * 32 * C::LDU + t_u]) = wv4[il][1] + wv4[il][2];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 2) * 32 * C::LDU + t_u]) = wv4[il][2] - wv4[il][1];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 3) * 32 * C::LDU + t_u]) = wv4[il][1] - wv4[il][3];
        }
    }
    __syncthreads();
    // A fragments of the first two steps of the first phase; every phase requests those of the phase after it (steps 30, 31)
    f32x4 av[4];
    av[0] = *reinterpret_cast<const f32x4*>(&s_u[a_off]);
    av[1] = *reinterpret_cast<const f32x4*>(&s_u[a_off + 8]);

    // one (chunk, sub-kernel) step = the two phases.  LAST (compile time): the last sub-kernel of a chunk, whose second phase transforms the
    // first window of the NEXT chunk -- the raw halo is replaced in between, spread over the free side slots so that the matrix pipe never
    // waits for it: phase 0 reads the old halo in slots 2..13, then one barrier (slot 14: every wave is done with the old halo) and one
    // ds_write_b128 of the new halo per slot from slot 40 on (the phase's barrier in step 30 publishes it); phase 1 issues one global
    // load of the chunk after next per slot from slot 40 on.  (Before: 10 / 20 stores + a barrier + the loads in one clump in slot 0 of
    // phase 1, exposed: +4 % per block with the 12 x 52 halo of the run geometry.)
    auto p1_step = [&](auto sub_c, int ch, bool more, unsigned chunk_b, unsigned next_b) {
        constexpr int sub = decltype(sub_c)::value;
        constexpr bool LAST = sub == C::NSUB - 1;
        constexpr int sub_n = LAST ? 0 : sub + 1;
        const bool repl = LAST && ((C::NDIR > 0 && !UNIT) ? true : more);      // chunks are staged round-robin over the passes (0 .. nch-1, then 0 ..
        int cn = ch + 2;                                                        // again for pass 2a, 2b): the registers hold the chunk after next
        if (UNIT) cn = cn < c1 ? cn : c1 - 1;
        if (cn >= nch) cn -= nch;
        if (cn >= nch) cn -= nch;
        const unsigned plane_b = chunk_b + (unsigned)(sub * 16) * freq_b;                       // plane sub * 16 + 0 of this chunk
        const unsigned nplane_b = (LAST ? next_b : chunk_b) + (unsigned)(sub_n * 16) * freq_b;     // plane 0 of the next step
        const int src_cur = t_raw + ((3 * (sub >> 1)) * C::HW + 3 * (sub & 1)) * C::LDR;
        const int src_nxt = t_raw + ((3 * (sub_n >> 1)) * C::HW + 3 * (sub_n & 1)) * C::LDR;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // side work of this phase: rows (2, 3) of the current window (r = 0) / rows (0, 1) of the next one (r = 1)
            const int q = r ^ 1;                                        // V row pair produced
            const int src = (r == 0 ? src_cur : src_nxt) + q * C::HW * C::LDR;      // d rows q .. q + 2
            float* const udst = s_u + (q * 8) * 32 * C::LDU + t_u;
            f32x4 dd[3][4], wv[2][4], vv, vvs[8];
#pragma unroll
            for (int s = 0; s < 32; ++s) {                              // step = (frequency r * 8 + s / 4, k8-step s % 4)
                const int f = r * 8 + (s >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 3][e], bw[s & 15][e], acc[f], 0, 0, 0);
                    if (e == 0) {                                       // weights of step s + lead
                        const int sn = s + PMX_WINO_WLEAD1;
                        unsigned so;
                        if (sn < 32) so = plane_b + (unsigned)(r * 8 + (sn >> 2)) * freq_b;
                        else if (r == 0) so = plane_b + (unsigned)(8 + ((sn - 32) >> 2)) * freq_b;
                        else so = nplane_b + (unsigned)((sn - 32) >> 2) * freq_b;
                        if (!(PMX_ABLATE & 2))
                        bw[sn & 15] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)(sn & 3) * st_v, so + (unsigned)(sn & 3) * st_b, 0));
                        // THE barrier of the phase sits here, eight MFMAs before its end: U half q is complete (its last store is in slot
                        // 39; LAST, r = 0: and the new raw halo, slot 40 + NHF - 1 <= 60), every wave has read all it needs of U half r (the
                        // fragments of steps 30, 31 were requested at steps 28, 29).  The MFMAs that follow have their operands in
                        // registers, and the next phase's first two fragments are requested behind it -- at the phase boundary itself
                        // nothing waits (with the barrier there, the first MFMA of every phase waited for the barrier AND an LDS read)
                        if (s == 30 && !(PMX_ABLATE & 8)) __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (e == 1) {                                // A fragment of step s + 2 (steps 30, 31: of the next phase's steps 0, 1)
                        const int fn = s + 2 < 32 ? r * 8 + ((s + 2) >> 2) : q * 8, sn = (s + 2) & 3;
                        if (!(PMX_ABLATE & 4))
                        av[(s + 2) & 3] = *reinterpret_cast<const f32x4*>(&s_u[fn * 32 * C::LDU + a_off + sn * 8]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (!(PMX_ABLATE & 1)) {                     // transform slot t
                        const int t = 2 * s + (e - 2);
                        if (t >= 2 && t < 14) {                         // 12 reads: d rows q .. q + 2, column by column
                            const int jx = (t - 2) / 3, ri = (t - 2) % 3;
                            if (PMX_ABLATE & 16) asm volatile("" : "=v"(dd[ri][jx]));
                            else
                            dd[ri][jx] = *reinterpret_cast<const f32x4*>(&s_raw[src + (ri * C::HW + jx) * C::LDR]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (LAST && t == 14 && r == 0) {         // every wave has read what it needs of the old halo
                            if (repl) __syncthreads();
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (VCL && (t == 16 || t == 24)) {
                            // the 16 packed adds of B^T d (slot 16) / of (.) B (slot 24) in one gap each; the eight U stores follow one per slot
#pragma unroll
                            for (int k8 = 0; k8 < 8; ++k8) {
                                if (t == 16) {
                                    const int jx = k8 >> 1, wi = k8 & 1;
                                    if (PMX_ABLATE & 32) asm volatile("" : "=v"(wv[wi][jx]));
                                    else if (q == 0) wv[wi][jx] = wi == 0 ? pk_sub4(dd[0][jx], dd[2][jx]) : pk_add4(dd[1][jx], dd[2][jx]);
                                    else wv[wi][jx] = wi == 0 ? pk_sub4(dd[1][jx], dd[0][jx]) : pk_sub4(dd[0][jx], dd[2][jx]);
                                } else {
                                    const int il = k8 >> 2, jv = k8 & 3;
                                    if (PMX_ABLATE & 32) asm volatile("" : "=v"(vvs[k8]));
                                    else vvs[k8] = jv == 0 ? pk_sub4(wv[il][0], wv[il][2]) : jv == 1 ? pk_add4(wv[il][1], wv[il][2]) : jv == 2 ? pk_sub4(wv[il][2], wv[il][1]) : pk_sub4(wv[il][1], wv[il][3]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (VCL && t >= 25 && t < 33) {
                            const int k8 = t - 25, il = k8 >> 2, jv = k8 & 3;
                            if (PMX_ABLATE & 64) asm volatile("" :: "v"(vvs[k8]));
                            else *reinterpret_cast<f32x4*>(&udst[(4 * il + jv) * 32 * C::LDU]) = vvs[k8];
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (VCL && t >= 16 && t < 40) {
                            // (nothing: the spread schedule's slots)
                        } else if (t >= 16 && t < 24) {                 // B^T d: q = 0: (d0 - d2, d1 + d2); q = 1: (d2 - d1, d1 - d3)
                            const int jx = (t - 16) >> 1, wi = (t - 16) & 1;
                            if (PMX_ABLATE & 32) asm volatile("" : "=v"(wv[wi][jx]));
                            else if (q == 0) wv[wi][jx] = wi == 0 ? pk_sub4(dd[0][jx], dd[2][jx]) : pk_add4(dd[1][jx], dd[2][jx]);
                            else wv[wi][jx] = wi == 0 ? pk_sub4(dd[1][jx], dd[0][jx]) : pk_sub4(dd[0][jx], dd[2][jx]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (t >= 24 && t < 40) {                 // (.) B, one column per two slots: compute, store
                            const int pidx = (t - 24) >> 1, il = pidx >> 2, jv = pidx & 3;
                            if (((t - 24) & 1) == 0) {
                                if (PMX_ABLATE & 32) asm volatile("" : "=v"(vv));
                                else
                                vv = jv == 0 ? pk_sub4(wv[il][0], wv[il][2]) : jv == 1 ? pk_add4(wv[il][1], wv[il][2]) : jv == 2 ? pk_sub4(wv[il][2], wv[il][1]) : pk_sub4(wv[il][1], wv[il][3]);
                            } else if (PMX_ABLATE & 64) {
                                asm volatile("" :: "v"(vv));
                            } else {
                                *reinterpret_cast<f32x4*>(&udst[(4 * il + jv) * 32 * C::LDU]) = vv;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (LAST && ((t >= 40 && t < 40 + (C::NHF < 20 ? C::NHF : 20)) || (C::NHF > 20 && t == 15))) {
                            // the halo of the next chunk -> LDS (r = 0) / of the one after it -> registers: slots 40 .. 59 (before the
                            // phase's barrier in step 30), a 21st staging slot (merged tails, 7x7) in slot 15 right behind the barrier
                            const int hs = t == 15 ? 20 : t - 40;       // (a constant once the loops are unrolled)
                            if (repl) {
                                if (r == 0) { if (!(PMX_ABLATE & 128)) halo_store_slot(hreg, hs); }
                                else if (!(PMX_ABLATE & 256)) halo_load_slot(hreg, cn, hs);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
    };
    static_assert(C::NHF <= 21, "halo slots");
    PMX_T(2);
    for (int ch = c0; ch < c1; ++ch) {
        const bool more = ch + 1 < c1;
        const unsigned chunk_b = (unsigned)ch * panel_b;
        const unsigned next_b = (unsigned)(more ? ch + 1 : ch) * panel_b;
        // (the sub-kernels are unrolled: a run-time loop over three of them + a peeled last one made the register allocator shuttle
        //  accumulator tiles between the two copies with v_accvgpr_mov + s_nop 15)
        p1_step(std::integral_constant<int, 0>{}, ch, more, chunk_b, next_b);
        if constexpr (C::NSUB > 1) {
            p1_step(std::integral_constant<int, 1>{}, ch, more, chunk_b, next_b);
            p1_step(std::integral_constant<int, 2>{}, ch, more, chunk_b, next_b);
            p1_step(std::integral_constant<int, 3>{}, ch, more, chunk_b, next_b);
        }
    }

    }   // do_p1

    // ---- output transform Y = A^T M A per (tile, channel): y[2 * i + j] = pixel (i, j) of the tile
    f32x16 y[4];
    if (!UNIT || do_p1) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = (acc[0 + j][reg] + acc[4 + j][reg]) + acc[8 + j][reg];
                t1[j] = (acc[4 + j][reg] - acc[8 + j][reg]) - acc[12 + j][reg];
            }
            y[0][reg] = (t0[0] + t0[1]) + t0[2]; y[1][reg] = (t0[1] - t0[2]) - t0[3];
            y[2][reg] = (t1[0] + t1[1]) + t1[2]; y[3][reg] = (t1[1] - t1[2]) - t1[3];
        }
    } else {
        // a unit block without pass 1 (row 6 / column 6 / tap (6, 6)): the transform of all-zero accumulators is +0 -- no accumulator
        // is zeroed or read for it
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) y[pp][reg] = 0.f;
    }

    if (C::NDIR > 0) {
        // ---- pass 2 (7x7): the 13 taps of row 6 and column 6.  Row 6 = two 1x3 sub-kernels (kx 0..2, 3..5) as 1-D F(2,3) along x:
        // per output row i of the tile 4 horizontal frequencies -> 8 planes (i * 4 + f), both sub-kernels summed in the same planes;
        // column 6 = two 3x1 sub-kernels (ky 0..2, 3..5) as 1-D F(2,3) along y: 8 planes (j * 4 + f); tap (6, 6) direct into the four
        // pixel planes.  2 * 8 + 2 * 8 + 4 = 36 products per tile and channel pair instead of 13 * 4 = 52.  Weight planes (same
        // [plane][chunk32][cout_pad][32] array as pass 1): 64 + sub * 4 + f (row 6), 72 + sub * 4 + f (column 6), 80 (tap (6, 6)).
        // Pass 2a per chunk: D phase (16 steps x 4 MFMAs from the raw halo; the threads transform row-6 sub-kernel 0 meanwhile), H0 and
        // H1 phases (16 steps x 8 MFMAs; during H0 sub-kernel 1 is transformed, during H1 the raw halo of the next chunk replaces this
        // one); then y += A^T-transform of the row planes.  Pass 2b per chunk: V0, V1 phases; then y += transform of the column planes.
        constexpr int PH = 64, PV = 72, PD = 80;
        constexpr int HBAR = C::NHF > 20 ? C::NHF : 20;      // H1 / V1: the slot of the barrier behind the halo stores (slots 0 .. NHF - 1)
        f32x16 e8[8];
        f32x4 bwr[8], bd[4], av[4];
        auto zero8 = [&]() {
#pragma unroll
            for (int pl = 0; pl < 8; ++pl)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) e8[pl][r16] = 0.f;
        };
        auto wload = [&](int plane, unsigned chb, int st) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)st * st_v, chb + (unsigned)plane * freq_b + (unsigned)st * st_b, 0));
        };
        // 1-D transform of this thread's (tile, 4 channels): two lines (output rows i for the row class, output columns j for the
        // column class) of 4 samples each -> (d0 - d2, d1 + d2, d2 - d1, d1 - d3), slot by slot
        f32x4 dd[2][4], vv[2][4];
        auto side1d = [&](int t, int base, int line_stride, int samp_stride, float* udst) {     // t compile-time
            if (t >= 2 && t < 10) {
                const int l = (t - 2) >> 2, c = (t - 2) & 3;
                dd[l][c] = *reinterpret_cast<const f32x4*>(&s_raw[base + l * line_stride + c * samp_stride]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (VCL && t == 12) {                 // the 16 packed adds in one gap
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int l = k8 >> 2, f = k8 & 3;
                    vv[l][f] = f == 0 ? pk_sub4(dd[l][0], dd[l][2]) : f == 1 ? pk_add4(dd[l][1], dd[l][2]) : f == 2 ? pk_sub4(dd[l][2], dd[l][1]) : pk_sub4(dd[l][1], dd[l][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (!VCL && t >= 12 && t < 20) {
                const int l = (t - 12) >> 2, f = (t - 12) & 3;
                vv[l][f] = f == 0 ? pk_sub4(dd[l][0], dd[l][2]) : f == 1 ? pk_add4(dd[l][1], dd[l][2]) : f == 2 ? pk_sub4(dd[l][2], dd[l][1]) : pk_sub4(dd[l][1], dd[l][3]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (t >= 20 && t < 28) {
                const int l = (t - 20) >> 2, f = (t - 20) & 3;
                *reinterpret_cast<f32x4*>(&udst[(l * 4 + f) * 32 * C::LDU]) = vv[l][f];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // one 8-plane phase: 16 steps s = f * 4 + st, per step the two lines l = 0, 1 x 4 MFMAs; A fragments one step ahead, weights
        // four steps ahead (ring of 8; `wnext(s)` loads step s of whatever phase follows), side slots m = 2, 3, 6, 7 of every step
        auto phase8 = [&](const float* ub, int wplane, unsigned chb, auto&& wnext, auto&& side) {
            av[0] = *reinterpret_cast<const f32x4*>(&ub[a_off]);
            av[1] = *reinterpret_cast<const f32x4*>(&ub[4 * 32 * C::LDU + a_off]);
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int f = s2 >> 2, st = s2 & 3;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int l = m >> 2, e = m & 3;
                    e8[l * 4 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(s2 & 1) * 2 + l][e], bwr[s2 & 7][e], e8[l * 4 + f], 0, 0, 0);
                    if (m == 0) {
                        constexpr int L2 = PMX_WINO_WLEAD2;
                        if (PMX_ABLATE & 2) {}
                        else if (s2 + L2 < 16) bwr[(s2 + L2) & 7] = wload(wplane + ((s2 + L2) >> 2), chb, (s2 + L2) & 3);
                        else wnext(s2 + L2 - 16);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if ((m == 1 || m == 5) && s2 + 1 < 16) {
                        const int ln = m == 1 ? 0 : 1, fn = (s2 + 1) >> 2, sn = (s2 + 1) & 3;
                        if (!(PMX_ABLATE & 4))
                        av[((s2 + 1) & 1) * 2 + ln] = *reinterpret_cast<const f32x4*>(&ub[(ln * 4 + fn) * 32 * C::LDU + a_off + sn * 8]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if ((m == 2 || m == 3 || m == 6 || m == 7) && !(PMX_ABLATE & 1)) {
                        side(4 * s2 + (m < 4 ? m - 2 : m - 4));
                    }
                }
            }
        };
        float* const u0 = s_u + t_u;
        float* const u1 = s_u + 8 * 32 * C::LDU + t_u;
        const int a2_off = tile_px(li) * C::LDR + kh * 4;

        // ================= pass 2a: tap (6, 6) + row 6 =================
        if (do_p2a) {
        zero8();                                    // (pass 1's last phase already staged the raw halo of chunk 0 again)
#pragma unroll
        for (int st = 0; st < 4; ++st) bd[st] = wload(PD, 0u, st);
        if (UNIT) {                                 // standalone: stage chunk 0, keep chunk 1 in the registers
            if (GEOM) {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
            halo_store(hreg);
            __syncthreads();
            if (nch > 1) {
                halo_load(hreg, 1);
            }
        }
        for (int ch = 0; ch < nch; ++ch) {
            const bool more = ch + 1 < nch;
            const unsigned chb = (unsigned)ch * panel_b, nxb = (unsigned)(more ? ch + 1 : ch) * panel_b;
            // ---- D phase: step q = st * 4 + p; side: row-6 sub-kernel 0 -> U half 0; weights of H0's first four steps
            {
                f32x4 ad[4];
                ad[0] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + 0) * C::HW + 6 + 0) * C::LDR]);
                ad[1] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + 0) * C::HW + 6 + 1) * C::LDR]);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int st = q >> 2, pp = q & 3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!UNIT) y[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[q & 3][e], bd[st][e], y[pp], 0, 0, 0);
                        if (e == 0) {
                            if (q < PMX_WINO_WLEAD2) { bwr[q] = wload(PH + (q >> 2), chb, q & 3); __builtin_amdgcn_sched_barrier(0); }
                        } else if (e == 1) {
                            if (q + 2 < 16) {
                                const int qn = q + 2, pn = qn & 3, sn = qn >> 2;
                                ad[qn & 3] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + (pn >> 1)) * C::HW + 6 + (pn & 1)) * C::LDR + sn * 8]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                            side1d(2 * q + (e - 2), t_raw + (6 * C::HW + 0) * C::LDR, C::HW * C::LDR, C::LDR, u0);
                        }
                    }
                }
            }
            __syncthreads();                        // U half 0 = row-6 sub-kernel 0
            // The raw halo is replaced without ever stopping the matrix pipe and without the staging registers meeting the transform's:
            // ---- H0: side = sub-kernel 1 (kx 3..5) -> U half 1 (the last reads of this chunk's raw halo, slots 2..27); then the halo of the
            // next chunk (chunk 0 again after the last one: pass 2b starts from it) global -> registers, one load per slot from slot 28
            // on; afterwards H1's first weights
            const int cnx = more ? ch + 1 : 0;
            phase8(s_u, PH + 0, chb,
                   [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PH + 4 + (s2n >> 2), chb, s2n & 3); },
                   [&](int t) {
                       if (t >= 28 && t < 28 + C::NHF) { halo_load_slot(hreg, cnx, t - 28); __builtin_amdgcn_sched_barrier(0); }
                       else side1d(t, t_raw + (6 * C::HW + 3) * C::LDR, C::HW * C::LDR, C::LDR, u1);
                   });
            __syncthreads();                        // U half 1 = row-6 sub-kernel 1; nobody reads the old raw halo any more
            // ---- H1: side = registers -> LDS, one ds_write_b128 per slot from slot 0 on, a barrier (slot 20), then column-6 sub-kernel 0
            // of the new chunk -> U half 0 (free: H1 reads half 1; needed after the last chunk, otherwise unused and overwritten by the
            // next D phase -- unconditional, because a branch per slot would cut the schedule into pieces); afterwards the next chunk's
            // tap-(6,6) weights
            phase8(s_u + 8 * 32 * C::LDU, PH + 4, chb,
                   [&](int s2n) {
                       if (s2n < 4) bd[s2n] = wload(PD, nxb, s2n);
                       bwr[(s2n + 16) & 7] = wload(PV + (s2n >> 2), 0u, s2n & 3);  // pass 2b's first weights (used after the last chunk)
                   },
                   [&](int t) {
                       if (t < C::NHF) { halo_store_slot(hreg, t); __builtin_amdgcn_sched_barrier(0); }
                       else if (t == HBAR) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
                       else if (t >= HBAR + 2) side1d(t - HBAR, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
                   });
        }
        __syncthreads();                            // U half 0 = column-6 sub-kernel 0 of chunk 0
        // y += A^T-transform of the row planes: e8[i * 4 + f]
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                y[i2 * 2 + 0][reg] = y[i2 * 2 + 0][reg] + ((e8[i2 * 4 + 0][reg] + e8[i2 * 4 + 1][reg]) + e8[i2 * 4 + 2][reg]);
                y[i2 * 2 + 1][reg] = y[i2 * 2 + 1][reg] + ((e8[i2 * 4 + 1][reg] - e8[i2 * 4 + 2][reg]) - e8[i2 * 4 + 3][reg]);
            }

        }   // do_p2a

        // ================= pass 2b: column 6 =================
        if (do_p2b) {
        zero8();
        if (UNIT) {                                 // standalone: stage chunk 0, first weights, sub-kernel 0 of chunk 0 (not overlapped)
            if (GEOM) {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
#pragma unroll
            for (int s2n = 0; s2n < PMX_WINO_WLEAD2; ++s2n) bwr[s2n] = wload(PV + (s2n >> 2), 0u, s2n & 3);
            halo_store(hreg);
            __syncthreads();
            if (nch > 1) {
                halo_load(hreg, 1);
            }
#pragma unroll
            for (int t = 0; t < 28; ++t) side1d(t, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
            __syncthreads();
        }
        // one chunk; MORE (compile time): another chunk follows -- its raw halo replaces this one inside V0 (barrier in slot 10 after
        // the last reads of the old halo, one ds_write_b128 per slot from slot 28 on), the halo after it is requested inside V1
        auto p2b_chunk = [&](auto more_c, int ch) {
            constexpr bool MORE = decltype(more_c)::value;
            // y rests during pass 2b: pin it to the accumulator file (64 of its registers are free here) -- left in VGPRs next to the 20-slot
            // halo it pushed the halo addresses to scratch, each reload with an s_waitcnt vmcnt(0) that also drains the weight ring
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) asm volatile("" : "+a"(y[pp]));
            const unsigned chb = (unsigned)ch * panel_b, nxb = (unsigned)(MORE ? ch + 1 : ch) * panel_b;
            // ---- V0: side = sub-kernel 1 (ky 3..5) -> U half 1, then the next chunk's halo global -> registers (slots 28 ..)
            phase8(s_u, PV + 0, chb,
                   [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 4 + (s2n >> 2), chb, s2n & 3); },
                   [&](int t) {
                       if (MORE && t >= 28 && t < 28 + C::NHF) { halo_load_slot(hreg, ch + 1, t - 28); __builtin_amdgcn_sched_barrier(0); }
                       else side1d(t, t_raw + (3 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u1);
                   });
            __syncthreads();
            // ---- V1: side = registers -> LDS (slots 0 ..), barrier (slot 20), the next chunk's sub-kernel 0 -> U half 0
            if constexpr (MORE)
                phase8(s_u + 8 * 32 * C::LDU, PV + 4, chb,
                       [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 0 + (s2n >> 2), nxb, s2n & 3); },
                       [&](int t) {
                           if (t < C::NHF) { halo_store_slot(hreg, t); __builtin_amdgcn_sched_barrier(0); }
                           else if (t == HBAR) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
                           else if (t >= HBAR + 2) side1d(t - HBAR, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
                       });
            else
                phase8(s_u + 8 * 32 * C::LDU, PV + 4, chb,
                       [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 0 + (s2n >> 2), nxb, s2n & 3); },
                       [&](int) {});
            __syncthreads();
        };
        static_assert(HBAR + 28 <= 64 && 28 + C::NHF <= 64, "halo slots");
        for (int ch = 0; ch < nch - 1; ++ch) p2b_chunk(std::true_type{}, ch);
        p2b_chunk(std::false_type{}, nch - 1);
        // y += A^T-transform of the column planes: e8[j * 4 + f]
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                y[0 * 2 + j2][reg] = y[0 * 2 + j2][reg] + ((e8[j2 * 4 + 0][reg] + e8[j2 * 4 + 1][reg]) + e8[j2 * 4 + 2][reg]);
                y[1 * 2 + j2][reg] = y[1 * 2 + j2][reg] + ((e8[j2 * 4 + 1][reg] - e8[j2 * 4 + 2][reg]) - e8[j2 * 4 + 3][reg]);
            }
        }   // do_p2b

        if (do_pd) {
            // ================= unit mode: tap (6, 6) over all chunks, straight from the raw halo =================
            f32x4 bdn[4];
            if (GEOM) {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
#pragma unroll
            for (int st = 0; st < 4; ++st) bd[st] = wload(PD, 0u, st);
            for (int ch = 0; ch < nch; ++ch) {
                if (ch) __syncthreads();
                halo_store(hreg);
                __syncthreads();
                const int cn = ch + 1 < nch ? ch + 1 : ch;
                halo_load(hreg, cn);
#pragma unroll
                for (int st = 0; st < 4; ++st) bdn[st] = wload(PD, (unsigned)cn * panel_b, st);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int st = q >> 2, pp = q & 3;
                    const f32x4 ad = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + (pp >> 1)) * C::HW + 6 + (pp & 1)) * C::LDR + st * 8]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[e], bd[st][e], y[pp], 0, 0, 0);
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) bd[st] = bdn[st];
            }
        }
    }

    // ---- bias, ReLU, (pool), store.  The stores go through a buffer resource that spans exactly this image's output (32-bit byte offsets,
    // an out-of-range offset = the store is dropped): no 64-bit address arithmetic and no branch per store.  (Written with pointers and
    // `if (inside) out[...] = v` this epilogue compiled to ~1100 instructions -- 270 quarter-rate integer multiplies / 64-bit mads, 80
    // exec-mask branches -- and took 4-5 us of a block that lasts 23 us (conv2_1) to 200 us (7x7): tools/block_timing.py.)
    PMX_T(5);
    const bool nok = n < G.cout;
    const int Hp = H >> 1, Wp = W >> 1;
    const int opix = POOL ? Hp * Wp : H * W;                         // output pixels per image
    const int ldc_b = a.ldc * 4;
    if (UNIT && GEOM) {
        // unit mode of a run (the part-filled last block of an image): compact slab [image][block of the launch][tile][pixel][cout_pad];
        // conv_wino_tail_reduce_kernel adds the units in order and drops the tiles past the end of the map
        // (MERGE: bslab * 1 + 0 = the block of the stream)
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(G.out + (size_t)(bslab * (MERGE ? 1 : a.run_nb) + trem) * (32 * 4) * a.ldc, 0,
                                                                                  (unsigned)(32 * 4 * ldc_b), 0x00020000);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mr = (reg & 3) + 8 * (reg >> 2) + 4 * kh;      // Winograd tile of this register row
            const int o = (int)__umul24(mr * 4, ldc_b) + n * 4;
            // (through float temporaries: __builtin_bit_cast applied directly to the vector element y[k][reg] compiled to element 0 for every reg)
            const float y00 = y[0][reg], y01 = y[1][reg], y10 = y[2][reg], y11 = y[3][reg];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), srsrc, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), srsrc, o + ldc_b, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), srsrc, o + 2 * ldc_b, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), srsrc, o + 3 * ldc_b, 0, 0);
        }
    } else {
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(G.out + (size_t)bimg * opix * a.ldc, 0, (unsigned)(opix * ldc_b), 0x00020000);
        const int n_b = nok ? n * 4 : -1;                            // (a lane without a real output channel: every offset out of range)
        // run geometry: (tile row, tile column) of the lane's first tile by one division, then stepped from register row to register row
        // (the rows of a lane are the tiles tb + 0, 1, 2, 3, 8, 9, ...: steps of 1 or 5 < 23, at most one wrap)
        const unsigned tb = (unsigned)(t0 + 4 * kh);
        int rty = (int)(tb / (unsigned)C::RUN_TX), rtx = (int)(tb - (unsigned)rty * (unsigned)C::RUN_TX);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mrc = (reg & 3) + 8 * (reg >> 2);              // (+ 4 kh): Winograd tile of this register row
            float y00 = y[0][reg], y01 = y[1][reg], y10 = y[2][reg], y11 = y[3][reg];
            int gy, gx;
            if (GEOM) {
                if (reg) {
                    rtx += (reg & 3) ? 1 : 5;
                    const bool wrap = rtx >= C::RUN_TX;
                    rtx = wrap ? rtx - C::RUN_TX : rtx;
                    rty += wrap ? 1 : 0;
                }
                gy = 2 * rty; gx = x0 + 2 * rtx;                     // tiles past the end of the map land on rows >= H
            } else {
                gy = y0 + 2 * ((mrc >> 3)) ; gx = x0 + 2 * ((mrc & 7) + 4 * kh);
            }
            if (POOL) {
                float v = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                const int py = gy >> 1, px = gx >> 1;
                const int o = (py < Hp && px < Wp && nok) ? (int)__umul24(__umul24(py, Wp) + px, ldc_b) + n_b : -1;      // (24-bit operands: full-rate multiplies)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, o, 0, 0);
            } else {
                y00 += bias; y01 += bias; y10 += bias; y11 += bias;
                if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                const int o00 = (int)__umul24(__umul24(gy, W) + gx, ldc_b) + n_b;      // (24-bit operands: full-rate multiplies; < 2^31 by the launcher's check)
                // (run geometry: the map is a whole number of 46-column slabs and a tile column is < 23, so both pixel columns are inside)
                const bool r0 = gy < H && nok, r1 = gy + 1 < H && nok, c0v = GEOM ? true : gx < W, c1v = GEOM ? true : gx + 1 < W;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), orsrc, (r0 && c0v) ? o00 : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), orsrc, (r0 && c1v) ? o00 + ldc_b : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), orsrc, (r1 && c0v) ? o00 + W * ldc_b : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), orsrc, (r1 && c1v) ? o00 + (W + 1) * ldc_b : -1, 0, 0);
            }
        }
    }
    PMX_T(6);
}

// ---- conv1_1: 3 input channels ---------------------------------------------------------------------------------------
// The generic kernels spend a whole 16-channel chunk (8 MFMA k-pairs per tap) on 3 real channels.  Here K = 27 is packed
// tap-major / channel-minor into 14 k-pairs (the 28th k is zero) - the order in which the generic kernels meet the three
// non-zero channels, so the fp32 FMA chain per output is the same.  Block = 16 x 16 pixels x 64 channels (wave w: rows
// 4w .. 4w+3 as two 32-pixel row tiles x two 32-channel tiles); the 18 x 18 x 3 input patch goes through LDS (pitch 3 floats:
// conflict-free ds_read_b32), weights and bias live in registers, blocks loop over tiles.  Memory-bound (writes 64 ch/pixel).
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const ConvArgs a)
{
    constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2;
    __shared__ float s_patch[PH * PW * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const float* wp = a.g[0].w;                     // packed [tap][1 chunk][cout_pad = 64][16]
    float wv[14][2], biasv[2];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k = 2 * s + kh;
        const int kk = k < 27 ? k : 26;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float w = wp[(size_t)((kk / 3) * 64 + u * 32 + li) * 16 + kk % 3];
            wv[s][u] = k < 27 ? w : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) biasv[u] = a.g[0].bias[u * 32 + li];
    // A gather: row tile T = 2 * wave + t covers tile rows 2T, 2T+1; lane's pixel = (li >> 4, li & 15); k -> (ky, kx, c)
    int aoff[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k = 2 * s + kh, kk = k < 27 ? k : 26;
        const int tap = kk / 3, cc = kk % 3;
        aoff[s] = ((4 * wave + (li >> 4) + tap / 3) * PW + (li & 15) + tap % 3) * 3 + cc;
    }
    const int H = a.H, W = a.W, cout = a.g[0].cout;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bimg = tile / (a.tiles_x * a.tiles_y);
        const int trem = tile - bimg * a.tiles_x * a.tiles_y;
        const int y0 = (trem / a.tiles_x) * TH, x0 = (trem % a.tiles_x) * TW;
        const float* in_b = a.g[0].in + (size_t)bimg * H * W * a.lda;
        __syncthreads();
        for (int f = tid; f < PH * PW; f += 256) {
            const int hy = f / PW, hx = f - hy * PW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v = *reinterpret_cast<const float4*>(in_b + ((size_t)gy * W + gx) * a.lda);
            s_patch[f * 3 + 0] = v.x; s_patch[f * 3 + 1] = v.y; s_patch[f * 3 + 2] = v.z;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float av = s_patch[aoff[s] + t * 2 * PW * 3];
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv[s][u], acc[t][u], 0, 0, 0);
            }
        }
        float* out_b = a.g[0].out + (size_t)bimg * H * W * a.ldc;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = u * 32 + li;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    const int gy = y0 + 4 * wave + 2 * t + (m >> 4), gx = x0 + (m & 15);
                    float v = acc[t][u][reg] + biasv[u];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (n < cout && gy < H && gx < W) out_b[((size_t)gy * W + gx) * a.ldc + n] = v;
                }
            }
    }
}

// ---- conv1_1 + conv1_2 in one launch ----------------------------------------------------------------------------------------
// conv1_1 (3 -> 64) is HBM-bound: it writes 64 channels per input pixel (1.1 GB per batch of 32 at 368 x 368) that conv1_2 reads
// straight back.  Here a block owns an 8 x 16 tile of conv1_2's (pre-pool) output and first RECOMPUTES conv1_1 on the
// 10 x 18 halo of that tile from a 12 x 20 x 3 input patch (12 MFMA tiles x 14 k-pairs: +7 % matrix work), bias + ReLU, zero
// outside the image (= conv1_2's zero padding), into an LDS tile holding all 64 channels; conv1_2 then runs its 4 chunks x 9
// taps from LDS with no staging and no barrier, weights L2 -> registers one tap ahead, epilogue (bias, ReLU, 2x2 max-pool) as
// every other kernel.  Both layers walk K exactly like conv3x3_c3_kernel and the v5 kernels -> bit-identical to running them apart.
// Arguments: a.g[0] = conv1_2 (w, bias, out, cout; in = the 16-channel padded network input), a.g[1].w / .bias = conv1_1's.
__global__ __launch_bounds__(256, 3) void conv1_fused_kernel(const ConvArgs a)
{
    using C = ConvCfg<3, 8, 16, 64, 16, 2, 2>;
    constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, NPX = HH * HW, PH = HH + 2, PW = HW + 2, LDA = 68;
    __shared__ float s_patch[PH * PW * 3];
    extern __shared__ float4 smem4[];
    float* const s_act = reinterpret_cast<float*>(smem4);                  // [180][LDA]: conv1_1 output on the halo, 64 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int H = a.H, W = a.W;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int bimg = tile / tiles_per_img;
    const int trem = tile - bimg * tiles_per_img;
    const int y0 = (trem / a.tiles_x) * TH, x0 = (trem % a.tiles_x) * TW;
    const float* in_b = a.g[0].in + (size_t)bimg * H * W * a.lda;

    // ---- input patch: 12 x 20 pixels x 3 channels, zero outside the image (conv1_1's padding)
    for (int f = tid; f < PH * PW; f += 256) {
        const int py = f / PW, px = f - py * PW;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *reinterpret_cast<const float4*>(in_b + ((size_t)gy * W + gx) * a.lda);
        s_patch[f * 3 + 0] = v.x; s_patch[f * 3 + 1] = v.y; s_patch[f * 3 + 2] = v.z;
    }
    // conv1_1 weights of this wave's 32 output channels (column tile ct = wave & 1), K = 27 packed into 14 k-pairs
    const int ct = wave & 1;
    float wv[14];
    int koff[14];
    {
        const float* wp = a.g[1].w;                     // packed [tap][1 chunk][64][16]
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k = 2 * s + kh, kk = k < 27 ? k : 26;
            const float w = wp[(size_t)((kk / 3) * 64 + ct * 32 + li) * 16 + kk % 3];
            wv[s] = k < 27 ? w : 0.f;
            const int tap = kk / 3;
            koff[s] = ((tap / 3) * PW + tap % 3) * 3 + kk % 3;
        }
    }
    const float bias1 = a.g[1].bias[ct * 32 + li];
    // conv1_2: this lane's output channel, bias, weight stream
    float biasv[1];
    conv_load_bias<C>(biasv, a.g[0].bias, 0, wn, li);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[0].w), 0, 0x7fffffff, 0x00020000);
    const unsigned b_off = (unsigned)(((wn * 32 + li) * 16 + kh * 4) * 4);
    const unsigned panel_b = 64u * 16u * 4u;           // bytes of one (tap, chunk) panel: [64][16] floats
    f32x4 bA[2];
    bA[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off, 0u, 0));
    bA[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + 32, 0u, 0));
    __syncthreads();

    // ---- conv1_1 on the halo: row tiles rt = (wave >> 1) + 2 i, i = 0..2 (192 rows cover the 180 halo pixels)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int rt = (wave >> 1) + 2 * i;
        int m = rt * 32 + li;
        if (m >= NPX) m = NPX - 1;
        const int hy = m / HW, hx = m - hy * HW;
        const int pbase = (hy * PW + hx) * 3;
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s_patch[pbase + koff[s]], wv[s], acc1, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mr = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (mr < NPX) {
                const int ry = mr / HW, rx = mr - ry * HW;
                const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
                float v = fmaxf(acc1[reg] + bias1, 0.f);
                if (!((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) v = 0.f;       // conv1_2's zero padding
                s_act[mr * LDA + ct * 32 + li] = v;
            }
        }
    }
    __syncthreads();

    // ---- conv1_2 from the LDS tile: 4 chunks x 9 taps, no barrier; the v5 tap schedule (one memory instruction per MFMA gap,
    //      weights one tap ahead through the buffer resource, A fragments one k-step ahead)
    int a_base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = (wm * 2 + t) * 32 + li, q = m >> 2, r = m & 3;
        const int py = 2 * (q / (TW / 2)) + (r >> 1), px = 2 * (q % (TW / 2)) + (r & 1);
        a_base[t] = (py * HW + px) * LDA + kh * 4;
    }
    f32x16 acc[2][1];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] = 0.f;
    f32x4 bX[1][2], bY[1][2];
    bX[0][0] = bA[0]; bX[0][1] = bA[1];
    const unsigned b_offs[1] = {b_off};
    f32x4 av0[2], av1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) av0[t] = *reinterpret_cast<const f32x4*>(&s_act[a_base[t]]);
#pragma unroll
    for (int u = 0; u < 36; ++u) {                      // u = chunk * 9 + tap; panels are [tap][chunk] in memory
        const int ch = u / 9, tap = u % 9;
        const int un = u + 1 < 36 ? u + 1 : u;
        const int toff = ((tap / 3) * HW + tap % 3) * LDA + ch * 16;
        const int toff_n = (((un % 9) / 3) * HW + (un % 9) % 3) * LDA + (un / 9) * 16;
        const unsigned wnext = (unsigned)((un % 9) * 4 + un / 9) * panel_b;
        if (u & 1) TapBody<C, 3>::run_u(acc, av0, av1, bY, bX, wrsrc, wnext, b_offs, s_act, a_base, toff, toff_n);
        else TapBody<C, 3>::run_u(acc, av0, av1, bX, bY, wrsrc, wnext, b_offs, s_act, a_base, toff, toff_n);
    }
    conv_epilogue<C, TW>(acc, biasv, a, a.g[0].out, a.g[0].cout, bimg, y0, x0, 0, wm, wn, li, kh);
}

int conv1_fused_launch(const ConvArgs& a0, hipStream_t stream)
{
    ConvArgs a = a0;
    PMX_CHECK(a.cout_pad == 64 && a.nch == 4 && a.lda >= 4, PMX_ERR_INVALID, "conv1 fused: needs conv1_2 = 64 -> 64 and a >= 4-channel padded input");
    PMX_CHECK(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 7) / 8;
    a.ksplit = 1;
    hipLaunchKernelGGL(conv1_fused_kernel, dim3((unsigned)(a.tiles_x * a.tiles_y * a.B)), dim3(256), 180 * 68 * 4, stream, a);      // 49 KB + 2.9 KB static: three blocks per CU
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

static int launch_c3(const ConvArgs& a0, int groups, hipStream_t stream)
{
    ConvArgs a = a0;
    PMX_CHECK(groups == 1 && !a.pool && a.nch == 1 && a.cout_pad == 64, PMX_ERR_INVALID,
              "conv c3: needs one group, no pooling, one 16-channel chunk and 64 padded output channels");
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 15) / 16;
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.B;
    const unsigned grid = (unsigned)(ntiles < 256 * 8 ? ntiles : 256 * 8);
    hipLaunchKernelGGL(conv3x3_c3_kernel, dim3(grid), dim3(256), 0, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// dynamic LDS above 64 KB must be allowed per kernel AND per device (a process may hold contexts on several GPUs)
constexpr int PMX_MAX_DEVICES = 64;
// (`done` is a plain flag per kernel and device: two host threads racing here both call hipFuncSetAttribute with the same value -- idempotent)
static int conv_allow_big_lds(const void* kern, bool (&done)[PMX_MAX_DEVICES])
{
    int dev = 0;
    PMX_HIP(hipGetDevice(&dev));
    PMX_CHECK(dev >= 0 && dev < PMX_MAX_DEVICES, PMX_ERR_INVALID, "device index %d out of range", dev);
    if (!done[dev]) {
        PMX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done[dev] = true;
    }
    return PMX_OK;
}

// minimum dynamic LDS per block: caps the number of co-resident blocks per CU (the fp32 MFMA pipe
// loses ~20% with 3+ waves per SIMD)
static int g_min_lds = 0;
void conv_set_min_lds(int bytes) { g_min_lds = bytes; }
static int g_v5_lds = 56 * 1024;     // dynamic LDS floor of the v5 kernels: 3 x 56 KB > 160 KB -> at most 2 blocks per CU
void conv_set_v5_lds(int bytes) { g_v5_lds = bytes; }

// ---- variant table ---------------------------------------------------------------------------------------
// {ksize, tile rows (v6: row tiles per block), tile cols (v6: 32), BN, CK, name}
static const ConvVariant g_variants[] = {
    // v1 kernels: weight panel through LDS, one barrier per tap (1x1 layers; reference implementation of the others)
    {7, 8, 16, 128, 16, "conv7x7_t8x16_n128"},    // 0
    {3, 8, 16, 128, 16, "conv3x3_t8x16_n128"},    // 1
    {3, 8, 16, 64, 16, "conv3x3_t8x16_n64"},      // 2
    {1, 8, 16, 128, 16, "conv1x1_t8x16_n128"},    // 3
    {1, 8, 16, 64, 16, "conv1x1_t8x16_n64"},      // 4
    {7, 8, 8, 64, 16, "conv7x7_t8x8_n64"},        // 5: small batches (more blocks)
    {3, 8, 8, 64, 16, "conv3x3_t8x8_n64"},        // 6
    {1, 8, 8, 64, 16, "conv1x1_t8x8_n64"},        // 7
    {7, 2, 46, 128, 16, "conv7x7_t2x46_n128"},    // 8: zero-waste row strips for 46-wide maps (368x368 input)
    {3, 2, 46, 128, 16, "conv3x3_t2x46_n128"},    // 9
    // v5 kernels: weights L2 -> registers, software-pipelined, taps unrolled, two blocks per CU (see conv_mfma_v5_kernel)
    {7, 2, 46, 128, 16, "conv7x7_v5_t2x46_n128"},  // 10
    {3, 2, 46, 128, 16, "conv3x3_v5_t2x46_n128"},  // 11
    {7, 8, 16, 128, 16, "conv7x7_v5_t8x16_n128"},  // 12
    {3, 8, 16, 128, 16, "conv3x3_v5_t8x16_n128"},  // 13
    {3, 8, 16, 64, 16, "conv3x3_v5_t8x16_n64"},    // 14
    {7, 8, 8, 64, 16, "conv7x7_v5_t8x8_n64"},      // 15: small launches (single images)
    {3, 8, 8, 64, 16, "conv3x3_v5_t8x8_n64"},      // 16
    // v6 kernels: one block per CU, 17 x 32 consecutive pixels of a 46-column slab x 128 channels (see conv_mfma_v6_kernel)
    {7, 17, 32, 128, 16, "conv7x7_v6_t17x32_n128"},     // 17
    {3, 17, 32, 128, 16, "conv3x3_v6_t17x32_n128"},     // 18
    {3, 17, 32, 128, 16, "conv3x3_v6p_t17x32_n128"},    // 19: with the fused 2x2 max-pool (row-pair pixel order)
    {3, 16, 16, 64, 16, "conv3x3_c3_t16x16_n64"},       // 20: conv1_1 (3 input channels, K packed to 28)
    // v6 with 9 row tiles per block (288 px): fills the chip at batch 16 / 48 (8 blocks per 46x46 map)
    {7, 9, 32, 128, 16, "conv7x7_v6_t9x32_n128"},       // 21
    {3, 9, 32, 128, 16, "conv3x3_v6_t9x32_n128"},       // 22
    {3, 9, 32, 128, 16, "conv3x3_v6p_t9x32_n128"},      // 23
    // v5 with 16 x 8 tiles: less padding than 8 x 16 on maps like 46 x 82 (368 x 656 inputs)
    {7, 16, 8, 128, 16, "conv7x7_v5_t16x8_n128"},       // 24
    {3, 16, 8, 128, 16, "conv3x3_v5_t16x8_n128"},       // 25
    {3, 16, 8, 64, 16, "conv3x3_v5_t16x8_n64"},         // 26
    // v7: the v6 geometry on the bf16 matrix cores, fp32 values split into three bf16 terms (opt-in, see conv_bf16x3_kernel)
    {7, 17, 32, 128, 16, "conv7x7_v7bf16x3_t17x32_n128"},   // 27
    {3, 17, 32, 128, 16, "conv3x3_v7bf16x3_t17x32_n128"},   // 28
    {3, 17, 32, 128, 16, "conv3x3_v7bf16x3p_t17x32_n128"},  // 29
    {7, 9, 32, 128, 16, "conv7x7_v7bf16x3_t9x32_n128"},     // 30
    {3, 9, 32, 128, 16, "conv3x3_v7bf16x3_t9x32_n128"},     // 31
    {3, 9, 32, 128, 16, "conv3x3_v7bf16x3p_t9x32_n128"},    // 32
    // v8: bf16x3 on the small v5 tiles (single images / small batches, split-K capable)
    {7, 8, 8, 64, 16, "conv7x7_v8bf16x3_t8x8_n64"},         // 33
    {3, 8, 8, 64, 16, "conv3x3_v8bf16x3_t8x8_n64"},         // 34
};
enum { V5_K7_STRIP = 10, V5_K3_STRIP = 11, V5_K7 = 12, V5_K3 = 13, V5_K3_N64 = 14, V5_K7_SMALL = 15, V5_K3_SMALL = 16,
       V6_K7 = 17, V6_K3 = 18, V6_K3_POOL = 19, C3 = 20, V6M9_K7 = 21, V6M9_K3 = 22, V6M9_K3_POOL = 23, V5T_K7 = 24, V5T_K3 = 25, V5T_K3_N64 = 26,
       V7_K7 = 27, V7_K3 = 28, V7_K3_POOL = 29, V7M9_K7 = 30, V7M9_K3 = 31, V7M9_K3_POOL = 32, V8_K7_SMALL = 33, V8_K3_SMALL = 34 };

// the bf16x3 twin of a v6 variant (same block geometry), or -1
int conv_bf16x3_twin(int v)
{
    switch (v) {
        case V6_K7: return V7_K7; case V6_K3: return V7_K3; case V6_K3_POOL: return V7_K3_POOL;
        case V6M9_K7: return V7M9_K7; case V6M9_K3: return V7M9_K3; case V6M9_K3_POOL: return V7M9_K3_POOL;
        case V5_K7_SMALL: return V8_K7_SMALL; case V5_K3_SMALL: return V8_K3_SMALL;
        case V5_K3_N64: case V5T_K3_N64: return V8_K3_SMALL;      // 64-output-channel layers (conv1_2): 2.45 -> 1.87 ms at batch 32
    }
    return -1;
}

int conv_num_variants() { return (int)(sizeof(g_variants) / sizeof(g_variants[0])); }
const ConvVariant& conv_variant(int idx) { return g_variants[idx]; }

static std::atomic<int> g_num_cus{256};       // process-wide (contexts on different devices of one process are expected to be the same part)
void conv_set_num_cus(int n) { if (n > 0) g_num_cus.store(n, std::memory_order_relaxed); }
int conv_num_cus() { return g_num_cus.load(std::memory_order_relaxed); }

// gen: 1 = v1 kernels everywhere, 5 = v5 for 3x3 / 7x7, 6 (default) = v6 / c3 where they apply, else v5
int conv_pick_variant(int ks, int cout, int H, int W, int B, int forced, int gen, int pool, int cin, int bf16x3)
{
    const int ncu = g_num_cus;      // compute units of the device (256 on an MI355X in SPX mode)
    // `cout` is the padded channel count of the layer
    const bool forced_v6 = (forced >= V6_K7 && forced <= V6_K3_POOL) || (forced >= V6M9_K7 && forced <= V6M9_K3_POOL);
    if (forced >= V7_K7) forced = -1;       // the bf16x3 kernels are chosen through their v6 twins (they need the bf16x3 weight pack)
    if (forced >= 0 && forced < conv_num_variants() && g_variants[forced].ks == ks && cout % g_variants[forced].bn == 0 &&
        !(forced_v6 && (W % 46 != 0 || !!pool != (forced == V6_K3_POOL || forced == V6M9_K3_POOL))) &&      // v6: 46-column slabs
        !(forced == C3 && (cin > 3 || cout != 64 || pool)))                                                  // c3: conv1_1-shaped layers only
        return forced;
    if (gen >= 6 && ks == 3 && cin <= 3 && cout == 64 && !pool) return C3;
    // enough 8x16 tiles to fill the CUs a few times over?  otherwise use the small tiles
    const long tiles816 = (long)((H + 7) / 8) * ((W + 15) / 16) * B;
    const bool small = tiles816 * ((cout + 127) / 128) < 2 * ncu;
    // 2 x 46 row strips tile 46-wide maps exactly (8 x 16 tiles waste 8.9 %); on 92-wide maps they measured neutral
    const bool strip = (W == 46) && (cout % 128 == 0) && ((long)((H + 1) / 2) * B * (cout / 128) >= 2 * ncu);
    // (3x3 layers with fewer than 8 input chunks have too little work per chunk transition for one wave per SIMD: measured
    //  slower than v5 on conv2_1)
    // (with the bf16x3 twins the matrix work shrinks 2.7x and the one-block-per-CU geometry also wins at 4 input chunks: conv2_1
    //  1.35 -> 0.91 ms)
    if (gen >= 6 && cout % 128 == 0 && W % 46 == 0 && ((ks == 3 && cin >= (bf16x3 ? 64 : 128)) || (ks == 7 && !pool)) && (!pool || H % 2 == 0)) {
        // v6 (one block of 17 or 9 row tiles per CU, 46-column slabs) when its blocks fill whole rounds of the CUs and
        // the pixel padding is small: efficiency = useful pixels / (rounds * CUs * block pixels) >= 0.88; else v5
        const long useful = (long)H * W * B * (cout / 128);
        double best = 0.0;
        int best_mt = 0;
        for (int mt : {17, 9}) {
            const long nblk = (long)((H * 46 + 32 * mt - 1) / (32 * mt)) * (W / 46) * B * (cout / 128);
            const long rounds = (nblk + ncu - 1) / ncu;
            const double eff = (double)useful / ((double)rounds * ncu * 32 * mt);
            if (eff > best + 1e-9) { best = eff; best_mt = mt; }
        }
        if (best >= 0.88) {
            if (best_mt == 17) return ks == 7 ? V6_K7 : (pool ? V6_K3_POOL : V6_K3);
            return ks == 7 ? V6M9_K7 : (pool ? V6M9_K3_POOL : V6M9_K3);
        }
    }
    if (gen >= 5) {      // v5 for 3x3 / 7x7; launches that would not fill the chip with 8x16 tiles use the 8x8 / BN64 tiles
        // 8 x 16 or 16 x 8 tiles, whichever pads the map less (46 x 82: 48 x 96 vs 48 x 88)
        const long pad816 = (long)((H + 7) / 8 * 8) * ((W + 15) / 16 * 16), pad168 = (long)((H + 15) / 16 * 16) * ((W + 7) / 8 * 8);
        const bool tall = pad168 < pad816;
        if (ks == 7) return strip ? V5_K7_STRIP : (small ? V5_K7_SMALL : (tall ? V5T_K7 : V5_K7));
        if (ks == 3) return strip ? V5_K3_STRIP : (small ? V5_K3_SMALL : (cout <= 64 ? (tall ? V5T_K3_N64 : V5_K3_N64) : (tall ? V5T_K3 : V5_K3)));
        return small ? 7 : (cout <= 64 ? 4 : 3);
    }
    if (ks == 7) return strip ? 8 : (small ? 5 : 0);
    if (ks == 3) return strip ? 9 : (small ? 6 : (cout <= 64 ? 2 : 1));
    return small ? 7 : (cout <= 64 ? 4 : 3);
}

static bool is_v5_variant(int v) { return (v >= V5_K7_STRIP && v <= V5_K3_SMALL) || (v >= V5T_K7 && v <= V5T_K3_N64) || v == V8_K7_SMALL || v == V8_K3_SMALL; }

// Split-K for launches that cannot fill the chip (single images): nblk blocks over ncu CUs leave CUs idle or quantise badly
// (144 blocks of a 7x7 layer at batch 1: 112 CUs idle; 288 blocks of conv4_2: a second round on 32 CUs).  With S K-slices
// there are nblk * S smaller blocks.  Which split is best depends on how the blocks land on the CUs, so the candidates are
// SIMULATED: blocks are dispatched in grid order (all blocks of slice 0, then slice 1, ...: z is the slowest grid dimension),
// two resident per CU (the LDS floor of the v5 kernels), round-robin at launch and then to whichever CU frees a slot; two
// co-resident blocks share the matrix pipe.  Uneven slices, largest first, let the late small blocks fill the gaps (7x7 at
// batch 1: slices of 3 + 2 + 2 + 1 chunks finish in 5 chunk times where 2 + 3 + 3 or 2 + 2 + 2 + 2 need 6).  The combine
// kernel costs about one launch boundary plus S slab reads.  The schedule is a speed guess only -- any split computes the
// same defined sum (slices added left to right).
static double sk_simulate(const int* sizes, int S, long nblk, int ncu, double t_chunk, double t_fixed)
{
    // per CU: up to two active blocks (remaining pipe work), local clock
    struct CU { double rem[2]; int n; double t; };
    std::vector<CU> cu((size_t)ncu);
    for (auto& c : cu) { c.rem[0] = c.rem[1] = 0; c.n = 0; c.t = 0; }
    const long total = nblk * S;
    long next = 0;
    auto work = [&](long i) { return sizes[i / nblk] * t_chunk + t_fixed; };
    for (int slot = 0; slot < 2 && next < total; ++slot)
        for (int c = 0; c < ncu && next < total; ++c) { cu[c].rem[cu[c].n++] = work(next++); }
    double makespan = 0;
    for (;;) {
        // CU whose next block completion comes first
        int best = -1;
        double bt = 1e300;
        for (int c = 0; c < ncu; ++c) {
            if (!cu[c].n) continue;
            const double m = cu[c].n == 2 ? (cu[c].rem[0] < cu[c].rem[1] ? cu[c].rem[0] : cu[c].rem[1]) : cu[c].rem[0];
            const double tc = cu[c].t + m * cu[c].n;
            if (tc < bt) { bt = tc; best = c; }
        }
        if (best < 0) break;
        CU& c = cu[best];
        const double m = c.n == 2 ? (c.rem[0] < c.rem[1] ? c.rem[0] : c.rem[1]) : c.rem[0];
        c.t = bt;
        if (c.n == 2) {
            c.rem[0] -= m; c.rem[1] -= m;
            if (c.rem[0] <= 1e-12) { c.rem[0] = c.rem[1]; }
            c.n = 1;
            if (c.rem[0] <= 1e-12) c.n = 0;
        } else c.n = 0;
        if (bt > makespan) makespan = bt;
        while (c.n < 2 && next < total) c.rem[c.n++] = work(next++);
    }
    return makespan;
}

static void sk_partitions(int n, int parts, int maxpart, int* cur, int depth, std::vector<std::vector<int>>& out)
{
    if (parts == 0) { if (n == 0) out.emplace_back(cur, cur + depth); return; }
    for (int v = (n - (parts - 1) < maxpart ? n - (parts - 1) : maxpart); v >= 1 && v * parts >= n; --v) {
        cur[depth] = v;
        sk_partitions(n - v, parts - 1, v, cur, depth + 1, out);
    }
}

SplitPlan conv_pick_ksplit(int variant, int H, int W, int B, int groups, int cout_pad, int nch, int pool, int forced)
{
    (void)pool;
    SplitPlan none;
    memset(&none, 0, sizeof none);
    none.S = 1; none.sizes[0] = nch;
    auto make = [&](const std::vector<int>& sizes) {
        SplitPlan p;
        memset(&p, 0, sizeof p);
        p.S = (int)sizes.size();
        int c = 0;
        for (int s = 0; s < p.S; ++s) { p.bounds |= (unsigned long long)c << (8 * s); p.sizes[s] = sizes[s]; c += sizes[s]; }
        return p;
    };
    if (!is_v5_variant(variant) || nch < 2 || nch > 255) return none;
    if (forced == 1) return none;
    if (forced < 0) {      // tuning: explicit plan, decimal digits = chunks per slice (must sum to nch, else ignored)
        std::vector<int> sizes;
        for (long d = -(long)forced; d > 0; d /= 10) sizes.insert(sizes.begin(), (int)(d % 10));
        int sum = 0;
        for (int v : sizes) sum += v;
        if (sum != nch || sizes.size() > 8) return none;
        for (int v : sizes) if (v < 1) return none;
        return make(sizes);
    }
    if (forced > 1) {      // (near-)even slices, the larger ones first
        const int S = forced < nch ? (forced < 8 ? forced : 8) : (nch < 8 ? nch : 8);
        std::vector<int> sizes(S);
        for (int s = 0; s < S; ++s) sizes[s] = nch / S + (s < nch % S ? 1 : 0);
        return make(sizes);
    }
    const ConvVariant& v = g_variants[variant];
    const long nblk = (long)((H + v.th - 1) / v.th) * ((W + v.tw - 1) / v.tw) * B * (cout_pad / v.bn) * groups;
    const int ncu = g_num_cus;
    if (nblk >= 8 * ncu) return none;
    // measured plans (tools/splitk_tune.py, in-network HIP-event times on an MI355X, profiles/r02_splitk_tune.txt) for the
    // launch shapes of 368 x 368 inputs at batch 1-6; {ksize, chunks, blocks before the split, slices...}.  How the
    // dispatcher really places late blocks is not modelled well enough by the simulation below (it is right about
    // 3-2-2-1 at batch 1 and wrong about the 12-chunk layers), so shapes that were measured use the measurement.
    static const int tuned[][12] = {
        {7, 8, 144, 3, 2, 2, 1}, {7, 12, 144, 5, 3, 3, 1},                                  // batch 1: 106.6 -> 70.2 us, 168.7 -> 103.2
        {3, 16, 576, 8, 8}, {3, 16, 288, 6, 5, 5}, {3, 32, 288, 7, 7, 6, 6, 6}, {3, 32, 144, 11, 11, 10}, {3, 16, 72, 3, 3, 3, 3, 2, 2},
        {7, 8, 288, 5, 1, 1, 1}, {7, 12, 288, 6, 2, 2, 2},                                  // batch 2: 182.5 -> 124.4, 270.4 -> 176.2
        {3, 32, 576, 11, 11, 10}, {3, 16, 144, 6, 5, 5},
        {7, 8, 432, 4, 3, 1}, {7, 12, 432, 7, 3, 1, 1}, {3, 32, 864, 16, 16}, {3, 16, 216, 8, 8},      // batch 3
        {7, 8, 576, 5, 2, 1}, {7, 12, 576, 7, 4, 1}, {3, 32, 1152, 16, 16},                 // batch 4: 273.5 -> 231.2, 406.8 -> 332.7
        {7, 8, 864, 6, 2}, {7, 12, 864, 5, 5, 2}, {3, 32, 1728, 11, 11, 10}, {3, 16, 1728, 8, 8},      // batch 6
    };
    // 3x3 layers that measured no better split than unsplit at these block counts
    static const int tuned_unsplit[][3] = {{3, 8, 1058}, {3, 8, 576}, {3, 8, 144}, {3, 16, 1152}, {3, 8, 2116}, {3, 8, 1152}, {3, 16, 2304}};
    if (ncu == 256) {
        for (const auto& u : tuned_unsplit)
            if (u[0] == v.ks && u[1] == nch && u[2] == nblk) return none;
        for (const auto& t : tuned)
            if (t[0] == v.ks && t[1] == nch && t[2] == nblk) {
                std::vector<int> sizes;
                for (int i = 3; i < 12 && t[i] > 0; ++i) sizes.push_back(t[i]);
                return make(sizes);
            }
    }
    // one answer per launch shape (process-wide: contexts of several host threads share it -- the header allows one context per
    // thread and ctypes releases the GIL -- hence the lock; the simulation below runs outside it, two threads may both compute a
    // missing entry, the results are equal)
    static std::map<std::vector<long>, SplitPlan> cache;
    static std::mutex cache_mu;
    const std::vector<long> key = {variant, H, W, B, groups, cout_pad, nch, ncu};
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    // per-wave MFMAs of one chunk in this block shape (32x32x2 MFMA = 64 cycles), microseconds at 2.4 GHz, ~91 % pipe rate
    const int mt_nt = ((v.th * v.tw + 31) / 32) * (v.bn / 32) / 4;
    const double t_chunk = (double)v.ks * v.ks * 8 * mt_nt * 64 / 2400.0 / 0.91;
    const double t_fixed = 1.5;                         // prologue + epilogue of one block that the co-resident block cannot hide, us
    const double out_mb = (double)H * W * B * cout_pad * groups * 4 / 1e6;
    int one[1] = {nch};
    double best = sk_simulate(one, 1, nblk, ncu, t_chunk, t_fixed);
    SplitPlan best_p = none;
    for (int S = 2; S <= 8 && S <= nch; ++S) {
        std::vector<std::vector<int>> cands;
        if (nch <= 12) {
            int cur[8];
            sk_partitions(nch, S, nch, cur, 0, cands);          // all splits into S parts, parts in descending order
        } else {
            std::vector<int> sizes(S);
            for (int s = 0; s < S; ++s) sizes[s] = nch / S + (s < nch % S ? 1 : 0);
            cands.push_back(sizes);
        }
        for (const auto& sizes : cands) {
            const double t = sk_simulate(sizes.data(), S, nblk, ncu, t_chunk, t_fixed) + 2.5 + (S + 1) * out_mb / 3.0;
            if (t < best * 0.97 - 1e-9 && (best_p.S == 1 || t < best - 1e-9)) { best = t; best_p = make(sizes); }
        }
    }
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        cache[key] = best_p;
    }
    return best_p;
}

template <int KS, int TH, int TW, int BN, int CK, int WM, int WN>
static int launch_cfg(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = ConvCfg<KS, TH, TW, BN, CK, WM, WN>;
    ConvArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    PMX_CHECK(a.cout_pad % BN == 0, PMX_ERR_INVALID, "conv: cout_pad %d not a multiple of BN %d", a.cout_pad, BN);
    PMX_CHECK(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    auto kern = conv_mfma_kernel<KS, TH, TW, BN, CK, WM, WN>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    const int lds = C::LDS_BYTES > g_min_lds ? C::LDS_BYTES : g_min_lds;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / BN), (unsigned)groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int KS, int TH, int TW, int BN, int CK, int WM, int WN>
static int launch_v5(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = ConvCfg<KS, TH, TW, BN, CK, WM, WN>;
    ConvArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    PMX_CHECK(a.cout_pad % BN == 0, PMX_ERR_INVALID, "conv: cout_pad %d not a multiple of BN %d", a.cout_pad, BN);
    PMX_CHECK(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    auto kern = conv_mfma_v5_kernel<KS, TH, TW, BN, CK, WM, WN>;
    int lds = 2 * C::IN_ELEMS * 4;
    if (lds < g_v5_lds) lds = g_v5_lds;
    if (lds < g_min_lds) lds = g_min_lds;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    if (a.ksplit < 1) a.ksplit = 1;
    a.ngroups = groups;
    PMX_CHECK(a.ksplit <= a.nch && a.ksplit <= 8, PMX_ERR_INVALID, "conv: %d K slices for %d chunks", a.ksplit, a.nch);
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / BN), (unsigned)(groups * a.ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- split-K: combine the partial-sum slabs ------------------------------------------------------------------------------
// out = [pool2x2]( slab_0 + slab_1 + ... + slab_{S-1} ) + bias, ReLU -- slabs added in slice order (left to right), so the
// result is a defined function of (ksplit, chunk order): oracle/conv_fma_ref.c reproduces it bit for bit.  One thread per
// 4 output channels of one output pixel; the kernel boundary in front of it is the release / acquire between the slice
// blocks (any XCD) and the combiner.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const SplitKReduceArgs r)
{
    const int g = blockIdx.z;
    const float* slabs = g ? r.slabs[1] : r.slabs[0];
    const float* bias = g ? r.bias[1] : r.bias[0];
    float* out = g ? r.out[1] : r.out[0];
    const int cout = g ? r.cout[1] : r.cout[0];
    const int c4n = cout >> 2;
    const int Ho = r.pool ? r.H >> 1 : r.H, Wo = r.pool ? r.W >> 1 : r.W;
    const long long total = (long long)r.B * Ho * Wo * c4n;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    const long long p = i / c4n;                        // output pixel (b, oy, ox)
    const int ox = (int)(p % Wo);
    const long long q = p / Wo;
    const int oy = (int)(q % Ho), b = (int)(q / Ho);
    float4 best;
    const int nwin = r.pool ? 4 : 1;
    for (int wi = 0; wi < nwin; ++wi) {
        const int y = r.pool ? 2 * oy + (wi >> 1) : oy, x = r.pool ? 2 * ox + (wi & 1) : ox;
        const float* src = slabs + (((long long)b * r.H + y) * r.W + x) * r.ld_slab + c;
        float4 acc = *reinterpret_cast<const float4*>(src);
        for (int s = 1; s < r.ksplit; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long long)s * r.slab_stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (wi == 0) best = acc;
        else { best.x = fmaxf(best.x, acc.x); best.y = fmaxf(best.y, acc.y); best.z = fmaxf(best.z, acc.z); best.w = fmaxf(best.w, acc.w); }
    }
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    best.x += bv.x; best.y += bv.y; best.z += bv.z; best.w += bv.w;
    if (r.relu) { best.x = fmaxf(best.x, 0.f); best.y = fmaxf(best.y, 0.f); best.z = fmaxf(best.z, 0.f); best.w = fmaxf(best.w, 0.f); }
    *reinterpret_cast<float4*>(out + p * r.ldc + c) = best;
}

int conv_splitk_reduce(const SplitKReduceArgs& r, int groups, hipStream_t stream)
{
    PMX_CHECK(r.cout[0] % 4 == 0 && (groups < 2 || r.cout[1] == r.cout[0]) && r.ldc % 4 == 0 && r.ld_slab % 4 == 0, PMX_ERR_INVALID,
              "split-K reduce: channel counts / strides must be multiples of 4");
    const int Ho = r.pool ? r.H / 2 : r.H, Wo = r.pool ? r.W / 2 : r.W;
    const long long total = (long long)r.B * Ho * Wo * (r.cout[0] / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, (unsigned)groups), dim3(256), 0, stream, r);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- pair: two chained 1x1 convolutions in one launch ------------------------------------------------------------------------
// The last two layers of every stage (CocoPoseNet.py: conv5_4 -> conv5_5, Mconv6 -> Mconv7; FaceNet / HandNet: conv6_1 -> conv6_2,
// Mconv6 -> Mconv7) are 1x1 convolutions: 128 -> CMID (+ReLU) -> cout.  As separate launches they are short, latency-bound
// kernels (24-120 us at batch 32, 13 us each at batch 1) that write and re-read the hidden map.  Here a block owns 32 pixels
// (1x1 has no spatial structure: the whole batch is one pixel list): X tile -> LDS, wave w computes hidden channels
// [w * CMID / 4, (w + 1) * CMID / 4) (weights L2 -> registers), bias + ReLU, hidden tile -> LDS, then waves 0 .. N2T - 1 each
// compute one 32-channel tile of the second layer over all CMID hidden channels.  Every output walks K in the same order
// as the single-layer kernels (chunk -> half -> k, one sequential FMA chain) -> bit-identical to running the two layers apart.
template <int CMID, int N2T>
__global__ __launch_bounds__(256) void conv1x1_pair_kernel(const PairArgs a)
{
    constexpr int CIN = 128, LDX = CIN + 4, LDH = CMID + 4, NT1 = CMID / 128;     // col tiles of the hidden layer per wave
    extern __shared__ float4 smem4[];
    // the hidden tile is written only after every wave has finished reading X (barrier below), so the two tiles share the
    // LDS space: 66 KB instead of 83 KB for CMID = 512 -> two blocks per CU
    float* const sX = reinterpret_cast<float*>(smem4);          // [32][LDX]
    float* const sH = sX;                                       // [32][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const bool g1 = blockIdx.z != 0;
    const float* in = g1 ? a.g[1].in : a.g[0].in;
    const float* w1 = g1 ? a.g[1].w1 : a.g[0].w1;
    const float* b1 = g1 ? a.g[1].b1 : a.g[0].b1;
    const float* w2 = g1 ? a.g[1].w2 : a.g[0].w2;
    const float* b2 = g1 ? a.g[1].b2 : a.g[0].b2;
    float* out = g1 ? a.g[1].out : a.g[0].out;
    const int cout = g1 ? a.g[1].cout : a.g[0].cout;
    const long long p0 = (long long)blockIdx.x * 32;

    // X tile: 32 pixels x 128 channels (zero rows past the end)
    for (int f = tid; f < 32 * (CIN / 4); f += 256) {
        const int r = f / (CIN / 4), c4 = f % (CIN / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + r < a.npix) v = *reinterpret_cast<const float4*>(in + (p0 + r) * a.lda + c4 * 4);
        *reinterpret_cast<float4*>(&sX[r * LDX + c4 * 4]) = v;
    }
    // hidden layer: wave -> NT1 column tiles
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w1), 0, 0x7fffffff, 0x00020000);
    f32x16 acc[NT1];
    float bias1[NT1];
#pragma unroll
    for (int u = 0; u < NT1; ++u) {
        bias1[u] = b1[(wave * NT1 + u) * 32 + li];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    }
    // weights one chunk ahead (double-buffered in registers): loaded at the top of the chunk that uses them, every chunk waited out an L2
    // round trip -- the kernel is latency-bound (the first chunk's fragments are requested before the barrier that publishes X)
    f32x4 bw[2][NT1][2];
    auto load_w1 = [&](f32x4 (&dst)[NT1][2], int ch) {
#pragma unroll
        for (int u = 0; u < NT1; ++u) {
            const unsigned off = (unsigned)(((ch * CMID + (wave * NT1 + u) * 32 + li) * 16 + kh * 4) * 4);
            dst[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0u, 0));
            dst[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, off + 32, 0u, 0));
        }
    };
    load_w1(bw[0], 0);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < CIN / 16; ++ch) {
        if (ch + 1 < CIN / 16) load_w1(bw[(ch + 1) & 1], ch + 1);
        __builtin_amdgcn_sched_barrier(0);          // (left alone the compiler sinks the loads back next to their use)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&sX[li * LDX + ch * 16 + s * 8 + kh * 4]);
#pragma unroll
            for (int u = 0; u < NT1; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bw[ch & 1][u][s][e], acc[u], 0, 0, 0);
        }
    }
    __syncthreads();            // all waves are done with the X tile: the hidden tile may overwrite it
    // bias + ReLU -> hidden tile in LDS (C layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * kh)
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            sH[row * LDH + (wave * NT1 + u) * 32 + li] = fmaxf(acc[u][reg] + bias1[u], 0.f);
        }
    __syncthreads();
    if (wave >= N2T) return;
    // second layer: wave -> one 32-channel tile over all CMID hidden channels
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w2), 0, 0x7fffffff, 0x00020000);
    const int n = wave * 32 + li;
    const float bias2 = b2[n];
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
    f32x4 bq[2][2];
    {
        const unsigned off = (unsigned)(((0 * a.cout_pad + n) * 16 + kh * 4) * 4);
        bq[0][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, off, 0u, 0));
        bq[0][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, off + 32, 0u, 0));
    }
#pragma unroll 2
    for (int ch = 0; ch < CMID / 16; ++ch) {
        const int cn = ch + 1 < CMID / 16 ? ch + 1 : ch;          // next chunk's weights under this chunk's MFMAs
        const unsigned off = (unsigned)(((cn * a.cout_pad + n) * 16 + kh * 4) * 4);
        bq[(ch + 1) & 1][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, off, 0u, 0));
        bq[(ch + 1) & 1][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, off + 32, 0u, 0));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&sH[li * LDH + ch * 16 + s * 8 + kh * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) y = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bq[ch & 1][s][e], y, 0, 0, 0);
        }
    }
    if (n < cout) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            float v = y[reg] + bias2;
            if (a.relu2) v = fmaxf(v, 0.f);
            if (p0 + row < a.npix) out[(p0 + row) * a.ldc + n] = v;
        }
    }
}

bool conv_pair_supported(int cin, int cmid, int cout_pad)
{
    return cin == 128 && (cmid == 128 || cmid == 512) && (cout_pad == 64 || cout_pad == 128);
}

template <int CMID, int N2T>
static int launch_pair(const PairArgs& a, int groups, hipStream_t stream)
{
    constexpr int LDS = 32 * ((CMID > 128 ? CMID : 128) + 4) * 4;       // X tile and hidden tile share the space
    auto kern = conv1x1_pair_kernel<CMID, N2T>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    const long long nblk = (a.npix + 31) / 32;
    PMX_CHECK(nblk < (1ll << 31), PMX_ERR_INVALID, "conv pair: too many pixels");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, 1, (unsigned)groups), dim3(256), LDS, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int conv_pair_launch(const PairArgs& a, int groups, hipStream_t stream)
{
    PMX_CHECK(conv_pair_supported(128, a.cmid, a.cout_pad), PMX_ERR_INVALID, "conv pair: unsupported shape (cmid %d, cout_pad %d)", a.cmid, a.cout_pad);
    PMX_CHECK(a.npix * (long long)(a.lda > a.ldc ? a.lda : a.ldc) < (1ll << 40), PMX_ERR_INVALID, "conv pair: map too large");
    if (a.cmid == 128) return a.cout_pad == 64 ? launch_pair<128, 2>(a, groups, stream) : launch_pair<128, 4>(a, groups, stream);
    return a.cout_pad == 64 ? launch_pair<512, 2>(a, groups, stream) : launch_pair<512, 4>(a, groups, stream);
}

template <int KS, int MT, int POOL>
static int launch_v6(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = V6Cfg<KS, MT, POOL>;
    ConvArgs a = a0;
    PMX_CHECK(a.W % C::W == 0 && !!a.pool == !!POOL, PMX_ERR_INVALID, "conv v6: needs a map width that is a multiple of %d (W = %d) and pool = %d",
              C::W, a.W, POOL);
    PMX_CHECK(!POOL || a.H % 2 == 0, PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv v6: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    a.tiles_x = (a.H * C::W + C::M - 1) / C::M;       // blocks per slab
    a.tiles_y = a.W / C::W;                            // slabs per image
    auto kern = conv_mfma_v6_kernel<KS, MT, POOL>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / 128), (unsigned)groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int KS, int MT, int POOL>
static int launch_v7(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = V7Cfg<KS, MT, POOL>;
    ConvArgs a = a0;
    PMX_CHECK(a.W % C::W == 0 && !!a.pool == !!POOL, PMX_ERR_INVALID, "conv v7: needs a map width that is a multiple of %d (W = %d) and pool = %d",
              C::W, a.W, POOL);
    PMX_CHECK(!POOL || a.H % 2 == 0, PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv v7: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    a.tiles_x = (a.H * C::W + C::M - 1) / C::M;
    a.tiles_y = a.W / C::W;
    auto kern = conv_bf16x3_kernel<KS, MT, POOL>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / 128), (unsigned)groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int KS>
static int launch_v8(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = ConvCfg<KS, 8, 8, 64, 16, 2, 2>;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 7) / 8;
    a.tiles_y = (a.H + 7) / 8;
    PMX_CHECK(a.cout_pad % 64 == 0, PMX_ERR_INVALID, "conv: cout_pad %d not a multiple of 64", a.cout_pad);
    PMX_CHECK(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    if (a.ksplit < 1) a.ksplit = 1;
    a.ngroups = groups;
    PMX_CHECK(a.ksplit <= a.nch && a.ksplit <= 8, PMX_ERR_INVALID, "conv: %d K slices for %d chunks", a.ksplit, a.nch);
    auto kern = conv_bf16x3_small_kernel<KS>;
    int lds = 2 * C::HALO_H * C::HALO_W * 112;
    if (lds < g_v5_lds) lds = g_v5_lds;         // at most two blocks per CU, as for the v5 kernels
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / 64), (unsigned)(groups * a.ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int conv_launch(int variant, const ConvArgs& a, int groups, hipStream_t stream)
{
    switch (variant) {
        case V8_K7_SMALL: return launch_v8<7>(a, groups, stream);
        case V8_K3_SMALL: return launch_v8<3>(a, groups, stream);
        case V7_K7: return launch_v7<7, 17, 0>(a, groups, stream);
        case V7_K3: return launch_v7<3, 17, 0>(a, groups, stream);
        case V7_K3_POOL: return launch_v7<3, 17, 1>(a, groups, stream);
        case V7M9_K7: return launch_v7<7, 9, 0>(a, groups, stream);
        case V7M9_K3: return launch_v7<3, 9, 0>(a, groups, stream);
        case V7M9_K3_POOL: return launch_v7<3, 9, 1>(a, groups, stream);
        case 0: return launch_cfg<7, 8, 16, 128, 16, 2, 2>(a, groups, stream);
        case 1: return launch_cfg<3, 8, 16, 128, 16, 2, 2>(a, groups, stream);
        case 2: return launch_cfg<3, 8, 16, 64, 16, 4, 1>(a, groups, stream);
        case 3: return launch_cfg<1, 8, 16, 128, 16, 2, 2>(a, groups, stream);
        case 4: return launch_cfg<1, 8, 16, 64, 16, 4, 1>(a, groups, stream);
        case 5: return launch_cfg<7, 8, 8, 64, 16, 2, 2>(a, groups, stream);
        case 6: return launch_cfg<3, 8, 8, 64, 16, 2, 2>(a, groups, stream);
        case 7: return launch_cfg<1, 8, 8, 64, 16, 2, 2>(a, groups, stream);
        case 8: return launch_cfg<7, 2, 46, 128, 16, 1, 4>(a, groups, stream);
        case 9: return launch_cfg<3, 2, 46, 128, 16, 1, 4>(a, groups, stream);
        case V5_K7_STRIP: return launch_v5<7, 2, 46, 128, 16, 1, 4>(a, groups, stream);
        case V5_K3_STRIP: return launch_v5<3, 2, 46, 128, 16, 1, 4>(a, groups, stream);
        case V5_K7: return launch_v5<7, 8, 16, 128, 16, 1, 4>(a, groups, stream);
        case V5_K3: return launch_v5<3, 8, 16, 128, 16, 1, 4>(a, groups, stream);
        case V5_K3_N64: return launch_v5<3, 8, 16, 64, 16, 2, 2>(a, groups, stream);
        case V5_K7_SMALL: return launch_v5<7, 8, 8, 64, 16, 2, 2>(a, groups, stream);
        case V5_K3_SMALL: return launch_v5<3, 8, 8, 64, 16, 2, 2>(a, groups, stream);
        case V6_K7: return launch_v6<7, 17, 0>(a, groups, stream);
        case V6_K3: return launch_v6<3, 17, 0>(a, groups, stream);
        case V6_K3_POOL: return launch_v6<3, 17, 1>(a, groups, stream);
        case C3: return launch_c3(a, groups, stream);
        case V6M9_K7: return launch_v6<7, 9, 0>(a, groups, stream);
        case V6M9_K3: return launch_v6<3, 9, 0>(a, groups, stream);
        case V6M9_K3_POOL: return launch_v6<3, 9, 1>(a, groups, stream);
        case V5T_K7: return launch_v5<7, 16, 8, 128, 16, 1, 4>(a, groups, stream);
        case V5T_K3: return launch_v5<3, 16, 8, 128, 16, 1, 4>(a, groups, stream);
        case V5T_K3_N64: return launch_v5<3, 16, 8, 64, 16, 2, 2>(a, groups, stream);
    }
    pmx_set_error("conv_launch: unknown variant %d", variant);
    return PMX_ERR_INVALID;
}

template <int KS, int POOL, int UNIT = 0>
static int launch_wino(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, 0>;
    ConvArgs a = a0;
    PMX_CHECK(!!a.pool == !!POOL, PMX_ERR_INVALID, "conv wino: pool mismatch");
    PMX_CHECK(!POOL || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.H * a.W * a.lda < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    PMX_CHECK((long long)(a.H + 2) * (a.W + 2) * a.ldc * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: output image too large for 32-bit byte offsets");
    a.tiles_x = (a.W + C::TW - 1) / C::TW;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.run_j0 = a.run_nb = 0;
    auto kern = conv_wino_kernel<KS, POOL, UNIT, 0>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    if (UNIT) { a.ngroups = groups; PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan"); }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / 128), (unsigned)(groups * (UNIT ? a.ksplit : 1)));
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// a.nch = input channels / 32 (chunks of the Winograd kernel), a.g[].w = transformed weights (pmx_api.hip::pack_wino)
// a.ksplit > 1: unit mode -- a.ksplit = ceil(nch / g) (+ 3 for 7x7: row 6, column 6, tap (6, 6)) slabs at a.g[].out + unit * a.slab_stride, g = a.kbounds
int conv_wino_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    if (a.ksplit > 1) return ks == 7 ? launch_wino<7, 0, 1>(a, groups, stream) : launch_wino<3, 0, 1>(a, groups, stream);
    if (ks == 7) return launch_wino<7, 0>(a, groups, stream);
    return a.pool ? launch_wino<3, 1>(a, groups, stream) : launch_wino<3, 0>(a, groups, stream);
}

// run geometry (46-pixel-wide maps): the blocks [a.run_j0, a.run_j0 + a.run_nb) of every image, 32 consecutive Winograd tiles each
// Events for the NEXT run-geometry launch of this thread (profile mode 2): handed to hipExtLaunchKernelGGL, which stamps them from the
// dispatch's own completion signal -- the kernel's execution time without the two barrier packets of hipEventRecord (~6 us of idle stream
// per pair, 25 pairs per step inside bench.py's timed region)
static thread_local hipEvent_t g_launch_ev0 = nullptr, g_launch_ev1 = nullptr;
void conv_set_launch_events(hipEvent_t e0, hipEvent_t e1) { g_launch_ev0 = e0; g_launch_ev1 = e1; }

template <int KS, int POOL, int UNIT, int GEOM>
static int launch_wino_run_g(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, GEOM>;
    ConvArgs a = a0;
    PMX_CHECK(!!a.pool == !!POOL && a.W % C::RUN_W == 0 && a.W > 0, PMX_ERR_INVALID, "conv wino runs: map width must be a multiple of %d (W %d)", C::RUN_W, a.W);
    PMX_CHECK(!POOL || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    a.run_nslab = a.W / C::RUN_W;
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.H * a.W * a.lda * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    PMX_CHECK((long long)(a.H + 2) * (a.W + 2) * a.ldc * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: output image too large for 32-bit byte offsets");
    const int nblk = (C::RUN_TX * ((a.H + 1) / 2) + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES;
    PMX_CHECK(a.run_j0 >= 0 && a.run_nb >= 1 && a.run_j0 + a.run_nb <= nblk, PMX_ERR_INVALID, "conv wino runs: blocks [%d, %d) of %d", a.run_j0, a.run_j0 + a.run_nb, nblk);
    a.tiles_x = a.tiles_y = 0;
    PMX_CHECK(GEOM == 2 || a.run_nslab == 1, PMX_ERR_INVALID, "conv wino runs: single-slab kernel on a %d-wide map", a.W);
    auto kern = conv_wino_kernel<KS, POOL, UNIT, GEOM>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    if (UNIT) { a.ngroups = groups; PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan"); }
    dim3 grid((unsigned)(a.run_nb * a.B * a.run_nslab), (unsigned)(a.cout_pad / 128), (unsigned)(groups * (UNIT ? a.ksplit : 1)));
    if (g_launch_ev0) {
        hipExtLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, g_launch_ev0, g_launch_ev1, 0, a);
        g_launch_ev0 = g_launch_ev1 = nullptr;
    } else
        hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int KS, int POOL, int UNIT>
static int launch_wino_run(const ConvArgs& a, int groups, hipStream_t stream)
{
    return a.W == 2 * PMX_WINO_RUN_TX ? launch_wino_run_g<KS, POOL, UNIT, 1>(a, groups, stream) : launch_wino_run_g<KS, POOL, UNIT, 2>(a, groups, stream);
}

template <int KS>
static int launch_wino_merged(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, 3>;
    ConvArgs a = a0;
    PMX_CHECK(wino_tail_mergeable(a.B, a.H, a.W) && a.run_j0 == C::RUN_TX * ((a.H + 1) / 2) / PMX_WINO_RUN_TILES, PMX_ERR_INVALID,
              "conv wino merged tails: not a mergeable tail (B %d, %d x %d, full blocks %d)", a.B, a.H, a.W, a.run_j0);
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.B * a.H * a.W * a.lda * 4 < (1ll << 31), PMX_ERR_INVALID, "conv wino merged tails: batch too large for 32-bit offsets");
    PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan");
    a.run_nslab = 1; a.run_nb = wino_tail_merged_blocks(a.B, a.H); a.tiles_x = a.tiles_y = 0; a.ngroups = groups;
    auto kern = conv_wino_kernel<KS, 0, 1, 3>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)a.run_nb, (unsigned)(a.cout_pad / 128), (unsigned)(groups * a.ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int conv_wino_merged_tail_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    return ks == 7 ? launch_wino_merged<7>(a, groups, stream) : launch_wino_merged<3>(a, groups, stream);
}

int conv_wino_run_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    if (a.ksplit > 1) return ks == 7 ? launch_wino_run<7, 0, 1>(a, groups, stream) : launch_wino_run<3, 0, 1>(a, groups, stream);      // (the combine pools)
    if (ks == 7) return launch_wino_run<7, 0, 0>(a, groups, stream);
    return a.pool ? launch_wino_run<3, 1, 0>(a, groups, stream) : launch_wino_run<3, 0, 0>(a, groups, stream);
}

// ---- combine of the unit-mode slabs of a block range of the run geometry (see WinoTailReduceArgs) -------------------------------------------
// One thread per (image, block, tile, pixel of the tile, 4 output channels): slabs added in unit order (left to right), then bias, ReLU --
// the arithmetic of conv_splitk_reduce_kernel on the compact slab layout of conv_wino_kernel<KS, 0, 1, 1>.
__global__ __launch_bounds__(256) void conv_wino_tail_reduce_kernel(const WinoTailReduceArgs r)
{
    const int g = blockIdx.z;
    const float* slabs = g ? r.slabs[1] : r.slabs[0];
    const float* bias = g ? r.bias[1] : r.bias[0];
    float* out = g ? r.out[1] : r.out[0];
    const int cout = g ? r.cout[1] : r.cout[0];
    const int c4n = cout >> 2;
    // pooled layers: one thread per TILE (its four pixels are the pooling window), else one per pixel
    const int ppt = r.pool ? 1 : 4;
    const int nt = PMX_WINO_RUN_TX * ((r.H + 1) >> 1) - r.run_j0 * PMX_WINO_RUN_TILES;     // merged: tail tiles per image
    const long long total = r.merged ? (long long)r.B * nt * ppt * c4n : (long long)r.B * r.nslab * r.run_nb * (PMX_WINO_RUN_TILES * ppt) * c4n;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    const long long q = i / c4n;                        // ((image-slab * run_nb + block) * 32 + tile) [* 4 + pixel]
    const int k = r.pool ? 0 : (int)(q & 3);
    const long long qt = r.pool ? q : q >> 2;           // (image-slab * run_nb + block) * 32 + tile
    int b, sx0, t;
    if (r.merged) {                                     // qt = stream position: image * nt + tail tile (= block of the stream * 32 + row)
        b = (int)(qt / nt); sx0 = 0;
        t = r.run_j0 * PMX_WINO_RUN_TILES + (int)(qt - (long long)b * nt);
    } else {
        const int m = (int)(qt & (PMX_WINO_RUN_TILES - 1));
        const long long bj = qt >> 5;
        const int jl = (int)(bj % r.run_nb);
        const long long bs = bj / r.run_nb;
        b = (int)(bs / r.nslab); sx0 = (int)(bs % r.nslab) * (2 * PMX_WINO_RUN_TX);
        t = (r.run_j0 + jl) * PMX_WINO_RUN_TILES + m;
    }
    const int ty = t / PMX_WINO_RUN_TX, tx = t - ty * PMX_WINO_RUN_TX;
    float4 best;
    for (int kk = 0; kk < (r.pool ? 4 : 1); ++kk) {
        const int kq = r.pool ? kk : k;
        const float* src = slabs + (qt * 4 + kq) * r.ld_slab + c;
        float4 acc = *reinterpret_cast<const float4*>(src);
        for (int s = 1; s < r.S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long long)s * r.slab_stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (kk == 0) best = acc;
        else { best.x = fmaxf(best.x, acc.x); best.y = fmaxf(best.y, acc.y); best.z = fmaxf(best.z, acc.z); best.w = fmaxf(best.w, acc.w); }
    }
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    best.x += bv.x; best.y += bv.y; best.z += bv.z; best.w += bv.w;
    if (r.relu) { best.x = fmaxf(best.x, 0.f); best.y = fmaxf(best.y, 0.f); best.z = fmaxf(best.z, 0.f); best.w = fmaxf(best.w, 0.f); }
    if (r.pool) {
        const int Hp = r.H >> 1, Wp = r.W >> 1, py = ty, px = (sx0 >> 1) + tx;
        if (py >= Hp || px >= Wp) return;
        *reinterpret_cast<float4*>(out + (((long long)b * Hp + py) * Wp + px) * r.ldc + c) = best;
    } else {
        const int gy = 2 * ty + (k >> 1), gx = sx0 + 2 * tx + (k & 1);
        if (gy >= r.H || gx >= r.W) return;                 // tiles past the end of the map, the odd last row
        *reinterpret_cast<float4*>(out + (((long long)b * r.H + gy) * r.W + gx) * r.ldc + c) = best;
    }
}

int conv_wino_tail_reduce(const WinoTailReduceArgs& r, int groups, hipStream_t stream)
{
    PMX_CHECK(r.cout[0] % 4 == 0 && (groups < 2 || r.cout[1] == r.cout[0]) && r.ldc % 4 == 0 && r.ld_slab % 4 == 0, PMX_ERR_INVALID,
              "winograd tail reduce: channel counts / strides must be multiples of 4");
    static_assert(PMX_WINO_RUN_TILES == 32, "tile index bits");
    const int nt = PMX_WINO_RUN_TX * ((r.H + 1) / 2) - r.run_j0 * PMX_WINO_RUN_TILES;
    const long long total = (r.merged ? (long long)r.B * nt * (r.pool ? 1 : 4) : (long long)r.B * r.nslab * r.run_nb * (PMX_WINO_RUN_TILES * (r.pool ? 1 : 4))) * (r.cout[0] / 4);
    hipLaunchKernelGGL(conv_wino_tail_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, (unsigned)groups), dim3(256), 0, stream, r);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

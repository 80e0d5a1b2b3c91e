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

#include "conv_direct.h"

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
// (`done` is a plain flag per kernel and device: two host threads racing here both call hipFuncSetAttribute with the same value -- idempotent)
int conv_allow_big_lds(const void* kern, bool (&done)[PMX_MAX_DEVICES])
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
int conv_v5_lds() { return g_v5_lds; }

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
    // second layer: N2T of the four waves take one 32-channel tile each over all CMID hidden channels.  WHICH waves rotates with the block:
    // with waves 0 .. N2T - 1 of every block the SIMDs those waves sit on carried 2x the matrix work of the others (two blocks per CU:
    // -3 %, profiles/r05_pair_rotation_ab.json); a tile that holds padding channels only (the 19 heat-map channels of a 64-channel pad)
    // is not computed at all
    const int w2t = (wave + 2 * (int)(blockIdx.x & 1) + (int)((blockIdx.x >> 1) & 1)) & 3;
    if (w2t >= N2T || w2t * 32 >= cout) return;
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w2), 0, 0x7fffffff, 0x00020000);
    const int n = w2t * 32 + li;
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

int conv_launch(int variant, const ConvArgs& a, int groups, hipStream_t stream)
{
    switch (variant) {
        case V8_K7_SMALL: case V8_K3_SMALL: case V7_K7: case V7_K3: case V7_K3_POOL: case V7M9_K7: case V7M9_K3: case V7M9_K3_POOL: {
            // the opt-in bf16x3 kernels live in conv_bf16x3.hip, which the default build leaves out (weak symbol: null when absent)
            PMX_CHECK(conv_bf16x3_launch != nullptr, PMX_ERR_INVALID, "conv_launch: this build carries no bf16x3 kernels (PMX_BUILD_BF16X3=1)");
            const int ks = conv_variant(variant).ks;
            const int mt = (variant == V8_K7_SMALL || variant == V8_K3_SMALL) ? 0 : (variant >= V7M9_K7 ? 9 : 17);
            return conv_bf16x3_launch(ks, mt, variant == V7_K3_POOL || variant == V7M9_K3_POOL, a, groups, stream);
        }
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


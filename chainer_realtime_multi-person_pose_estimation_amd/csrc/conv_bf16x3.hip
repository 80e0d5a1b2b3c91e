// conv_bf16x3.hip -- the opt-in bf16x3 kernels (option "precision" = 1; DESIGN.md 4.1.5: frozen after round 5, NOT part of the default
// build -- native.build() compiles this file only with PMX_BUILD_BF16X3=1 in the environment; without it pmx_set_option(ctx, "precision",
// 1) fails with a clear message).  fp32 is the arithmetic of record; these kernels compute the same convolutions from three bf16 terms
// per fp32 value on v_mfma_f32_32x32x16_bf16: fp32-grade accuracy, another summation.
#include <hip/hip_ext.h>
#include "conv_direct.h"

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
    if (lds < conv_v5_lds()) lds = conv_v5_lds();         // at most two blocks per CU, as for the v5 kernels
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.cout_pad / 64), (unsigned)(groups * a.ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// which = the kernel's block geometry: mt = 17 / 9 (v7: the v6 geometry, one block per CU) or 0 (v8: the small v5 tiles, split-K capable)
int conv_bf16x3_launch(int ks, int mt, int pool, const ConvArgs& a, int groups, hipStream_t stream)
{
    if (mt == 0) return ks == 7 ? launch_v8<7>(a, groups, stream) : launch_v8<3>(a, groups, stream);
    if (mt == 17) {
        if (ks == 7) return launch_v7<7, 17, 0>(a, groups, stream);
        return pool ? launch_v7<3, 17, 1>(a, groups, stream) : launch_v7<3, 17, 0>(a, groups, stream);
    }
    if (ks == 7) return launch_v7<7, 9, 0>(a, groups, stream);
    return pool ? launch_v7<3, 9, 1>(a, groups, stream) : launch_v7<3, 9, 0>(a, groups, stream);
}

// what native.has_bf16x3() looks for in the shared library
extern "C" const int conv_bf16x3_probe = 1;

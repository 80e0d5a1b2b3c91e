// conv1_wino.hip -- conv1_1 (3 -> 64) + conv1_2 (64 -> 64, ReLU, 2x2 max-pool) of the VGG stem (models/CocoPoseNet.py:26-27,133-135; the
// same two layers open FaceNet / HandNet) as ONE launch with conv1_2 in Winograd F(2x2, 3x3) fp32: 16 instead of 36 products per output
// and channel pair on the largest maps of the network.  Until round 4 conv1_2 was the one big layer still on the direct kernel
// (conv1_fused_kernel, 2.57 ms of a 35.3 ms step at batch 32: 0.83 of the fp32-MFMA peak in DIRECT products).
//
// Why a kernel of its own: conv_wino_kernel's block is 32 Winograd tiles x 128 output channels (one wave per SIMD, 16 accumulator tiles =
// 256 AGPRs per wave); with 64 output channels half of its waves would idle.  Here a block is 64 tiles (8 x 8 = a 16 x 16 pixel output
// square) x 64 channels: wave w = tile half (w >> 1) x channel half (w & 1), still 32 tiles x 32 channels x 16 frequencies per wave.
// Twice the tiles means twice the transformed input per output channel, so U is kept as two halves of FOUR planes (one row of V = B^T d B)
// instead of eight: phases of 4 planes x 4 k8-steps x 4 MFMAs = 64 MFMAs per wave, during which every thread transforms its two (tile, 4
// channels) items of the next row (8 raw reads, 8 packed-add pairs, 4 U stores each), one instruction per slot between two MFMAs.
// Cin = 64 is two chunks of 32: 8 phases = 512 MFMAs per wave and block.
// conv1_1 is RECOMPUTED on the block's 18 x 18 halo from a 20 x 20 x 3 input patch (as conv1_fused_kernel does on its 10 x 18 halo): 10
// row tiles of 32 pixels x 2 channel halves x 14 MFMAs (K = 27 packed tap-major into 14 k-pairs; the last 4 of the 324 pixels on the
// vector ALU), bias + ReLU, zero outside the image
// (= conv1_2's padding), into LDS, where all 64 channels stay for the whole block -- the 1.1 GB round trip of conv1_1's output (batch 32)
// never happens and the phases have no halo staging at all.
// LDS: raw halo 324 pixels x 68 floats (88 128 B) + U 2 x 4 x 64 x 36 floats (73 728 B) + 64 bias values = 162 112 of 163 840 bytes.
// Arithmetic = conv3x3_c3_kernel's chain for conv1_1 and conv_wino_kernel<3, 1>'s for conv1_2 (same transforms in the same order, one FMA
// chain per frequency over chunk -> k8-step -> k, same output transform, pool, bias, ReLU): bit-identical to running the two layers apart,
// and to oracle/conv_fma_ref (conv_fma for conv1_1, conv_wino_ref for conv1_2).
#include <type_traits>
#include "pmx_common.h"
#include "wino_util.h"

// Diagnostic builds only (tools/kernel_variants.py; the product is built with 0; results are wrong, only the time is read): leave out
// 1: conv1_1 (tiles and the vector-ALU pixels), 2: the phases' transform slots, 4: their weight loads, 8: the phases altogether, 16: the
// output stores, 32: the per-lane gather of conv1_1's weights
#ifndef PMX_C1W_ABLATE
#define PMX_C1W_ABLATE 0
#endif
namespace {
constexpr int TT = 8;                     // Winograd tiles per block side (16 x 16 output pixels before the pool)
constexpr int HW1 = 2 * TT + 2;           // halo of conv1_2's input = conv1_1's output: 18 x 18 pixels
constexpr int NPX = HW1 * HW1;            // 324
constexpr int PW = HW1 + 2, PPX = PW * PW;     // input patch of conv1_1: 20 x 20 pixels x 3 channels
constexpr int LDA = 68;                   // raw-halo pixel pitch (floats): 64 channels + 4 (16-lane b128 reads / writes of 8 pixels tile the banks)
constexpr int LDU = 36;                   // U pitch per (plane, tile): 32 channels + 4
constexpr int U_HALF = 4 * 64 * LDU;      // four planes x 64 tiles
constexpr int RAW_ELEMS = NPX * LDA, U_ELEMS = 2 * U_HALF;
constexpr int LDS_BYTES = (RAW_ELEMS + U_ELEMS + 64) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "conv1 Winograd kernel: LDS");
static_assert(PPX * 3 <= U_ELEMS, "the input patch lives in U before the first transform");
}

__global__ __launch_bounds__(256, 1) void conv1_wino_kernel(const ConvArgs a)
{
    extern __shared__ float4 smem4[];
    float* const s_raw = reinterpret_cast<float*>(smem4);
    float* const s_u = s_raw + RAW_ELEMS;
    float* const s_bias = s_u + U_ELEMS;
    float* const s_patch = s_u;            // (dead before U half 0 is first written)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    const int th = wave >> 1, chh = wave & 1;
    int tile;
    {   // consecutive blocks of an XCD = neighbouring squares (their halos overlap in that XCD's L2)
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // heterogeneous launch (a.nseg > 0; pmx_common.h::ConvSeg): the segment this block's square belongs to -- scalar work on block-uniform
    // values; from here on the block is a block of a launch of that segment alone
    int H = a.H, W = a.W, s_tiles_x = a.tiles_x, tiles_per_img = a.tiles_x * a.tiles_y;
    size_t s_pix_in = 0, s_pix_out = 0;
    if (a.nseg > 0) {
        int sg = 0;
        for (int k = 1; k < a.nseg; ++k) sg = tile >= a.segs[k].tile0 ? k : sg;
        const ConvSeg S = a.segs[sg];
        H = S.H; W = S.W; s_tiles_x = S.tiles_x; tiles_per_img = S.tiles_img;
        s_pix_in = (size_t)(unsigned)S.pix0; s_pix_out = (size_t)(unsigned)S.pixo;
        tile -= S.tile0;
    }
    const int bimg = tile / tiles_per_img;
    const int trem = tile - bimg * tiles_per_img;
    const int y0 = (trem / s_tiles_x) * (2 * TT), x0 = (trem % s_tiles_x) * (2 * TT);
    constexpr int OUTSIDE = (int)0x80000000;

    // ---- input patch: 20 x 20 pixels x 3 channels through a buffer resource spanning the image (outside = 0 = conv1_1's padding).
    // Either the preprocessed float input (NHWC, a.lda floats per pixel) or -- a.g[1].in set -- the uint8 BGR image itself, preprocessed
    // here: x / divisor - 0.5 in float32, the reference's two operations (pose_detector.py:428-429; prep_u8_kernel's arithmetic), which
    // saves the 64-bytes-per-pixel float copy of the network input (277 MB written and read back per batch of 32) and a launch.
    if (a.g[1].in) {
        const uint8_t* src8 = reinterpret_cast<const uint8_t*>(a.g[1].in) + (s_pix_in + (size_t)bimg * H * W) * 3;
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src8), 0, (unsigned)(H * W * 3), 0x00020000);
        const float divisor = __builtin_bit_cast(float, (unsigned)a.kbounds);
        unsigned char pb[2][3];
        bool pin[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + r * 256;
            const int py = f / PW, px = f - py * PW;
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            pin[r] = f < PPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const int o = pin[r] ? (gy * W + gx) * 3 : OUTSIDE;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) pb[r][ch] = __builtin_amdgcn_raw_buffer_load_b8(irsrc, o, ch, 0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + r * 256;
            if (f < PPX) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    s_patch[f * 3 + ch] = pin[r] ? __fsub_rn(__fdiv_rn((float)pb[r][ch], divisor), 0.5f) : 0.f;
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[0].in + (s_pix_in + (size_t)bimg * H * W) * a.lda), 0,
                                                                               (unsigned)(H * W * a.lda) * 4u, 0x00020000);
        float4 pv[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + r * 256;
            const int py = f / PW, px = f - py * PW;
            const int gy = y0 - 2 + py, gx = x0 - 2 + px;
            const bool in = f < PPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            pv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, in ? (gy * W + gx) * a.lda * 4 : OUTSIDE, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int f = tid + r * 256;
            if (f < PPX) { s_patch[f * 3 + 0] = pv[r].x; s_patch[f * 3 + 1] = pv[r].y; s_patch[f * 3 + 2] = pv[r].z; }
        }
    }
    // conv1_1 weights of this wave's 32 output channels (channel half chh), K = 27 packed into 14 k-pairs; its bias: the 16 channels of a
    // lane's registers (transposed tile: register reg = channel 8 (reg >> 2) + 4 kh + (reg & 3) of the half)
    float wv[14];
    int koff[14];
    {
        const float* wp = a.g[1].w + chh * (14 * 64) + lane;      // packed for this kernel [half][k-pair 14][k of the pair 2][32] (pmx_api.hip::ensure_conv1_pack)
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k = 2 * s + kh, kk = k < 27 ? k : 26;
            wv[s] = (PMX_C1W_ABLATE & 32) ? 1.f : wp[s * 64];     // (k = 27: the pack holds 0)
            const int tap = kk / 3;
            koff[s] = ((tap / 3) * PW + tap % 3) * 3 + kk % 3;
        }
    }
    f32x4 b1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1[q] = *reinterpret_cast<const f32x4*>(a.g[1].bias + chh * 32 + 8 * q + 4 * kh);
    if (tid < 64) s_bias[tid] = a.g[0].bias[tid];
    // conv1_2's transformed weights [plane 16][chunk 2][channel half 2][k8-step 4][32][8] (pmx_api.hip::pack_wino, cout_pad = 64): a ring of
    // 16 fragments, requested 8 steps (32 MFMAs) ahead across all phase boundaries
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[0].w), 0, 0x7fffffff, 0x00020000);
    const unsigned b_off = (unsigned)((chh * 1024 + li * 8 + kh * 4) * 4);
    constexpr unsigned panel_b = 64u * 32u * 4u, freq_b = panel_b * 2u;
    auto wload = [&](int g) -> f32x4 {                  // global step g = phase * 16 + plane-in-row * 4 + k8-step (compile-time)
        const int P = g >> 4, s = g & 15;
        const unsigned so = (unsigned)(4 * (P & 3) + (s >> 2)) * freq_b + (unsigned)(P >> 2) * panel_b;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)(s & 3) * 1024u, so, 0));
    };
    f32x4 bw[16];
#pragma unroll
    for (int g = 0; g < 8; ++g) bw[g] = wload(g);
    __syncthreads();

    // ---- conv1_1 on the halo.  Pixels 0..319 = ten row tiles of 32 on the matrix core: wave = channel half chh x row tiles (wave >> 1)
    // + 2 i, i = 0..4, as a software pipeline -- while the 14 MFMAs of tile i run, the lane gathers the patch values of tile i + 1 (one
    // ds_read_b32 per gap) and finishes tile i - 1 (accumulator -> bias, ReLU, padding mask -> LDS), a few instructions per gap; left as
    // five sequential chains every tile paid its gather latency, an s_nop 15 and its 70-instruction epilogue with the matrix pipe idle.
    // The weights are the matrix core's ROW operand, so a lane ends up with 16 channels of ITS pixel: four 16-byte LDS stores per tile.
    // The last four pixels (320..323) are one (pixel, channel) per thread on the vector ALU, the same fused-multiply-add chain k = 0..26
    // (an eleventh row tile would be two more 14-MFMA jobs for four pixels, on two of the four waves only).
    const float b1s = a.g[1].bias[chh * 32 + li];
    int c1_m[5], c1_pb[5];
    float c1_hi[5];                                    // upper clamp of the tile's pixel: +inf inside the image, 0 outside (conv1_2's zero padding)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int m = ((wave >> 1) + 2 * i) * 32 + li;
        const int hy = m / HW1, hx = m - hy * HW1;
        c1_m[i] = m;
        c1_pb[i] = (hy * PW + hx) * 3;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        c1_hi[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? __builtin_inff() : 0.f;
    }
    if (!(PMX_C1W_ABLATE & 1)) {
        const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float pa[2][14];
        f32x16 acc1[2];
        f32x4 ev;
#pragma unroll
        for (int sg = 0; sg < 14; ++sg) pa[0][sg] = s_patch[c1_pb[0] + koff[sg]];
        // piece `sg` of the epilogue of a finished tile: quad q = sg / 3 of its 16 channels: (0) out of the accumulator + bias, (1) ReLU
        // and mask in one median -- med3(x, 0, +inf) = max(x, 0), med3(x, 0, 0) = 0 --, (2) the 16-byte store
        auto c1_finish = [&](const f32x16& acc, int i, int sg) {
            const int q = sg / 3, part = sg - 3 * q;
            if (q >= 4) return;
            if (part == 0) ev = pk_add4(f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]}, b1[q]);
            else if (part == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ev[e] = __builtin_amdgcn_fmed3f(ev[e], 0.f, c1_hi[i]);
            } else *reinterpret_cast<f32x4*>(&s_raw[c1_m[i] * LDA + chh * 32 + 8 * q + 4 * kh]) = ev;
        };
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int sg = 0; sg < 14; ++sg) {
                acc1[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[sg], pa[i & 1][sg], sg == 0 ? z16 : acc1[i & 1], 0, 0, 0);
                if (i + 1 < 5) pa[(i + 1) & 1][sg] = s_patch[c1_pb[i + 1] + koff[sg]];
                if (i > 0) c1_finish(acc1[(i - 1) & 1], i - 1, sg);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // pixels 320..323 (halo row 17, columns 14..17) while the last tile's MFMAs drain: thread = channel chh * 32 + li of pixel
        // 320 + 2 (wave >> 1) + kh; the odd / even k's of the channel sit in the other half of the wave
        {
            const int p4 = 2 * (wave >> 1) + kh, m = 320 + p4;
            const int hy = m / HW1, hx = m - hy * HW1, pb = (hy * PW + hx) * 3;
            float accv = 0.f;
#pragma unroll
            for (int sg = 0; sg < 14; ++sg) {
                const float other = __shfl_xor(wv[sg], 32);
                const float we = kh ? other : wv[sg], wo = kh ? wv[sg] : other;
                const int k0 = 2 * sg, k1 = 2 * sg + 1;
                accv = __builtin_fmaf(s_patch[pb + ((k0 / 9) * PW + (k0 / 3) % 3) * 3 + k0 % 3], we, accv);
                if (k1 < 27) accv = __builtin_fmaf(s_patch[pb + ((k1 / 9) * PW + (k1 / 3) % 3) * 3 + k1 % 3], wo, accv);
            }
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const float hi = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? __builtin_inff() : 0.f;
            s_raw[m * LDA + chh * 32 + li] = __builtin_amdgcn_fmed3f(accv + b1s, 0.f, hi);
        }
#pragma unroll
        for (int sg = 0; sg < 12; ++sg) c1_finish(acc1[0], 4, sg);      // (tile 4 = acc1[4 & 1])
    }
    __syncthreads();

    // ---- conv1_2.  Transform items of this thread: tiles tt and tt + 32 (tile = 8 ty + tx of the block), channels 4 tc .. + 3 of the chunk.
    // Row i of V = B^T d B of a 4 x 4 window d:  w[jx] = d[ra][jx] (-|+) d[rb][jx] with (ra, rb, op) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -);
    // V[i][0..3] = (w0 - w2, w1 + w2, w2 - w1, w1 - w3) -- conv_wino_kernel's operations, row by row.
    // (the two tiles whose items share a 16-lane LDS access are FOUR tile columns apart, not neighbours: 8 pixels x 68 floats = 32 banks,
    //  so their 128-byte runs of the raw halo no longer overlap -- neighbours, 2 pixels apart, overlapped in 24 of 32 banks: PMC 37 % of the
    //  LDS cycles were conflicts; a relabelling of which thread transforms which tile, nothing else)
    const int tg = tid >> 3, tc = tid & 7;
    const int tt = (tg & ~7) + ((tg & 7) >> 1) + 4 * (tg & 1);
    int t_raw[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int t = tt + 32 * it;
        t_raw[it] = ((2 * (t >> 3)) * HW1 + 2 * (t & 7)) * LDA + tc * 4;
    }
    const int t_u = tt * LDU + tc * 4;                 // (+ 32 * LDU for the second item)
    auto row_a = [](int i) { return i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : 1; };
    auto row_b = [](int i) { return i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3; };
    f32x4 dA[2][4], dB[2][4], wq[2][4], vq[2][4];
    auto tr_read = [&](int it, int k, int ch, int i) {          // k = 0..7: row ra (k < 4) / rb, column k & 3
        const int row = k < 4 ? row_a(i) : row_b(i), jx = k & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&s_raw[t_raw[it] + (row * HW1 + jx) * LDA + ch * 32]);
        if (k < 4) dA[it][jx] = v; else dB[it][jx] = v;
    };
    auto tr_w = [&](int it, int jx, int i) { wq[it][jx] = i == 1 ? pk_add4(dA[it][jx], dB[it][jx]) : pk_sub4(dA[it][jx], dB[it][jx]); };
    auto tr_v = [&](int it, int jv) {
        vq[it][jv] = jv == 0 ? pk_sub4(wq[it][0], wq[it][2]) : jv == 1 ? pk_add4(wq[it][1], wq[it][2]) : jv == 2 ? pk_sub4(wq[it][2], wq[it][1]) : pk_sub4(wq[it][1], wq[it][3]);
    };
    auto tr_store = [&](int it, int jv, int half) {
        *reinterpret_cast<f32x4*>(&s_u[half * U_HALF + (jv * 64 + 32 * it) * LDU + t_u]) = vq[it][jv];
    };
    // side slot t (0..31) of a phase: the transform of row i of chunk ch into U half `half`, one LDS instruction or one packed-add pair (or
    // both) per slot: t 0..7 reads of item 0; 8..15 reads of item 1 + w (8..11) and V (12..15) of item 0; 16..19 stores of item 0 + w of
    // item 1; 20..23 V of item 1; 24..27 stores of item 1 (all before the phase's barrier in step 14)
    auto side = [&](int t, int ch, int i, int half) {
        if (t < 8) tr_read(0, t, ch, i);
        else if (t < 16) {
            tr_read(1, t - 8, ch, i);
            if (t < 12) tr_w(0, t - 8, i); else tr_v(0, t - 12);
        } else if (t < 20) { tr_store(0, t - 16, half); tr_w(1, t - 16, i); }
        else if (t < 24) tr_v(1, t - 20);
        else if (t < 28) tr_store(1, t - 24, half);
    };
    // first row (chunk 0, row 0) -> U half 0, not overlapped
#pragma unroll
    for (int t = 0; t < 28; ++t) side(t, 0, 0, 0);
    __syncthreads();

    f32x16 acc[16];
    f32x4 av[4];
    const int a_off = (32 * th + li) * LDU + kh * 4;
    av[0] = *reinterpret_cast<const f32x4*>(&s_u[a_off]);
    av[1] = *reinterpret_cast<const f32x4*>(&s_u[a_off + 8]);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- 8 phases P = chunk * 4 + row: 16 steps s = plane-in-row * 4 + k8-step, 4 MFMAs each; e = 0: weights of step g + 8 (and, in step
    // 14, THE barrier of the phase: the next U half is complete, every wave has issued and -- s_waitcnt -- received its last reads of this
    // one); e = 1: A fragment of step s + 2 (steps 14, 15: of the next phase, behind the barrier); e = 2, 3: side slots 2 s, 2 s + 1
    auto phase = [&](auto P_c) {
        constexpr int P = decltype(P_c)::value;
        constexpr int ch = P >> 2, i = P & 3, half = P & 1;
        constexpr int Pn = P + 1, chn = Pn >> 2, in = Pn & 3, halfn = Pn & 1;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int f = 4 * i + (s >> 2), g = P * 16 + s;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (a chain starts from the constant 0: no accumulator is zeroed, and none is live while conv1_1 uses the matrix core)
                acc[f] = wino_mfma(av[s & 3][e], bw[g & 15][e], (ch == 0 && (s & 3) == 0 && e == 0) ? zero16 : acc[f]);
                if (e == 0) {
                    if (g + 8 < 128 && !(PMX_C1W_ABLATE & 4)) bw[(g + 8) & 15] = wload(g + 8);
                    if (s == 14 && P < 7) __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                } else if (e == 1) {
                    const int sn = s + 2;
                    if (sn < 16) av[sn & 3] = *reinterpret_cast<const f32x4*>(&s_u[half * U_HALF + ((sn >> 2) * 64) * LDU + a_off + (sn & 3) * 8]);
                    else if (P < 7) av[sn & 3] = *reinterpret_cast<const f32x4*>(&s_u[halfn * U_HALF + a_off + (sn & 3) * 8]);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    if (P < 7 && !(PMX_C1W_ABLATE & 2)) side(2 * s + (e - 2), chn, in, halfn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // (explicit instances: left as a loop the compiler does not unroll all eight phases and the accumulator array goes to scratch)
    if (!(PMX_C1W_ABLATE & 8)) {
    phase(std::integral_constant<int, 0>{}); phase(std::integral_constant<int, 1>{}); phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{}); phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
    phase(std::integral_constant<int, 6>{}); phase(std::integral_constant<int, 7>{});
    } else {
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = zero16;
    }
    // (every accumulator stays an AGPR tile until all phases are done: left alone the register allocator starts reading finished tiles out
    //  between the MFMAs of the last chunk -- copies, write-backs and an s_nop 15 per tile in the middle of the matrix pipe's work)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 16; ++f) asm volatile("" : "+a"(acc[f]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- output transform Y = A^T M A per (tile, channel) (two registers at a time through the packed adds), 2 x 2 max-pool = the maximum
    // over the tile's four outputs, bias, ReLU; a lane stores four 16-byte channel runs of its pooled pixel
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(&s_bias[chh * 32 + 8 * q + 4 * kh]);
    float pooled[16];
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        f32x2 t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 m0 = {acc[0 + j][2 * rp], acc[0 + j][2 * rp + 1]}, m1 = {acc[4 + j][2 * rp], acc[4 + j][2 * rp + 1]};
            const f32x2 m2 = {acc[8 + j][2 * rp], acc[8 + j][2 * rp + 1]}, m3 = {acc[12 + j][2 * rp], acc[12 + j][2 * rp + 1]};
            t0[j] = pk_add2(pk_add2(m0, m1), m2);
            t1[j] = pk_sub2(pk_sub2(m1, m2), m3);
        }
        const f32x2 o0 = pk_add2(pk_add2(t0[0], t0[1]), t0[2]), o1 = pk_sub2(pk_sub2(t0[1], t0[2]), t0[3]);
        const f32x2 o2 = pk_add2(pk_add2(t1[0], t1[1]), t1[2]), o3 = pk_sub2(pk_sub2(t1[1], t1[2]), t1[3]);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) pooled[2 * rp + h2] = fmaxf(fmaxf(o0[h2], o1[h2]), fmaxf(o2[h2], o3[h2]));
    }
    const int Hp = H >> 1, Wp = W >> 1, ldc_b = a.ldc * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.g[0].out + (s_pix_out + (size_t)bimg * Hp * Wp) * a.ldc, 0, (unsigned)(Hp * Wp * ldc_b), 0x00020000);
    const int tb = 32 * th + li;
    const int py = (y0 >> 1) + (tb >> 3), px = (x0 >> 1) + (tb & 7);
    const int o = (py < Hp && px < Wp) ? (int)__umul24(__umul24(py, Wp) + px, ldc_b) + (chh * 32 + 4 * kh) * 4 : OUTSIDE;
    const float lo = a.relu ? 0.f : -__builtin_inff();
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(pooled[4 * q + e] + bq[q][e], lo);
        if (PMX_C1W_ABLATE & 16) asm volatile("" :: "v"(v));
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, o + q * 32, 0, 0);
    }
}

// a.g[0] = conv1_2 (in = the 16-channel padded network input, w = its TRANSFORMED weights, pack_wino with cout_pad 64), a.g[1].w = conv1_1's
// weights in lane order (pmx_api.hip::ensure_conv1_pack), a.g[1].bias = its bias; a.g[1].in != null: the uint8 BGR batch (B x H x W x 3) to
// preprocess on the fly instead of reading a.g[0].in, a.kbounds = the bits of the float32 divisor (255 | 256)
int conv1_wino_launch(const ConvArgs& a0, hipStream_t stream)
{
    ConvArgs a = a0;
    PMX_CHECK(a.cout_pad == 64 && a.g[0].cout == 64 && a.nch == 4 && a.lda >= 4 && a.pool && a.ldc % 4 == 0, PMX_ERR_INVALID,
              "conv1 winograd: needs conv1_2 = 64 -> 64 with the pool and a >= 4-channel padded input");
    PMX_CHECK(a.H % 2 == 0 && a.W % 2 == 0, PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK((long long)a.H * a.W * a.lda * 4 < (1ll << 31) && (long long)a.H * a.W * a.ldc < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    a.tiles_x = (a.W + 2 * TT - 1) / (2 * TT);
    a.tiles_y = (a.H + 2 * TT - 1) / (2 * TT);
    a.ksplit = 1;
    if (!a.g[1].in) a.kbounds = 0;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(conv1_wino_kernel), attr_set)) return rc;
    PMX_CHECK(a.nseg == 0 || (a.segs && a.seg_tiles > 0), PMX_ERR_INVALID, "conv1 winograd: segments without a table");
    hipLaunchKernelGGL(conv1_wino_kernel, dim3(a.nseg ? (unsigned)a.seg_tiles : (unsigned)(a.tiles_x * a.tiles_y * a.B)), dim3(256), LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

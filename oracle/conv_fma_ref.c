/* TEST INFRASTRUCTURE (oracle): order-defined fp32 convolution, bit-exact twin of the HIP conv kernels.
 *
 * The reference's convolution (chainer.links.Convolution2D, models/CocoPoseNet.py:26-129, called at :136-260) is an
 * im2col + BLAS contraction whose summation order is undefined, so no bit-level golden exists for it (oracle/network_ref.py
 * is the tolerance-based restatement).  The HIP kernels, however, have a DEFINED order: v_mfma_f32_32x32x2_f32 adds its two
 * k's as a sequential fused-multiply-add chain (tools/mfma_order.hip: identical to a host fmaf chain), and every kernel
 * generation walks K as
 *     for 16-channel chunk: for tap (ky, kx) row-major: for half (channels +0..7, +8..15): for e in 0..3: k = e, then e + 4
 * so this plain-C loop reproduces their outputs bit for bit.  Bias is added after the chain; ReLU / 2x2 max-pool follow.
 * Split-K (small launches, csrc/conv_mfma.hip::conv_pick_ksplit): with `splitk` = S > 1 slice s runs its own chain (from 0)
 * over `slice_chunks[s]` consecutive 16-channel chunks and the slices are added left to right (conv_splitk_reduce_kernel).
 * Compile with -ffp-contract=off (the fmaf calls are explicit); -mfma makes fmaf one instruction, -fopenmp spreads the
 * independent outputs over the host cores. */
#include <math.h>
#include <stddef.h>

/* x: [B][cin][H][W], w: [cout][cin][ks][ks], bias: [cout], y: [B][cout][Ho][Wo] (Ho = H or H/2), zero padding ks/2 */
void conv_fma_ref(const float* x, const float* w, const float* bias, float* y, int B, int cin, int H, int W, int cout, int ks,
                  int relu, int pool, int splitk, const int* slice_chunks)
{
    int start[9];
    if (splitk < 1 || !slice_chunks) splitk = 1;
    start[0] = 0;
    for (int s = 0; s < splitk; ++s) start[s + 1] = splitk == 1 ? (cin + 15) / 16 : start[s] + slice_chunks[s];
    const int pad = ks / 2, nch = (cin + 15) / 16;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < cout; ++n)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    float best = 0.f;
                    const int nwin = pool ? 4 : 1;
                    for (int wi = 0; wi < nwin; ++wi) {
                        const int py = pool ? 2 * oy + (wi >> 1) : oy, px = pool ? 2 * ox + (wi & 1) : ox;
                        float total = 0.f;
                        for (int sl = 0; sl < splitk; ++sl) {
                        float acc = 0.f;
                        for (int c16 = start[sl]; c16 < start[sl + 1]; ++c16)
                            for (int ky = 0; ky < ks; ++ky)
                                for (int kx = 0; kx < ks; ++kx) {
                                    const int iy = py + ky - pad, ix = px + kx - pad;
                                    const int inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
                                    for (int half = 0; half < 2; ++half)
                                        for (int e = 0; e < 4; ++e)
                                            for (int hi = 0; hi < 2; ++hi) {
                                                const int c = c16 * 16 + half * 8 + e + 4 * hi;
                                                if (c >= cin) continue;               /* zero-padded channel: fmaf(0, 0, acc) == acc */
                                                const float xv = inside ? x[(((size_t)b * cin + c) * H + iy) * W + ix] : 0.f;
                                                acc = fmaf(xv, w[(((size_t)n * cin + c) * ks + ky) * ks + kx], acc);
                                            }
                                }
                        total = sl == 0 ? acc : total + acc;
                        }
                        const float acc = total;
                        if (wi == 0 || acc > best) best = acc;
                    }
                    float v = best + bias[n];
                    if (relu) v = v > 0.f ? v : 0.f;
                    y[(((size_t)b * cout + n) * Ho + oy) * Wo + ox] = v;
                }
}

/* Winograd F(2x2, 3x3) twin of csrc/conv_wino.hip::conv_wino_kernel (option "conv_algo"; 3x3 and 7x7 layers, cin % 32 == 0):
 *   U = G g G^T in double, rounded once to fp32 (csrc/pmx_api.hip::pack_wino);
 *   V = B^T d B in fp32, rows first then columns, each entry one add/subtract of two terms;
 *   16 frequency-wise sequential fmaf chains over (32-channel chunk -> sub-kernel -> 8-channel step -> e in 0..3: k = e, then e + 4);
 *   Y = A^T M A in fp32: t0j = (m0j + m1j) + m2j, t1j = (m1j - m2j) - m3j, y_i0 = (t_i0 + t_i1) + t_i2, y_i1 = (t_i1 - t_i2) - t_i3;
 *   ks = 7: sub-kernel (sy, sx) = taps (3 sy .. 3 sy + 2, 3 sx .. 3 sx + 2) on the window shifted by (3 sy, 3 sx), all four summed in the
 *   frequency domain; after the output transform, tap (6, 6) is chained directly onto each y (chunk -> 8-channel step -> k), row 6 is two
 *   1x3 sub-kernels as 1-D F(2,3) along x (weights G g in double -> fp32; per output row four chains over chunk -> sub-kernel -> channel;
 *   y[i][0] += (m0 + m1) + m2, y[i][1] += (m1 - m2) - m3), then column 6 the same along y;
 *   max-pool (3x3 only) = max of the tile's four outputs (before the bias, like the kernel), + bias, ReLU.
 * The window of output tile (ty, tx), sub-kernel (sy, sx), covers input rows 2 ty - pad + 3 sy .. + 3 (columns alike), zeros outside. */
#include <stdlib.h>
/* unit_g_all > 0: the tiles with row-major index >= unit_from are summed unit by unit (the kernel's unit mode: every tile of a single
 * image, unit_from = 0; or the part-filled last block of every (image, slab) in the run geometry, profile label ".../t<g>"); all other
 * tiles are one chain (plain kernel).  run_tx = 0: the index runs over the whole tile grid (ty * TX + tx); run_tx > 0 (run geometry: 23):
 * the map is cut into slabs of run_tx tile columns and the index runs inside a slab (ty * run_tx + tx % run_tx; unit_from =
 * 32 * (TY * run_tx / 32)). */
void conv_wino_ref2(const float* x, const float* w, const float* bias, float* y, int B, int cin, int H, int W, int cout, int ks,
                    int relu, int pool, int unit_g_all, int unit_from, int run_tx)
{
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int pad = ks / 2, nsub = ks == 3 ? 1 : 4, ndir = ks == 3 ? 0 : 13;
    const int TY = (H + 1) / 2, TX = (W + 1) / 2;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    float* U = (float*)malloc((size_t)cout * cin * nsub * 16 * sizeof(float));
    float* V = (float*)malloc((size_t)B * cin * TY * TX * nsub * 16 * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int sub = 0; sub < nsub; ++sub) {
                const float* g = w + ((size_t)n * cin + c) * ks * ks + (3 * (sub >> 1)) * ks + 3 * (sub & 1);
                double gg[4][3];
                for (int i = 0; i < 4; ++i)
                    for (int kx = 0; kx < 3; ++kx)
                        gg[i][kx] = (Gm[i][0] * (double)g[kx] + Gm[i][1] * (double)g[ks + kx]) + Gm[i][2] * (double)g[2 * ks + kx];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        U[(((size_t)n * cin + c) * nsub + sub) * 16 + 4 * i + j] =
                            (float)((gg[i][0] * Gm[j][0] + gg[i][1] * Gm[j][1]) + gg[i][2] * Gm[j][2]);
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < cin; ++c)
            for (int ty = 0; ty < TY; ++ty)
                for (int tx = 0; tx < TX; ++tx)
                    for (int sub = 0; sub < nsub; ++sub) {
                        float d[4][4], r[4][4];
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) {
                                const int iy = 2 * ty - pad + 3 * (sub >> 1) + i, ix = 2 * tx - pad + 3 * (sub & 1) + j;
                                d[i][j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)b * cin + c) * H + iy) * W + ix] : 0.f;
                            }
                        for (int j = 0; j < 4; ++j) {
                            r[0][j] = d[0][j] - d[2][j];
                            r[1][j] = d[1][j] + d[2][j];
                            r[2][j] = d[2][j] - d[1][j];
                            r[3][j] = d[1][j] - d[3][j];
                        }
                        float* v = V + (((((size_t)b * cin + c) * TY + ty) * TX + tx) * nsub + sub) * 16;
                        for (int i = 0; i < 4; ++i) {
                            v[4 * i + 0] = r[i][0] - r[i][2];
                            v[4 * i + 1] = r[i][1] + r[i][2];
                            v[4 * i + 2] = r[i][2] - r[i][1];
                            v[4 * i + 3] = r[i][1] - r[i][3];
                        }
                    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < cout; ++n)
            for (int ty = 0; ty < TY; ++ty)
                for (int tx = 0; tx < TX; ++tx) {
                    /* unit_g > 0 (unit mode of the kernel, single images): pass 1 in units of unit_g chunks, each with its own chains (from
                     * 0) and its own output transform; row 6, column 6 and tap (6, 6) as units starting from 0; the units are added in
                     * that order */
                    float yv[2][2], ysum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
                    const int tidx = run_tx > 0 ? ty * run_tx + tx % run_tx : ty * TX + tx;
                    const int unit_g = (unit_g_all > 0 && tidx >= unit_from) ? unit_g_all : 0;
                    const int ustep = unit_g > 0 ? 32 * unit_g : ((cin + 31) / 32) * 32;
                    int nunit = 0;
                    for (int cu = 0; cu < cin; cu += ustep, ++nunit) {
                    float m[16];
                    for (int f = 0; f < 16; ++f) m[f] = 0.f;
                    for (int c32 = cu; c32 < cin && c32 < cu + ustep; c32 += 32)
                        for (int sub = 0; sub < nsub; ++sub)
                            for (int c8 = c32; c8 < c32 + 32; c8 += 8)
                                for (int e = 0; e < 4; ++e)
                                    for (int hi = 0; hi < 2; ++hi) {
                                        const int c = c8 + e + 4 * hi;
                                        if (c >= cin) continue;
                                        const float* v = V + (((((size_t)b * cin + c) * TY + ty) * TX + tx) * nsub + sub) * 16;
                                        const float* u = U + (((size_t)n * cin + c) * nsub + sub) * 16;
                                        for (int f = 0; f < 16; ++f) m[f] = fmaf(v[f], u[f], m[f]);
                                    }
                    float t0[4], t1[4];
                    for (int j = 0; j < 4; ++j) {
                        t0[j] = (m[j] + m[4 + j]) + m[8 + j];
                        t1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
                    }
                    yv[0][0] = (t0[0] + t0[1]) + t0[2]; yv[0][1] = (t0[1] - t0[2]) - t0[3];
                    yv[1][0] = (t1[0] + t1[1]) + t1[2]; yv[1][1] = (t1[1] - t1[2]) - t1[3];
                    if (unit_g > 0)
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) ysum[i][j] = nunit == 0 ? yv[i][j] : ysum[i][j] + yv[i][j];
                    }
                    if (unit_g > 0 && ndir) yv[0][0] = yv[0][1] = yv[1][0] = yv[1][1] = 0.f;        /* pass 2a starts its own unit */
                    if (unit_g > 0 && !ndir) { yv[0][0] = ysum[0][0]; yv[0][1] = ysum[0][1]; yv[1][0] = ysum[1][0]; yv[1][1] = ysum[1][1]; }
                    if (ndir) {
                        /* pass 2a: tap (6, 6) chained directly onto y; row 6 as two 1x3 sub-kernels, 1-D F(2,3) along x: per output row
                         * i four frequency chains hm[i][f] over (chunk -> sub-kernel -> channel); then y[i][.] += A^T hm[i] */
                        float hm[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, vm[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define XIN(iy_, ix_, c_) (((iy_) >= 0 && (iy_) < H && (ix_) >= 0 && (ix_) < W) ? x[(((size_t)b * cin + (c_)) * H + (iy_)) * W + (ix_)] : 0.f)
#define CH_LOOP(c32_) for (int c8 = (c32_); c8 < (c32_) + 32; c8 += 8) for (int e = 0; e < 4; ++e) for (int hi = 0; hi < 2; ++hi)
                        float yd[2][2] = {{0.f, 0.f}, {0.f, 0.f}};     /* unit mode: tap (6, 6) is a unit of its own, added last */
                        for (int c32 = 0; c32 < cin; c32 += 32) {
                            CH_LOOP(c32) {
                                const int c = c8 + e + 4 * hi;
                                if (c >= cin) continue;
                                const float wv = w[(((size_t)n * cin + c) * ks + 6) * ks + 6];
                                for (int i = 0; i < 2; ++i)
                                    for (int j = 0; j < 2; ++j) {
                                        const float xv = XIN(2 * ty + i + 6 - pad, 2 * tx + j + 6 - pad, c);
                                        if (unit_g > 0) yd[i][j] = fmaf(xv, wv, yd[i][j]);
                                        else yv[i][j] = fmaf(xv, wv, yv[i][j]);
                                    }
                            }
                            for (int sub = 0; sub < 2; ++sub)
                                CH_LOOP(c32) {
                                    const int c = c8 + e + 4 * hi;
                                    if (c >= cin) continue;
                                    const float* g = w + (((size_t)n * cin + c) * ks + 6) * ks + 3 * sub;
                                    float u[4];
                                    for (int f = 0; f < 4; ++f) u[f] = (float)((Gm[f][0] * (double)g[0] + Gm[f][1] * (double)g[1]) + Gm[f][2] * (double)g[2]);
                                    for (int i = 0; i < 2; ++i) {
                                        float d[4];
                                        for (int k = 0; k < 4; ++k) d[k] = XIN(2 * ty + i + 6 - pad, 2 * tx + 3 * sub + k - pad, c);
                                        const float v4[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
                                        for (int f = 0; f < 4; ++f) hm[i][f] = fmaf(v4[f], u[f], hm[i][f]);
                                    }
                                }
                        }
                        for (int i = 0; i < 2; ++i) {
                            yv[i][0] = yv[i][0] + ((hm[i][0] + hm[i][1]) + hm[i][2]);
                            yv[i][1] = yv[i][1] + ((hm[i][1] - hm[i][2]) - hm[i][3]);
                        }
                        if (unit_g > 0)         /* unit "pass 2a" is complete; pass 2b starts from 0 */
                            for (int i = 0; i < 2; ++i)
                                for (int j = 0; j < 2; ++j) { ysum[i][j] = ysum[i][j] + yv[i][j]; yv[i][j] = 0.f; }
                        /* pass 2b: column 6 as two 3x1 sub-kernels, 1-D F(2,3) along y: per output column j four chains vm[j][f] */
                        for (int c32 = 0; c32 < cin; c32 += 32)
                            for (int sub = 0; sub < 2; ++sub)
                                CH_LOOP(c32) {
                                    const int c = c8 + e + 4 * hi;
                                    if (c >= cin) continue;
                                    const float* g = w + (((size_t)n * cin + c) * ks + 3 * sub) * ks + 6;
                                    float u[4];
                                    for (int f = 0; f < 4; ++f) u[f] = (float)((Gm[f][0] * (double)g[0] + Gm[f][1] * (double)g[ks]) + Gm[f][2] * (double)g[2 * ks]);
                                    for (int j = 0; j < 2; ++j) {
                                        float d[4];
                                        for (int k = 0; k < 4; ++k) d[k] = XIN(2 * ty + 3 * sub + k - pad, 2 * tx + j + 6 - pad, c);
                                        const float v4[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
                                        for (int f = 0; f < 4; ++f) vm[j][f] = fmaf(v4[f], u[f], vm[j][f]);
                                    }
                                }
                        for (int j = 0; j < 2; ++j) {
                            yv[0][j] = yv[0][j] + ((vm[j][0] + vm[j][1]) + vm[j][2]);
                            yv[1][j] = yv[1][j] + ((vm[j][1] - vm[j][2]) - vm[j][3]);
                        }
                        if (unit_g > 0)
                            for (int i = 0; i < 2; ++i)
                                for (int j = 0; j < 2; ++j) yv[i][j] = (ysum[i][j] + yv[i][j]) + yd[i][j];
#undef XIN
#undef CH_LOOP
                    }
                    if (pool) {
                        if (ty < Ho && tx < Wo) {
                            float best = yv[0][0];
                            if (yv[0][1] > best) best = yv[0][1];
                            if (yv[1][0] > best) best = yv[1][0];
                            if (yv[1][1] > best) best = yv[1][1];
                            float o = best + bias[n];
                            if (relu) o = o > 0.f ? o : 0.f;
                            y[(((size_t)b * cout + n) * Ho + ty) * Wo + tx] = o;
                        }
                    } else {
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) {
                                const int oy = 2 * ty + i, ox = 2 * tx + j;
                                if (oy >= H || ox >= W) continue;
                                float o = yv[i][j] + bias[n];
                                if (relu) o = o > 0.f ? o : 0.f;
                                y[(((size_t)b * cout + n) * H + oy) * W + ox] = o;
                            }
                    }
                }
    free(U); free(V);
}

void conv_wino_ref(const float* x, const float* w, const float* bias, float* y, int B, int cin, int H, int W, int cout, int ks,
                   int relu, int pool, int unit_g)
{
    conv_wino_ref2(x, w, bias, y, B, cin, H, W, cout, ks, relu, pool, unit_g, 0, 0);
}

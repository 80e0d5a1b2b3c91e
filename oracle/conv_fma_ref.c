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

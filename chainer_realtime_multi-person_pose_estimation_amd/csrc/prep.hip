// prep.hip -- input pre-processing and layout conversion kernels (HBM-bound, coalesced float4 stores).
//
//   prep_u8   : PoseDetector.preprocess (pose_detector.py:426-431) fused with the NHWC-16 packing the first
//               convolution reads: uint8 HWC BGR -> float32 (x / 255 - 0.5), channels 3..15 = 0.
//               The two float32 operations are the reference's (`x_data /= 255; x_data -= 0.5`), applied to
//               each of the 256 possible byte values -> exact by construction (IEEE divide, then subtract).
//   prep_f32  : the inner seam `model(x)` (pose_detector.py:499): float32 NCHW (B,3,H,W) -> NHWC-16.
//   nchw<->nhwc: test / accessor helpers (pmx_set_maps, pmx_get_maps, pmx_conv2d).
#include "pmx_common.h"

__global__ __launch_bounds__(256) void prep_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                      long long npix, float divisor)
{
    // one thread per (pixel, float4 slot of the 16 output channels)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix * 4) return;
    const long long p = i >> 2;
    const int slot = (int)(i & 3);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot == 0) {
        const uint8_t* s = src + p * 3;
        // pose: /255 (pose_detector.py:428); face / hand: /256 (face_detector.py:32, hand_detector.py:36)
        v.x = (float)s[0] / divisor - 0.5f;   // fp-contract is off for this file: divide, then subtract
        v.y = (float)s[1] / divisor - 0.5f;
        v.z = (float)s[2] / divisor - 0.5f;
    }
    reinterpret_cast<float4*>(dst)[i] = v;
}

__global__ __launch_bounds__(256) void prep_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       int B, long long hw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npix = (long long)B * hw;
    if (i >= npix * 4) return;
    const long long p = i >> 2;
    const int slot = (int)(i & 3);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot == 0) {
        const long long b = p / hw, r = p - b * hw;
        const float* s = src + b * 3 * hw + r;
        v.x = s[0];
        v.y = s[hw];
        v.z = s[2 * hw];
    }
    reinterpret_cast<float4*>(dst)[i] = v;
}

// dst[(b*H*W + p) * ldc + coff + c] = src[(b*C + c) * H*W + p]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int B, int C, long long hw, int ldc, int coff)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * C * hw;
    if (i >= total) return;
    // thread order follows the destination (c fastest) so stores coalesce
    const int c = (int)(i % C);
    const long long bp = i / C;
    const long long b = bp / hw, p = bp - b * hw;
    dst[bp * ldc + coff + c] = src[(b * C + c) * hw + p];
}

// dst[(b*C + c) * H*W + p] = src[(b*H*W + p) * lda + coff + c]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int B, int C, long long hw, int lda, int coff)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * C * hw;
    if (i >= total) return;
    const long long p = i % hw;
    const long long bc = i / hw;
    const long long b = bc / C;
    const int c = (int)(bc - b * C);
    dst[i] = src[(b * hw + p) * lda + coff + c];
}

// cv2.resize(img, (dw, dh)) for uint8 HWC images, INTER_LINEAR (reference pose_detector.py:493) -- OpenCV's fixed-point
// algorithm: horizontal pass S = s[sx]*a0 + s[sx1]*a1 (11-bit coefficients, int32), vertical pass
// (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  One thread per output pixel (3 channels), HBM-bound
// byte work; the per-axis tables (source indices, coefficients) are built on the host exactly as the restatement in
// oracle/resize_ref.py does (float32 coordinate math, round-half-even).
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                               const int* __restrict__ xtab, const int* __restrict__ ytab,
                                                               int B, int sh, int sw, int dh, int dw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npix = (long long)B * dh * dw;
    if (i >= npix) return;
    const int x = (int)(i % dw);
    const long long t = i / dw;
    const int y = (int)(t % dh);
    const long long b = t / dh;
    const int sx0 = xtab[x], sx1 = xtab[dw + x], a0 = xtab[2 * dw + x], a1 = xtab[3 * dw + x];
    const int sy0 = ytab[y], sy1 = ytab[dh + y], b0 = ytab[2 * dh + y], b1 = ytab[3 * dh + y];
    const uint8_t* r0 = src + ((b * sh + sy0) * sw) * 3;
    const uint8_t* r1 = src + ((b * sh + sy1) * sw) * 3;
    uint8_t* o = dst + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = (int)r0[sx0 * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
        const int S1 = (int)r1[sx0 * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
        int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o[c] = (uint8_t)v;
    }
}

// ---- cv2.resize(..., interpolation=cv2.INTER_CUBIC) restated (detect_precise, reference pose_detector.py:443,461-467) ----
// Tables per axis (host, pmx_precise.hip::make_cubic_table): 4 clamped source indices and 4 float32 coefficients (OpenCV's
// bicubic, A = -0.75) per destination coordinate, laid out [k][dst].  Arithmetic order = the NumPy restatement
// (pose_detector.py::resize_cubic_*, oracle/precise_ref.py): horizontal 4-tap sums for the 4 source rows, then the
// vertical 4-tap sum, float32 products added left to right (this file is compiled with -ffp-contract=off).

// float32, PLANAR destination, a whole batch per launch: one thread per destination element with x the FASTEST thread index, so a wave's
// stores (and, in the accumulating form, its read-modify-writes) are one contiguous 256-byte run of the destination row and its gathers land
// in a handful of neighbouring cache lines of the source rows (round 4 ran one thread per element with the CHANNEL fastest and accumulated
// with a stride of dh * dw floats between neighbouring threads: 11 % of detect_precise's device time, 16.6 ms for one launch of the
// batch-of-8 form).  Source element (b, c, y, x) = src[b * sb + c * sc + y * sy + x * sx] (the network's NHWC cat buffer for the x8
// up-sampling, the planar intermediate for the resize to the original size); destination dst[((b * C + c) * dh + y) * dw + x],
// ACC: += (the per-scale sums of :463,467).  blockIdx = (x / 256, y, b * C + c); the row's y taps are uniform (scalar loads).
template <int ACC>
__global__ __launch_bounds__(256) void resize_cubic_f32_planar_kernel(const float* __restrict__ src, long long sb, long long sc, long long sy, long long sx,
                                                                      int C, float* __restrict__ dst, int dh, int dw,
                                                                      const int* __restrict__ xi, const float* __restrict__ xc,
                                                                      const int* __restrict__ yi, const float* __restrict__ yc)
{
    const int x = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int y = (int)blockIdx.y, bc = (int)blockIdx.z;
    if (x >= dw) return;
    const int b = bc / C, c = bc - b * C;
    const float* s = src + (long long)b * sb + (long long)c * sc;
    const long long x0 = (long long)xi[x] * sx, x1 = (long long)xi[dw + x] * sx, x2 = (long long)xi[2 * dw + x] * sx, x3 = (long long)xi[3 * dw + x] * sx;
    const float c0 = xc[x], c1 = xc[dw + x], c2 = xc[2 * dw + x], c3 = xc[3 * dw + x];
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* r = s + (long long)yi[k * dh + y] * sy;
        float a = r[x0] * c0;
        a = a + r[x1] * c1;
        a = a + r[x2] * c2;
        a = a + r[x3] * c3;
        rows[k] = a;
    }
    float v = rows[0] * yc[y];
    v = v + rows[1] * yc[dh + y];
    v = v + rows[2] * yc[2 * dh + y];
    v = v + rows[3] * yc[3 * dh + y];
    float* d = dst + ((long long)bc * dh + y) * dw + x;
    if (ACC) *d = *d + v;
    else *d = v;
}

// The same resize, separable through LDS: a block = 256 consecutive x of RB consecutive destination rows of one plane.  The horizontal
// 4-tap sum depends only on (source row, x), so a thread computes it ONCE per source row the block touches (rows yi[0][y0] .. yi[3][y0 + RB - 1]:
// RB / scale + 3 of them) into its own LDS column, then every destination row takes its four sums from there: 4 / RB ... 28 / RB gathers per
// output instead of 16 (x8 up-sampling: 1.5; the largest down-scale of detect_precise: ~7).  Same float32 operations in the same order per
// output -> bit-identical to the one-thread-per-element form above, which stays as the fallback for blocks that need more than NR source rows.
template <int ACC, int RB, int NR>
__global__ __launch_bounds__(256) void resize_cubic_f32_rows_kernel(const float* __restrict__ src, long long sb, long long sc, long long sy, long long sx,
                                                                    int C, float* __restrict__ dst, int dh, int dw,
                                                                    const int* __restrict__ xi, const float* __restrict__ xc,
                                                                    const int* __restrict__ yi, const float* __restrict__ yc)
{
    __shared__ float sH[NR * 256];
    const int tid = (int)threadIdx.x, x = (int)blockIdx.x * 256 + tid;
    const int y0 = (int)blockIdx.y * RB, y1 = min(y0 + RB, dh) - 1, bc = (int)blockIdx.z;
    const int b = bc / C, c = bc - b * C;
    const float* s = src + (long long)b * sb + (long long)c * sc;
    const int r_lo = yi[y0], r_hi = yi[3 * dh + y1];          // (tap indices are clamped and non-decreasing in k and y)
    const int nrows = r_hi - r_lo + 1;                        // block-uniform
    const bool live = x < dw;
    const int xq = live ? x : dw - 1;
    const long long x0 = (long long)xi[xq] * sx, x1 = (long long)xi[dw + xq] * sx, x2 = (long long)xi[2 * dw + xq] * sx, x3 = (long long)xi[3 * dw + xq] * sx;
    const float c0 = xc[xq], c1 = xc[dw + xq], c2 = xc[2 * dw + xq], c3 = xc[3 * dw + xq];
    auto hsum = [&](int sr) -> float {
        const float* r = s + (long long)sr * sy;
        float a = r[x0] * c0;
        a = a + r[x1] * c1;
        a = a + r[x2] * c2;
        a = a + r[x3] * c3;
        return a;
    };
    if (nrows <= NR) {
        for (int r = 0; r < nrows; ++r) sH[r * 256 + tid] = hsum(r_lo + r);     // (own column: written and read by this thread only)
    }
    if (!live) return;
    for (int y = y0; y <= y1; ++y) {
        float h0, h1, h2, h3;
        if (nrows <= NR) {
            h0 = sH[(yi[y] - r_lo) * 256 + tid]; h1 = sH[(yi[dh + y] - r_lo) * 256 + tid];
            h2 = sH[(yi[2 * dh + y] - r_lo) * 256 + tid]; h3 = sH[(yi[3 * dh + y] - r_lo) * 256 + tid];
        } else {
            h0 = hsum(yi[y]); h1 = hsum(yi[dh + y]); h2 = hsum(yi[2 * dh + y]); h3 = hsum(yi[3 * dh + y]);
        }
        float v = h0 * yc[y];
        v = v + h1 * yc[dh + y];
        v = v + h2 * yc[2 * dh + y];
        v = v + h3 * yc[3 * dh + y];
        float* d = dst + ((long long)bc * dh + y) * dw + x;
        if (ACC) *d = *d + v;
        else *d = v;
    }
}

// uint8 HWC, 11-bit fixed point: rows = sum(src * ax) (int32), out = (sum(rows * ay) + (1 << 21)) >> 22, saturated.
// dst has row pitch `dpitch` pixels (the resized image is written into the top-left corner of the padded image, :445)
// blockIdx.y = image of the batch (source / destination images `sbytes` / `dbytes` apart)
__global__ __launch_bounds__(256) void resize_cubic_u8_kernel(const uint8_t* __restrict__ src, int sw, uint8_t* __restrict__ dst,
                                                              int dh, int dw, int dpitch, const int* __restrict__ xi,
                                                              const int* __restrict__ xa, const int* __restrict__ yi, const int* __restrict__ ya,
                                                              long long sbytes, long long dbytes)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)dh * dw) return;
    src += (long long)blockIdx.y * sbytes; dst += (long long)blockIdx.y * dbytes;
    const int x = (int)(i % dw), y = (int)(i / dw);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        long long acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint8_t* r = src + ((long long)yi[k * dh + y] * sw) * 3 + c;
            const long long row = (long long)r[xi[x] * 3] * xa[x] + (long long)r[xi[dw + x] * 3] * xa[dw + x] +
                                  (long long)r[xi[2 * dw + x] * 3] * xa[2 * dw + x] + (long long)r[xi[3 * dw + x] * 3] * xa[3 * dw + x];
            acc += row * ya[k * dh + y];
        }
        long long v = (acc + (1ll << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        dst[((long long)y * dpitch + x) * 3 + c] = (uint8_t)v;
    }
}

// The padding of detect_precise's network inputs (pose_detector.py:445: pad bottom / right to a multiple of 8 with (104, 117, 123)): only
// the strips outside the resized image are written -- the columns [sw, pw) of the rows above sh, the whole rows [sh, ph) -- instead of
// filling the padded image before the resize overwrites most of it (until round 6: 2.2 MB of byte stores at the head of the largest
// scale's chain, 6 % of detect_precise's kernel time in profiles/r06_precise_kernel_stats.csv).  grid (ph, n); one block per row.
__global__ __launch_bounds__(256) void fill_pad_bgr_kernel(uint8_t* __restrict__ dst, int ph, int pw, int sh, int sw, long long img_bytes, int b, int g, int r)
{
    const int row = blockIdx.x;
    uint8_t* const p = dst + (long long)blockIdx.y * img_bytes + (long long)row * pw * 3;
    for (int x = (row < sh ? sw : 0) + threadIdx.x; x < pw; x += 256) {
        p[x * 3] = (uint8_t)b; p[x * 3 + 1] = (uint8_t)g; p[x * 3 + 2] = (uint8_t)r;
    }
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ p, long long n, float divisor)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] / divisor;
}

// out[i] = ((((0 + p0[i + off]) + p1[i + off]) + ...) / divisor: the per-scale parts of detect_precise added left to right from zero
// (`sum = sum + resized`, pose_detector.py:463,467) and averaged (:469-470) in one pass over the data
struct SumPartsArgs { const float* part[8]; int nparts; };
__global__ __launch_bounds__(256) void sum_parts_f32_kernel(float* __restrict__ out, SumPartsArgs a, long long off, long long n, float divisor)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < a.nparts) v = v + a.part[k][off + i];
    out[i] = v / divisor;
}

static inline unsigned nblocks(long long n) { return (unsigned)((n + 255) / 256); }

int launch_prep_u8(const uint8_t* bgr, float* out16, int B, int H, int W, float divisor, hipStream_t s)
{
    const long long npix = (long long)B * H * W;
    hipLaunchKernelGGL(prep_u8_kernel, dim3(nblocks(npix * 4)), dim3(256), 0, s, bgr, out16, npix, divisor);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_prep_f32(const float* x_nchw, float* out16, int B, int H, int W, hipStream_t s)
{
    const long long hw = (long long)H * W;
    hipLaunchKernelGGL(prep_f32_kernel, dim3(nblocks((long long)B * hw * 4)), dim3(256), 0, s, x_nchw, out16, B, hw);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int ldc, int coff, hipStream_t s)
{
    const long long total = (long long)B * C * H * W;
    if (total == 0) return PMX_OK;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks(total)), dim3(256), 0, s, src, dst, B, C, (long long)H * W, ldc, coff);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int lda, int coff, hipStream_t s)
{
    const long long total = (long long)B * C * H * W;
    if (total == 0) return PMX_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks(total)), dim3(256), 0, s, src, dst, B, C, (long long)H * W, lda, coff);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_resize_linear_u8(const uint8_t* src, uint8_t* dst, const int* xtab, const int* ytab, int B, int sh, int sw, int dh, int dw,
                            hipStream_t s)
{
    const long long npix = (long long)B * dh * dw;
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(nblocks(npix)), dim3(256), 0, s, src, dst, xtab, ytab, B, sh, sw, dh, dw);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// 1: the separable form through LDS (default); 0: one thread per element (A/B and tests: the two must agree bit for bit)
static int g_cubic_rows = 1;
void prep_set_cubic_rows(int on) { g_cubic_rows = on; }

int launch_resize_cubic_f32_planar(const float* src, long long sb, long long sc, long long sy, long long sx, int B, int C, float* dst, int dh, int dw,
                                   const int* xi, const float* xc, const int* yi, const float* yc, int accumulate, hipStream_t s)
{
    PMX_CHECK(dh >= 1 && dh <= 65535 && (long long)B * C >= 1 && (long long)B * C <= 65535, PMX_ERR_INVALID,
              "cubic resize: %d rows x %d planes outside the launch grid", dh, B * C);
    if (g_cubic_rows) {
        constexpr int RB = 16, NR = 32;
        const dim3 grid((unsigned)((dw + 255) / 256), (unsigned)((dh + RB - 1) / RB), (unsigned)(B * C));
        if (accumulate) hipLaunchKernelGGL((resize_cubic_f32_rows_kernel<1, RB, NR>), grid, dim3(256), 0, s, src, sb, sc, sy, sx, C, dst, dh, dw, xi, xc, yi, yc);
        else hipLaunchKernelGGL((resize_cubic_f32_rows_kernel<0, RB, NR>), grid, dim3(256), 0, s, src, sb, sc, sy, sx, C, dst, dh, dw, xi, xc, yi, yc);
        PMX_HIP(hipGetLastError());
        return PMX_OK;
    }
    const dim3 grid((unsigned)((dw + 255) / 256), (unsigned)dh, (unsigned)(B * C));
    if (accumulate) hipLaunchKernelGGL(resize_cubic_f32_planar_kernel<1>, grid, dim3(256), 0, s, src, sb, sc, sy, sx, C, dst, dh, dw, xi, xc, yi, yc);
    else hipLaunchKernelGGL(resize_cubic_f32_planar_kernel<0>, grid, dim3(256), 0, s, src, sb, sc, sy, sx, C, dst, dh, dw, xi, xc, yi, yc);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_resize_cubic_u8(const uint8_t* src, int sw, uint8_t* dst, int dh, int dw, int dpitch, const int* xi, const int* xa,
                           const int* yi, const int* ya, int B, long long sbytes, long long dbytes, hipStream_t s)
{
    hipLaunchKernelGGL(resize_cubic_u8_kernel, dim3(nblocks((long long)dh * dw), (unsigned)B), dim3(256), 0, s, src, sw, dst, dh, dw, dpitch, xi, xa, yi, ya,
                       sbytes, dbytes);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_fill_pad_bgr(uint8_t* dst, int n, int ph, int pw, int sh, int sw, int b, int g, int r, hipStream_t s)
{
    if (ph == sh && pw == sw) return PMX_OK;          // nothing to pad
    PMX_CHECK(n >= 1 && n <= 65535 && ph >= sh && pw >= sw && sh >= 0 && sw >= 0, PMX_ERR_INVALID, "pad fill: bad geometry");
    hipLaunchKernelGGL(fill_pad_bgr_kernel, dim3((unsigned)ph, (unsigned)n), dim3(256), 0, s, dst, ph, pw, sh, sw, (long long)ph * pw * 3, b, g, r);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_sum_parts_f32(float* out, const float* const* parts, int nparts, long long off, long long n, float divisor, hipStream_t s)
{
    PMX_CHECK(nparts >= 1 && nparts <= 8, PMX_ERR_INVALID, "sum of parts: %d parts outside 1..8", nparts);
    SumPartsArgs a;
    for (int k = 0; k < 8; ++k) a.part[k] = k < nparts ? parts[k] : nullptr;
    a.nparts = nparts;
    hipLaunchKernelGGL(sum_parts_f32_kernel, dim3(nblocks(n)), dim3(256), 0, s, out, a, off, n, divisor);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int launch_scale_f32(float* p, long long n, float divisor, hipStream_t s)
{
    hipLaunchKernelGGL(scale_f32_kernel, dim3(nblocks(n)), dim3(256), 0, s, p, n, divisor);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

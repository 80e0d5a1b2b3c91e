// postproc.hip -- the reference post-process (pose_detector.py:501-517) as HIP kernels.
//
//   pp_peaks_kernel   F.resize_images (:501-502, corner-aligned bilinear, computed on the fly from the
//                     low-resolution network output) + scipy gaussian_filter(sigma=2.5) semantics (:86:
//                     separable taps, 'reflect' border, float64 accumulate per pass in SciPy's order, float32
//                     store between passes, rows first) + 4-neighbour strict NMS against zero-padded shifts and
//                     the 0.05 threshold (:87-102) on LDS tiles; wave-ballot compaction of the peaks.
//   pp_sort_kernel    restores the reference's row-major peak order per joint type (:104) and assigns the
//                     running global ids (:106-108).
//   pp_limbs_kernel   compute_candidate_connections (:135-159): 16 lanes per candidate pair, 10 of them sample
//                     the PAF line integral (bilinear PAF fetched on the fly), ballot for the n_valid count,
//                     shuffle-gather + NumPy-order float64 sum; then compute_connections' greedy matching
//                     (:172-177) as repeated block-wide arg-max over the accepted candidates (equivalent to
//                     the stable descending sort + scan, ties resolved by the (a, b) loop order of :137-138).
//   pp_group_kernel   grouping_key_points (:183-250) incl. the `[-2:] += score` quirk (:217), the final filter
//                     (:248-249), the rescale to image pixels (:513-514) and subsets_to_pose_array (:252-265);
//                     one wavefront per image, subset table in LDS, lane-parallel subset search via ballot.
//
// Bit-exactness: every float32 / float64 operation below is written in the operation order of the
// reference's NumPy / SciPy code and this file is compiled with -ffp-contract=off (no FMA fusion), so peak
// coordinates, ids, matching and grouping decisions are reproduced exactly for identical network outputs.
#include "pmx_common.h"

#pragma clang fp contract(off)

#define PK_TS 32                                   // NMS output tile (pixels)
#define PK_UW_MAX (PK_TS + 2 + 2 * PMX_GAUSS_MAX_RADIUS)   // 66
#define PK_US (PK_UW_MAX + 1)                      // LDS row stride (floats)

__constant__ int c_limbs[PMX_N_LIMBS][2] = {
    {1, 8}, {8, 9}, {9, 10}, {1, 11}, {11, 12}, {12, 13}, {1, 2}, {2, 3}, {3, 4}, {2, 16},
    {1, 5}, {5, 6}, {6, 7}, {5, 17}, {1, 0}, {0, 14}, {0, 15}, {14, 16}, {15, 17}};

// scipy 'reflect' (d c b a | a b c d), any distance
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// one corner-aligned bilinear sample of channel `ch` of image `b` at integer map coordinates (y, x):
// ((w1*x00 + w2*x01) + w3*x10) + w4*x11 in float32, weights = float32(float64 product) -- see
// oracle/postprocess_ref.py::resize_images_ref for the restated Chainer formula.
__device__ __forceinline__ float bilinear_at(const float* __restrict__ base, long long sy, long long sx,
                                             const PPTables& t, int y, int x)
{
    const int y0 = t.yi0[y], y1 = t.yi1[y], x0 = t.xi0[x], x1 = t.xi1[x];
    const double ylo = t.ylo[y], yhi = t.yhi[y], xlo = t.xlo[x], xhi = t.xhi[x];
    const float w1 = (float)(ylo * xlo);
    const float w2 = (float)(ylo * xhi);
    const float w3 = (float)(yhi * xlo);
    const float w4 = (float)(yhi * xhi);
    const float x00 = base[y0 * sy + x0 * sx];
    const float x01 = base[y0 * sy + x1 * sx];
    const float x10 = base[y1 * sy + x0 * sx];
    const float x11 = base[y1 * sy + x1 * sx];
    float v = w1 * x00;
    v = v + w2 * x01;
    v = v + w3 * x10;
    v = v + w4 * x11;
    return v;
}

// ============================================================================================== peaks
__global__ __launch_bounds__(256) void pp_peaks_kernel(PPMaps maps, PPTables tab, PPBuffers buf, int map_h, int map_w,
                                                       int tiles_x, int keep_smoothed, int n_ch, int do_nms)
{
    __shared__ float sU[PK_UW_MAX * PK_US];          // upsampled (+reflect) tile
    __shared__ float sV[(PK_TS + 2) * PK_US];        // after the vertical (axis 0) pass
    __shared__ float sS[(PK_TS + 2) * (PK_TS + 3)];  // smoothed, 1-pixel NMS halo
    __shared__ double sG[2 * PMX_GAUSS_MAX_RADIUS + 1];

    const int tid = threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * PK_TS, x0 = tx * PK_TS;
    const int R = tab.radius;
    const int UW = PK_TS + 2 + 2 * R;
    const int SW = PK_TS + 3;

    if (tid < 2 * R + 1) sG[tid] = tab.gauss[tid];

    const float* base = maps.heat + (long long)b * maps.sbh + (long long)ch * maps.sc;
    for (int i = tid; i < UW * UW; i += 256) {
        const int ur = i / UW, uc = i - ur * UW;
        const int ry = y0 - 1 - R + ur, rx = x0 - 1 - R + uc;
        float v;
        if (tab.border_zero) {      // reference GPU branch: F.convolution_2d zero padding (:112-113)
            v = (ry >= 0 && ry < map_h && rx >= 0 && rx < map_w) ? bilinear_at(base, maps.sy, maps.sx, tab, ry, rx) : 0.f;
        } else {
            v = bilinear_at(base, maps.sy, maps.sx, tab, reflect_idx(ry, map_h), reflect_idx(rx, map_w));
        }
        sU[ur * PK_US + uc] = v;
    }
    __syncthreads();

    // axis-0 pass (SciPy order: centre tap, then pairs from the outermost inwards), float64 accumulate
    for (int i = tid; i < (PK_TS + 2) * UW; i += 256) {
        const int vr = i / UW, vc = i - vr * UW;
        const float* col = &sU[(vr + R) * PK_US + vc];
        double acc = (double)col[0] * sG[R];
        for (int j = R; j >= 1; --j)
            acc = acc + ((double)col[-j * PK_US] + (double)col[j * PK_US]) * sG[R - j];
        sV[vr * PK_US + vc] = (float)acc;
    }
    __syncthreads();

    // axis-1 pass; positions outside the map are the NMS zero padding (pose_detector.py:87-94)
    for (int i = tid; i < (PK_TS + 2) * (PK_TS + 2); i += 256) {
        const int sr = i / (PK_TS + 2), sc = i - sr * (PK_TS + 2);
        const int y = y0 - 1 + sr, x = x0 - 1 + sc;
        float out = 0.f;
        if (y >= 0 && y < map_h && x >= 0 && x < map_w) {
            const float* row = &sV[sr * PK_US + sc + R];
            double acc = (double)row[0] * sG[R];
            for (int j = R; j >= 1; --j)
                acc = acc + ((double)row[-j] + (double)row[j]) * sG[R - j];
            out = (float)acc;
            if (keep_smoothed && sr >= 1 && sr <= PK_TS && sc >= 1 && sc <= PK_TS)
                buf.smoothed[(((long long)b * n_ch + ch) * map_h + y) * map_w + x] = out;
        }
        sS[sr * SW + sc] = out;
    }
    __syncthreads();

    if (!do_nms) return;      // smoothing-only mode (face / hand key points)
    // NMS + compaction
    const int cap_pk = buf.cap_pk;
    unsigned* keys = buf.pk_raw_key + ((long long)b * PMX_N_JOINTS + ch) * cap_pk;
    float* scores = buf.pk_raw_score + ((long long)b * PMX_N_JOINTS + ch) * cap_pk;
    int* counter = buf.pk_count + b * PMX_N_JOINTS + ch;
    const int lane = tid & 63;
    for (int i = tid; i < PK_TS * PK_TS; i += 256) {
        const int r = i / PK_TS, c = i - r * PK_TS;
        const int y = y0 + r, x = x0 + c;
        bool peak = false;
        float p = 0.f;
        if (y < map_h && x < map_w) {
            p = sS[(r + 1) * SW + (c + 1)];
            const float up = sS[r * SW + (c + 1)], dn = sS[(r + 2) * SW + (c + 1)], lf = sS[(r + 1) * SW + c], rt = sS[(r + 1) * SW + (c + 2)];
            peak = tab.nms_ge ? (p > PMX_HEATMAP_PEAK_THRESH && p >= up && p >= dn && p >= lf && p >= rt)
                              : (p > PMX_HEATMAP_PEAK_THRESH && p > up && p > dn && p > lf && p > rt);
        }
        const unsigned long long m = __ballot(peak);
        if (m) {
            int basei = 0;
            if (lane == 0) basei = atomicAdd(counter, __popcll(m));
            basei = __shfl(basei, 0);
            if (peak) {
                const int slot = basei + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cap_pk) {
                    keys[slot] = (unsigned)(y * map_w + x);
                    scores[slot] = p;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ peaks (fast path)
// Same arithmetic as pp_peaks_kernel, specialised for a compile-time Gaussian radius R (10 for sigma 2.5):
//   * the tile's row / column resize tables (corner indices, float64 weights, reflect applied) are staged in LDS once,
//     so a bilinear sample costs 4 map loads instead of 4 + 12 table loads from global memory;
//   * both 1-D passes use a sliding window: a thread produces SEG consecutive outputs of one column (row) from
//     SEG + 2R LDS reads held in registers instead of 2R + 1 reads per output.
// The float32 / float64 operation order per output is unchanged (SciPy's: centre tap, then pairs from the outermost
// inwards), so the results are bit-identical to the generic kernel (tests compare both against the oracle).
template <int R>
__global__ __launch_bounds__(256) void pp_peaks_fast_kernel(PPMaps maps, PPTables tab, PPBuffers buf, int map_h, int map_w,
                                                            int tiles_x, int keep_smoothed, int n_ch, int do_nms)
{
    constexpr int UW = PK_TS + 2 + 2 * R;        // 54
    constexpr int US = UW + 1;
    constexpr int VR = PK_TS + 2;                // 34 rows/cols that feed the NMS
    constexpr int SEG = 9;                       // outputs per thread and pass
    constexpr int NSEG = (VR + SEG - 1) / SEG;   // 4
    constexpr int SW = PK_TS + 3;
    __shared__ float sU[UW * US];
    __shared__ float sV[VR * US];
    __shared__ float sS[VR * SW];
    __shared__ double sG[2 * R + 1];
    __shared__ double sYlo[UW], sYhi[UW], sXlo[UW], sXhi[UW];
    __shared__ int sY0[UW], sY1[UW], sX0[UW], sX1[UW];
    constexpr int PW = 56;                       // low-resolution patch staged in LDS (rows x cols), covers in == out
    __shared__ float sP[PW * PW];
    __shared__ int sBox[4];                      // patch origin (row, col) and extent
    __shared__ float sPmax[4];                   // per-wave maximum of the patch (threshold pruning)

    const int tid = threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * PK_TS, x0 = tx * PK_TS;

    if (tid < 2 * R + 1) sG[tid] = tab.gauss[tid];
    if (tid < UW) {
        const int gy = reflect_idx(y0 - 1 - R + tid, map_h);
        sY0[tid] = tab.yi0[gy]; sY1[tid] = tab.yi1[gy]; sYlo[tid] = tab.ylo[gy]; sYhi[tid] = tab.yhi[gy];
    } else if (tid >= 64 && tid < 64 + UW) {
        const int k = tid - 64;
        const int gx = reflect_idx(x0 - 1 - R + k, map_w);
        sX0[k] = tab.xi0[gx]; sX1[k] = tab.xi1[gx]; sXlo[k] = tab.xlo[gx]; sXhi[k] = tab.xhi[gx];
    }
    __syncthreads();

    const float* base = maps.heat + (long long)b * maps.sbh + (long long)ch * maps.sc;
    // bounding box of the low-resolution pixels this tile samples (a 7x upsampled tile touches ~10 x 10 of them):
    // stage it in LDS once instead of 4 scattered global loads per full-resolution sample
    if (tid < 64) {
        int lo_y = 1 << 30, hi_y = -1, lo_x = 1 << 30, hi_x = -1;
        if (tid < UW) { lo_y = sY0[tid]; hi_y = sY1[tid]; lo_x = sX0[tid]; hi_x = sX1[tid]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo_y = min(lo_y, __shfl_xor(lo_y, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); hi_x = max(hi_x, __shfl_xor(hi_x, off));
        }
        if (tid == 0) { sBox[0] = lo_y; sBox[1] = lo_x; sBox[2] = hi_y - lo_y + 1; sBox[3] = hi_x - lo_x + 1; }
    }
    __syncthreads();
    const int py0 = sBox[0], px0 = sBox[1], ph = sBox[2], pw = sBox[3];
    const bool patched = ph <= PW && pw <= PW;       // block-uniform
    if (patched) {
        float pmax = -3.0e38f;
        for (int i = tid; i < ph * pw; i += 256) {
            const int r = i / pw, c = i - r * pw;
            const float v = base[(long long)(py0 + r) * maps.sy + (long long)(px0 + c) * maps.sx];
            sP[r * PW + c] = v;
            pmax = fmaxf(pmax, v);
        }
        // Threshold pruning.  Every smoothed value of this tile is a convex combination (bilinear weights, then Gaussian taps: all
        // non-negative, each set summing to 1 up to rounding) of the low-resolution pixels of the patch, so it cannot exceed their maximum
        // by more than rounding noise; a peak needs smoothed > 0.05 (pose_detector.py:97).  If the patch maximum is below the threshold
        // with a 1e-5 relative safety margin (the rounded weight sums exceed 1 by < 1e-6), no pixel of the tile can be a peak: skip the
        // two float64 passes and the NMS.  Results are unchanged by construction (real heat maps are ~0 away from the joints: most
        // tiles take this exit).  Not taken when the smoothed map itself is wanted (keep_smoothed, key-point nets).
        if (do_nms && !keep_smoothed) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, off));
            if ((tid & 63) == 0) sPmax[tid >> 6] = pmax;
        }
        __syncthreads();
        if (do_nms && !keep_smoothed) {
            const float m = fmaxf(fmaxf(sPmax[0], sPmax[1]), fmaxf(sPmax[2], sPmax[3]));
            if (m * 1.00001f < PMX_HEATMAP_PEAK_THRESH && m < PMX_HEATMAP_PEAK_THRESH) return;      // block-uniform
        }
    }
    for (int i = tid; i < UW * UW; i += 256) {
        const int ur = i / UW, uc = i - ur * UW;
        const double ylo = sYlo[ur], yhi = sYhi[ur], xlo = sXlo[uc], xhi = sXhi[uc];
        const float w1 = (float)(ylo * xlo), w2 = (float)(ylo * xhi), w3 = (float)(yhi * xlo), w4 = (float)(yhi * xhi);
        float x00, x01, x10, x11;
        if (patched) {
            const int r0 = (sY0[ur] - py0) * PW, r1 = (sY1[ur] - py0) * PW, c0 = sX0[uc] - px0, c1 = sX1[uc] - px0;
            x00 = sP[r0 + c0]; x01 = sP[r0 + c1]; x10 = sP[r1 + c0]; x11 = sP[r1 + c1];
        } else {
            const long long r0 = sY0[ur] * maps.sy, r1 = sY1[ur] * maps.sy, c0 = sX0[uc] * maps.sx, c1 = sX1[uc] * maps.sx;
            x00 = base[r0 + c0]; x01 = base[r0 + c1]; x10 = base[r1 + c0]; x11 = base[r1 + c1];
        }
        float v = w1 * x00;
        v = v + w2 * x01;
        v = v + w3 * x10;
        v = v + w4 * x11;
        sU[ur * US + uc] = v;
    }
    __syncthreads();

    // axis-0 pass: thread = (column, segment of SEG rows)
    for (int i = tid; i < UW * NSEG; i += 256) {
        const int vc = i % UW, sg = i / UW;
        const int r0 = sg * SEG;
        float win[SEG + 2 * R];
#pragma unroll
        for (int k = 0; k < SEG + 2 * R; ++k) win[k] = (r0 + k < UW) ? sU[(r0 + k) * US + vc] : 0.f;
#pragma unroll
        for (int o = 0; o < SEG; ++o) {
            if (r0 + o < VR) {
                double acc = (double)win[o + R] * sG[R];
#pragma unroll
                for (int j = R; j >= 1; --j) acc = acc + ((double)win[o + R - j] + (double)win[o + R + j]) * sG[R - j];
                sV[(r0 + o) * US + vc] = (float)acc;
            }
        }
    }
    __syncthreads();

    // axis-1 pass: thread = (row, segment of SEG columns); positions outside the map are the NMS zero padding
    for (int i = tid; i < VR * NSEG; i += 256) {
        const int sr = i % VR, sg = i / VR;
        const int c0 = sg * SEG;
        const int y = y0 - 1 + sr;
        float win[SEG + 2 * R];
#pragma unroll
        for (int k = 0; k < SEG + 2 * R; ++k) win[k] = (c0 + k < UW) ? sV[sr * US + c0 + k] : 0.f;
#pragma unroll
        for (int o = 0; o < SEG; ++o) {
            const int sc = c0 + o;
            if (sc < VR) {
                const int x = x0 - 1 + sc;
                float out = 0.f;
                if (y >= 0 && y < map_h && x >= 0 && x < map_w) {
                    double acc = (double)win[o + R] * sG[R];
#pragma unroll
                    for (int j = R; j >= 1; --j) acc = acc + ((double)win[o + R - j] + (double)win[o + R + j]) * sG[R - j];
                    out = (float)acc;
                    if (keep_smoothed && sr >= 1 && sr <= PK_TS && sc >= 1 && sc <= PK_TS)
                        buf.smoothed[(((long long)b * n_ch + ch) * map_h + y) * map_w + x] = out;
                }
                sS[sr * SW + sc] = out;
            }
        }
    }
    __syncthreads();

    if (!do_nms) return;      // smoothing-only mode (face / hand key points)
    const int cap_pk = buf.cap_pk;
    unsigned* keys = buf.pk_raw_key + ((long long)b * PMX_N_JOINTS + ch) * cap_pk;
    float* scores = buf.pk_raw_score + ((long long)b * PMX_N_JOINTS + ch) * cap_pk;
    int* counter = buf.pk_count + b * PMX_N_JOINTS + ch;
    const int lane = tid & 63;
    for (int i = tid; i < PK_TS * PK_TS; i += 256) {
        const int r = i / PK_TS, c = i - r * PK_TS;
        const int y = y0 + r, x = x0 + c;
        bool peak = false;
        float p = 0.f;
        if (y < map_h && x < map_w) {
            p = sS[(r + 1) * SW + (c + 1)];
            peak = p > PMX_HEATMAP_PEAK_THRESH && p > sS[r * SW + (c + 1)] && p > sS[(r + 2) * SW + (c + 1)] &&
                   p > sS[(r + 1) * SW + c] && p > sS[(r + 1) * SW + (c + 2)];
        }
        const unsigned long long m = __ballot(peak);
        if (m) {
            int basei = 0;
            if (lane == 0) basei = atomicAdd(counter, __popcll(m));
            basei = __shfl(basei, 0);
            if (peak) {
                const int slot = basei + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cap_pk) {
                    keys[slot] = (unsigned)(y * map_w + x);
                    scores[slot] = p;
                }
            }
        }
    }
}

// ============================================================================================== sort
__global__ __launch_bounds__(256) void pp_sort_kernel(PPBuffers buf, int map_w, int keys_in_lds)
{
    extern __shared__ unsigned sKeyDyn[];             // [4 waves][cap_pk] when keys_in_lds (else the keys are read from L2)
    __shared__ int sStart[PMX_N_JOINTS + 1];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cap_pk = buf.cap_pk;
    const int* cnt = buf.pk_count + b * PMX_N_JOINTS;
    if (tid == 0) {
        int s = 0, ovf = 0;
        for (int j = 0; j < PMX_N_JOINTS; ++j) {
            sStart[j] = s;
            int n = cnt[j];
            if (n > cap_pk) { n = cap_pk; ovf = 1; }
            s += n;
        }
        sStart[PMX_N_JOINTS] = s;
        if (ovf) atomicOr(buf.status + b, PMX_IMG_PEAK_OVERFLOW);
    }
    __syncthreads();
    if (tid <= PMX_N_JOINTS) buf.pk_start[b * (PMX_N_JOINTS + 1) + tid] = sStart[tid];
    const long long pbase = (long long)b * PMX_N_JOINTS * cap_pk;
    for (int j = wave; j < PMX_N_JOINTS; j += 4) {     // trip count is wave-uniform
        const int n = min(cnt[j], cap_pk);
        const unsigned* keys = buf.pk_raw_key + ((long long)b * PMX_N_JOINTS + j) * cap_pk;
        const float* scores = buf.pk_raw_score + ((long long)b * PMX_N_JOINTS + j) * cap_pk;
        const unsigned* kk = keys;
        if (keys_in_lds) {
            unsigned* mine = sKeyDyn + wave * cap_pk;
            for (int e = lane; e < n; e += 64) mine[e] = keys[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            kk = mine;
        }
        for (int e = lane; e < n; e += 64) {
            const unsigned k = kk[e];
            int rank = 0;
            for (int o = 0; o < n; ++o) rank += (kk[o] < k) ? 1 : 0;   // keys are unique pixels
            const int id = sStart[j] + rank;
            buf.pk_x[pbase + id] = (int)(k % (unsigned)map_w);
            buf.pk_y[pbase + id] = (int)(k / (unsigned)map_w);
            buf.pk_score[pbase + id] = scores[e];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ============================================================================================= limbs
__device__ __forceinline__ bool cand_better(double s1, unsigned i1, double s2, unsigned i2)
{
    return s1 > s2 || (s1 == s2 && i1 < i2);
}

// MODE 0: one block per (limb, image) scans all nA x nB candidate pairs and matches (the batch path: a handful of peaks per joint type).
// MODE 1 / 2 (crowds at full resolution -- detect_precise: ~45 peaks per type, 2000 pairs per limb, 19 blocks for 256 CUs): MODE 1 =
// blockIdx.z slices the pair range, accepted candidates go to a per-(image, limb) list in device memory (slots by atomicAdd: the order
// of the list is arbitrary, the matching below is a total order on (score, pair index) and does not depend on it); MODE 2 = the matching
// of MODE 0 on that list.  Same arithmetic per pair, same result.
template <int MODE>
__global__ __launch_bounds__(256) void pp_limbs_kernel(PPMaps maps, PPTables tab, PPBuffers buf, double img_len)
{
    // fast path: accepted candidates and the endpoint "used" flags of one limb live in LDS; large mode (buf.cap_cand > 0,
    // entered by the host after an overflow): both live in device memory, same algorithm
    __shared__ double sScore[PMX_LDS_CANDIDATES];
    __shared__ unsigned sIdx[PMX_LDS_CANDIDATES];
    __shared__ unsigned char sUsed[2][PMX_LDS_USED];
    __shared__ double sRedS[4];
    __shared__ unsigned sRedI[4];
    __shared__ int sRedE[4];
    __shared__ int sNC, sBestE;

    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ja = c_limbs[l][0], jb = c_limbs[l][1];
    const int cap_pk = buf.cap_pk;
    const int* start = buf.pk_start + b * (PMX_N_JOINTS + 1);
    const int sA = start[ja], nA = start[ja + 1] - sA;
    const int sB = start[jb], nB = start[jb + 1] - sB;
    int* out_cnt = buf.cn_count + b * PMX_N_LIMBS + l;
    if (nA == 0 || nB == 0) {        // pose_detector.py:170,179-180
        if (tid == 0) { *out_cnt = 0; buf.cn_need[b * PMX_N_LIMBS + l] = 0; }
        return;
    }
    const bool large = buf.cap_cand > 0;
    const long long lb = (long long)b * PMX_N_LIMBS + l;
    double* const scanScore = buf.scan_score + lb * buf.scan_cap;
    unsigned* const scanIdx = buf.scan_idx + lb * buf.scan_cap;
    // (MODE 1 writes the device list; MODE 2 in the large mode matches straight on it, else on a copy in LDS)
    double* const cScore = MODE == 1 ? scanScore : large ? (MODE == 2 ? scanScore : buf.cand_score + lb * buf.cap_cand) : sScore;
    unsigned* const cIdx = MODE == 1 ? scanIdx : large ? (MODE == 2 ? scanIdx : buf.cand_idx + lb * buf.cap_cand) : sIdx;
    unsigned char* const usedA = large ? buf.cand_used + lb * 2 * cap_pk : sUsed[0];
    unsigned char* const usedB = large ? usedA + cap_pk : sUsed[1];
    const int ccap = MODE == 1 ? buf.scan_cap : large ? buf.cap_cand : PMX_LDS_CANDIDATES;
    if (tid == 0) sNC = 0;
    if (MODE != 1) {
        for (int i = tid; i < nA; i += 256) usedA[i] = 0;
        for (int i = tid; i < nB; i += 256) usedB[i] = 0;
    }
    __syncthreads();

    const long long pbase = (long long)b * PMX_N_JOINTS * cap_pk;
    const int* px = buf.pk_x + pbase;
    const int* py = buf.pk_y + pbase;
    const float* pafx = maps.paf + (long long)b * maps.sbp + (long long)(2 * l) * maps.sc;
    const float* pafy = pafx + maps.sc;
    const long long P = (long long)nA * nB;
    const int g = tid >> 4, k = tid & 15;        // 16 pair slots per block iteration, 16 lanes per pair
    const int gl = lane & ~15;                   // first lane of my group inside the wave
    // MODE 1: slice blockIdx.z of gridDim.z takes the pairs [p_lo, p_hi) (multiples of 16 pairs); MODE 2: no scan at all
    long long p_lo = 0, p_hi = P;
    if (MODE == 1) {
        const long long per = ((P + 15) / 16 + gridDim.z - 1) / gridDim.z * 16;
        p_lo = per * blockIdx.z; p_hi = min(P, p_lo + per);
    }
    if (MODE == 2) p_hi = 0;
    for (long long base = p_lo; base < p_hi; base += 16) {   // uniform trip count
        const long long p = base + g;
        const bool pv = p < p_hi;
        const int ia = pv ? (int)(p / nB) : 0, ib = pv ? (int)(p - (long long)ia * nB) : 0;
        const double ax = (double)px[sA + ia], ay = (double)py[sA + ia];
        const double bx = (double)px[sB + ib], by = (double)py[sB + ib];
        const double vx = bx - ax, vy = by - ay;                 // :139
        const double norm = sqrt(vx * vx + vy * vy);             // :140
        const bool live = pv && norm != 0.0;                     // :141-142
        double ip = 0.0;
        if (live && k < PMX_N_INTEG_POINTS) {
            // np.linspace(a, b, 10): step = (b-a)/9; y_k = k*step + a; y_9 = b   (:144-145)
            const double stepx = vx / 9.0, stepy = vy / 9.0;
            const double xs = (k == 9) ? bx : (double)k * stepx + ax;
            const double ys = (k == 9) ? by : (double)k * stepy + ay;
            const int xi = (int)rint(xs), yi = (int)rint(ys);    // .round().astype('i') (:146)
            const float fx = bilinear_at(pafx, maps.sy, maps.sx, tab, yi, xi);   // paf[0][ys, xs] (:147)
            const float fy = bilinear_at(pafy, maps.sy, maps.sx, tab, yi, xi);
            const double ux = vx / norm, uy = vy / norm;         // :148
            ip = (double)fx * ux + (double)fy * uy;              // :149
        }
        const bool valid = live && k < PMX_N_INTEG_POINTS && ip > PMX_INNER_PRODUCT_THRESH;   // :155
        const unsigned long long m = __ballot(valid);
        const int n_valid = __popcll((m >> gl) & 0x3FFull);
        // gather the 10 inner products to every lane of the group; sum in NumPy's pairwise order for n = 10
        double v[PMX_N_INTEG_POINTS];
#pragma unroll
        for (int j = 0; j < PMX_N_INTEG_POINTS; ++j) v[j] = __shfl(ip, gl + j);
        if (live && k == 0) {
            double s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            s = s + v[8];
            s = s + v[9];
            const double integ = s / 10.0;                                        // :151
            double prior = PMX_LIMB_LENGTH_RATIO * img_len / norm - PMX_LENGTH_PENALTY_VALUE;   // :153
            if (!(prior < 0.0)) prior = 0.0;
            const double score = integ + prior;
            if (n_valid > PMX_N_INTEG_POINTS_THRESH && score > 0.0) {              // :156
                const int slot = MODE == 1 ? atomicAdd(buf.scan_cnt + lb, 1) : atomicAdd(&sNC, 1);
                if (slot < ccap) {
                    cScore[slot] = score;
                    cIdx[slot] = (unsigned)p;
                }
            }
        }
    }
    if (MODE == 1) return;                       // (the list is complete at the kernel boundary)
    __syncthreads();
    int nc = MODE == 2 ? buf.scan_cnt[lb] : sNC;
    if (MODE == 2 && !large && nc <= PMX_LDS_CANDIDATES) {      // the list -> LDS (the matching re-reads it once per accepted connection)
        for (int e = tid; e < nc; e += 256) { sScore[e] = scanScore[e]; sIdx[e] = scanIdx[e]; }
        __syncthreads();
    }
    if (tid == 0) buf.cn_need[b * PMX_N_LIMBS + l] = nc;
    if (nc > ccap || (MODE == 2 && nc > buf.scan_cap) || (!large && max(nA, nB) > PMX_LDS_USED)) {
        // does not fit: the host grows the candidate store (large mode) and re-runs the post-process of this batch
        if (tid == 0) { atomicOr(buf.status + b, PMX_IMG_CAND_OVERFLOW); *out_cnt = 0; }
        return;
    }

    // greedy matching (:172-177): repeatedly take the best remaining candidate whose endpoints are both free
    const int K = min(nA, nB);
    int* oa = buf.cn_a + lb * cap_pk;
    int* ob = buf.cn_b + lb * cap_pk;
    double* os = buf.cn_score + lb * cap_pk;
    int count = 0;
    while (count < K) {
        double bs = -1.0;
        unsigned bi = 0xFFFFFFFFu;
        int be = -1;
        for (int e = tid; e < nc; e += 256) {
            const unsigned idx = cIdx[e];
            const int ia = idx / nB, ib = idx - ia * nB;
            if (!usedA[ia] && !usedB[ib]) {
                const double s = cScore[e];
                if (cand_better(s, idx, bs, bi)) { bs = s; bi = idx; be = e; }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double s2 = __shfl_xor(bs, off);
            const unsigned i2 = __shfl_xor(bi, off);
            const int e2 = __shfl_xor(be, off);
            if (cand_better(s2, i2, bs, bi)) { bs = s2; bi = i2; be = e2; }
        }
        if (lane == 0) { sRedS[wave] = bs; sRedI[wave] = bi; sRedE[wave] = be; }
        __syncthreads();
        if (tid == 0) {
            double s = sRedS[0]; unsigned i = sRedI[0]; int e = sRedE[0];
            for (int w = 1; w < 4; ++w)
                if (cand_better(sRedS[w], sRedI[w], s, i)) { s = sRedS[w]; i = sRedI[w]; e = sRedE[w]; }
            sBestE = e;
            if (e >= 0) {
                const int ia = i / nB, ib = i - ia * nB;
                usedA[ia] = 1;
                usedB[ib] = 1;
                oa[count] = sA + ia;        // global peak ids (:157)
                ob[count] = sB + ib;
                os[count] = s;
            }
        }
        __syncthreads();
        if (sBestE < 0) break;
        ++count;
        __syncthreads();
    }
    if (tid == 0) *out_cnt = count;
}

// ============================================================================================= group
// LDS_TABLE: the live subset rows sit in LDS (cap_sub <= PMX_LDS_SUBSETS = 896 rows: also the crowds of detect_precise); otherwise in device memory
// (buf.sub_work) -- same code, entered by the host after a subset overflow.
template <bool LDS_TABLE>
__global__ __launch_bounds__(64) void pp_group_kernel(PPBuffers buf, const double* __restrict__ scale_xy)
{
    extern __shared__ double sS[];        // LDS_TABLE: cap_sub rows of 20 doubles (dynamic: up to PMX_LDS_SUBSETS rows = 140 KB for crowds)
    const int b = blockIdx.x, lane = threadIdx.x;
    const int cap_pk = buf.cap_pk, cap_sub = buf.cap_sub;
    double* const S = LDS_TABLE ? sS : buf.sub_work + (long long)b * cap_sub * 20;      // row r at S + 20 * r
    const int* start = buf.pk_start + b * (PMX_N_JOINTS + 1);
    const int n_peaks = start[PMX_N_JOINTS];
    const long long pbase = (long long)b * PMX_N_JOINTS * cap_pk;
    const float* pscore = buf.pk_score + pbase;
    pmx_image_info* info = reinterpret_cast<pmx_image_info*>(buf.results + (size_t)b * buf.rec_bytes);
    const int cap_ppl = buf.cap_ppl;
    double* res_scores = reinterpret_cast<double*>(info + 1);
    double* res_poses = res_scores + cap_ppl;
    int n = 0;            // number of live subsets (wave-uniform)
    int status = 0;
    bool aborted = false;

    for (int l = 0; l < PMX_N_LIMBS && !aborted; ++l) {
        const int ja = c_limbs[l][0], jb = c_limbs[l][1];
        const int cnt = buf.cn_count[b * PMX_N_LIMBS + l];
        const long long cbase = ((long long)b * PMX_N_LIMBS + l) * cap_pk;
        for (int c = 0; c < cnt; ++c) {
            const int ind_a = buf.cn_a[cbase + c], ind_b = buf.cn_b[cbase + c];
            const double score = buf.cn_score[cbase + c];
            const double da = (double)ind_a, db = (double)ind_b;
            // :194-198 which subsets hold either endpoint (rows in ascending order; the third match aborts like the reference)
            int found = 0, idx1 = -1, idx2 = -1;
            for (int base = 0; base < n && found < 3; base += 64) {
                const int r = base + lane;
                const bool m = r < n && (S[r * 20 + ja] == da || S[r * 20 + jb] == db);
                unsigned long long km = __ballot(m);
                while (km && found < 3) {
                    const int i = base + __ffsll((long long)km) - 1;
                    km &= km - 1;
                    if (found == 0) idx1 = i; else if (found == 1) idx2 = i;
                    ++found;
                }
            }
            if (found >= 3) {                       // reference: IndexError at :197
                status |= PMX_IMG_TRIPLE_MATCH;
                aborted = true;
                break;
            }
            if (found == 1) {                       // :200-206
                if (lane == 0 && S[idx1 * 20 + jb] != db) {
                    S[idx1 * 20 + jb] = db;
                    S[idx1 * 20 + 19] += 1.0;
                    S[idx1 * 20 + 18] += (double)pscore[ind_b] + score;
                }
            } else if (found == 2) {                // :208-235
                const bool both = lane < 18 && S[idx1 * 20 + lane] >= 0.0 && S[idx2 * 20 + lane] >= 0.0;   // :213
                if (__ballot(both) == 0ull) {       // merge (:214-218)
                    __syncthreads();
                    if (lane < 18) {
                        S[idx1 * 20 + lane] += S[idx2 * 20 + lane] + 1.0;
                    } else if (lane < 20) {
                        S[idx1 * 20 + lane] += S[idx2 * 20 + lane];
                        S[idx1 * 20 + lane] += score;     // (sic) score AND count, :217
                    }
                    __syncthreads();
                    if (lane < 20)
                        for (int r = idx2; r < n - 1; ++r) S[r * 20 + lane] = S[(r + 1) * 20 + lane];   // np.delete (:218)
                    n -= 1;
                } else if (lane == 0) {             // :219-235
                    double* s1 = S + idx1 * 20;
                    double* s2 = S + idx2 * 20;
                    if (s1[ja] == -1.0) {
                        s1[ja] = da; s1[19] += 1.0; s1[18] += (double)pscore[ind_a] + score;
                    } else if (s1[jb] == -1.0) {
                        s1[jb] = db; s1[19] += 1.0; s1[18] += (double)pscore[ind_b] + score;
                    }
                    if (s2[ja] == -1.0) {
                        s2[ja] = da; s2[19] += 1.0; s2[18] += (double)pscore[ind_a] + score;
                    } else if (s2[jb] == -1.0) {
                        s2[jb] = db; s2[19] += 1.0; s2[18] += (double)pscore[ind_b] + score;
                    }
                }
            } else if (found == 0 && l != 9 && l != 13) {   // :237-243
                if (n >= cap_sub) {
                    status |= PMX_IMG_SUBSET_OVERFLOW;      // host doubles the capacity and re-runs
                    aborted = true;
                    break;
                }
                if (lane < 20) {
                    double v = -1.0;
                    if (lane == ja) v = da;
                    if (lane == jb) v = db;
                    if (lane == 19) v = 2.0;
                    if (lane == 18) v = ((0.0 + (double)pscore[ind_a]) + (double)pscore[ind_b]) + score;
                    S[n * 20 + lane] = v;
                }
                n += 1;
            }
            __syncthreads();
        }
    }
    __syncthreads();

    // final filter (:248-249), rescale (:513-514), pose array (:252-265)
    const double sx = scale_xy ? scale_xy[2 * b] : 1.0;
    const double sy = scale_xy ? scale_xy[2 * b + 1] : 1.0;
    const int* pkx = buf.pk_x + pbase;
    const int* pky = buf.pk_y + pbase;
    double* subs_out = buf.subsets + (long long)b * cap_sub * 20;
    int n_keep = 0;
    if (!aborted) {
        for (int base = 0; base < n; base += 64) {
            const int r = base + lane;
            const bool keep = r < n && S[r * 20 + 19] >= PMX_N_SUBSET_LIMBS_THRESH &&
                              S[r * 20 + 18] / S[r * 20 + 19] >= PMX_SUBSET_SCORE_THRESH;
            const unsigned long long km = __ballot(keep);
            if (keep) {
                const int o = n_keep + __popcll(km & ((1ull << lane) - 1ull));     // o <= r < cap_sub
                for (int j = 0; j < 20; ++j) subs_out[o * 20 + j] = S[r * 20 + j];
                if (o < cap_ppl) res_scores[o] = S[r * 20 + 18];
                double* pose = res_poses + (long long)o * PMX_N_JOINTS * 3;
                for (int j = 0; j < PMX_N_JOINTS && o < cap_ppl; ++j) {
                    const int idx = (int)S[r * 20 + j];
                    if (idx >= 0) {
                        pose[j * 3 + 0] = (double)pkx[idx] * sx;
                        pose[j * 3 + 1] = (double)pky[idx] * sy;
                        pose[j * 3 + 2] = 2.0;
                    } else {
                        pose[j * 3 + 0] = 0.0;
                        pose[j * 3 + 1] = 0.0;
                        pose[j * 3 + 2] = 0.0;
                    }
                }
            }
            n_keep += __popcll(km);
        }
    }
    if (lane == 0) {
        if (n_keep > cap_ppl) status |= PMX_IMG_PEOPLE_OVERFLOW;      // host grows the record and re-runs
        const int prev = atomicOr(buf.status + b, status);
        info->n_people = n_keep;
        info->n_peaks = n_peaks;
        info->status = prev | status;
        info->n_subsets_raw = n;
    }
}

// counters / status words / records of the batch <- 0 (the records are a multiple of 8 bytes: pmx_image_info is 16, the rest doubles)
__global__ __launch_bounds__(256) void pp_clear_kernel(PPBuffers buf, int B, int scan)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < buf.rec_bytes * (size_t)B / 8) reinterpret_cast<unsigned long long*>(buf.results)[i] = 0ull;
    if (i < (size_t)B * PMX_N_JOINTS) buf.pk_count[i] = 0;
    if (i < (size_t)B) buf.status[i] = 0;
    if (scan && i < (size_t)B * PMX_N_LIMBS) buf.scan_cnt[i] = 0;       // (the sliced candidate scan of pp_limbs_kernel<1>)
}

// ============================================================================================== host
static int g_pp_generic = 0;      // 1: always use the generic-radius peaks kernel (tests compare both paths)
void pp_set_generic(int on) { g_pp_generic = on; }

int pp_launch(const PPMaps& maps, const PPTables& tab, const PPBuffers& buf, int B, int map_h, int map_w,
              double img_len, const double* d_scale_xy, int keep_smoothed, hipStream_t stream,
              void (*prof)(void*, const char*, int), void* prof_ctx, int limbs_slices)
{
    // peak counters, status words and records start from zero (unused rows of a record stay zero): one launch instead of three memsets
    // (a single image per call is ~100 launches of 2 ms in all: every launch counts)
    {
        const size_t n8 = buf.rec_bytes * (size_t)B / 8;
        hipLaunchKernelGGL(pp_clear_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, buf, B, limbs_slices > 1 ? 1 : 0);
        PMX_HIP(hipGetLastError());
    }
    const int tiles_x = (map_w + PK_TS - 1) / PK_TS, tiles_y = (map_h + PK_TS - 1) / PK_TS;

    if (prof) prof(prof_ctx, "pp_peaks", 1);
    if (tab.radius == 10 && !g_pp_generic && !tab.border_zero && !tab.nms_ge)
        hipLaunchKernelGGL(pp_peaks_fast_kernel<10>, dim3(tiles_x * tiles_y, PMX_N_JOINTS, B), dim3(256), 0, stream, maps, tab, buf,
                           map_h, map_w, tiles_x, keep_smoothed, PMX_N_JOINTS, 1);
    else
        hipLaunchKernelGGL(pp_peaks_kernel, dim3(tiles_x * tiles_y, PMX_N_JOINTS, B), dim3(256), 0, stream, maps, tab, buf,
                           map_h, map_w, tiles_x, keep_smoothed, PMX_N_JOINTS, 1);
    PMX_HIP(hipGetLastError());
    if (prof) prof(prof_ctx, "pp_peaks", 0);

    if (prof) prof(prof_ctx, "pp_sort", 1);
    {
        // rank sort of the keys of one joint type per wave: keys staged in LDS while 4 waves x cap_pk keys fit 32 KB
        const int in_lds = buf.cap_pk <= 2048;
        hipLaunchKernelGGL(pp_sort_kernel, dim3(B), dim3(256), in_lds ? sizeof(unsigned) * 4 * buf.cap_pk : 0, stream, buf, map_w, in_lds);
    }
    PMX_HIP(hipGetLastError());
    if (prof) prof(prof_ctx, "pp_sort", 0);

    if (prof) prof(prof_ctx, "pp_limbs", 1);
    if (limbs_slices > 1) {
        hipLaunchKernelGGL(pp_limbs_kernel<1>, dim3(PMX_N_LIMBS, B, limbs_slices), dim3(256), 0, stream, maps, tab, buf, img_len);
        hipLaunchKernelGGL(pp_limbs_kernel<2>, dim3(PMX_N_LIMBS, B), dim3(256), 0, stream, maps, tab, buf, img_len);
    } else {
        hipLaunchKernelGGL(pp_limbs_kernel<0>, dim3(PMX_N_LIMBS, B), dim3(256), 0, stream, maps, tab, buf, img_len);
    }
    PMX_HIP(hipGetLastError());
    if (prof) prof(prof_ctx, "pp_limbs", 0);

    if (prof) prof(prof_ctx, "pp_group", 1);
    if (buf.cap_sub <= PMX_LDS_SUBSETS) {
        const size_t lds = (size_t)buf.cap_sub * 20 * sizeof(double);
        if (lds > 64 * 1024) {
            static bool attr_set[PMX_MAX_DEVICES] = {};
            if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(pp_group_kernel<true>), attr_set)) return rc;
        }
        hipLaunchKernelGGL(pp_group_kernel<true>, dim3(B), dim3(64), lds, stream, buf, d_scale_xy);
    } else
        hipLaunchKernelGGL(pp_group_kernel<false>, dim3(B), dim3(64), 20 * sizeof(double), stream, buf, d_scale_xy);
    PMX_HIP(hipGetLastError());
    if (prof) prof(prof_ctx, "pp_group", 0);
    return PMX_OK;
}


// ======================================================================================= face / hand key points
// FaceDetector / HandDetector.compute_peaks_from_heatmaps, CPU branch (face_detector.py:58-68, hand_detector.py:68-78):
// per channel (background excluded) gaussian_filter(sigma) of the resized heat map, max_value = heatmap.max(); if it
// exceeds the threshold the key point is `np.array(np.where(heatmap == max_value)).flatten()` -> [coords[1], coords[0]]:
// (x, y) of the maximum when it is unique; with k > 1 equal maxima the flattened array is [y0, y1, .., x0, x1, ..], so the
// reference returns (y1, y0) -- reproduced here.
struct ArgMax { float v; int cnt; int i0; int i1; };   // max value, multiplicity, two smallest row-major indices

__device__ __forceinline__ ArgMax argmax_merge(const ArgMax& a, const ArgMax& b)
{
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    ArgMax r;
    r.v = a.v;
    r.cnt = a.cnt + b.cnt;
    // two smallest of {a.i0, a.i1, b.i0, b.i1} (INT_MAX = empty)
    const int lo = min(a.i0, b.i0), hi = max(a.i0, b.i0);
    r.i0 = lo;
    r.i1 = min(hi, min(a.i1, b.i1));
    return r;
}

__global__ __launch_bounds__(256) void pp_argmax_kernel(const float* __restrict__ smoothed, int n_ch, int map_h, int map_w,
                                                        double thresh, double* __restrict__ out)
{
    __shared__ ArgMax sred[4];
    const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* m = smoothed + ((long long)b * n_ch + ch) * map_h * map_w;
    const int n = map_h * map_w;
    ArgMax a;
    a.v = -INFINITY; a.cnt = 0; a.i0 = 0x7fffffff; a.i1 = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
        const float v = m[i];
        if (v > a.v) { a.v = v; a.cnt = 1; a.i0 = i; a.i1 = 0x7fffffff; }
        else if (v == a.v) { a.cnt += 1; if (a.i1 == 0x7fffffff) a.i1 = i; }     // own indices ascend
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ArgMax o;
        o.v = __shfl_xor(a.v, off); o.cnt = __shfl_xor(a.cnt, off); o.i0 = __shfl_xor(a.i0, off); o.i1 = __shfl_xor(a.i1, off);
        a = argmax_merge(a, o);
    }
    if (lane == 0) sred[wave] = a;
    __syncthreads();
    if (tid == 0) {
        ArgMax r = sred[0];
        for (int w = 1; w < 4; ++w) r = argmax_merge(r, sred[w]);
        double* o = out + ((long long)b * n_ch + ch) * 4;
        const bool valid = (double)r.v > thresh;        // np.float32 scalar > python float: float64 comparison
        int x = r.i0 % map_w, y = r.i0 / map_w;
        if (r.cnt >= 2) { x = r.i1 / map_w; }            // (sic) coords[1] is the SECOND maximum's row when k >= 2
        o[0] = valid ? (double)x : 0.0;
        o[1] = valid ? (double)y : 0.0;
        o[2] = (double)r.v;
        o[3] = valid ? 1.0 : 0.0;
    }
}

int pp_keypoints_launch(const PPMaps& maps, const PPTables& tab, const PPBuffers& buf, int B, int n_ch, int map_h, int map_w,
                        double thresh, double* d_out, hipStream_t stream)
{
    const int tiles_x = (map_w + PK_TS - 1) / PK_TS, tiles_y = (map_h + PK_TS - 1) / PK_TS;
    if (tab.radius == 10 && !g_pp_generic && !tab.border_zero)
        hipLaunchKernelGGL(pp_peaks_fast_kernel<10>, dim3(tiles_x * tiles_y, n_ch, B), dim3(256), 0, stream, maps, tab, buf, map_h, map_w,
                           tiles_x, 1, n_ch, 0);
    else
        hipLaunchKernelGGL(pp_peaks_kernel, dim3(tiles_x * tiles_y, n_ch, B), dim3(256), 0, stream, maps, tab, buf, map_h, map_w,
                           tiles_x, 1, n_ch, 0);
    PMX_HIP(hipGetLastError());
    hipLaunchKernelGGL(pp_argmax_kernel, dim3(n_ch, B), dim3(256), 0, stream, buf.smoothed, n_ch, map_h, map_w, thresh, d_out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

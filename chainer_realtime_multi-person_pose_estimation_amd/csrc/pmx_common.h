// Shared internal declarations of libpose_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/pose_mi355x.h"

// ---- constants of the inference path (reference entity.py:71-105); the Python copy lives in entity.py --
#define PMX_HEATMAP_PEAK_THRESH 0.05f       // entity.py:79 (compared in float32, see pose_detector.py:97)
#define PMX_N_INTEG_POINTS 10               // entity.py:77
#define PMX_N_INTEG_POINTS_THRESH 8         // entity.py:78
#define PMX_INNER_PRODUCT_THRESH 0.05       // entity.py:80 (float64 compare, pose_detector.py:155)
#define PMX_LIMB_LENGTH_RATIO 1.0           // entity.py:81
#define PMX_LENGTH_PENALTY_VALUE 1.0        // entity.py:82
#define PMX_N_SUBSET_LIMBS_THRESH 3.0       // entity.py:83
#define PMX_SUBSET_SCORE_THRESH 0.2         // entity.py:84
#define PMX_GAUSS_SIGMA 2.5                 // entity.py:75
#define PMX_GAUSS_MAX_RADIUS 16

// limb table (entity.py:85-105): joint indices (from, to) of limb i; PAF channels (2i, 2i+1) = (x, y)
static const int PMX_LIMBS[PMX_N_LIMBS][2] = {
    {1, 8}, {8, 9}, {9, 10}, {1, 11}, {11, 12}, {12, 13}, {1, 2}, {2, 3}, {3, 4}, {2, 16},
    {1, 5}, {5, 6}, {6, 7}, {5, 17}, {1, 0}, {0, 14}, {0, 15}, {14, 16}, {15, 17}};

// ---- activation layout -------------------------------------------------------------------------
// All activations are NHWC float32.  The "concat" buffer that feeds stages 2-6 replaces F.concat
// (CocoPoseNet.py:168): [feature 0..127 | PAF 128..165 | pad 166,167 | heat 168..186 | pad 187..191].
#define PMX_CAT_C 192
#define PMX_CAT_FEAT 0
#define PMX_CAT_PAF 128
#define PMX_CAT_HEAT 168
#define PMX_IN_C 16            // network input padded 3 -> 16 channels (zeros)

// ---- error handling ----------------------------------------------------------------------------
void pmx_set_error(const char* fmt, ...);
#define PMX_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            pmx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PMX_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)
#define PMX_CHECK(cond, code, ...)      \
    do {                                \
        if (!(cond)) {                  \
            pmx_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

// ---- convolution kernel interface (conv_mfma.hip) ------------------------------------------------
struct ConvGroupArgs {
    const float* in;    // NHWC, already offset to the group's first input channel
    const float* w;     // packed [tap][chunk][cout_pad][CK]
    const float* bias;  // [cout_pad]
    float* out;         // NHWC, already offset to the group's first output channel
    int cout;           // real output channels (<= cout_pad)
    int pad_;
};
// Heterogeneous launches: a SEGMENT = n images of one map size.  The segments of a launch lie end to end in the activation buffers
// (segment s, image i at pixel pix0 + i * H * W of the input, pixo + i * (output pixels per image) of the output) and own consecutive
// tiles of the launch.  The table lives in device memory (one per resolution level and tile shape of a forward: conv_build_segs).
struct ConvSeg {
    int H, W;            // map size of the segment's images at this layer (before the optional pool)
    int tiles_x;         // tile columns per image
    int tiles_img;       // tiles per image
    int tile0;           // first tile of the segment in the launch
    int n;               // images
    int pix0, pixo;      // first input / output pixel of the segment in the launch's buffers
};
struct ConvArgs {
    ConvGroupArgs g[2];
    int B, H, W;        // conv input == output spatial size (before the optional 2x2 pool)
    int lda, ldc;       // channel strides (floats) of input / output pixels
    int nch;            // cin_pad / CK
    int cout_pad;       // multiple of the kernel's BN
    int tiles_x, tiles_y;
    int relu, pool;
    // split-K (v5 kernels, small launches): blockIdx.z = slice * ngroups + group (slice-major, so the blocks of the first --
    // largest -- slice are dispatched first whatever the group); slice s accumulates the 16-channel chunks
    // [kbounds byte s, kbounds byte s+1) and stores its raw partial sums (the host passes zero bias, relu = pool = 0)
    // to g[].out + s * slab_stride; conv_splitk_reduce then adds the slabs in slice order, the bias, ReLU and the pool
    int ksplit;              // >= 1 (<= 8)
    int ngroups;             // 1 | 2 (split-K launches only)
    long long slab_stride;   // floats between the partial-sum slabs of one group (0 when ksplit == 1)
    unsigned long long kbounds;   // byte s = first chunk of slice s (s = 0 .. ksplit - 1); slice s ends where s + 1 starts / at nch
    // Winograd kernel, run geometry (46-pixel-wide maps): a block owns 32 consecutive Winograd tiles (row-major over the 23 x ceil(H / 2)
    // tile grid of one image); this launch covers the blocks [run_j0, run_j0 + run_nb) of every image
    int run_j0, run_nb;
    int run_nslab;           // slabs of 46 columns per image (W / 46)
    // heterogeneous launch (rectangles of conv_wino_kernel, squares of conv1_wino_kernel; plain mode): nseg segments, seg_tiles tiles in all
    int nseg, seg_tiles;
    const ConvSeg* segs;     // device memory
};
// Transformed Winograd weights (pmx_api.hip::pack_wino -> conv_wino_kernel): [plane][chunk32][cout_pad / 32][k8-step 4][32][8] -- the four
// k8-steps of a wave's 32 channels are 1 KB apart, an immediate offset of the load
// Winograd run geometry: tile columns of a 46-pixel-wide map / tiles per block
#define PMX_WINO_RUN_TX 23
#define PMX_WINO_RUN_TILES 32
// combine of the unit-mode slabs of a block range (the part-filled last block of every image): out = slab_0 + slab_1 + ... (unit order),
// [2x2 max-pool], + bias, ReLU; compact slabs [unit][image][slab][block - run_j0][tile 32][pixel 4][ld_slab]
struct WinoTailReduceArgs {
    const float* slabs[2];
    const float* bias[2];
    float* out[2];           // NHWC, channel stride ldc, already offset to the group's first output channel
    int cout[2];
    long long slab_stride;   // floats between the unit slabs of one group
    int S, B, H, W, ld_slab, ldc, relu, run_j0, run_nb;
    int nslab, pool;         // slabs per image; pool: 2x2 max over the tile's four pixels before the bias
    int merged;              // 1: slabs of the merged-tail launch -- [unit][stream position b * nt + tail tile][pixel 4][ld_slab] (run_nb unused)
};
// Merged tails (conv_wino_kernel<KS, 0, 1, 3>): the part-filled last blocks of all images of a launch as one stream of tiles, 32 per block.
// Possible when the tail lies in one tile row of a single-slab map and is long enough that a block meets at most three images.
// `lda` (input channel stride, floats): the merged launch reads every image through ONE buffer resource, so the batch must stay below 2^31
// bytes -- part of the predicate, so that the selection (conv_select.hip) and the launch (pmx_api.hip) can never disagree; 0 = not checked
inline bool wino_tail_mergeable(int B, int H, int W, int lda = 0)
{
    const int ntiles = PMX_WINO_RUN_TX * ((H + 1) / 2), t0 = ntiles / PMX_WINO_RUN_TILES * PMX_WINO_RUN_TILES, nt = ntiles - t0;
    return B >= 2 && W == 2 * PMX_WINO_RUN_TX && nt >= 16 && t0 % PMX_WINO_RUN_TX + nt <= PMX_WINO_RUN_TX &&
           (long long)B * H * W * lda * 4 < (1ll << 31);
}
inline int wino_tail_merged_blocks(int B, int H)
{
    const int ntiles = PMX_WINO_RUN_TX * ((H + 1) / 2), nt = ntiles % PMX_WINO_RUN_TILES;
    return (B * nt + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES;
}
int conv_wino_tail_reduce(const WinoTailReduceArgs& r, int groups, hipStream_t stream);
// dynamic LDS above 64 KB must be allowed per kernel AND per device (conv_mfma.hip; shared with conv_wino.hip)
constexpr int PMX_MAX_DEVICES = 64;
int conv_allow_big_lds(const void* kern, bool (&done)[PMX_MAX_DEVICES]);
// launch of the run-geometry Winograd kernel (a.W % 46 == 0); a.ksplit > 1: unit mode writing compact slabs (see WinoTailReduceArgs)
int conv_wino_run_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream);
// launch of the merged-tail kernel (unit mode; a.run_j0 = the number of full blocks per image; wino_tail_mergeable(a.B, a.H, a.W))
int conv_wino_merged_tail_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream);
// start / stop events for the next conv_wino_run_launch of this thread (stamped by the dispatch itself: hipExtLaunchKernelGGL)
void conv_set_launch_events(hipEvent_t e0, hipEvent_t e1);
// ---- kernel selection for the 3x3 / 7x7 layers (conv_select.hip) ------------------------------------------------------------
struct WinoSelectOpts {      // the context options the choice depends on (pmx_set_option keys of the same names)
    int conv_algo, precision, forced_variant, ksplit, wino_unit_eff, wino_min_fill, wino_geom, wino_tail, wino_tail_g;
    int wino_tail_merge = 1;     // the tails of all images as one stream of tiles (0: one part-filled block per image)
    int wino_split = 1;          // 1: a batch whose plain launch ends in a part-filled round may be cut in two by images (wino_split_images)
    int wino_unit_g = 0;         // chunks per pass-1 unit of a launch in unit mode: 0 = the plan that finishes first (dispatch simulation), > 0 = forced, -1 = as many units as 8 slabs allow (the rule until round 6)
    int groups = 1;              // branch groups in the launch (`images` counts images x groups)
    int lda = 0;                 // input channel stride (floats) of the launch: bounds the merged-tail form (wino_tail_mergeable)
};
bool wino_eligible(int ks, int cin_pad, int cout_pad);
// returns 0 = direct kernels (+ split-K), 1 = the Winograd kernel (*run = 1: run geometry; *tail_g > 0: chunks per pass-1 unit of its
// part-filled last blocks, which then run in unit mode), 2 = the Winograd kernel in unit mode (*unit_g = chunks per pass-1 unit)
int wino_select(const WinoSelectOpts& o, int ks, int cin_pad, int cout_pad, int cout, int ldc, int images, int H, int W, int pool, int* unit_g,
                int* run, int* tail_g);
// images [0, n0) of a batch of B through the plain kernel, the rest through the selection of their own count; 0 = the whole batch at once
int wino_split_images(const WinoSelectOpts& o, int ks, int cin_pad, int cout_pad, int cout, int ldc, int B, int groups, int H, int W, int pool);
struct SplitPlan { int S; unsigned long long bounds; int sizes[8]; };
struct SplitKReduceArgs {
    const float* slabs[2];   // per group: ksplit slabs of B x H x W x ld_slab floats
    const float* bias[2];
    float* out[2];           // NHWC, channel stride ldc, already offset to the group's first output channel
    int cout[2];
    long long slab_stride;
    int ksplit, B, H, W, ld_slab, ldc, relu, pool;
};
int conv_splitk_reduce(const SplitKReduceArgs& r, int groups, hipStream_t stream);
// two chained 1x1 convolutions (x -> relu(W1 x + b1) -> W2 . + b2 [relu2]) in one launch: the last two layers of every stage
struct PairGroupArgs {
    const float* in;     // NHWC, channel stride lda, already offset to the group's first input channel (128 channels read)
    const float* w1; const float* b1;   // packed [chunk][cmid][16], [cmid]
    const float* w2; const float* b2;   // packed [chunk][cout_pad][16], [cout_pad]
    float* out;          // NHWC, channel stride ldc, already offset
    int cout;            // real output channels of the second layer
    int pad_;
};
struct PairArgs {
    PairGroupArgs g[2];
    long long npix;      // B * H * W
    int lda, ldc;
    int cmid;            // hidden channels (128 | 512)
    int cout_pad;        // padded output channels of the second layer (64 | 128)
    int relu2;
    int pad_;
};
int conv_pair_launch(const PairArgs& a, int groups, hipStream_t stream);
bool conv_pair_supported(int cin, int cmid, int cout_pad);
// conv1_1 (3 -> 64) recomputed on the halo + conv1_2 (64 -> 64 [+ pool]) in one launch: a.g[0] = conv1_2 (in = padded network input),
// a.g[1].w / .bias = conv1_1's packed weights / bias
int conv1_fused_launch(const ConvArgs& a, hipStream_t stream);
// the same pair with conv1_2 in Winograd F(2x2, 3x3) (conv1_wino.hip): a.g[0].w = conv1_2's TRANSFORMED weights (pack_wino, cout_pad 64)
int conv1_wino_launch(const ConvArgs& a, hipStream_t stream);
// Winograd F(2x2, 3x3): a.nch = cin / 32, a.g[].w = transformed weights [plane][chunk32][k8-step][cout_pad][8] (G g G^T, host);
// ks = 7: planes 0..63 = four 3x3 sub-kernels (taps 0..5 x 0..5), 64..71 row 6, 72..79 column 6 (1-D G g), 80 = tap (6, 6)
int conv_wino_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream);
// K slices for a launch of `variant` (S = 1: no split); forced > 0 asks for that many (near-)even slices
SplitPlan conv_pick_ksplit(int variant, int H, int W, int B, int groups, int cout_pad, int nch, int pool, int forced);

struct ConvVariant {
    int ks, th, tw, bn, ck;
    const char* name;
};
// picks a kernel variant for (ksize, cout, B*H*W); returns index into the variant table
int conv_pick_variant(int ks, int cout, int H, int W, int B, int forced, int gen, int pool, int cin, int bf16x3);
const ConvVariant& conv_variant(int idx);
int conv_num_variants();
// launches the variant; groups = 1 or 2 (blockIdx.z)
int conv_launch(int variant, const ConvArgs& a, int groups, hipStream_t stream);
void conv_set_min_lds(int bytes);
void conv_set_v5_lds(int bytes);        // LDS floor of the v5 / v8 kernels (caps the blocks per CU; tuning)
int conv_v5_lds();
// conv_bf16x3.hip (opt-in: built only with PMX_BUILD_BF16X3=1; a weak reference, null in the default library)
int conv_bf16x3_launch(int ks, int mt, int pool, const ConvArgs& a, int groups, hipStream_t stream) __attribute__((weak));
int conv_bf16x3_twin(int variant);      // the bf16x3 kernel with the geometry of a v6 variant, or -1
void conv_set_num_cus(int n);     // compute units of the device the contexts run on (tile / kernel selection heuristics)
int conv_num_cus();
// packed weight geometry helpers
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// other kernels (prep.hip)
int launch_prep_u8(const uint8_t* bgr, float* out16, int B, int H, int W, float divisor, hipStream_t s);
int launch_prep_f32(const float* x_nchw, float* out16, int B, int H, int W, hipStream_t s);
int launch_resize_linear_u8(const uint8_t* src, uint8_t* dst, const int* xtab, const int* ytab, int B, int sh, int sw, int dh, int dw,
                            hipStream_t s);
// cubic resize of B x C planes in one launch: source element (b, c, y, x) = src[b * sb + c * sc + y * sy + x * sx], planar destination
// dst[((b * C + c) * dh + y) * dw + x] (accumulate: +=)
int launch_resize_cubic_f32_planar(const float* src, long long sb, long long sc, long long sy, long long sx, int B, int C, float* dst, int dh, int dw,
                                   const int* xi, const float* xc, const int* yi, const float* yc, int accumulate, hipStream_t s);
void prep_set_cubic_rows(int on);     // 1 (default): separable form through LDS; 0: one thread per element (same bits)
int launch_resize_cubic_u8(const uint8_t* src, int sw, uint8_t* dst, int dh, int dw, int dpitch, const int* xi, const int* xa,
                           const int* yi, const int* ya, int B, long long sbytes, long long dbytes, hipStream_t s);
// n padded images of ph x pw (dense, one after the other): the pixels outside the top-left sh x sw of each <- (b, g, r)
int launch_fill_pad_bgr(uint8_t* dst, int n, int ph, int pw, int sh, int sw, int b, int g, int r, hipStream_t s);
int launch_scale_f32(float* p, long long n, float divisor, hipStream_t s);
// out[i] = (((0 + parts[0][off + i]) + parts[1][off + i]) + ...) / divisor, nparts <= 8
int launch_sum_parts_f32(float* out, const float* const* parts, int nparts, long long off, long long n, float divisor, hipStream_t s);
int launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int ldc, int coff, hipStream_t s);
int launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int lda, int coff, hipStream_t s);

// ---- post-process interface (postproc.hip) -------------------------------------------------------
struct PPTables {            // device pointers, per context, sized for the largest map
    int* xi0; int* xi1; double* xlo; double* xhi;   // per output column
    int* yi0; int* yi1; double* ylo; double* yhi;   // per output row
    double* gauss;                                   // 2r+1 taps
    int radius;
    int border_zero;   // 0: scipy 'reflect' (CPU branch, golden); 1: zero padding (reference GPU branch, :112-113)
    int nms_ge;        // 0: strict '>' against the 4 neighbours (:98-101); 1: '>=' (GPU branch, :123-126)
};
struct PPMaps {              // where the low-resolution network outputs live
    const float* heat; const float* paf;
    long long sbh, sbp;          // batch strides (floats) of the heat / PAF maps
    long long sy, sx, sc;        // row, column, channel strides (floats), common to both maps
    int fh, fw;
};
struct PPBuffers {
    // capacities (runtime, per context; grown and the post-process re-run when an image needs more: the reference has none)
    int cap_pk;              // peaks per joint type
    int cap_sub;             // live subsets during grouping
    int cap_ppl;             // persons per result record (<= cap_sub)
    int cap_cand;            // 0: accepted candidates of a limb live in LDS (PMX_LDS_CANDIDATES); else slots per limb in `cand_*`
    // peaks
    unsigned* pk_raw_key;    // [B][18][cap_pk]   y*W+x, unsorted
    float* pk_raw_score;     // [B][18][cap_pk]
    int* pk_count;           // [B][18] (raw, may exceed cap_pk)
    int* pk_x; int* pk_y; float* pk_score;   // [B][18 * cap_pk] sorted, global ids
    int* pk_start;           // [B][19] first id of each joint type; [18] = total
    // connections
    int* cn_a; int* cn_b; double* cn_score;  // [B][19][cap_pk]
    int* cn_count;           // [B][19]
    int* cn_need;            // [B][19] accepted candidates before greedy matching (may exceed the candidate capacity)
    double* scan_score; unsigned* scan_idx;  // [B][19][scan_cap] candidates of the sliced scan (pp_limbs_kernel<1> -> <2>), arbitrary order
    int* scan_cnt;                           // [B][19] accepted candidates of the sliced scan (may exceed scan_cap)
    int scan_cap;                            // max(PMX_LDS_CANDIDATES, cap_cand)
    double* cand_score; unsigned* cand_idx;  // [B][19][cap_cand] (large mode only)
    unsigned char* cand_used;                // [B][19][2][cap_pk] (large mode only)
    // grouping
    double* sub_work;        // [B][cap_sub][20] live subsets when they do not fit the LDS table (cap_sub > PMX_LDS_SUBSETS)
    double* subsets;         // [B][cap_sub][20] (filtered, for parity accessors)
    int* status;             // [B]
    unsigned char* results;  // [B] records of rec_bytes: pmx_image_info | double scores[cap_ppl] | double poses[cap_ppl][18][3]
    size_t rec_bytes;
    float* smoothed;         // optional [B][18][map_h][map_w]
};
#define PMX_LDS_CANDIDATES 4096     // candidate slots per limb in the LDS fast path
#define PMX_LDS_SUBSETS 896         // subset rows in the LDS fast path (20 doubles each, dynamic LDS: 140 KB at most; the initial capacity is 128)
#define PMX_LDS_USED 4096           // peaks per joint type whose "used" flags fit the LDS fast path
void pp_set_generic(int on);
int pp_keypoints_launch(const PPMaps& maps, const PPTables& tab, const PPBuffers& buf, int B, int n_ch, int map_h, int map_w,
                        double thresh, double* d_out, hipStream_t stream);
int pp_launch(const PPMaps& maps, const PPTables& tab, const PPBuffers& buf, int B, int map_h, int map_w,
              double img_len, const double* d_scale_xy, int keep_smoothed, hipStream_t stream,
              void (*prof)(void*, const char*, int), void* prof_ctx, int limbs_slices = 0);      // limbs_slices > 1: the candidate scan of a limb over that many blocks

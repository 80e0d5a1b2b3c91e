// pmx_ctx.h -- the context behind the opaque `pmx_ctx*` of the C ABI and the few internal entry points the translation units of the
// ABI share (pmx_api.hip: context, weights, forward plan, post-process, results; pmx_precise.hip: detect_precise and the key-point nets).
#pragma once
#include "pmx_common.h"

#include <map>
#include <tuple>
#include <string>
#include <vector>

struct LayerDesc { std::string name; int cin, cout, ks; };

enum NetKind { NET_POSE = 0, NET_FACE = 1, NET_HAND = 2 };     // params['archs'] (entity.py:50-54)

static const int CK = 16;   // channel chunk of every kernel variant

struct PackedLayer {
    bool set = false;
    float* d_w = nullptr;
    float* d_b = nullptr;
    float* d_ww = nullptr;       // Winograd F(2x2,3x3) pack [freq 16][chunk32][cout_pad][32] = G g G^T (3x3 layers; option "conv_algo" = 1)
    void* d_w3 = nullptr;        // bf16x3 pack [tap][chunk][plane hi|mid|lo][cout_pad][16] (3x3 / 7x7 layers; option "precision" = 1)
    int cin = 0, cout = 0, ks = 0, cin_pad = 0, cout_pad = 0, nch = 0;
};

// ------------------------------------------------------------------------------------------ profiler
struct ProfEntry {
    std::string name;
    double total_ms = 0;
    int64_t launches = 0;
    double flops = 0, bytes = 0;   // per launch: algorithmic FLOP of the convolution, compulsory bytes
    double issued = 0;             // per launch: FLOP the kernel issues to the matrix cores for real outputs (Winograd forms: 16/36, 100/196 of
                                   // the algorithmic figure; direct kernels: all of it; tile padding is not counted)
};
struct ProfPending { int entry; hipEvent_t e0, e1; };

// detect_precise runs its inference scales CONCURRENTLY: one image at 0.5x / 1x / 1.5x is 12 ... 108 one-per-CU blocks per layer for 256
// CUs, so the four forward passes go to four streams ("lanes"), each with its own working set; a lane's fields are swapped into the
// context while its scale is enqueued (the forward code keeps using c->stream / c->act0 / ...).  The first scale enqueued in a
// sequence runs on lane 3 (highest stream priority), the others round-robin on the remaining lanes in use (pmx_precise.hip); the scale in
// slot k leaves its maps, resized to the original size, in pr_part[k]; pmx_precise_finish adds the parts IN SLOT ORDER
// (the reference's left-to-right sum, pose_detector.py:463,467) and divides.
constexpr int PMX_PR_LANES = 4;
constexpr int PMX_SK_ZERO_BIAS = 1024;      // floats of the shared zero-bias vector of the split-K / unit-mode launches
struct PrLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float *in16 = nullptr, *act0 = nullptr, *act1 = nullptr, *cat = nullptr, *brA = nullptr, *brB = nullptr, *brT = nullptr;
    uint8_t* u8_tmp = nullptr;
    size_t cap_px = 0;                       // n * padded_h * padded_w the activation buffers hold
    float* pr_tmp = nullptr; size_t pr_tmp_cap = 0;
    float* sk_scratch = nullptr; size_t sk_floats = 0;
};

// Heterogeneous batches (pmx_multi.hip): a SEGMENT = n images of one network-input size.  The segments of a forward lie end to end in
// every activation buffer; per resolution level (0 = input .. 3 = 1/8) and tile shape a device table tells the kernels which tiles
// belong to which segment (pmx_common.h::ConvSeg).  Tables of one forward: conv1's 16 x 16 squares at level 0 (output = level 1), the
// 8 x 16 rectangles of levels 1, 2 (un-pooled / pooled output) and 3.
struct SegDesc { int n, H, W; };
enum { PMX_SEG_CONV1 = 0, PMX_SEG_L1 = 1, PMX_SEG_L1P = 2, PMX_SEG_L2 = 3, PMX_SEG_L2P = 4, PMX_SEG_L3 = 5, PMX_SEG_TABLES = 6 };
#define PMX_SEG_RECT(level, pool) ((level) == 1 ? ((pool) ? PMX_SEG_L1P : PMX_SEG_L1) : (level) == 2 ? ((pool) ? PMX_SEG_L2P : PMX_SEG_L2) : PMX_SEG_L3)
// one post-process launch set of the current results: a uniform batch is one call (base 0), a mixed batch one per segment
struct PPCall { PPMaps maps; PPTables tab; int base, B, map_h, map_w; double img_len; bool has_scale; int limbs_slices; };

// ------------------------------------------------------------------------------------------- context
struct pmx_ctx {
    // heterogeneous forward / post-process state
    std::vector<SegDesc> segs;       // non-empty only WHILE a heterogeneous forward is being enqueued (run_conv / run_pair / run_conv1 look at it)
    std::vector<SegDesc> cur_segs;   // layout of the current network output when it came from a heterogeneous forward (else empty)
    ConvSeg* d_segs = nullptr; size_t d_segs_cap = 0;     // device: PMX_SEG_TABLES tables of segs.size() entries
    int seg_tiles[PMX_SEG_TABLES] = {};                   // tiles per launch, per table
    long long seg_pix[4] = {};                            // pixels of all segments, per level
    std::vector<PPCall> pp_calls;                         // non-empty: the post-process of the current results ran per segment
    std::map<std::tuple<int, int, int, int>, PPTables> tab_cache;   // up-sampling tables per (in_h, in_w, out_h, out_w) of the per-segment calls
    uint8_t* mi_src = nullptr; size_t mi_src_cap = 0;     // pmx_detect_images: original-size images awaiting the device resize
    int* mi_tab = nullptr; size_t mi_tab_cap = 0;         // ... and their resize tables
    int kind = NET_POSE;             // architecture: posenet | facenet | handnet
    int n_heat = PMX_N_HEAT;         // heat-map channels of the last layer (19 | 71 | 22)
    int cat_c = PMX_CAT_C;           // channels of the cat buffer (192 | 208 | 160)
    int cat_heat = PMX_CAT_HEAT;     // first heat-map channel in the cat buffer (168 | 128 | 128)
    // detect_precise accumulation state (pmx_precise_*)
    int pr_h = 0, pr_w = 0, pr_scales = 0, pr_n = 0;      // original size, scales accumulated so far, images of the batch
    unsigned pr_mask = 0;                                 // slots (positions in the reference's scale loop) filled so far
    float* pr_tmp = nullptr; size_t pr_tmp_cap = 0;      // x8 up-sampled maps of one scale, planar [n][38][ph][pw] | [n][19][ph][pw]
    std::map<std::tuple<int, int, int>, int*> pr_tabs;   // cubic tables per axis, keyed (src, dst, fixed point?): built once, kept
    const uint8_t* pr_src = nullptr;                     // host images of the current begin / finish sequence already in u8_src
    PrLane pr_lane[PMX_PR_LANES];                        // lanes 1 .. : own buffers; lane 0 = the context's own stream and buffers
    std::vector<float*> pr_part; size_t pr_part_cap = 0; // per scale: [n][57][orig_h][orig_w] (PAF planes, then heat planes of every image)
    hipEvent_t pr_src_ready = nullptr, pr_fin = nullptr; // originals uploaded / parts consumed by the last finish
    int opt_precise_lanes = PMX_PR_LANES;                // 1: every scale on the context's own stream (A/B, tests)
    int opt_precise_plain = -1;                          // detect_precise's forwards on the plain Winograd kernels: -1 = when all four lanes are in use, 0 never, 1 always
    int opt_precise_lane_priority = 1;                   // lane streams with priorities (largest scale first); read when a lane's stream is created
    int opt_precise_table_cap = 208;                     // cached cubic tables at which the next pmx_precise_begin* starts the cache over (256 - 48)
    int pr_tabs_trims = 0;                               // how often that happened (diagnostics: option query "precise_table_trims")
    double* d_kp = nullptr;          // key-point records of pmx_keypoints
    size_t kp_cap = 0;
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    int max_batch = 0, max_h = 0, max_w = 0;
    std::vector<LayerDesc> table;
    std::map<std::string, int> index;
    std::vector<PackedLayer> layers;
    // buffers
    float *in16 = nullptr, *act0 = nullptr, *act1 = nullptr, *cat = nullptr, *brA = nullptr, *brB = nullptr, *brT = nullptr;
    float* nchw_tmp = nullptr;       // staging for NCHW host <-> NHWC device conversions
    size_t nchw_tmp_bytes = 0;
    uint8_t* u8_tmp = nullptr;
    const uint8_t* in_u8 = nullptr; float in_div = 255.0f;      // set for ONE forward: conv1_wino_kernel preprocesses this uint8 batch itself
    uint8_t* u8_src = nullptr;       // original-size images awaiting the on-device resize
    size_t u8_src_cap = 0;
    int* rs_tab = nullptr;           // resize tables: x (4 * dw ints) then y (4 * dh ints)
    size_t rs_tab_cap = 0;
    // state of the last forward / set_maps
    bool maps_valid = false, maps_external = false;
    int cur_B = 0, cur_fh = 0, cur_fw = 0;
    float *ext_paf = nullptr, *ext_heat = nullptr;   // NCHW copies installed by pmx_set_maps
    size_t ext_cap = 0;
    // post-process
    PPTables tab{};
    int tab_cap = 0;
    int tab_in_h = -1, tab_in_w = -1, tab_out_h = -1, tab_out_w = -1;
    std::vector<double> gauss;
    PPBuffers pp{};
    double* d_scale = nullptr;
    unsigned char* h_results = nullptr;         // pinned staging for pmx_get_results (pageable D2H is slow and jittery)
    size_t h_results_bytes = 0;
    // pmx_results_snapshot / pmx_snapshot_wait: PMX_SNAPSHOT_SLOTS slots (event, pinned status words, layout at snapshot time)
    hipEvent_t snap_ev[PMX_SNAPSHOT_SLOTS] = {};
    int* snap_status[PMX_SNAPSHOT_SLOTS] = {};
    int snap_B[PMX_SNAPSHOT_SLOTS] = {}, snap_cap_ppl[PMX_SNAPSHOT_SLOTS] = {};
    size_t snap_rec[PMX_SNAPSHOT_SLOTS] = {};
    bool pp_valid = false;
    bool pp_final = false;                      // statuses checked: no image of the last post-process overflowed a capacity
    int pp_B = 0, pp_h = 0, pp_w = 0;
    // arguments of the last post-process, kept for the grow-and-re-run
    PPMaps pp_maps{};
    double pp_img_len = 0;
    bool pp_has_scale = false;
    int pp_regrown = 0;                         // number of capacity growths so far (diagnostics)
    size_t smoothed_cap = 0;
    // options
    int opt_force[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // by ksize
    int opt_gpu_branch_peaks = 0;    // reference GPU-branch peak extraction (non-golden variant)
    int opt_limbs_slices = -1;       // blocks per (limb, image) of the candidate scan: -1 = 8 for full-resolution (external) maps, else one; same results
    int pp_limbs_slices = 0;         // ... of the last post-process (kept for the grow-and-re-run)
    int opt_kp_flip_x = 0;           // pmx_keypoints: mirror the resized heat maps left-right before the peaks (hand_detector.py:46-47)
    int tab_flip = 0;
    int opt_keep_smoothed = 0, opt_stop_stage = 6, opt_kernel_gen = 6;
    int opt_conv_algo = 1;           // 1 (default): Winograd F(2x2,3x3) fp32 kernel for the 3x3 / 7x7 layers of launches that fill the chip
                                     // (>= 2 blocks per CU: batches); 0: direct kernels everywhere; 2: Winograd on every eligible layer
                                     // (tests).  Both are fp32 with a defined order and a C twin; they differ by fp32 rounding (~1e-6)
    int opt_wino_unit_eff = 80;      // unit mode: in-round efficiency of the 7x7 unit blocks relative to the plain kernel, percent (cost model;
                                     // measured with tools/wino_batch_sweep.py: 75 - 90 alike, 60 loses batch 4 and 8, 105 loses batch 16+)
    int opt_wino_min_fill = 50;      // conv_algo 1: percent of ceil(blocks / CUs) * CUs block slots a launch must fill to take the Winograd kernel
    int opt_wino_geom = -1;          // Winograd block geometry on 46-pixel-wide maps: -1 / 1 runs of 32 consecutive tiles, 0 the 8 x 16 pixel rectangles
                                     // of every other map size (same bits either way)
    int opt_wino_tail = -1;          // run geometry: the part-filled last block of every image in unit mode (K units + combine) -- -1 by the cost
                                     // model (conv_algo 1), 0 never, 1 wherever a unit plan exists.  Changes the summation of those tiles (C twin: unit_from)
    int opt_wino_tail_merge = 1;     // the tails of all images of a launch as one stream of tiles, 32 per block (0: one part-filled block per image)
    int opt_wino_tail_g = 0;         // tuning: chunks per pass-1 unit of the tail (0 = automatic)
    int opt_wino_split = 1;          // a batch whose plain launch ends in a part-filled round of the CUs is cut in two by images (0: never)
    std::string split_suffix;        // run_conv inside a split: "@<first image>+<count>", appended to the profile labels
    int opt_wino_unit_g = 0;         // tuning: chunks per pass-1 unit of a launch in unit mode (0 = automatic, -1 = as many units as 8 slabs allow)
    int opt_precision = 0;           // 0: fp32 MFMA everywhere (the path whose results are specified); 1: bf16x3 kernels where a
                                     // v6 kernel would run (fp32-grade accuracy at 2.67x the matrix rate, NOT the fp32 FMA chain)
    int opt_conv1_wino = 1;          // fused conv1_1 + conv1_2 with conv1_2 as Winograd F(2x2, 3x3) (conv1_wino_kernel) where conv_algo allows Winograd
    int opt_fuse_conv1 = 1;          // conv1_1 recomputed on conv1_2's halo tile, one launch (conv1_fused_kernel); identical bits
    int opt_fuse_pairs = 1;          // the two 1x1 layers that end every stage run as one launch (conv1x1_pair_kernel)
    int opt_ksplit = 0;              // 0: automatic split-K for small launches; n > 0: force n K slices where split-K applies
    // split-K scratch: partial-sum slabs of the current launch + a zero bias vector for the slice blocks
    float* sk_scratch = nullptr; size_t sk_floats = 0;
    float* sk_zero_bias = nullptr;
    // timing / profiling
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int prof_on = 0;                 // 0 off | 1 every launch | 2 only the 7x7 convolutions (the dominant kernel: fewest events in a timed region)
    std::vector<ProfEntry> prof;
    std::map<std::string, int> prof_index;
    std::vector<ProfPending> pending;
    std::vector<hipEvent_t> ev_pool;   // recycled events (creating two per launch inside the timed region costs ~0.5 %)
    int prof_open = -1;
};

#define PMX_DEV(c) PMX_HIP(hipSetDevice((c)->device))

// pmx_api.hip
// the network on a uint8 BGR batch on the device: preprocess (x / divisor - 0.5) + forward; where conv1 runs as conv1_wino_kernel the
// preprocessing happens inside it (no float copy of the input), else prep_u8 fills c->in16 first
int pmx_forward_from_u8(pmx_ctx* c, const uint8_t* d_u8, int B, int H, int W, float divisor);
int pmx_forward_from_in16(pmx_ctx* c, int B, int H, int W);      // the network on the padded float input already in c->in16
int pmx_ensure_tables(pmx_ctx* c, int in_h, int in_w, int out_h, int out_w, int flip_x = 0);   // up-sampling tables of the post-process
PPBuffers pmx_pp_view(const PPBuffers& p, int base);         // the post-process buffers of the images [base, ...) (every per-image array offset)
void pmx_make_resize_table(int dst, int src, int* tab);      // OpenCV INTER_LINEAR uint8 table of one axis: [idx0 | idx1 | coef0 | coef1] x dst
int pmx_check_weights(pmx_ctx* c);                            // PMX_ERR_WEIGHTS unless every layer has weights

/*
 * pose_mi355x.h -- C ABI of the MI355X-native OpenPose inference path (libpose_mi355x.so).
 *
 * The reference (DeNA/Chainer_Realtime_Multi-Person_Pose_Estimation) is pure Python and has no FFI of
 * its own: the boundary its hot path sits behind is the Python class `PoseDetector`
 * (pose_detector.py:15-517).  This header is the C ABI a binding for that class would use; each entry
 * point names the reference interface it replaces (file:line relative to the reference repo).
 * The Python mirror of `PoseDetector` in this repo binds exactly these symbols through ctypes
 * (see INTEGRATION.md for the stub).
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types in signatures; a HIP stream is
 * passed as void*).  Every function returns PMX_OK (0) or an error code; no exceptions cross the ABI;
 * pmx_last_error() returns a thread-local message for the last failing call.  One context per host
 * thread / stream; calls on a context are stream-ordered and asynchronous unless stated.
 * There is NO CPU fallback: pmx_create fails with PMX_ERR_NO_DEVICE when no gfx950 device is present.
 */
#ifndef POSE_MI355X_H
#define POSE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_ABI_VERSION 2

/* The reference has no limits on peaks, candidate connections, person hypotheses or persons (np.vstack / Python lists,
 * pose_detector.py:104-110,157,243).  The device-side buffers have CAPACITIES instead, owned by the context: they start at
 * the values below and, when an image needs more, are grown and the post-process of that batch is re-run before any result
 * is handed out (pmx_get_results / pmx_results_layout / the parity accessors) -- callers never see a truncated result. */
#define PMX_N_JOINTS 18          /* entity.py:9-45 */
#define PMX_N_LIMBS 19           /* entity.py:85-105 */
#define PMX_N_PAF 38
#define PMX_N_HEAT 19
#define PMX_INIT_PEAKS_PER_JOINT 128   /* initial capacity: peaks of one joint type per image */
#define PMX_INIT_SUBSETS 128           /* initial capacity: live person hypotheses during grouping */
#define PMX_INIT_PEOPLE 64             /* initial capacity: persons per result record */

enum pmx_status {
    PMX_OK = 0,
    PMX_ERR_INVALID = 1,     /* bad argument */
    PMX_ERR_HIP = 2,         /* HIP runtime error (message in pmx_last_error) */
    PMX_ERR_NO_DEVICE = 3,   /* no usable gfx950 GPU */
    PMX_ERR_WEIGHTS = 4,     /* forward called before all 92 layers were set / unknown layer */
    PMX_ERR_CAPACITY = 5,    /* batch / image larger than the context was created for */
    PMX_ERR_STATE = 6        /* call sequence error (e.g. postprocess before forward) */
};

/* per-image status bits (pmx_image_info.status).  The capacity bits are internal: they trigger the grow-and-re-run and are
 * never set in a record that is handed out. */
enum pmx_image_status {
    PMX_IMG_OK = 0,
    PMX_IMG_PEAK_OVERFLOW = 1,      /* more peaks of one joint type than the current capacity */
    PMX_IMG_CAND_OVERFLOW = 2,      /* more accepted candidates for one limb than the current capacity */
    PMX_IMG_SUBSET_OVERFLOW = 4,    /* more live subsets than the current capacity */
    PMX_IMG_TRIPLE_MATCH = 8,       /* third subset matches a connection: the reference raises IndexError
                                       (pose_detector.py:193,197); the binding re-raises it */
    PMX_IMG_PEOPLE_OVERFLOW = 16    /* more persons after the final filter than the record capacity */
};

typedef struct pmx_ctx pmx_ctx;

typedef struct pmx_image_info {
    int32_t n_people;   /* rows of poses/scores (pose_detector.py:515-516) */
    int32_t n_peaks;    /* len(all_peaks) (pose_detector.py:508); 0 => the reference's early return :509-510 */
    int32_t status;     /* pmx_image_status bits */
    int32_t n_subsets_raw; /* subsets alive before the final filter (pose_detector.py:248) */
} pmx_image_info;

/* Result record, one per image, contiguous on the device (what multi-GPU runs gather with RCCL) and in pmx_get_results:
 *     pmx_image_info info;
 *     double scores[people_cap];                  subsets[:, -2]        (pose_detector.py:516)
 *     double poses[people_cap][PMX_N_JOINTS][3];  [x, y, 2] | [0, 0, 0] (pose_detector.py:252-265)
 * people_cap is the context's current person capacity; query it (and the record size) with pmx_results_layout AFTER the
 * post-process and BEFORE sizing the output buffer: it grows when an image needs it. */
#define PMX_RECORD_BYTES(people_cap) (sizeof(pmx_image_info) + (size_t)(people_cap) * (1 + PMX_N_JOINTS * 3) * sizeof(double))

/* ---- library / device ------------------------------------------------------------------------ */
const char* pmx_version(void);
const char* pmx_last_error(void);
int pmx_device_count(int* n);

/* ---- context: PoseDetector.__init__ (pose_detector.py:16-35) ----------------------------------
 * Replaces model construction (`params['archs'][arch]()`, :23), `model.to_gpu()` (:31) and the device
 * selection (:30).  max_h/max_w bound the network input size (multiples of 8), max_batch the batch. */
int pmx_create(pmx_ctx** out, int device, int max_batch, int max_h, int max_w);
/* `params['archs'][arch]()` (entity.py:50-54; pose_detector.py:23, face_detector.py:15, hand_detector.py:15):
 * arch = "posenet" (models/CocoPoseNet.py), "facenet" (models/FaceNet.py, 71 maps) or "handnet" (models/HandNet.py, 22 maps). */
int pmx_create_net(pmx_ctx** out, const char* arch, int device, int max_batch, int max_h, int max_w);
void pmx_destroy(pmx_ctx* ctx);
int pmx_set_stream(pmx_ctx* ctx, void* hip_stream);   /* NULL -> the context's own stream */
int pmx_synchronize(pmx_ctx* ctx);                    /* cuda.get_device_from_id().synchronize(), :506 */
/* Test / measurement knobs (defaults are the product configuration):
 *   "kernel_gen" 1 | 5 | 6     conv kernel generation: 1 = LDS-staged weights (v1) everywhere, 5 = v5 for 3x3 / 7x7,
 *                              6 = default (v6 / conv1_1 kernel where they apply, else v5); all compute identical bits
 *   "force_variant_k1|k3|k7"   force one entry of the conv variant table for that kernel size (-1 = automatic)
 *   "conv_algo" 1 | 0 | 2 | 3  fp32 algorithm of the 3x3 / 7x7 layers.  The Winograd F(2x2, 3x3) kernel needs 16 instead of 36 products per
 *                              2x2 output tile and channel pair for a 3x3 layer; a 7x7 layer = four 3x3 sub-kernels summed in the
 *                              transformed domain + four 1-D F(2,3) sub-kernels for row 6 / column 6 + one direct tap: 100 instead
 *                              of 196.  Its blocks are equal and run one per CU, so a launch costs whole CU rounds.
 *                              1 (default): by launch size (pmx_api.hip::wino_mode) -- the Winograd kernel when the blocks fill at
 *                              least half of their rounds (batches); its UNIT mode (a tile's passes / chunk groups as separate
 *                              blocks writing slabs + the split-K combine kernel) when a cost model says the rounds would stay
 *                              mostly empty (the 46x46 layers of 1 - 4 and 8 images); else the direct kernels with split-K.
 *                              0: direct kernels everywhere.  2: the Winograd kernel on every eligible layer; 3: unit mode wherever
 *                              it applies (tests).  All forms are fp32 fused-multiply-add chains in a defined order with a plain-C
 *                              twin (oracle/conv_fma_ref.c); the Winograd forms differ from the direct chain by fp32 rounding (~1e-6
 *                              of the map scale; they are the closer ones to float64)
 *   "wino_min_fill" percent    tuning (default 50): share of the block slots of its CU rounds a launch must fill to take the Winograd kernel
 *   "wino_unit_eff" percent    tuning (default 80): in-round efficiency assumed for unit-mode blocks in the selection cost model
 *   "wino_geom" -1 | 0 | 1     block geometry of the plain Winograd kernel on 46-pixel-wide maps (the 46 x 46 maps of a 368 x 368 input):
 *                              -1 / 1 (default) = runs of 32 consecutive Winograd tiles (529 tiles = 16.5 blocks per map), 0 = the
 *                              8 x 16 pixel rectangles every other map size uses (18 per map).  Same bits either way
 *   "wino_tail" -1 | 0 | 1     run geometry: the part-filled last block of every image as K units + a combine kernel (the 16 full
 *                              blocks of 32 images x 2 branches are exactly 4 rounds of 256 CUs): -1 (default) by the cost model,
 *                              0 never, 1 wherever a unit plan exists.  The tiles of that block (row-major index >= 32 * full blocks)
 *                              are then summed unit by unit (C twin: `unit_from`); profile label "...r/t<chunks per unit>"
 *   "wino_tail_g" n            tuning: chunks per pass-1 unit of those tails (0 = automatic)
 *   "wino_split" 1 | 0 | pct   a batch whose plain launch of a 3x3 / 7x7 layer would end in a part-filled round of the CUs (equal
 *                              one-per-CU blocks: the round costs as much as a full one) is cut in two by images: the images of the
 *                              whole rounds through the plain kernel, the rest through the launch form of THEIR count (unit mode) --
 *                              1 (default) where the cost model gains >= 3 %, 0 never, 50 .. 100 = that threshold in percent.
 *                              Eight 368 x 496 frames: 5 + 3, 14.4 -> 13.4 ms.  Part of the arithmetic (an image's rounding then depends
 *                              on which side of the cut it lies): profile label "...@<first image>+<count>"
 *   "wino_unit_g" n            chunks per pass-1 unit of a launch that runs in unit mode as a whole (single images, small batches):
 *                              0 (default) = the plan a dispatch simulation over the device's CUs finishes first (one 368 x 368 image:
 *                              conv4_2 as 3 units of 6 / 6 / 4 chunks = 216 blocks in one round), n > 0 = n chunks, -1 = as many
 *                              units as 8 slabs allow (the rule until round 6).  Part of the arithmetic: profile label ".../u<n>"
 *   "wino_tail_merge" 1 | 0    batches of 46-wide maps whose tail lies in one tile row (46 x 46: 17 tiles per image): 1 (default) = the
 *                              tails of all images of the launch as one stream of tiles, 32 per block (every MFMA row a real tile);
 *                              0 = one part-filled block per image.  Same units, same bits; profile label "...r/t<g>m"
 *   "precision" 0 | 1          0 (default): every convolution is the fp32 FMA chain the parity tests specify.  1: the 3x3 / 7x7
 *                              layers that run on the one-block-per-CU kernels use the bf16 matrix cores with every fp32 value
 *                              split into three bf16 terms (six products, fp32 accumulate): fp32-grade accuracy, 2.67x the
 *                              matrix rate, results equal to the fp32 path only to summation-order-sized noise
 *   "fuse_conv1" 1 | 0         conv1_1 recomputed on conv1_2's halo tiles, one launch instead of two (default 1); identical bits
 *   "precise_lanes" 1..4       detect_precise: inference scales in flight at once, each on its own stream and working set (default 4;
 *                              1: one after the other on the context's stream); same bits
 *   "precise_plain" -1 | 0 | 1  detect_precise's forward passes on the plain Winograd kernels ("conv_algo" 2 for their duration): -1 (default) =
 *                              when all four lanes are in use (scales that share the chip should spend as little CU time as possible),
 *                              0 = the default selection, 1 = always.  Changes the fp32 rounding of the maps like any kernel choice
 *   "precise_lane_priority" 1 | 0   detect_precise: the lanes' streams get priorities, the last lane (the reference's largest scale) the highest
 *                              (default 1; takes effect for lanes created afterwards, i.e. set it before the first detect_precise); same bits
 *   "precise_table_cap" n      detect_precise: cached cubic tables at which the next pmx_precise_begin* starts the cache over (default 208)
 *   "cubic_rows" 1 | 0         detect_precise's float32 cubic resizes: separable through LDS (default) | one thread per element; same bits
 *   "conv1_wino" 1 | 0 | 2     that launch with conv1_2 as Winograd F(2x2, 3x3) on 16 x 16 squares (default 1: where "conv_algo" >= 1
 *                              and the launch has a block per CU; 2: whatever the launch size); 0: the direct 8 x 16 tiles everywhere
 *   "fuse_pairs" 1 | 0         the two 1x1 layers that end every stage as one launch (default) or as two; identical bits
 *   "ksplit" 0 | 1 | n         split-K of the 3x3 / 7x7 launches that cannot fill the chip (single images): 0 = automatic,
 *                              1 = never, n = n K slices wherever split-K applies.  The slices are combined in a fixed
 *                              order, so results stay deterministic; they differ in the last bits from the unsplit sum
 *   "ksplit_plan" digits       tuning: explicit slices, decimal digits = 16-channel chunks per slice (3221 = 3 + 2 + 2 + 1);
 *                              applied to launches whose chunk count equals the digit sum
 *   "keep_smoothed"            keep the smoothed heat maps for pmx_get_smoothed
 *   "stop_stage" 1..6          stop the network after this stage (profiling)
 *   "kp_flip_x" 0 | 1          pmx_keypoints: mirror the resized heat maps left-right before the peaks are taken (the reference's
 *                              `cv2.flip(heatmaps, 1)` for left hands, hand_detector.py:46-47)
 *   "peaks_gpu_branch"         the reference's GPU-branch peak extraction (17 x 17 un-normalised kernel, zero pad, >=)
 *   "pp_limbs_slices" -1 | n   blocks per (limb, image) of the candidate-pair scan: -1 (default) = 8 where the maps come at full
 *                              resolution (detect_precise, pmx_set_maps), one block otherwise; 0 / 1 = one block; same results
 *   "conv_min_lds", "conv_v5_lds", "pp_generic"   ablation switches, PROCESS-wide (not per context) */
int pmx_set_option(pmx_ctx* ctx, const char* key, int value);

/* ---- weights: serializers.load_npz (pose_detector.py:26) --------------------------------------
 * One call per Chainer-NPZ entry pair `<layer>/W` (float32 OIHW) + `<layer>/b`; names and shapes are the
 * 92 links of models/CocoPoseNet.py:26-129.  Host pointers; packed and uploaded immediately. */
int pmx_set_layer(pmx_ctx* ctx, const char* name, const float* w_oihw, const float* bias,
                  int cout, int cin, int ksize);
int pmx_weights_missing(pmx_ctx* ctx, int* n_missing);

/* ---- network forward: `self.model(x)` (pose_detector.py:499; models/CocoPoseNet.py:132-262) ----
 * pmx_forward_u8 also fuses `preprocess` (pose_detector.py:426-431): uint8 HWC BGR -> float32, /255 - 0.5.
 * pmx_forward_f32 is the reference's inner seam (x = float32 NCHW as produced by preprocess).
 * `on_device` != 0: the pointer is device memory on the context's device. */
int pmx_forward_u8(pmx_ctx* ctx, const uint8_t* bgr_nhwc, int batch, int h, int w, int on_device);
int pmx_forward_f32(pmx_ctx* ctx, const float* x_nchw, int batch, int h, int w, int on_device);
/* `cv2.resize(orig_img, (w, h))` (pose_detector.py:493; INTER_LINEAR uint8, OpenCV's fixed-point algorithm restated) on
 * the device, then pmx_forward_u8.  All images of the batch share one source size; identity when the sizes agree. */
int pmx_forward_u8_resized(pmx_ctx* ctx, const uint8_t* bgr_nhwc, int batch, int src_h, int src_w, int h, int w, int on_device);
int pmx_get_resized(pmx_ctx* ctx, uint8_t* out_nhwc, int batch, int h, int w);   /* parity accessor */
/* last-stage outputs h1s[-1] (PAF, B x 38 x h/8 x w/8) and h2s[-1] (heat, B x 19 x h/8 x w/8), float32 NCHW,
 * copied to host (synchronises).  Either pointer may be NULL. */
int pmx_get_maps(pmx_ctx* ctx, float* paf_nchw, float* heat_nchw);
/* test seam = the reference's `model=` constructor argument (pose_detector.py:19-20): install network
 * outputs directly (host float32 NCHW, B x 38|19 x fh x fw) instead of running the network. */
int pmx_set_maps(pmx_ctx* ctx, const float* paf_nchw, const float* heat_nchw, int batch, int fh, int fw);

/* ---- post-process (pose_detector.py:501-517) --------------------------------------------------
 * F.resize_images to (map_h, map_w) (:501-502) + compute_peaks_from_heatmaps CPU-branch semantics
 * (:75-110) + compute_connections (:135-181) + grouping_key_points (:183-250) + the rescale to original
 * image pixels (:513-514) + subsets_to_pose_array (:252-265), all on the device.
 * img_len: :511 passes map_w (fast path), :478 orig_img_w (precise path).
 * scale_xy: per image (sx, sy) = (orig_w / map_w, orig_h / map_h) as float64, or NULL for (1, 1).
 * gauss_w: the 2*radius+1 float64 taps scipy's gaussian_filter(sigma=2.5) uses (NULL -> computed in C). */
int pmx_set_gaussian(pmx_ctx* ctx, const double* taps, int radius);
int pmx_postprocess(pmx_ctx* ctx, int batch, int map_h, int map_w, double img_len, const double* scale_xy);

/* ---- detect_precise (pose_detector.py:433-482) accumulated on the device ------------------------------------
 * begin(orig size) -> add_scale(host uint8 orig image, scaled size = ceil(orig * multiplier), :442-443) per inference scale ->
 * finish() (average, :469-470; installs the full-resolution maps as a batch of one) -> pmx_postprocess(ctx, 1, orig_h,
 * orig_w, img_len = orig_w, NULL) (:475-481).  cv2.resize(INTER_CUBIC) is restated (uint8 fixed-point and float32 paths).
 * Every add_scale of one begin / finish sequence resizes the SAME original image(s): the host buffer is uploaded by the first call and
 * a later call that passes the same pointer reuses the device copy, so the pixels must not change between begin and finish, and the
 * buffer must stay allocated until a synchronising call (pmx_get_results, pmx_get_maps, pmx_synchronize) has returned: add_scale only enqueues. */
int pmx_precise_begin(pmx_ctx* ctx, int orig_h, int orig_w);
int pmx_precise_add_scale(pmx_ctx* ctx, const uint8_t* bgr_hwc, int scaled_h, int scaled_w);
int pmx_precise_finish(pmx_ctx* ctx);
/* the same for n images of ONE original size (n <= the context's batch capacity; the reference handles one image per call): every
 * scale runs the n images as one batch through the network; finish() installs a batch of n -> pmx_postprocess(ctx, n, orig_h, orig_w,
 * orig_w, NULL).  bgr_nhwc: n x orig_h x orig_w x 3, contiguous.  Per image the results equal the single-image calls up to the
 * kernel-choice-by-launch-size rounding of the network. */
int pmx_precise_begin_batch(pmx_ctx* ctx, int n_images, int orig_h, int orig_w);      /* (at most 8 scales per sequence) */
int pmx_precise_add_scale_batch(pmx_ctx* ctx, const uint8_t* bgr_nhwc, int scaled_h, int scaled_w);
/* the same with the scale's POSITION in the reference's loop given (slot 0 .. 7, each once per sequence, no gaps at finish): the parts are
 * summed in slot order (:463,467) whatever order the scales are enqueued in, so the caller can enqueue the largest scale -- the longest
 * chain of the sequence -- first.  pmx_precise_add_scale* without a slot take the next free one. */
int pmx_precise_add_scale_at(pmx_ctx* ctx, const uint8_t* bgr_nhwc, int scaled_h, int scaled_w, int slot);
/* the per-axis cubic tables a context caches (~22 per distinct original size) and how often the cache was started over.  That happens
 * only inside pmx_precise_begin*, behind a device synchronisation, once `cached` has reached option "precise_table_cap" (default 208):
 * never while a sequence holds table pointers. */
int pmx_precise_table_stats(pmx_ctx* ctx, int* cached, int* trims);

/* FaceDetector / HandDetector post-process (face_detector.py:37-38,58-68; hand_detector.py:41,68-78) for facenet / handnet
 * contexts: F.resize_images(hs[-1], (out_h, out_w)) + gaussian_filter + per-channel arg-max.  out: batch x (maps - 1) x 4
 * float64 rows (x, y, confidence, valid); valid = 0 where the reference appends None.  Synchronises. */
int pmx_keypoints(pmx_ctx* ctx, int batch, int out_h, int out_w, double thresh, double* out);

/* fused: forward_u8 + postprocess (PoseDetector.__call__, pose_detector.py:484-517, for images already
 * at the network input size; cv2.resize at :493 is the identity for them) */
int pmx_detect_batch(pmx_ctx* ctx, const uint8_t* bgr_nhwc, int batch, int h, int w, int on_device,
                     int map_h, int map_w, double img_len, const double* scale_xy);

/* ---- batches of images of DIFFERENT sizes ------------------------------------------------------------------------
 * The reference takes any image in any call and picks the network size per image (pose_detector.py:490-493, compute_optimal_size
 * :57-73), so a stream of frames has a new size every few images.  pmx_detect_images is `PoseDetector.__call__` (:484-517) for B
 * such images in ONE call: per image the cv2.resize of :493 on the device, the network over all images as one launch per layer
 * (consecutive images of one network size form a segment; the segments lie end to end in the activation buffers and a per-level
 * table tells a block which segment its tile belongs to), the post-process per run of images with equal sizes -- with the image's
 * own map size (:491, :501-502), img_len = map_w (:511) and coordinate rescale orig / map (:513-514).  Records in image order
 * (pmx_get_results as usual).  Capacity: B <= max_batch and the sum of net_h * net_w <= max_batch * max_h * max_w of pmx_create.
 * Per image the maps equal those of a single-image call that runs the plain Winograd kernels (options "conv_algo" 2, "conv1_wino" 2)
 * bit for bit; against the default single-image call (unit-mode / split-K kernels) they differ by fp32 rounding like any two batch
 * sizes do.  Callers sort their images by size to get few segments; any order is valid. */
typedef struct pmx_image {
    const uint8_t* bgr;      /* src_h x src_w x 3 uint8 BGR, host memory; read before the call returns */
    int src_h, src_w;        /* original size */
    int net_h, net_w;        /* network input size: compute_optimal_size(orig, inference_img_size), multiples of 8 */
    int map_h, map_w;        /* size the network maps are up-sampled to: compute_optimal_size(orig, heatmap_size) */
} pmx_image;
int pmx_detect_images(pmx_ctx* ctx, const pmx_image* images, int batch);
/* the two halves apart: the network on uint8 images already at their network sizes, pixels end to end (image i: net_hw[2 i] x
 * net_hw[2 i + 1] x 3; host or device memory), and the post-process of those maps (image i up-sampled to map_hw[2 i] x map_hw[2 i + 1],
 * img_len = that width; scale_xy: batch x 2 doubles or NULL as in pmx_postprocess) */
int pmx_forward_u8_images(pmx_ctx* ctx, const uint8_t* bgr, const int* net_hw, int batch, int on_device);
int pmx_postprocess_images(pmx_ctx* ctx, const int* map_hw, int batch, const double* scale_xy);
/* parity accessor: the network output of ONE image of the current batch (uniform or mixed) as NCHW float32, paf 38 x fh x fw and heat
 * 19 x fh x fw (either may be NULL); fh x fw must be the image's map size (network size / 8).  Synchronises. */
int pmx_get_image_maps(pmx_ctx* ctx, int image, float* paf, float* heat, int fh, int fw);

/* results.  pmx_results_layout synchronises, grows the capacities and re-runs the post-process if an image overflowed them,
 * and returns the layout of the (now final) records; pmx_get_results does the same and copies `batch` records to `out`
 * (out_bytes >= batch * bytes_per_record, else PMX_ERR_CAPACITY). */
int pmx_results_layout(pmx_ctx* ctx, int* people_cap, size_t* bytes_per_record);
int pmx_get_results(pmx_ctx* ctx, int batch, void* out, size_t out_bytes);
/* device pointer of the record array (for RCCL gathers): synchronises and finalises the records like pmx_results_layout (PMX_ERR_STATE
 * before any post-process); valid until the next post-process, capacity change or pmx_destroy */
int pmx_results_device_ptr(pmx_ctx* ctx, void** dev_ptr, size_t* bytes_per_record);
/* pipelined consumers (a multi-GPU gather that runs one step behind the compute): pmx_results_snapshot enqueues, stream-ordered after
 * the last post-process and WITHOUT synchronising, a device-to-device copy of its `batch` records to dst_device (>= batch *
 * bytes_per_record at the context's current layout, else PMX_ERR_CAPACITY) plus a copy of the per-image status words to pinned host
 * memory, and marks the point with the event of `slot` (0 .. PMX_SNAPSHOT_SLOTS - 1: a consumer that runs d steps behind needs d + 1).  The caller
 * may then enqueue the next pmx_detect_batch.
 * pmx_snapshot_wait blocks until slot's copies are done (not until the stream is idle) and returns the layout the snapshot was
 * taken at and the OR of the status words: if it carries a capacity bit (PMX_IMG_*_OVERFLOW) the snapshot is NOT final -- run the
 * step again and fetch it through pmx_results_layout / pmx_get_results, which grow the capacities (docs: INTEGRATION.md). */
#define PMX_SNAPSHOT_SLOTS 4
int pmx_results_snapshot(pmx_ctx* ctx, int slot, void* dst_device, size_t dst_bytes);
int pmx_snapshot_wait(pmx_ctx* ctx, int slot, int* batch, int* people_cap, size_t* bytes_per_record, int* status_or);
/* capacities: pre-size a context for crowds (or shrink them in tests to exercise the growth path); 0 keeps a value.
 * candidates: 0 = accepted candidates of a limb are kept in LDS (4096 slots), > 0 = that many slots in device memory.
 * people <= subsets, every value <= 2^20 (candidates <= 2^24), else PMX_ERR_INVALID; if the device cannot hold the new buffers the
 * call fails with PMX_ERR_HIP and the context keeps its previous buffers and capacities. */
int pmx_set_capacities(pmx_ctx* ctx, int peaks_per_joint, int subsets, int people, int candidates);
int pmx_get_capacities(pmx_ctx* ctx, int* peaks_per_joint, int* subsets, int* people, int* candidates);

/* parity accessors for one image of the last post-process (host copies; synchronise).
 * peaks: rows (type, x, y, score, id) float64 = all_peaks (pose_detector.py:76,110), BEFORE the rescale;
 * conns: rows (limb, id_a, id_b, score) = all_connections (:161-181) flattened in limb order;
 * subsets: rows of 20 float64 after the final filter (:248-249).
 * n_rows is always set; PMX_ERR_CAPACITY if it exceeds cap_rows (call again with a larger buffer). */
int pmx_get_peaks(pmx_ctx* ctx, int image, double* peaks5, int cap_rows, int* n_rows);
int pmx_get_connections(pmx_ctx* ctx, int image, double* conns4, int cap_rows, int* n_rows);
int pmx_get_subsets(pmx_ctx* ctx, int image, double* subsets20, int cap_rows, int* n_rows);
/* smoothed heat map (gaussian_filter output, pose_detector.py:86) of one image/joint: only produced when
 * option "keep_smoothed" is 1; float32 map_h x map_w. */
int pmx_get_smoothed(pmx_ctx* ctx, int image, int joint, float* out, int map_h, int map_w);

/* ---- measurement ------------------------------------------------------------------------------
 * HIP-event timing on the stream the kernels are launched on. */
int pmx_timer_start(pmx_ctx* ctx);
int pmx_timer_stop(pmx_ctx* ctx, double* ms);          /* synchronises */
int pmx_profile_enable(pmx_ctx* ctx, int on);          /* 1: event pairs around every kernel launch; 2: only around the 7x7
                                                          convolutions (the dominant kernel; fewest events inside a timed region) */
int pmx_profile_reset(pmx_ctx* ctx);
int pmx_profile_count(pmx_ctx* ctx, int* n);
int pmx_profile_entry(pmx_ctx* ctx, int i, char* name, int name_cap, double* total_ms,
                      int64_t* launches, double* flop_per_launch, double* bytes_per_launch);
/* FLOP per launch the entry's kernel ISSUES to the matrix cores for real outputs: the algorithmic figure for the direct kernels,
 * 16/36 (3x3) or 100/196 (7x7) of it for the Winograd forms -- the numerator of a roofline fraction that cannot exceed 1 */
int pmx_profile_issued(pmx_ctx* ctx, int i, double* issued_flop_per_launch);

/* ---- kernel unit-test entry (T0): one convolution layer through the product kernels -----------
 * x: float32 NCHW host (B, cin, h, w); w: OIHW; y: NCHW host (B, cout, h', w') with h' = h/2 if pool.
 * Semantics = L.Convolution2D(ksize, stride 1, pad ksize/2) [+ F.relu] [+ F.max_pooling_2d(2, 2)]. */
int pmx_conv2d(pmx_ctx* ctx, const float* x_nchw, const float* w_oihw, const float* bias,
               int batch, int cin, int h, int w, int cout, int ksize, int relu, int pool,
               float* y_nchw, int iters, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* POSE_MI355X_H */

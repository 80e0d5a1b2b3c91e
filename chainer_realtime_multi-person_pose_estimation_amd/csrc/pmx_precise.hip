// pmx_precise.hip -- the C-ABI entry points of the multi-scale path and of the key-point networks (include/pose_mi355x.h):
//   pmx_precise_begin / _add_scale / _finish   detect_precise (reference pose_detector.py:433-482) accumulated on the device
//   pmx_keypoints                              FaceDetector / HandDetector post-process (face_detector.py:37-68, hand_detector.py:41-78)
// Context, weights, the forward plan and the pose post-process live in pmx_api.hip; the shared context type in pmx_ctx.h.
#include "pmx_ctx.h"

#include <math.h>
#include <string.h>
#include <tuple>
#include <utility>

// ---------------------------------------------------------------------------------------- detect_precise on the device
// OpenCV bicubic tables for one axis (A = -0.75): idx[k][d] (clamped, replicate border) and coef[k][d] float32, k = 0..3.
// Same float32 expression order as pose_detector.py::_cubic_taps / oracle/precise_ref.py::_coeffs.
#pragma clang fp contract(off)
static void make_cubic_table(int dst, int src, int* idx, float* coef)
{
    const double scale = 1.0 / ((double)dst / (double)src);
    const float A = -0.75f;
    for (int d = 0; d < dst; ++d) {
        const float f = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        const float x = f - (float)s;
        const float x1 = x + 1.0f, xm = 1.0f - x;
        float c0 = A * x1;  c0 = c0 - 5.0f * A;  c0 = c0 * x1;  c0 = c0 + 8.0f * A;  c0 = c0 * x1;  c0 = c0 - 4.0f * A;
        float c1 = (A + 2.0f) * x;  c1 = c1 - (A + 3.0f);  c1 = c1 * x;  c1 = c1 * x;  c1 = c1 + 1.0f;
        float c2 = (A + 2.0f) * xm;  c2 = c2 - (A + 3.0f);  c2 = c2 * xm;  c2 = c2 * xm;  c2 = c2 + 1.0f;
        float c3 = 1.0f - c0;  c3 = c3 - c1;  c3 = c3 - c2;
        const float cs[4] = {c0, c1, c2, c3};
        for (int k = 0; k < 4; ++k) {
            int i = s - 1 + k;
            i = i < 0 ? 0 : (i > src - 1 ? src - 1 : i);
            idx[k * dst + d] = i;
            coef[k * dst + d] = cs[k];
        }
    }
}

// Device copy of the cubic table of ONE axis for (src -> dst): [4 dst indices | 4 dst coefficients (float32, or 11-bit fixed point when
// `fixed`)].  Tables are a pure function of (src, dst, fixed) and detect_precise asks for the same dozen on every call (4 scales x 3 resizes x
// 2 axes), so they are built once and kept: no stream synchronisation and no blocking copy per scale (round 4 rebuilt and re-uploaded
// them three times per scale behind a hipStreamSynchronize each).  A fresh table is a fresh allocation, so nothing in flight can be
// reading the memory it is copied to.  NOTHING is ever freed here: a scale takes six tables and keeps their raw pointers for launches it
// enqueues afterwards, on a lane whose stream this function does not see -- the cache is trimmed only between sequences
// (cubic_tables_trim, called by pmx_precise_begin_batch behind a device synchronisation).
static int cubic_table(pmx_ctx* c, int src, int dst, bool fixed, const int** idx, const void** coef)
{
    const auto key = std::make_tuple(src, dst, fixed ? 1 : 0);
    auto it = c->pr_tabs.find(key);
    if (it == c->pr_tabs.end()) {
        const size_t n = (size_t)4 * dst;
        std::vector<int> hi(2 * n);
        std::vector<float> hc(n);
        make_cubic_table(dst, src, hi.data(), hc.data());
        if (fixed) {
            for (size_t k = 0; k < n; ++k) {
                long v = lrintf(hc[k] * 2048.0f);              // saturate_cast<short>(coef * INTER_RESIZE_COEF_SCALE)
                hi[n + k] = (int)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
        } else {
            memcpy(hi.data() + n, hc.data(), n * sizeof(float));
        }
        int* d = nullptr;
        PMX_HIP(hipMalloc((void**)&d, 2 * n * sizeof(int)));
        if (hipMemcpy(d, hi.data(), 2 * n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            pmx_set_error("cubic table upload failed");
            return PMX_ERR_HIP;
        }
        it = c->pr_tabs.emplace(key, d).first;
    }
    *idx = it->second;
    *coef = (const void*)(it->second + (size_t)4 * dst);
    return PMX_OK;
}

// A context fed ever new image sizes (detect_precise over a data set: ~22 tables per distinct size) starts its cache over rather than grow
// without bound -- at the START of a sequence only, and only after every lane has drained: no launch of an earlier sequence can still
// be reading a table, and the new sequence has not taken a pointer yet.  `cap` = option "precise_table_cap" (entries; one sequence of 8
// scales adds at most 48).
static int cubic_tables_trim(pmx_ctx* c)
{
    if ((int)c->pr_tabs.size() < c->opt_precise_table_cap) return PMX_OK;
    PMX_HIP(hipDeviceSynchronize());
    for (auto& kv : c->pr_tabs) (void)hipFree(kv.second);
    c->pr_tabs.clear();
    c->pr_tabs_trims += 1;
    return PMX_OK;
}

// ---- lanes: one inference scale in flight per lane (pmx_ctx.h::PrLane) --------------------------------------------------------------
template <typename T>
static int lane_alloc(T** q, size_t count)
{
    if (*q) (void)hipFree(*q);
    *q = nullptr;
    PMX_HIP(hipMalloc((void**)q, count * sizeof(T)));
    return PMX_OK;
}
// lane i >= 1 holds n x ph x pw pixels of working set (the sizes pmx_create gives the context's own buffers); lane 0 IS the context
static int lane_ensure(pmx_ctx* c, int i, size_t n, size_t ph, size_t pw)
{
    PrLane& l = c->pr_lane[i];
    if (!l.done) PMX_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
    if (i == 0) return PMX_OK;
    if (!l.stream) {
        // Stream priorities (option "precise_lane_priority", default on): the reference's scale order is 0.5, 1, 1.5, 2, so the last lane carries
        // the largest scale -- the longest chain of the sequence (8.4 of ~19 ms of single-lane time for a 482 x 642 frame), whose 7x7 layers
        // are 192 one-per-CU blocks: it finishes a layer in ONE round only if it gets its CUs the moment it asks for them.  With equal
        // priorities the lanes take CUs from each other in arrival order, the large lane's launches split over two rounds and its chain --
        // the critical path -- doubles.  The highest priority for lane 3, the lowest for lane 1: the small scales fill what the large one
        // leaves free.
        int lo = 0, hi = 0;
        PMX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));          // (lo = numerically greatest = least urgent)
        int prio = 0;
        if (c->opt_precise_lane_priority) prio = i == PMX_PR_LANES - 1 ? hi : i == 1 ? lo : (lo + hi) / 2;
        PMX_HIP(hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, prio));
    }
    const size_t px = n * ph * pw;
    if (px <= l.cap_px) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(l.stream));
    l.cap_px = 0;
    const size_t px8 = px / 64;
    int rc;
    if ((rc = lane_alloc(&l.in16, px * PMX_IN_C)) || (rc = lane_alloc(&l.act0, px * 64)) || (rc = lane_alloc(&l.act1, px * 16)) ||
        (rc = lane_alloc(&l.cat, px8 * PMX_CAT_C)) || (rc = lane_alloc(&l.brA, px8 * 256)) || (rc = lane_alloc(&l.brB, px8 * 256)) ||
        (rc = lane_alloc(&l.brT, px8 * 1024)) || (rc = lane_alloc(&l.u8_tmp, px * 3))) return rc;
    PMX_HIP(hipMemsetAsync(l.cat, 0, px8 * PMX_CAT_C * sizeof(float), l.stream));      // pad channels stay zero (stream-ordered)
    l.cap_px = px;
    return PMX_OK;
}
// the lane's stream and working set <-> the context's (called in pairs around the enqueue of one scale; lane 0: nothing to do)
static void lane_swap(pmx_ctx* c, int i)
{
    if (i == 0) return;
    PrLane& l = c->pr_lane[i];
    std::swap(c->stream, l.stream);
    std::swap(c->in16, l.in16); std::swap(c->act0, l.act0); std::swap(c->act1, l.act1); std::swap(c->cat, l.cat);
    std::swap(c->brA, l.brA); std::swap(c->brB, l.brB); std::swap(c->brT, l.brT); std::swap(c->u8_tmp, l.u8_tmp);
    std::swap(c->pr_tmp, l.pr_tmp); std::swap(c->pr_tmp_cap, l.pr_tmp_cap);
    std::swap(c->sk_scratch, l.sk_scratch); std::swap(c->sk_floats, l.sk_floats);
}

// detect_precise (pose_detector.py:433-470) on the device, for a batch of n images of ONE original size (the reference handles one image
// per call; n = 1 is that call).  Every scale runs the n images as one batch through the network, the scales of a sequence run
// concurrently on up to four lanes.  begin: reset the sequence.
extern "C" int pmx_precise_begin_batch(pmx_ctx* c, int n_images, int orig_h, int orig_w)
{
    PMX_CHECK(c && c->kind == NET_POSE, PMX_ERR_INVALID, "pmx_precise_begin: posenet context required");
    PMX_CHECK(orig_h >= 1 && orig_w >= 1, PMX_ERR_INVALID, "pmx_precise_begin: bad size");
    PMX_CHECK(n_images >= 1 && n_images <= c->max_batch, PMX_ERR_CAPACITY, "pmx_precise_begin: %d images outside 1..%d (the context's batch capacity)",
              n_images, c->max_batch);
    PMX_DEV(c);
    if (int rc = cubic_tables_trim(c)) return rc;
    const size_t need = (size_t)n_images * orig_h * orig_w;
    if (need > c->ext_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->ext_paf) (void)hipFree(c->ext_paf);
        if (c->ext_heat) (void)hipFree(c->ext_heat);
        c->ext_paf = c->ext_heat = nullptr; c->ext_cap = 0;
        PMX_HIP(hipMalloc((void**)&c->ext_paf, need * PMX_N_PAF * 4));
        PMX_HIP(hipMalloc((void**)&c->ext_heat, need * PMX_N_HEAT * 4));
        c->ext_cap = need;
    }
    if (!c->pr_src_ready) PMX_HIP(hipEventCreateWithFlags(&c->pr_src_ready, hipEventDisableTiming));
    if (!c->pr_fin) PMX_HIP(hipEventCreateWithFlags(&c->pr_fin, hipEventDisableTiming));
    // the zero-bias vector of the unit-mode / split-K launches is shared by all lanes: it must exist (and be zero) before two streams can meet it
    if (!c->sk_zero_bias) {
        PMX_HIP(hipMalloc((void**)&c->sk_zero_bias, PMX_SK_ZERO_BIAS * sizeof(float)));
        PMX_HIP(hipMemset(c->sk_zero_bias, 0, PMX_SK_ZERO_BIAS * sizeof(float)));
    }
    c->pr_h = orig_h; c->pr_w = orig_w; c->pr_scales = 0; c->pr_mask = 0; c->pr_n = n_images; c->pr_src = nullptr;
    c->maps_valid = false;
    return PMX_OK;
}
extern "C" int pmx_precise_table_stats(pmx_ctx* c, int* cached, int* trims)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null arg");
    if (cached) *cached = (int)c->pr_tabs.size();
    if (trims) *trims = c->pr_tabs_trims;
    return PMX_OK;
}
extern "C" int pmx_precise_begin(pmx_ctx* c, int orig_h, int orig_w) { return pmx_precise_begin_batch(c, 1, orig_h, orig_w); }

// one scale of the loop at :441-467 for every image of the batch: cubic resize of the uint8 image to (scaled_h, scaled_w) (:443), pad to a
// multiple of 8 with (104, 117, 123) (:445), forward (:451), x8 cubic up-sampling of both outputs (:461,465), crop of the padding
// (:462,466), cubic resize to the original size (:463,467) into this scale's part.  `imgs`: host uint8, n x orig_h x orig_w x 3, contiguous.
// Everything is enqueued on the scale's lane; nothing here waits for the device.
static int precise_add_scale(pmx_ctx* c, const uint8_t* imgs, int scaled_h, int scaled_w, int slot);
extern "C" int pmx_precise_add_scale_batch(pmx_ctx* c, const uint8_t* imgs, int scaled_h, int scaled_w)
{
    return precise_add_scale(c, imgs, scaled_h, scaled_w, -1);
}
// The same with the POSITION of the scale in the reference's loop given explicitly (slot 0 .. 7; each once per sequence): the parts are
// summed in slot order (:463,467) whatever order the scales were enqueued in -- so a caller can enqueue the LARGEST scale first.  Its
// chain of ~100 launches is the critical path of the sequence; enqueued last (the reference's order 0.5, 1, 1.5, 2) it starts only after
// the host has spent ~1 ms enqueueing the three smaller ones.  Slot s runs on lane s % lanes (the priorities follow the slot).
extern "C" int pmx_precise_add_scale_at(pmx_ctx* c, const uint8_t* imgs, int scaled_h, int scaled_w, int slot)
{
    PMX_CHECK(slot >= 0 && slot < 8, PMX_ERR_INVALID, "pmx_precise_add_scale_at: slot %d outside 0..7", slot);
    return precise_add_scale(c, imgs, scaled_h, scaled_w, slot);
}
static int precise_add_scale(pmx_ctx* c, const uint8_t* imgs, int scaled_h, int scaled_w, int slot)
{
    PMX_CHECK(c && imgs && c->pr_h > 0 && c->pr_n > 0, PMX_ERR_STATE, "pmx_precise_add_scale: call pmx_precise_begin first");
    PMX_CHECK(scaled_h >= 1 && scaled_w >= 1, PMX_ERR_INVALID, "bad size");
    PMX_DEV(c);
    const int oh = c->pr_h, ow = c->pr_w, n = c->pr_n;
    const int ph = round_up(scaled_h, 8), pw = round_up(scaled_w, 8);
    PMX_CHECK((size_t)ph * pw <= (size_t)c->max_h * c->max_w && c->max_batch >= n, PMX_ERR_CAPACITY,
              "pmx_precise_add_scale: %d x padded size %d x %d exceeds the context capacity %d x %d x %d", n, ph, pw, c->max_batch, c->max_h, c->max_w);
    int missing = 0;
    for (auto& l : c->layers) missing += l.set ? 0 : 1;
    PMX_CHECK(missing == 0, PMX_ERR_WEIGHTS, "pmx_precise_add_scale: %d layers have no weights", missing);
    int rc;
    PMX_CHECK(c->pr_scales < 8, PMX_ERR_CAPACITY, "pmx_precise_add_scale: at most 8 scales per sequence");
    if (slot < 0) {                                   // the next free position
        slot = 0;
        while (slot < 8 && ((c->pr_mask >> slot) & 1u)) ++slot;
    }
    PMX_CHECK(slot < 8 && !((c->pr_mask >> slot) & 1u), PMX_ERR_STATE, "pmx_precise_add_scale: slot %d already holds a scale of this sequence", slot);
    const bool first = c->pr_mask == 0;
    // Lane of this scale, by ENQUEUE order: the first scale of a sequence -- the largest one when the caller follows the advice above --
    // gets lane 3, whose stream carries the highest priority, to itself; the others go round-robin over the remaining lanes in use
    // (0 = the context's own stream, then 2, then 1 = the lowest priority).  One lane: everything on the context's stream.
    const int L = c->opt_precise_lanes < 1 ? 1 : (c->opt_precise_lanes > PMX_PR_LANES ? PMX_PR_LANES : c->opt_precise_lanes);
    static const int others[PMX_PR_LANES - 1] = {0, 2, 1};
    const int k = slot, e = c->pr_scales;
    const int li = L == 1 ? 0 : (e == 0 ? PMX_PR_LANES - 1 : others[(e - 1) % (L - 1)]);
    hipStream_t main_stream = c->stream;
    // original images -> device, ONCE per begin / finish sequence, on the context's stream: every scale resizes the same originals (the
    // caller passes the same images to every pmx_precise_add_scale* of a sequence -- include/pose_mi355x.h)
    const size_t img_bytes = (size_t)oh * ow * 3, nsrc = img_bytes * n;
    if (first || c->pr_src != imgs) {
        if (nsrc > c->u8_src_cap) {
            PMX_HIP(hipDeviceSynchronize());          // (lanes of an earlier sequence may still read the old buffer)
            if (c->u8_src) (void)hipFree(c->u8_src);
            c->u8_src = nullptr; c->u8_src_cap = 0;
            PMX_HIP(hipMalloc((void**)&c->u8_src, nsrc));
            c->u8_src_cap = nsrc;
        }
        if (!first) {                                 // (a different buffer mid-sequence: the lanes still reading the old copy finish first)
            for (int i = 1; i < PMX_PR_LANES; ++i)
                if (c->pr_lane[i].done && c->pr_lane[i].stream) PMX_HIP(hipStreamWaitEvent(main_stream, c->pr_lane[i].done, 0));
        }
        PMX_HIP(hipMemcpyAsync(c->u8_src, imgs, nsrc, hipMemcpyHostToDevice, main_stream));
        PMX_HIP(hipEventRecord(c->pr_src_ready, main_stream));
        c->pr_src = imgs;
    }
    // this scale's part: [n][38][oh][ow] | [n][19][oh][ow]
    const size_t opx = (size_t)oh * ow, part_floats = opx * 57 * n;
    if (part_floats > c->pr_part_cap || (size_t)k >= c->pr_part.size()) {
        if (part_floats > c->pr_part_cap) {           // a larger sequence than any before: all parts are re-made (nothing of the old ones is pending: k == 0 or synced)
            PMX_HIP(hipDeviceSynchronize());
            for (float*& q : c->pr_part) { if (q) (void)hipFree(q); q = nullptr; }
            c->pr_part_cap = part_floats;
        }
        if ((size_t)k >= c->pr_part.size()) c->pr_part.resize(k + 1, nullptr);
    }
    for (size_t j = 0; j <= (size_t)k; ++j)
        if (!c->pr_part[j]) PMX_HIP(hipMalloc((void**)&c->pr_part[j], c->pr_part_cap * sizeof(float)));
    float* const part_paf = c->pr_part[k];
    float* const part_heat = part_paf + opx * PMX_N_PAF * n;
    if ((rc = lane_ensure(c, li, (size_t)n, (size_t)ph, (size_t)pw))) return rc;
    const int *xi, *yi; const void *xc, *yc;
    // cubic tables first (a cache miss is a blocking copy into fresh memory: before the swap, so that an error leaves the context intact)
    const int *t1xi = nullptr, *t1yi = nullptr; const void *t1xc = nullptr, *t1yc = nullptr;
    const bool same = scaled_h == oh && scaled_w == ow;
    if (!same && ((rc = cubic_table(c, ow, scaled_w, true, &t1xi, &t1xc)) || (rc = cubic_table(c, oh, scaled_h, true, &t1yi, &t1yc)))) return rc;
    const int fh = ph / 8, fw = pw / 8;
    const int *t3xi, *t3yi, *t4xi, *t4yi; const void *t3xc, *t3yc, *t4xc, *t4yc;
    if ((rc = cubic_table(c, fw, pw, false, &t3xi, &t3xc)) || (rc = cubic_table(c, fh, ph, false, &t3yi, &t3yc)) ||
        (rc = cubic_table(c, scaled_w, ow, false, &t4xi, &t4xc)) || (rc = cubic_table(c, scaled_h, oh, false, &t4yi, &t4yc))) return rc;
    (void)xi; (void)yi; (void)xc; (void)yc;

    lane_swap(c, li);                                 // from here on c->stream / c->in16 / ... are the lane's; every exit swaps back
    auto body = [&]() -> int {
        int rc2;
        PMX_HIP(hipStreamWaitEvent(c->stream, c->pr_src_ready, 0));      // the originals are on the device
        PMX_HIP(hipStreamWaitEvent(c->stream, c->pr_fin, 0));            // the last finish has read this scale's part
        // (1) uint8 cubic resize into the padded images (all images in one launch)
        const size_t pad_bytes = (size_t)ph * pw * 3;
        if ((rc2 = launch_fill_pad_bgr(c->u8_tmp, n, ph, pw, scaled_h, scaled_w, 104, 117, 123, c->stream))) return rc2;
        if (same) {
            for (int b = 0; b < n; ++b)
                PMX_HIP(hipMemcpy2DAsync(c->u8_tmp + b * pad_bytes, (size_t)pw * 3, c->u8_src + b * img_bytes, (size_t)ow * 3, (size_t)ow * 3, oh,
                                         hipMemcpyDeviceToDevice, c->stream));
        } else if ((rc2 = launch_resize_cubic_u8(c->u8_src, ow, c->u8_tmp, scaled_h, scaled_w, pw, t1xi, (const int*)t1xc, t1yi, (const int*)t1yc, n,
                                                 (long long)img_bytes, (long long)pad_bytes, c->stream))) return rc2;
        // (2) network, the n images as one batch.  With several scales in flight the chip is shared: what counts is the CU time a scale
        // consumes, not how fast it would finish alone -- so the lanes run the PLAIN Winograd kernel on every eligible layer ("conv_algo" 2:
        // 0.83 of the matrix peak per block) instead of the unit-mode / split-K forms the selection gives a launch that has the chip to
        // itself (0.45: they trade CU time for latency).  Round 6, 482 x 642 frame, lanes with priorities: 14.2 -> 13.5 ms per image
        // (profiles/r06_precise_priority_ab.json).  Option "precise_plain": -1 (default) = when all four lanes are in use, 0 / 1 = never / always.
        const int algo_saved = c->opt_conv_algo;
        // (-1 = only with all four lanes in use: a lane that carries two scales one after the other wants the latency-oriented forms --
        //  three lanes 15.0 -> 20.4 ms, two 16.0 -> 25.0 with the plain kernels, profiles/r06_precise_probe_lanes.json)
        const bool plain = c->opt_precise_plain < 0 ? c->opt_precise_lanes >= PMX_PR_LANES : c->opt_precise_plain != 0;
        if (plain && c->opt_conv_algo == 1) c->opt_conv_algo = 2;
        rc2 = pmx_forward_from_u8(c, c->u8_tmp, n, ph, pw, 255.0f);
        c->opt_conv_algo = algo_saved;
        if (rc2) return rc2;
        // (3) x8 cubic up-sampling of the PAF (38) and heat (19) channels of all images into PLANAR temporaries [n][38][ph][pw] | [n][19][ph][pw]
        // (two cv2.resize calls per image in the reference; planar so that step (4) reads rows of one channel and both steps store full rows)
        const size_t ppx = (size_t)ph * pw, ntmp = ppx * 57 * n;
        if (ntmp > c->pr_tmp_cap) {
            PMX_HIP(hipStreamSynchronize(c->stream));
            if (c->pr_tmp) (void)hipFree(c->pr_tmp);
            c->pr_tmp = nullptr; c->pr_tmp_cap = 0;
            PMX_HIP(hipMalloc((void**)&c->pr_tmp, ntmp * sizeof(float)));
            c->pr_tmp_cap = ntmp;
        }
        float* const t_paf = c->pr_tmp;
        float* const t_heat = c->pr_tmp + ppx * PMX_N_PAF * n;
        const long long sy = (long long)fw * PMX_CAT_C, sx = PMX_CAT_C, sb = (long long)fh * fw * PMX_CAT_C;
        if ((rc2 = launch_resize_cubic_f32_planar(c->cat + PMX_CAT_PAF, sb, 1, sy, sx, n, PMX_N_PAF, t_paf, ph, pw, t3xi, (const float*)t3xc, t3yi, (const float*)t3yc, 0, c->stream))) return rc2;
        if ((rc2 = launch_resize_cubic_f32_planar(c->cat + PMX_CAT_HEAT, sb, 1, sy, sx, n, PMX_N_HEAT, t_heat, ph, pw, t3xi, (const float*)t3xc, t3yi, (const float*)t3yc, 0, c->stream))) return rc2;
        // (4) crop the padding (source extent scaled_h x scaled_w of the padded maps) and cubic resize to the original size -> this scale's part
        if ((rc2 = launch_resize_cubic_f32_planar(t_paf, (long long)ppx * PMX_N_PAF, (long long)ppx, pw, 1, n, PMX_N_PAF, part_paf, oh, ow, t4xi, (const float*)t4xc, t4yi,
                                                  (const float*)t4yc, 0, c->stream))) return rc2;
        if ((rc2 = launch_resize_cubic_f32_planar(t_heat, (long long)ppx * PMX_N_HEAT, (long long)ppx, pw, 1, n, PMX_N_HEAT, part_heat, oh, ow, t4xi, (const float*)t4xc, t4yi,
                                                  (const float*)t4yc, 0, c->stream))) return rc2;
        PMX_HIP(hipEventRecord(c->pr_lane[li].done, c->stream));
        return PMX_OK;
    };
    rc = body();
    lane_swap(c, li);
    if (rc) return rc;
    c->pr_scales += 1;
    c->pr_mask |= 1u << slot;
    c->maps_valid = false;       // the cat buffers hold single scales only; the averaged maps become valid in pmx_precise_finish
    return PMX_OK;
}
extern "C" int pmx_precise_add_scale(pmx_ctx* c, const uint8_t* img, int scaled_h, int scaled_w)
{
    PMX_CHECK(c && c->pr_n == 1, PMX_ERR_STATE, "pmx_precise_add_scale: the batch was begun with %d images (use pmx_precise_add_scale_batch)", c ? c->pr_n : 0);
    return pmx_precise_add_scale_batch(c, img, scaled_h, scaled_w);
}

// :463,467,469-470: sum the scales' parts left to right starting from zero (the reference's `sum = sum + resized`), divide by the number
// of scales and install the result as the maps of a batch of n at the original size
extern "C" int pmx_precise_finish(pmx_ctx* c)
{
    PMX_CHECK(c && c->pr_h > 0 && c->pr_scales > 0 && c->pr_n > 0, PMX_ERR_STATE, "pmx_precise_finish: nothing accumulated");
    PMX_DEV(c);
    int rc;
    const long long n = (long long)c->pr_n * c->pr_h * c->pr_w;
    // every lane that ever ran a scale (its last event covers all its scales; an event of an earlier sequence is complete and costs nothing):
    // independent of the `precise_lanes` option, which may have changed since the scales were enqueued
    for (int i = 0; i < PMX_PR_LANES; ++i)
        if (c->pr_lane[i].done) PMX_HIP(hipStreamWaitEvent(c->stream, c->pr_lane[i].done, 0));
    PMX_CHECK(c->pr_scales <= 8, PMX_ERR_CAPACITY, "pmx_precise_finish: %d scales (at most 8 per sequence)", c->pr_scales);
    PMX_CHECK(c->pr_mask == (1u << c->pr_scales) - 1u, PMX_ERR_STATE, "pmx_precise_finish: the slots of the sequence have gaps (mask 0x%x, %d scales)", c->pr_mask, c->pr_scales);
    if ((rc = launch_sum_parts_f32(c->ext_paf, c->pr_part.data(), c->pr_scales, 0, n * PMX_N_PAF, (float)c->pr_scales, c->stream))) return rc;
    if ((rc = launch_sum_parts_f32(c->ext_heat, c->pr_part.data(), c->pr_scales, n * PMX_N_PAF, n * PMX_N_HEAT, (float)c->pr_scales, c->stream))) return rc;
    PMX_HIP(hipEventRecord(c->pr_fin, c->stream));
    c->maps_valid = true; c->maps_external = true;
    c->cur_B = c->pr_n; c->cur_fh = c->pr_h; c->cur_fw = c->pr_w;
    c->pp_valid = false;
    c->pr_scales = 0; c->pr_mask = 0;
    return PMX_OK;
}

// FaceDetector / HandDetector.__call__ post-process (face_detector.py:37-38,58-68; hand_detector.py:41,68-78):
// F.resize_images(hs[-1], (out_h, out_w)) + gaussian_filter + per-channel arg-max over the n_heat - 1 key-point channels.
// out: batch x (n_heat - 1) x 4 float64 rows (x, y, confidence, valid); valid = 0 where the reference appends None.
extern "C" int pmx_keypoints(pmx_ctx* c, int B, int out_h, int out_w, double thresh, double* out)
{
    PMX_CHECK(c && out, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(c->kind != NET_POSE, PMX_ERR_STATE, "pmx_keypoints: facenet / handnet only");
    PMX_CHECK(c->maps_valid && B == c->cur_B, PMX_ERR_STATE, "pmx_keypoints: no network output for batch %d", B);
    PMX_CHECK(out_h >= 1 && out_w >= 1 && (long long)out_h * out_w < (1ll << 31), PMX_ERR_INVALID, "pmx_keypoints: bad size");
    PMX_DEV(c);
    int rc;
    if ((rc = pmx_ensure_tables(c, c->cur_fh, c->cur_fw, out_h, out_w, c->opt_kp_flip_x))) return rc;
    const int n_ch = c->n_heat - 1;
    const long long fhw = (long long)c->cur_fh * c->cur_fw;
    PPMaps m;
    if (c->maps_external) {
        m.heat = c->ext_heat; m.paf = nullptr; m.sx = 1; m.sy = c->cur_fw; m.sc = fhw; m.sbh = c->n_heat * fhw; m.sbp = 0;
    } else {
        m.heat = c->cat + c->cat_heat; m.paf = nullptr; m.sc = 1; m.sx = c->cat_c; m.sy = (long long)c->cur_fw * c->cat_c;
        m.sbh = fhw * c->cat_c; m.sbp = 0;
    }
    m.fh = c->cur_fh; m.fw = c->cur_fw;
    const size_t need = (size_t)B * n_ch * out_h * out_w;
    if (need > c->smoothed_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->pp.smoothed) (void)hipFree(c->pp.smoothed);
        c->pp.smoothed = nullptr;
        PMX_HIP(hipMalloc((void**)&c->pp.smoothed, need * sizeof(float)));
        c->smoothed_cap = need;
    }
    const size_t nkp = (size_t)B * n_ch * 4;
    if (nkp > c->kp_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->d_kp) (void)hipFree(c->d_kp);
        c->d_kp = nullptr;
        PMX_HIP(hipMalloc((void**)&c->d_kp, nkp * sizeof(double)));
        c->kp_cap = nkp;
    }
    if ((rc = pp_keypoints_launch(m, c->tab, c->pp, B, n_ch, out_h, out_w, thresh, c->d_kp, c->stream))) return rc;
    PMX_HIP(hipMemcpyAsync(out, c->d_kp, nkp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PMX_HIP(hipStreamSynchronize(c->stream));
    c->pp_valid = true; c->pp_final = true; c->pp_B = B; c->pp_h = out_h; c->pp_w = out_w;
    return PMX_OK;
}


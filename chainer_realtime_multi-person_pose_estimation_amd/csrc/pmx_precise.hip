// pmx_precise.hip -- the C-ABI entry points of the multi-scale path and of the key-point networks (include/pose_mi355x.h):
//   pmx_precise_begin / _add_scale / _finish   detect_precise (reference pose_detector.py:433-482) accumulated on the device
//   pmx_keypoints                              FaceDetector / HandDetector post-process (face_detector.py:37-68, hand_detector.py:41-78)
// Context, weights, the forward plan and the pose post-process live in pmx_api.hip; the shared context type in pmx_ctx.h.
#include "pmx_ctx.h"

#include <math.h>
#include <string.h>
#include <tuple>

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
// 2 axes), so they are built once and kept for the life of the context: no stream synchronisation and no blocking copy per scale (round 4
// rebuilt and re-uploaded them three times per scale behind a hipStreamSynchronize each).  A fresh table is a fresh allocation, so nothing
// in flight can be reading the memory it is copied to.
static int cubic_table(pmx_ctx* c, int src, int dst, bool fixed, const int** idx, const void** coef)
{
    const auto key = std::make_tuple(src, dst, fixed ? 1 : 0);
    auto it = c->pr_tabs.find(key);
    if (it == c->pr_tabs.end()) {
        if (c->pr_tabs.size() >= 256) {                       // (a context fed ever new sizes: start over rather than grow without bound)
            PMX_HIP(hipStreamSynchronize(c->stream));
            for (auto& kv : c->pr_tabs) (void)hipFree(kv.second);
            c->pr_tabs.clear();
        }
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

// detect_precise (pose_detector.py:433-470) accumulated on the device, for a batch of n images of ONE original size (the reference
// handles one image per call; n = 1 is that call).  Every scale runs the n images as one batch through the network -- a 184 x 248 input of
// a single image is 23 x 31 maps, far too little for 256 CUs, eight of them are not -- and the (tiny, shared) resize tables are uploaded
// once per step instead of once per image.  begin: zero the per-channel sums at the original size.
extern "C" int pmx_precise_begin_batch(pmx_ctx* c, int n_images, int orig_h, int orig_w)
{
    PMX_CHECK(c && c->kind == NET_POSE, PMX_ERR_INVALID, "pmx_precise_begin: posenet context required");
    PMX_CHECK(orig_h >= 1 && orig_w >= 1, PMX_ERR_INVALID, "pmx_precise_begin: bad size");
    PMX_CHECK(n_images >= 1 && n_images <= c->max_batch, PMX_ERR_CAPACITY, "pmx_precise_begin: %d images outside 1..%d (the context's batch capacity)",
              n_images, c->max_batch);
    PMX_DEV(c);
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
    PMX_HIP(hipMemsetAsync(c->ext_paf, 0, need * PMX_N_PAF * 4, c->stream));
    PMX_HIP(hipMemsetAsync(c->ext_heat, 0, need * PMX_N_HEAT * 4, c->stream));
    c->pr_h = orig_h; c->pr_w = orig_w; c->pr_scales = 0; c->pr_n = n_images; c->pr_src = nullptr;
    c->maps_valid = false;
    return PMX_OK;
}
extern "C" int pmx_precise_begin(pmx_ctx* c, int orig_h, int orig_w) { return pmx_precise_begin_batch(c, 1, orig_h, orig_w); }

// one scale of the loop at :441-467 for every image of the batch: cubic resize of the uint8 image to (scaled_h, scaled_w) (:443), pad to a
// multiple of 8 with (104, 117, 123) (:445), forward (:451), x8 cubic up-sampling of both outputs (:461,465), crop of the padding
// (:462,466), cubic resize to the original size and accumulation (:463,467).  `imgs`: host uint8, n x orig_h x orig_w x 3, contiguous.
extern "C" int pmx_precise_add_scale_batch(pmx_ctx* c, const uint8_t* imgs, int scaled_h, int scaled_w)
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
    // original images -> device, ONCE per begin / finish sequence: every scale resizes the same originals (the caller passes the same
    // images to every pmx_precise_add_scale* of a sequence -- include/pose_mi355x.h; round 4 uploaded them again from pageable memory per scale)
    const size_t img_bytes = (size_t)oh * ow * 3, nsrc = img_bytes * n;
    if (c->pr_scales == 0 || c->pr_src != imgs) {
        if (nsrc > c->u8_src_cap) {
            PMX_HIP(hipStreamSynchronize(c->stream));
            if (c->u8_src) (void)hipFree(c->u8_src);
            c->u8_src = nullptr; c->u8_src_cap = 0;
            PMX_HIP(hipMalloc((void**)&c->u8_src, nsrc));
            c->u8_src_cap = nsrc;
        }
        PMX_HIP(hipMemcpyAsync(c->u8_src, imgs, nsrc, hipMemcpyHostToDevice, c->stream));
        c->pr_src = imgs;
    }
    const int *xi, *yi; const void *xc, *yc;
    // (1) uint8 cubic resize into the padded images (all images in one launch)
    const size_t pad_bytes = (size_t)ph * pw * 3;
    if ((rc = launch_fill_bgr(c->u8_tmp, (long long)n * ph * pw, 104, 117, 123, c->stream))) return rc;
    if (scaled_h == oh && scaled_w == ow) {
        for (int b = 0; b < n; ++b)
            PMX_HIP(hipMemcpy2DAsync(c->u8_tmp + b * pad_bytes, (size_t)pw * 3, c->u8_src + b * img_bytes, (size_t)ow * 3, (size_t)ow * 3, oh,
                                     hipMemcpyDeviceToDevice, c->stream));
    } else {
        if ((rc = cubic_table(c, ow, scaled_w, true, &xi, &xc)) || (rc = cubic_table(c, oh, scaled_h, true, &yi, &yc))) return rc;
        if ((rc = launch_resize_cubic_u8(c->u8_src, ow, c->u8_tmp, scaled_h, scaled_w, pw, xi, (const int*)xc, yi, (const int*)yc, n, (long long)img_bytes,
                                         (long long)pad_bytes, c->stream))) return rc;
    }
    // (2) network, the n images as one batch
    if ((rc = launch_prep_u8(c->u8_tmp, c->in16, n, ph, pw, 255.0f, c->stream))) return rc;
    if ((rc = pmx_forward_from_in16(c, n, ph, pw))) return rc;
    const int fh = ph / 8, fw = pw / 8;
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
    if ((rc = cubic_table(c, fw, pw, false, &xi, &xc)) || (rc = cubic_table(c, fh, ph, false, &yi, &yc))) return rc;
    const long long sy = (long long)fw * PMX_CAT_C, sx = PMX_CAT_C, sb = (long long)fh * fw * PMX_CAT_C;
    if ((rc = launch_resize_cubic_f32_planar(c->cat + PMX_CAT_PAF, sb, 1, sy, sx, n, PMX_N_PAF, t_paf, ph, pw, xi, (const float*)xc, yi, (const float*)yc, 0, c->stream))) return rc;
    if ((rc = launch_resize_cubic_f32_planar(c->cat + PMX_CAT_HEAT, sb, 1, sy, sx, n, PMX_N_HEAT, t_heat, ph, pw, xi, (const float*)xc, yi, (const float*)yc, 0, c->stream))) return rc;
    // (4) crop the padding (source extent scaled_h x scaled_w of the padded maps) and cubic resize to the original size, accumulating
    if ((rc = cubic_table(c, scaled_w, ow, false, &xi, &xc)) || (rc = cubic_table(c, scaled_h, oh, false, &yi, &yc))) return rc;
    if ((rc = launch_resize_cubic_f32_planar(t_paf, (long long)ppx * PMX_N_PAF, (long long)ppx, pw, 1, n, PMX_N_PAF, c->ext_paf, oh, ow, xi, (const float*)xc, yi,
                                             (const float*)yc, 1, c->stream))) return rc;
    if ((rc = launch_resize_cubic_f32_planar(t_heat, (long long)ppx * PMX_N_HEAT, (long long)ppx, pw, 1, n, PMX_N_HEAT, c->ext_heat, oh, ow, xi, (const float*)xc, yi,
                                             (const float*)yc, 1, c->stream))) return rc;
    c->pr_scales += 1;
    c->maps_valid = false;       // the cat buffer holds one scale only; the averaged maps become valid in pmx_precise_finish
    return PMX_OK;
}
extern "C" int pmx_precise_add_scale(pmx_ctx* c, const uint8_t* img, int scaled_h, int scaled_w)
{
    PMX_CHECK(c && c->pr_n == 1, PMX_ERR_STATE, "pmx_precise_add_scale: the batch was begun with %d images (use pmx_precise_add_scale_batch)", c ? c->pr_n : 0);
    return pmx_precise_add_scale_batch(c, img, scaled_h, scaled_w);
}

// :469-470: divide the sums by the number of scales and install them as the maps of a batch of n at the original size
extern "C" int pmx_precise_finish(pmx_ctx* c)
{
    PMX_CHECK(c && c->pr_h > 0 && c->pr_scales > 0 && c->pr_n > 0, PMX_ERR_STATE, "pmx_precise_finish: nothing accumulated");
    PMX_DEV(c);
    int rc;
    const long long n = (long long)c->pr_n * c->pr_h * c->pr_w;
    if ((rc = launch_scale_f32(c->ext_paf, n * PMX_N_PAF, (float)c->pr_scales, c->stream))) return rc;
    if ((rc = launch_scale_f32(c->ext_heat, n * PMX_N_HEAT, (float)c->pr_scales, c->stream))) return rc;
    c->maps_valid = true; c->maps_external = true;
    c->cur_B = c->pr_n; c->cur_fh = c->pr_h; c->cur_fw = c->pr_w;
    c->pp_valid = false;
    c->pr_scales = 0;
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


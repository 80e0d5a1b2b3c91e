// pmx_multi.hip -- batches of images of DIFFERENT sizes (include/pose_mi355x.h: pmx_detect_images, pmx_forward_u8_images,
// pmx_postprocess_images, pmx_get_image_maps).
// The reference takes any image in any call and picks the network size per image (pose_detector.py:490-493, :57-73); a stream of COCO-style
// frames therefore has a different network input size every few images, and a batch entry that demands one common size falls back to
// one image per call -- a 36-block launch per 7x7 layer on 256 CUs.  Here a batch is a list of SEGMENTS (consecutive images of one
// network-input size): the segments lie end to end in every activation buffer and every layer is ONE launch over the tiles of all
// segments (the plain Winograd kernel on 8 x 16 rectangles, conv1_wino_kernel's 16 x 16 squares, the 1x1 pairs on the flat pixel run);
// a per-level device table (pmx_common.h::ConvSeg) tells a block which segment its tile belongs to.  A block of such a launch computes
// exactly what a block of a plain launch of that image alone computes -- the per-pixel arithmetic of the plain Winograd kernel does not
// depend on the launch -- so the maps are bit-identical to per-image calls with the plain kernels (tests/test_gpu_multi.py), and the C
// twin needs no new case.  The post-process runs per segment on its slice of the post-process buffers; records come back in image order.
#include "pmx_ctx.h"

#include <string.h>
#include <algorithm>

namespace {

struct Geo { int n, H, W, mh, mw; };       // a segment: n images, network input H x W, up-sampled map mh x mw

int seg_capacity_check(pmx_ctx* c, const std::vector<Geo>& g, int B)
{
    PMX_CHECK(B >= 1 && B <= c->max_batch, PMX_ERR_CAPACITY, "mixed batch: %d images outside 1..%d (the context's batch capacity)", B, c->max_batch);
    size_t px = 0;
    for (const Geo& s : g) {
        PMX_CHECK(s.H >= 8 && s.W >= 8 && s.H % 8 == 0 && s.W % 8 == 0, PMX_ERR_INVALID, "mixed batch: network sizes must be multiples of 8 (got %d x %d)", s.H, s.W);
        PMX_CHECK((long long)s.H * s.W * 64 * 4 < (1ll << 31), PMX_ERR_INVALID, "mixed batch: image %d x %d too large for 32-bit offsets", s.H, s.W);
        px += (size_t)s.n * s.H * s.W;
    }
    PMX_CHECK(px <= (size_t)c->max_batch * c->max_h * c->max_w, PMX_ERR_CAPACITY,
              "mixed batch: %zu network-input pixels exceed the context capacity %d x %d x %d", px, c->max_batch, c->max_h, c->max_w);
    return PMX_OK;
}

// the six segment tables of a forward (pmx_ctx.h) -> device
int build_seg_tables(pmx_ctx* c, const std::vector<Geo>& g)
{
    const size_t ns = g.size();
    std::vector<ConvSeg> t(PMX_SEG_TABLES * ns);
    long long pix[5] = {0, 0, 0, 0, 0};
    int tiles[PMX_SEG_TABLES] = {};
    for (size_t s = 0; s < ns; ++s) {
        for (int tab = 0; tab < PMX_SEG_TABLES; ++tab) {
            const int level = tab == PMX_SEG_CONV1 ? 0 : tab <= PMX_SEG_L1P ? 1 : tab <= PMX_SEG_L2P ? 2 : 3;
            const bool pooled = tab == PMX_SEG_CONV1 || tab == PMX_SEG_L1P || tab == PMX_SEG_L2P;
            const int th = tab == PMX_SEG_CONV1 ? 16 : 8, tw = 16;
            ConvSeg& q = t[tab * ns + s];
            q.H = g[s].H >> level; q.W = g[s].W >> level;
            q.tiles_x = (q.W + tw - 1) / tw;
            q.tiles_img = q.tiles_x * ((q.H + th - 1) / th);
            q.tile0 = tiles[tab];
            q.n = g[s].n;
            PMX_CHECK(pix[level] < (1ll << 31) && pix[level + (pooled ? 1 : 0)] < (1ll << 31), PMX_ERR_CAPACITY, "mixed batch: pixel offsets exceed 31 bits");
            q.pix0 = (int)pix[level];
            q.pixo = (int)pix[level + (pooled ? 1 : 0)];
            PMX_CHECK((long long)tiles[tab] + (long long)q.tiles_img * q.n < (1ll << 31), PMX_ERR_CAPACITY, "mixed batch: too many tiles");
            tiles[tab] += q.tiles_img * q.n;
        }
        for (int level = 0; level < 4; ++level) pix[level] += (long long)g[s].n * (g[s].H >> level) * (g[s].W >> level);
    }
    for (int tab = 0; tab < PMX_SEG_TABLES; ++tab) c->seg_tiles[tab] = tiles[tab];
    for (int level = 0; level < 4; ++level) c->seg_pix[level] = pix[level];
    if (t.size() > c->d_segs_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->d_segs) (void)hipFree(c->d_segs);
        c->d_segs = nullptr; c->d_segs_cap = 0;
        PMX_HIP(hipMalloc((void**)&c->d_segs, t.size() * sizeof(ConvSeg)));
        c->d_segs_cap = t.size();
    }
    // (stream-ordered: the kernels of an earlier forward that still read the table run before this copy; the source is pageable
    //  memory, which the runtime stages before the call returns)
    PMX_HIP(hipMemcpyAsync(c->d_segs, t.data(), t.size() * sizeof(ConvSeg), hipMemcpyHostToDevice, c->stream));
    PMX_HIP(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

// up-sampling tables of one (network map, up-sampled map) size pair, built once per context and kept (a fresh allocation: nothing in
// flight reads it; the cache is started over only by pmx_detect_images / pmx_postprocess_images behind a device synchronisation)
int cached_tables(pmx_ctx* c, int in_h, int in_w, int out_h, int out_w, PPTables* out)
{
    // (the peak-branch option changes the Gaussian taps / border flags the table set carries: part of the key; a changed pmx_set_gaussian
    //  is not -- the mirror sets the taps once, right after pmx_create)
    const auto key = std::make_tuple(in_h, in_w, out_h, c->opt_gpu_branch_peaks ? -out_w : out_w);
    auto it = c->tab_cache.find(key);
    if (it != c->tab_cache.end()) { *out = it->second; return PMX_OK; }
    // the context's own single-size machinery builds the grids (np.linspace semantics, Gaussian taps, peak-branch flags) ...
    int rc = pmx_ensure_tables(c, in_h, in_w, out_h, out_w);
    if (rc) return rc;
    // ... and the cache keeps a private copy: one allocation [xi0 | xi1 | yi0 | yi1 | xlo | xhi | ylo | yhi | gauss]
    const PPTables& src = c->tab;
    const size_t ni = (size_t)2 * out_w + (size_t)2 * out_h, nd = (size_t)2 * out_w + (size_t)2 * out_h + (2 * PMX_GAUSS_MAX_RADIUS + 1);
    const size_t ibytes = (ni * sizeof(int) + 7) / 8 * 8;
    char* base = nullptr;
    PMX_HIP(hipMalloc((void**)&base, ibytes + nd * sizeof(double)));
    PPTables t = src;
    int* ip = reinterpret_cast<int*>(base);
    double* dp = reinterpret_cast<double*>(base + ibytes);
    t.xi0 = ip; t.xi1 = ip + out_w; t.yi0 = ip + 2 * out_w; t.yi1 = ip + 2 * out_w + out_h;
    t.xlo = dp; t.xhi = dp + out_w; t.ylo = dp + 2 * out_w; t.yhi = dp + 2 * out_w + out_h; t.gauss = dp + 2 * out_w + 2 * out_h;
    hipError_t e = hipSuccess;
    auto cp = [&](void* d, const void* s, size_t n) { if (e == hipSuccess) e = hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); };
    cp(t.xi0, src.xi0, out_w * sizeof(int)); cp(t.xi1, src.xi1, out_w * sizeof(int));
    cp(t.yi0, src.yi0, out_h * sizeof(int)); cp(t.yi1, src.yi1, out_h * sizeof(int));
    cp(t.xlo, src.xlo, out_w * sizeof(double)); cp(t.xhi, src.xhi, out_w * sizeof(double));
    cp(t.ylo, src.ylo, out_h * sizeof(double)); cp(t.yhi, src.yhi, out_h * sizeof(double));
    cp(t.gauss, src.gauss, (2 * PMX_GAUSS_MAX_RADIUS + 1) * sizeof(double));
    if (e != hipSuccess) { (void)hipFree(base); pmx_set_error("post-process table copy failed: %s", hipGetErrorString(e)); return PMX_ERR_HIP; }
    c->tab_cache.emplace(key, t);
    *out = t;
    return PMX_OK;
}

int trim_table_cache(pmx_ctx* c)
{
    if (c->tab_cache.size() < 512) return PMX_OK;
    PMX_HIP(hipDeviceSynchronize());
    for (auto& kv : c->tab_cache) (void)hipFree(kv.second.xi0);
    c->tab_cache.clear();
    return PMX_OK;
}

std::vector<Geo> segments_of(const int* net_hw, const int* map_hw, int B)
{
    std::vector<Geo> g;
    for (int i = 0; i < B; ++i) {
        const int H = net_hw[2 * i], W = net_hw[2 * i + 1], mh = map_hw ? map_hw[2 * i] : 0, mw = map_hw ? map_hw[2 * i + 1] : 0;
        if (!g.empty() && g.back().H == H && g.back().W == W && g.back().mh == mh && g.back().mw == mw) g.back().n += 1;
        else g.push_back(Geo{1, H, W, mh, mw});
    }
    return g;
}

// network forward over the segments; d_u8: the images' uint8 pixels end to end, on the device
int forward_segments(pmx_ctx* c, const uint8_t* d_u8, const std::vector<Geo>& g, int B)
{
    int rc = build_seg_tables(c, g);
    if (rc) return rc;
    int mh = 0, mw = 0;
    c->segs.clear();
    for (const Geo& s : g) { c->segs.push_back(SegDesc{s.n, s.H, s.W}); mh = std::max(mh, s.H); mw = std::max(mw, s.W); }
    // (H, W of the call = the largest segment: what the launch checks bound; the geometry itself comes from the tables)
    rc = pmx_forward_from_u8(c, d_u8, B, mh, mw, 255.0f);
    c->segs.clear();                 // heterogeneous mode ends with the enqueue; cur_segs keeps the layout of the maps
    if (rc) { c->cur_segs.clear(); c->maps_valid = false; }
    return rc;
}

}  // namespace

// The network on a mixed batch.  bgr: the B images' uint8 BGR pixels end to end (image i: net_hw[2 i] x net_hw[2 i + 1] x 3), host or
// device memory.  Consecutive images of one size form a segment.
extern "C" int pmx_forward_u8_images(pmx_ctx* c, const uint8_t* bgr, const int* net_hw, int B, int on_device)
{
    PMX_CHECK(c && bgr && net_hw, PMX_ERR_INVALID, "pmx_forward_u8_images: null arg");
    PMX_CHECK(c->kind == NET_POSE, PMX_ERR_STATE, "pmx_forward_u8_images: posenet contexts only");
    const std::vector<Geo> g = segments_of(net_hw, nullptr, B > 0 ? B : 0);
    int rc = seg_capacity_check(c, g, B);
    if (rc || (rc = pmx_check_weights(c))) return rc;
    PMX_DEV(c);
    const uint8_t* d = bgr;
    if (!on_device) {
        size_t bytes = 0;
        for (const Geo& s : g) bytes += (size_t)s.n * s.H * s.W * 3;
        PMX_HIP(hipMemcpyAsync(c->u8_tmp, bgr, bytes, hipMemcpyHostToDevice, c->stream));
        d = c->u8_tmp;
    }
    return forward_segments(c, d, g, B);
}

// pmx_postprocess for the maps of a mixed batch: image i's maps are up-sampled to map_hw[2 i] x map_hw[2 i + 1] (:501-502), img_len = that
// width (:511), scale_xy as in pmx_postprocess (B x 2 doubles or NULL).  One launch set per run of images with equal sizes.
extern "C" int pmx_postprocess_images(pmx_ctx* c, const int* map_hw, int B, const double* scale_xy)
{
    PMX_CHECK(c && map_hw, PMX_ERR_INVALID, "pmx_postprocess_images: null arg");
    PMX_CHECK(c->kind == NET_POSE, PMX_ERR_STATE, "pmx_postprocess_images: posenet only");
    PMX_CHECK(c->maps_valid && !c->cur_segs.empty(), PMX_ERR_STATE, "pmx_postprocess_images: the current maps are not those of a mixed batch (pmx_forward_u8_images first)");
    PMX_CHECK(B == c->cur_B, PMX_ERR_INVALID, "pmx_postprocess_images: batch %d != batch of the current maps %d", B, c->cur_B);
    PMX_CHECK(!c->opt_keep_smoothed, PMX_ERR_STATE, "pmx_postprocess_images: option keep_smoothed is not available for mixed batches");
    PMX_DEV(c);
    int rc = trim_table_cache(c);
    if (rc) return rc;
    // image -> (segment geometry, first pixel of its maps at level 3)
    std::vector<PPCall> calls;
    {
        int img = 0;
        long long pix = 0;
        for (const SegDesc& s : c->cur_segs) {
            const int fh = s.H / 8, fw = s.W / 8;
            for (int k = 0; k < s.n; ++k, ++img) {
                const int mh = map_hw[2 * img], mw = map_hw[2 * img + 1];
                PMX_CHECK(mh >= 1 && mw >= 1 && (long long)mh * mw < (1ll << 31), PMX_ERR_INVALID, "pmx_postprocess_images: bad map size of image %d", img);
                if (k > 0 && calls.back().map_h == mh && calls.back().map_w == mw) { calls.back().B += 1; continue; }
                PPCall q{};
                q.maps.heat = c->cat + (size_t)(pix + (long long)k * fh * fw) * PMX_CAT_C + PMX_CAT_HEAT;
                q.maps.paf = c->cat + (size_t)(pix + (long long)k * fh * fw) * PMX_CAT_C + PMX_CAT_PAF;
                q.maps.sc = 1; q.maps.sx = PMX_CAT_C; q.maps.sy = (long long)fw * PMX_CAT_C;
                q.maps.sbh = q.maps.sbp = (long long)fh * fw * PMX_CAT_C;
                q.maps.fh = fh; q.maps.fw = fw;
                q.base = img; q.B = 1; q.map_h = mh; q.map_w = mw; q.img_len = (double)mw; q.has_scale = scale_xy != nullptr;
                q.limbs_slices = c->opt_limbs_slices >= 0 ? c->opt_limbs_slices : 0;
                calls.push_back(q);
            }
            pix += (long long)s.n * fh * fw;
        }
        PMX_CHECK(img == B, PMX_ERR_STATE, "pmx_postprocess_images: segment bookkeeping (%d images, batch %d)", img, B);
    }
    for (PPCall& q : calls)
        if ((rc = cached_tables(c, q.maps.fh, q.maps.fw, q.map_h, q.map_w, &q.tab))) return rc;
    c->tab_in_h = -1;                 // (the context's single-size table was used as scratch by cached_tables)
    if (scale_xy) PMX_HIP(hipMemcpyAsync(c->d_scale, scale_xy, sizeof(double) * 2 * B, hipMemcpyHostToDevice, c->stream));
    for (const PPCall& q : calls)
        if ((rc = pp_launch(q.maps, q.tab, pmx_pp_view(c->pp, q.base), q.B, q.map_h, q.map_w, q.img_len, scale_xy ? c->d_scale + 2 * q.base : nullptr, 0,
                            c->stream, nullptr, nullptr, q.limbs_slices))) return rc;
    c->pp_calls = calls;
    c->pp_valid = true; c->pp_final = false; c->pp_B = B; c->pp_h = c->pp_w = 0;
    c->pp_has_scale = scale_xy != nullptr;
    return PMX_OK;
}

// `PoseDetector.__call__` (pose_detector.py:484-517) for B images of ANY sizes in one call: per image cv2.resize to its network size
// (:493, on the device), then the network over the size classes as one launch per layer, then the post-process per class with the
// image's own map size, img_len (:511) and coordinate rescale (:513-514).  Records in image order (pmx_get_results).
extern "C" int pmx_detect_images(pmx_ctx* c, const pmx_image* imgs, int B)
{
    PMX_CHECK(c && imgs, PMX_ERR_INVALID, "pmx_detect_images: null arg");
    PMX_CHECK(c->kind == NET_POSE, PMX_ERR_STATE, "pmx_detect_images: posenet contexts only");
    PMX_CHECK(B >= 1 && B <= c->max_batch, PMX_ERR_CAPACITY, "pmx_detect_images: %d images outside 1..%d", B, c->max_batch);
    std::vector<int> net_hw(2 * (size_t)B), map_hw(2 * (size_t)B);
    std::vector<double> scale(2 * (size_t)B);
    size_t src_bytes = 0, tab_ints = 0;
    for (int i = 0; i < B; ++i) {
        const pmx_image& m = imgs[i];
        PMX_CHECK(m.bgr && m.src_h >= 1 && m.src_w >= 1 && m.map_h >= 1 && m.map_w >= 1, PMX_ERR_INVALID, "pmx_detect_images: image %d: bad descriptor", i);
        net_hw[2 * i] = m.net_h; net_hw[2 * i + 1] = m.net_w; map_hw[2 * i] = m.map_h; map_hw[2 * i + 1] = m.map_w;
        scale[2 * i] = (double)m.src_w / (double)m.map_w; scale[2 * i + 1] = (double)m.src_h / (double)m.map_h;     // :513-514
        if (m.src_h != m.net_h || m.src_w != m.net_w) { src_bytes += (size_t)m.src_h * m.src_w * 3; tab_ints += (size_t)4 * (m.net_w + m.net_h); }
    }
    const std::vector<Geo> g = segments_of(net_hw.data(), map_hw.data(), B);
    int rc = seg_capacity_check(c, g, B);
    if (rc || (rc = pmx_check_weights(c))) return rc;
    PMX_DEV(c);
    if (src_bytes > c->mi_src_cap || tab_ints > c->mi_tab_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (src_bytes > c->mi_src_cap) {
            if (c->mi_src) (void)hipFree(c->mi_src);
            c->mi_src = nullptr; c->mi_src_cap = 0;
            PMX_HIP(hipMalloc((void**)&c->mi_src, src_bytes));
            c->mi_src_cap = src_bytes;
        }
        if (tab_ints > c->mi_tab_cap) {
            if (c->mi_tab) (void)hipFree(c->mi_tab);
            c->mi_tab = nullptr; c->mi_tab_cap = 0;
            PMX_HIP(hipMalloc((void**)&c->mi_tab, tab_ints * sizeof(int)));
            c->mi_tab_cap = tab_ints;
        }
    }
    // resize tables of all images in one upload, then per image: upload + resize into its place of the network-input run (identity: straight in)
    std::vector<int> tabs(tab_ints);
    {
        size_t o = 0;
        for (int i = 0; i < B; ++i) {
            const pmx_image& m = imgs[i];
            if (m.src_h == m.net_h && m.src_w == m.net_w) continue;
            pmx_make_resize_table(m.net_w, m.src_w, tabs.data() + o);
            pmx_make_resize_table(m.net_h, m.src_h, tabs.data() + o + 4 * (size_t)m.net_w);
            o += (size_t)4 * (m.net_w + m.net_h);
        }
    }
    if (tab_ints) PMX_HIP(hipMemcpyAsync(c->mi_tab, tabs.data(), tab_ints * sizeof(int), hipMemcpyHostToDevice, c->stream));
    size_t so = 0, to = 0, dst = 0;
    for (int i = 0; i < B; ++i) {
        const pmx_image& m = imgs[i];
        uint8_t* out = c->u8_tmp + dst;
        if (m.src_h == m.net_h && m.src_w == m.net_w) {
            PMX_HIP(hipMemcpyAsync(out, m.bgr, (size_t)m.net_h * m.net_w * 3, hipMemcpyHostToDevice, c->stream));
        } else {
            const size_t nb = (size_t)m.src_h * m.src_w * 3;
            PMX_HIP(hipMemcpyAsync(c->mi_src + so, m.bgr, nb, hipMemcpyHostToDevice, c->stream));
            if ((rc = launch_resize_linear_u8(c->mi_src + so, out, c->mi_tab + to, c->mi_tab + to + 4 * (size_t)m.net_w, 1, m.src_h, m.src_w, m.net_h, m.net_w, c->stream))) return rc;
            so += nb; to += (size_t)4 * (m.net_w + m.net_h);
        }
        dst += (size_t)m.net_h * m.net_w * 3;
    }
    PMX_HIP(hipStreamSynchronize(c->stream));      // the host tables (and the caller's images) have been read
    if ((rc = forward_segments(c, c->u8_tmp, g, B))) return rc;
    return pmx_postprocess_images(c, map_hw.data(), B, scale.data());
}

// parity accessor: the network output of ONE image of the current batch (uniform or mixed) as NCHW float32 -- paf 38 x fh x fw, heat
// 19 x fh x fw (either may be NULL); fh, fw must be the image's map size (network size / 8)
extern "C" int pmx_get_image_maps(pmx_ctx* c, int image, float* paf, float* heat, int fh, int fw)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(c->kind == NET_POSE && c->maps_valid && !c->maps_external, PMX_ERR_STATE, "pmx_get_image_maps: no posenet forward yet");
    PMX_CHECK(image >= 0 && image < c->cur_B, PMX_ERR_INVALID, "pmx_get_image_maps: image %d outside 0..%d", image, c->cur_B - 1);
    PMX_DEV(c);
    long long pix = 0;
    int ih = c->cur_fh, iw = c->cur_fw;
    if (c->cur_segs.empty()) {
        pix = (long long)image * ih * iw;
    } else {
        int first = 0;
        for (const SegDesc& s : c->cur_segs) {
            ih = s.H / 8; iw = s.W / 8;
            if (image < first + s.n) { pix += (long long)(image - first) * ih * iw; break; }
            pix += (long long)s.n * ih * iw; first += s.n;
        }
    }
    PMX_CHECK(fh == ih && fw == iw, PMX_ERR_INVALID, "pmx_get_image_maps: image %d has %d x %d maps (asked for %d x %d)", image, ih, iw, fh, fw);
    int rc;
    const float* src = c->cat + (size_t)pix * PMX_CAT_C;
    const size_t np = (size_t)PMX_N_PAF * fh * fw, nh = (size_t)PMX_N_HEAT * fh * fw;
    PMX_CHECK((np + nh) * sizeof(float) <= c->nchw_tmp_bytes, PMX_ERR_CAPACITY, "pmx_get_image_maps: staging buffer too small");
    if (paf) {
        if ((rc = launch_nhwc_to_nchw(src, c->nchw_tmp, 1, PMX_N_PAF, fh, fw, PMX_CAT_C, PMX_CAT_PAF, c->stream))) return rc;
        PMX_HIP(hipMemcpyAsync(paf, c->nchw_tmp, np * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (heat) {
        if ((rc = launch_nhwc_to_nchw(src, c->nchw_tmp + np, 1, PMX_N_HEAT, fh, fw, PMX_CAT_C, PMX_CAT_HEAT, c->stream))) return rc;
        PMX_HIP(hipMemcpyAsync(heat, c->nchw_tmp + np, nh * 4, hipMemcpyDeviceToHost, c->stream));
    }
    PMX_HIP(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

// pmx_api.hip -- C ABI of libpose_mi355x (include/pose_mi355x.h): context, weights, forward plan, accessors.
//
// Forward plan = models/CocoPoseNet.py:132-262 expressed as 47 convolution launches on NHWC buffers:
//   stem (12 launches; the three F.max_pooling_2d are fused into the epilogues of conv1_2, conv2_2, conv3_4),
//   stage 1 (5 launches) and stages 2-6 (7 launches each); the PAF branch (L1) and the heat-map branch (L2)
//   of a stage are the two groups (blockIdx.z) of one launch; F.concat (:168,...) is replaced by channel-slice
//   writes into the 192-channel "cat" buffer (layout in pmx_common.h).
#include "pmx_ctx.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>

// ------------------------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
void pmx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* pmx_last_error(void) { return g_err; }
extern "C" const char* pmx_version(void) { return "pose_mi355x 0.1 (gfx950, fp32 MFMA)"; }

extern "C" int pmx_device_count(int* n)
{
    PMX_CHECK(n, PMX_ERR_INVALID, "pmx_device_count: null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *n = c;
    return PMX_OK;
}

// -------------------------------------------------------------------------------------- layer table

static int net_out_channels(int kind) { return kind == NET_FACE ? 71 : (kind == NET_HAND ? 22 : 19); }

// FaceNet / HandNet (models/FaceNet.py:12-75, models/HandNet.py): VGG-19 stem to conv5_2, conv5_3_CPM, single-branch 6-stage CPM
static std::vector<LayerDesc> make_cpm_table(int C)
{
    std::vector<LayerDesc> t = {
        {"conv1_1", 3, 64, 3}, {"conv1_2", 64, 64, 3}, {"conv2_1", 64, 128, 3}, {"conv2_2", 128, 128, 3},
        {"conv3_1", 128, 256, 3}, {"conv3_2", 256, 256, 3}, {"conv3_3", 256, 256, 3}, {"conv3_4", 256, 256, 3},
        {"conv4_1", 256, 512, 3}, {"conv4_2", 512, 512, 3}, {"conv4_3", 512, 512, 3}, {"conv4_4", 512, 512, 3},
        {"conv5_1", 512, 512, 3}, {"conv5_2", 512, 512, 3}, {"conv5_3_CPM", 512, 128, 3},
        {"conv6_1_CPM", 128, 512, 1}, {"conv6_2_CPM", 512, C, 1}};
    char buf[64];
    for (int s = 2; s <= 6; ++s) {
        snprintf(buf, sizeof buf, "Mconv1_stage%d", s); t.push_back({buf, C + 128, 128, 7});
        for (int i = 2; i <= 5; ++i) { snprintf(buf, sizeof buf, "Mconv%d_stage%d", i, s); t.push_back({buf, 128, 128, 7}); }
        snprintf(buf, sizeof buf, "Mconv6_stage%d", s); t.push_back({buf, 128, 128, 1});
        snprintf(buf, sizeof buf, "Mconv7_stage%d", s); t.push_back({buf, 128, C, 1});
    }
    return t;
}

static std::vector<LayerDesc> make_layer_table()   // models/CocoPoseNet.py:26-129
{
    std::vector<LayerDesc> t = {
        {"conv1_1", 3, 64, 3}, {"conv1_2", 64, 64, 3}, {"conv2_1", 64, 128, 3}, {"conv2_2", 128, 128, 3},
        {"conv3_1", 128, 256, 3}, {"conv3_2", 256, 256, 3}, {"conv3_3", 256, 256, 3}, {"conv3_4", 256, 256, 3},
        {"conv4_1", 256, 512, 3}, {"conv4_2", 512, 512, 3}, {"conv4_3_CPM", 512, 256, 3}, {"conv4_4_CPM", 256, 128, 3}};
    const char* br[2] = {"L1", "L2"};
    const int bco[2] = {38, 19};
    char buf[64];
    for (int g = 0; g < 2; ++g) {
        for (int i = 1; i <= 3; ++i) { snprintf(buf, sizeof buf, "conv5_%d_CPM_%s", i, br[g]); t.push_back({buf, 128, 128, 3}); }
        snprintf(buf, sizeof buf, "conv5_4_CPM_%s", br[g]); t.push_back({buf, 128, 512, 1});
        snprintf(buf, sizeof buf, "conv5_5_CPM_%s", br[g]); t.push_back({buf, 512, bco[g], 1});
    }
    for (int s = 2; s <= 6; ++s)
        for (int g = 0; g < 2; ++g) {
            snprintf(buf, sizeof buf, "Mconv1_stage%d_%s", s, br[g]); t.push_back({buf, 185, 128, 7});
            for (int i = 2; i <= 5; ++i) { snprintf(buf, sizeof buf, "Mconv%d_stage%d_%s", i, s, br[g]); t.push_back({buf, 128, 128, 7}); }
            snprintf(buf, sizeof buf, "Mconv6_stage%d_%s", s, br[g]); t.push_back({buf, 128, 128, 1});
            snprintf(buf, sizeof buf, "Mconv7_stage%d_%s", s, br[g]); t.push_back({buf, 128, bco[g], 1});
        }
    return t;
}



static int cout_pad_of(int cout) { return cout <= 64 ? 64 : round_up(cout, 128); }

// pack OIHW -> [tap][chunk][cout_pad][CK]; cin_map[k] = source input channel of packed channel k (or -1 = zero)
static void pack_weights(const float* w, const float* bias, int cout, int cin, int ks, const std::vector<int>& cin_map,
                         int cout_pad, std::vector<float>& wp, std::vector<float>& bp)
{
    const int cin_pad = (int)cin_map.size(), nch = cin_pad / CK, T = ks * ks;
    wp.assign((size_t)T * nch * cout_pad * CK, 0.f);
    bp.assign((size_t)cout_pad, 0.f);
    for (int n = 0; n < cout; ++n) bp[n] = bias ? bias[n] : 0.f;
    for (int tap = 0; tap < T; ++tap)
        for (int k = 0; k < cin_pad; ++k) {
            const int src = cin_map[k];
            if (src < 0) continue;
            const int ch = k / CK, c = k % CK;
            for (int n = 0; n < cout; ++n)
                wp[(((size_t)tap * nch + ch) * cout_pad + n) * CK + c] = w[((size_t)n * cin + src) * T + tap];
        }
}

// fp32 -> three bf16 terms, each the round-to-nearest-even bf16 of what is left (as conv_bf16x3_kernel splits the activations)
static inline uint16_t bf16_rn(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_f(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// packed fp32 weights [tap][chunk][cout_pad][16] -> bf16x3 [tap][chunk][plane][cout_pad][16]
// Winograd F(2x2, 3x3) weights: U = G g G^T per (cout, cin), G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]], evaluated in double and
// rounded once to fp32 (oracle/conv_fma_ref.c::conv_wino_ref does the same); layout [plane = sub-kernel * 16 + 4i + j][chunk of 32
// cin][k8-step 4][cout_pad][8].  ks = 3: one sub-kernel; ks = 7: four, sub-kernel (sy, sx) = taps (3 sy .. 3 sy + 2, 3 sx .. 3 sx + 2); row 6 and
// column 6 of the 7x7 kernel are two 1x3 / two 3x1 sub-kernels with the 1-D transform G g (planes 64.., 72..), tap (6, 6) is plane 80
static void pack_wino(const std::vector<float>& wp, int ks, int nch16, int cout_pad, std::vector<float>& out)
{
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int cin_pad = nch16 * CK, nch32 = cin_pad / 32, nsub = ks == 3 ? 1 : 4;
    const int nplanes = ks == 3 ? 16 : 81;      // 7x7: 64 (four 3x3 sub-kernels) + 8 (row 6: two 1x3, G g) + 8 (column 6: two 3x1) + 1 (tap (6, 6))
    out.assign((size_t)nplanes * nch32 * cout_pad * 32, 0.f);
    auto tapw = [&](int ky, int kx, int n, int ci) -> double {
        return wp[(((size_t)(ky * ks + kx) * nch16 + ci / CK) * cout_pad + n) * CK + ci % CK];
    };
    auto put = [&](int plane, int n, int ci, double v) {
        // [plane][chunk32][cout_pad / 32][k8-step][32][8]: the four k8-steps of a wave's 32 channels are 1 KB apart (an immediate offset of the load)
        out[(((((size_t)plane * nch32 + ci / 32) * (cout_pad / 32) + n / 32) * 4 + (ci % 32) / 8) * 32 + n % 32) * 8 + ci % 8] = (float)v;
    };
    for (int n = 0; n < cout_pad; ++n)
        for (int ci = 0; ci < cin_pad; ++ci) {
            for (int sub = 0; sub < nsub; ++sub) {
                double gg[4][3];
                for (int i = 0; i < 4; ++i)
                    for (int kx = 0; kx < 3; ++kx)
                        gg[i][kx] = (Gm[i][0] * tapw(3 * (sub >> 1) + 0, 3 * (sub & 1) + kx, n, ci) + Gm[i][1] * tapw(3 * (sub >> 1) + 1, 3 * (sub & 1) + kx, n, ci)) +
                                    Gm[i][2] * tapw(3 * (sub >> 1) + 2, 3 * (sub & 1) + kx, n, ci);
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) put(sub * 16 + 4 * i + j, n, ci, (gg[i][0] * Gm[j][0] + gg[i][1] * Gm[j][1]) + gg[i][2] * Gm[j][2]);
            }
            if (ks == 7) {
                for (int sub = 0; sub < 2; ++sub)
                    for (int f = 0; f < 4; ++f) {
                        put(64 + sub * 4 + f, n, ci, (Gm[f][0] * tapw(6, 3 * sub + 0, n, ci) + Gm[f][1] * tapw(6, 3 * sub + 1, n, ci)) + Gm[f][2] * tapw(6, 3 * sub + 2, n, ci));
                        put(72 + sub * 4 + f, n, ci, (Gm[f][0] * tapw(3 * sub + 0, 6, n, ci) + Gm[f][1] * tapw(3 * sub + 1, 6, n, ci)) + Gm[f][2] * tapw(3 * sub + 2, 6, n, ci));
                    }
                put(80, n, ci, tapw(6, 6, n, ci));
            }
        }
}

static void pack_bf16x3(const std::vector<float>& wp, int T, int nch, int cout_pad, std::vector<uint16_t>& out)
{
    out.assign((size_t)T * nch * 3 * cout_pad * CK, 0);
    for (size_t pc = 0; pc < (size_t)T * nch; ++pc)
        for (int n = 0; n < cout_pad; ++n)
            for (int k = 0; k < CK; ++k) {
                const float x = wp[(pc * cout_pad + n) * CK + k];
                const uint16_t h = bf16_rn(x);
                const float r1 = x - bf16_f(h);
                const uint16_t m = bf16_rn(r1);
                const uint16_t l = bf16_rn(r1 - bf16_f(m));
                const size_t base = pc * 3 * cout_pad * CK + (size_t)n * CK + k;
                out[base] = h; out[base + (size_t)cout_pad * CK] = m; out[base + 2 * (size_t)cout_pad * CK] = l;
            }
}

static std::vector<int> identity_map(int cin)
{
    std::vector<int> m(round_up(cin, CK), -1);
    for (int i = 0; i < cin; ++i) m[i] = i;
    return m;
}

// F.concat((h1, h2, feature_map)) channel c of the reference (0..37 PAF, 38..56 heat, 57..184 feature) lives at
// cat-buffer channel: feature -> 0..127, PAF -> 128..165, heat -> 168..186
static std::vector<int> concat_map()
{
    std::vector<int> m(PMX_CAT_C, -1);
    for (int i = 0; i < 128; ++i) m[PMX_CAT_FEAT + i] = 57 + i;
    for (int i = 0; i < 38; ++i) m[PMX_CAT_PAF + i] = i;
    for (int i = 0; i < 19; ++i) m[PMX_CAT_HEAT + i] = 38 + i;
    return m;
}

// F.concat((h, feature_map)) of FaceNet / HandNet (models/FaceNet.py:107): reference channel c (0..C-1 heat, C..C+127
// feature) lives at cat-buffer channel: feature -> 0..127, heat -> 128..128+C-1, zero pad up to a multiple of 16
static std::vector<int> concat_map_cpm(int C)
{
    std::vector<int> m(round_up(128 + C, CK), -1);
    for (int i = 0; i < 128; ++i) m[i] = C + i;
    for (int i = 0; i < C; ++i) m[128 + i] = i;
    return m;
}


static int prof_begin(pmx_ctx* c, const std::string& name0, double flops, double bytes, double issued = -1.0, bool record = true)
{
    if (!c->prof_on) return PMX_OK;
    // (one half of a batch cut in two by images, run_conv: the label carries "@<first image>+<count>")
    const std::string name = c->split_suffix.empty() ? name0 : name0 + c->split_suffix;
    int idx;
    auto it = c->prof_index.find(name);
    if (it == c->prof_index.end()) {
        idx = (int)c->prof.size();
        ProfEntry e; e.name = name; e.flops = flops; e.bytes = bytes; e.issued = issued < 0 ? flops : issued;
        c->prof.push_back(e);
        c->prof_index[name] = idx;
    } else idx = it->second;
    ProfPending p; p.entry = idx;
    for (hipEvent_t* e : {&p.e0, &p.e1}) {
        if (!c->ev_pool.empty()) { *e = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else PMX_HIP(hipEventCreate(e));
    }
    if (record) PMX_HIP(hipEventRecord(p.e0, c->stream));
    c->pending.push_back(p);
    c->prof_open = (int)c->pending.size() - 1;
    return PMX_OK;
}
static int prof_end(pmx_ctx* c)
{
    if (!c->prof_on || c->prof_open < 0) return PMX_OK;
    PMX_HIP(hipEventRecord(c->pending[c->prof_open].e1, c->stream));
    c->prof_open = -1;
    return PMX_OK;
}
static int prof_collect(pmx_ctx* c)
{
    if (c->pending.empty()) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(c->stream));
    for (auto& p : c->pending) {
        float ms = 0.f;
        PMX_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        c->prof[p.entry].total_ms += ms;
        c->prof[p.entry].launches += 1;
        c->ev_pool.push_back(p.e0);
        c->ev_pool.push_back(p.e1);
    }
    c->pending.clear();
    return PMX_OK;
}
static void pp_prof_cb(void* vc, const char* name, int begin)
{
    pmx_ctx* c = (pmx_ctx*)vc;
    if (begin) (void)prof_begin(c, std::string(name) + "|" + name, 0, 0);
    else (void)prof_end(c);
}


template <typename T>
static int dev_alloc(T** p, size_t count)
{
    PMX_HIP(hipMalloc((void**)p, count * sizeof(T) ? count * sizeof(T) : 16));
    return PMX_OK;
}

// (re)allocates every post-process buffer whose size depends on the capacities in c->pp (cap_pk, cap_sub, cap_cand)
static void pp_free(pmx_ctx* c)
{
    PPBuffers& p = c->pp;
    void* ptrs[] = {p.pk_raw_key, p.pk_raw_score, p.pk_count, p.pk_x, p.pk_y, p.pk_score, p.pk_start, p.cn_a, p.cn_b, p.cn_score,
                    p.cn_count, p.cn_need, p.cand_score, p.cand_idx, p.cand_used, p.sub_work, p.subsets, p.status, p.results,
                    p.scan_score, p.scan_idx, p.scan_cnt};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    p.pk_raw_key = nullptr; p.pk_raw_score = nullptr; p.pk_count = nullptr; p.pk_x = p.pk_y = nullptr; p.pk_score = nullptr;
    p.pk_start = nullptr; p.cn_a = p.cn_b = nullptr; p.cn_score = nullptr; p.cn_count = p.cn_need = nullptr;
    p.cand_score = nullptr; p.cand_idx = nullptr; p.cand_used = nullptr; p.sub_work = p.subsets = nullptr; p.status = nullptr;
    p.results = nullptr;
    p.scan_score = nullptr; p.scan_idx = nullptr; p.scan_cnt = nullptr;
}

static int pp_alloc(pmx_ctx* c)
{
    PPBuffers& p = c->pp;
    const size_t B = c->max_batch, npk = (size_t)PMX_N_JOINTS * p.cap_pk;
    int rc;
    p.rec_bytes = PMX_RECORD_BYTES(p.cap_ppl);
    if ((rc = dev_alloc(&p.pk_raw_key, B * npk))) return rc;
    if ((rc = dev_alloc(&p.pk_raw_score, B * npk))) return rc;
    if ((rc = dev_alloc(&p.pk_count, B * PMX_N_JOINTS))) return rc;
    if ((rc = dev_alloc(&p.pk_x, B * npk))) return rc;
    if ((rc = dev_alloc(&p.pk_y, B * npk))) return rc;
    if ((rc = dev_alloc(&p.pk_score, B * npk))) return rc;
    if ((rc = dev_alloc(&p.pk_start, B * (PMX_N_JOINTS + 1)))) return rc;
    if ((rc = dev_alloc(&p.cn_a, B * PMX_N_LIMBS * p.cap_pk))) return rc;
    if ((rc = dev_alloc(&p.cn_b, B * PMX_N_LIMBS * p.cap_pk))) return rc;
    if ((rc = dev_alloc(&p.cn_score, B * PMX_N_LIMBS * p.cap_pk))) return rc;
    if ((rc = dev_alloc(&p.cn_count, B * PMX_N_LIMBS))) return rc;
    if ((rc = dev_alloc(&p.cn_need, B * PMX_N_LIMBS))) return rc;
    p.scan_cap = p.cap_cand > PMX_LDS_CANDIDATES ? p.cap_cand : PMX_LDS_CANDIDATES;
    if ((rc = dev_alloc(&p.scan_score, B * PMX_N_LIMBS * p.scan_cap))) return rc;
    if ((rc = dev_alloc(&p.scan_idx, B * PMX_N_LIMBS * p.scan_cap))) return rc;
    if ((rc = dev_alloc(&p.scan_cnt, B * PMX_N_LIMBS))) return rc;
    if (p.cap_cand > 0) {
        if ((rc = dev_alloc(&p.cand_score, B * PMX_N_LIMBS * p.cap_cand))) return rc;
        if ((rc = dev_alloc(&p.cand_idx, B * PMX_N_LIMBS * p.cap_cand))) return rc;
        if ((rc = dev_alloc(&p.cand_used, B * PMX_N_LIMBS * 2 * p.cap_pk))) return rc;
    }
    if (p.cap_sub > PMX_LDS_SUBSETS && (rc = dev_alloc(&p.sub_work, B * p.cap_sub * 20))) return rc;
    if ((rc = dev_alloc(&p.subsets, B * p.cap_sub * 20))) return rc;
    if ((rc = dev_alloc(&p.status, B))) return rc;
    if ((rc = dev_alloc(&p.results, B * p.rec_bytes))) return rc;
    return PMX_OK;
}

extern "C" int pmx_create_net(pmx_ctx** out, const char* arch, int device, int max_batch, int max_h, int max_w);
extern "C" int pmx_create(pmx_ctx** out, int device, int max_batch, int max_h, int max_w)
{
    return pmx_create_net(out, "posenet", device, max_batch, max_h, max_w);
}

extern "C" int pmx_create_net(pmx_ctx** out, const char* arch, int device, int max_batch, int max_h, int max_w)
{
    PMX_CHECK(out && arch, PMX_ERR_INVALID, "pmx_create: null arg");
    *out = nullptr;
    int kind;
    if (!strcmp(arch, "posenet")) kind = NET_POSE;
    else if (!strcmp(arch, "facenet")) kind = NET_FACE;
    else if (!strcmp(arch, "handnet")) kind = NET_HAND;
    else { pmx_set_error("pmx_create_net: unknown arch '%s' (posenet | facenet | handnet)", arch); return PMX_ERR_INVALID; }
    PMX_CHECK(max_batch >= 1 && max_h >= 8 && max_w >= 8 && max_h % 8 == 0 && max_w % 8 == 0, PMX_ERR_INVALID,
              "pmx_create: max_batch >= 1 and max_h/max_w positive multiples of 8 required (got %d, %d, %d)", max_batch, max_h, max_w);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        pmx_set_error("pmx_create: no HIP device visible (this library has no CPU fallback)");
        return PMX_ERR_NO_DEVICE;
    }
    PMX_CHECK(device >= 0 && device < ndev, PMX_ERR_NO_DEVICE, "pmx_create: device %d out of range (%d devices)", device, ndev);
    hipDeviceProp_t prop;
    PMX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pmx_set_error("pmx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return PMX_ERR_NO_DEVICE;
    }
    conv_set_num_cus(prop.multiProcessorCount);      // kernel / tile selection counts blocks against the CUs of this device
    PMX_HIP(hipSetDevice(device));
    pmx_ctx* c = new pmx_ctx();
    c->device = device;
    c->max_batch = max_batch; c->max_h = max_h; c->max_w = max_w;
    c->kind = kind;
    c->n_heat = net_out_channels(kind);
    if (kind != NET_POSE) { c->cat_c = round_up(128 + c->n_heat, CK); c->cat_heat = 128; }
    c->table = kind == NET_POSE ? make_layer_table() : make_cpm_table(c->n_heat);
    c->layers.resize(c->table.size());
    for (size_t i = 0; i < c->table.size(); ++i) c->index[c->table[i].name] = (int)i;

    const size_t B = max_batch, HW = (size_t)max_h * max_w, hw8 = HW / 64;
    // any failure below frees what was created so far (the caller only ever sees a complete context or NULL)
    auto build = [&]() -> int {
        int rc;
        PMX_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        PMX_HIP(hipEventCreate(&c->t0));
        PMX_HIP(hipEventCreate(&c->t1));
        if ((rc = dev_alloc(&c->in16, B * HW * PMX_IN_C))) return rc;
        if ((rc = dev_alloc(&c->act0, B * HW * 64))) return rc;      // conv1_1 out is the largest activation
        if ((rc = dev_alloc(&c->act1, B * HW * 16))) return rc;      // (H/2)(W/2) x 64 = (H/4)(W/4) x 256
        if ((rc = dev_alloc(&c->cat, B * hw8 * c->cat_c))) return rc;
        if ((rc = dev_alloc(&c->brA, B * hw8 * 256))) return rc;
        if ((rc = dev_alloc(&c->brB, B * hw8 * 256))) return rc;
        if ((rc = dev_alloc(&c->brT, B * hw8 * 1024))) return rc;
        PMX_HIP(hipMemsetAsync(c->cat, 0, B * hw8 * c->cat_c * sizeof(float), c->stream));   // pad channels stay zero forever (stream-ordered)
        c->nchw_tmp_bytes = B * HW * 3 * sizeof(float);
        if (c->nchw_tmp_bytes < B * hw8 * 80 * sizeof(float)) c->nchw_tmp_bytes = B * hw8 * 80 * sizeof(float);
        PMX_HIP(hipMalloc((void**)&c->nchw_tmp, c->nchw_tmp_bytes));
        PMX_HIP(hipMalloc((void**)&c->u8_tmp, B * HW * 3));
        // post-process buffers at the initial capacities
        c->pp.cap_pk = PMX_INIT_PEAKS_PER_JOINT; c->pp.cap_sub = PMX_INIT_SUBSETS; c->pp.cap_ppl = PMX_INIT_PEOPLE; c->pp.cap_cand = 0;
        if ((rc = pp_alloc(c))) return rc;
        if ((rc = dev_alloc(&c->d_scale, B * 2))) return rc;
        c->pp.smoothed = nullptr;
        return PMX_OK;
    };
    if (int rc = build()) { pmx_destroy(c); return rc; }

    // default Gaussian taps (sigma 2.5 -> radius 10); the Python binding overrides them with NumPy's values
    {
        const int r = (int)(4.0 * PMX_GAUSS_SIGMA + 0.5);
        std::vector<double> g(2 * r + 1);
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = -r; i <= r; ++i) g[i + r] = exp(-0.5 / (PMX_GAUSS_SIGMA * PMX_GAUSS_SIGMA) * (double)(i * i));
        // NumPy pairwise-sum order for n = 21: 8 running accumulators over the first 16, then the tail
        for (int j = 0; j < 8; ++j) acc[j] = g[j];
        for (int j = 0; j < 8; ++j) acc[j] += g[8 + j];
        double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        for (int i = 16; i < 2 * r + 1; ++i) s += g[i];
        for (auto& v : g) v /= s;
        c->gauss = g;
    }
    *out = c;
    return PMX_OK;
}

extern "C" void pmx_destroy(pmx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& l : c->layers) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); if (l.d_w3) (void)hipFree(l.d_w3); if (l.d_ww) (void)hipFree(l.d_ww); }
    pp_free(c);
    void* ptrs[] = {c->sk_scratch, c->sk_zero_bias, c->pr_tmp, c->d_kp, c->u8_src, c->rs_tab, c->in16, c->act0, c->act1, c->cat, c->brA, c->brB, c->brT, c->nchw_tmp, c->u8_tmp, c->ext_paf, c->ext_heat,
                    c->pp.smoothed, c->d_scale, c->tab.xi0, c->tab.xi1, c->tab.xlo, c->tab.xhi, c->tab.yi0, c->tab.yi1,
                    c->tab.ylo, c->tab.yhi, c->tab.gauss};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& kv : c->pr_tabs) (void)hipFree(kv.second);
    for (auto& kv : c->tab_cache) (void)hipFree(kv.second.xi0);        // (one allocation per cached table set: pmx_multi.hip)
    void* mptrs[] = {c->d_segs, c->mi_src, c->mi_tab};
    for (void* p : mptrs) if (p) (void)hipFree(p);
    for (float* q : c->pr_part) if (q) (void)hipFree(q);
    for (int i = 1; i < PMX_PR_LANES; ++i) {
        PrLane& l = c->pr_lane[i];
        void* lp[] = {l.in16, l.act0, l.act1, l.cat, l.brA, l.brB, l.brT, l.u8_tmp, l.pr_tmp, l.sk_scratch};
        for (void* q : lp) if (q) (void)hipFree(q);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (int i = 0; i < PMX_PR_LANES; ++i) if (c->pr_lane[i].done) (void)hipEventDestroy(c->pr_lane[i].done);
    if (c->pr_src_ready) (void)hipEventDestroy(c->pr_src_ready);
    if (c->pr_fin) (void)hipEventDestroy(c->pr_fin);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_results) (void)hipHostFree(c->h_results);
    for (int i = 0; i < PMX_SNAPSHOT_SLOTS; ++i) {
        if (c->snap_ev[i]) (void)hipEventDestroy(c->snap_ev[i]);
        if (c->snap_status[i]) (void)hipHostFree(c->snap_status[i]);
    }
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int pmx_set_stream(pmx_ctx* c, void* s)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return PMX_OK;
}

extern "C" int pmx_synchronize(pmx_ctx* c)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_set_option(pmx_ctx* c, const char* key, int value)
{
    PMX_CHECK(c && key, PMX_ERR_INVALID, "null arg");
    if (!strcmp(key, "force_variant_k7")) c->opt_force[7] = value;
    else if (!strcmp(key, "force_variant_k3")) c->opt_force[3] = value;
    else if (!strcmp(key, "force_variant_k1")) c->opt_force[1] = value;
    else if (!strcmp(key, "keep_smoothed")) c->opt_keep_smoothed = value;
    else if (!strcmp(key, "stop_stage")) c->opt_stop_stage = value;
    else if (!strcmp(key, "kernel_gen")) c->opt_kernel_gen = value;
    else if (!strcmp(key, "fuse_pairs")) c->opt_fuse_pairs = value;
    else if (!strcmp(key, "fuse_conv1")) c->opt_fuse_conv1 = value;
    else if (!strcmp(key, "conv1_wino")) c->opt_conv1_wino = value;
    else if (!strcmp(key, "precise_lanes")) c->opt_precise_lanes = value < 1 ? 1 : (value > PMX_PR_LANES ? PMX_PR_LANES : value);
    else if (!strcmp(key, "precise_plain")) c->opt_precise_plain = value;
    else if (!strcmp(key, "precise_lane_priority")) c->opt_precise_lane_priority = value != 0;
    else if (!strcmp(key, "precise_table_cap")) c->opt_precise_table_cap = value < 1 ? 1 : value;
    else if (!strcmp(key, "cubic_rows")) prep_set_cubic_rows(value);      // (process-wide, like the other kernel-form switches of prep / post-process)
    else if (!strcmp(key, "precision")) {
        PMX_CHECK(value == 0 || conv_bf16x3_launch != nullptr, PMX_ERR_INVALID,
                  "pmx_set_option: \"precision\" = %d needs the opt-in bf16x3 kernels, which this build does not carry (rebuild with PMX_BUILD_BF16X3=1)", value);
        c->opt_precision = value;
    }
    else if (!strcmp(key, "conv_algo")) c->opt_conv_algo = value;
    else if (!strcmp(key, "wino_min_fill")) c->opt_wino_min_fill = value;
    else if (!strcmp(key, "wino_unit_eff")) c->opt_wino_unit_eff = value;
    else if (!strcmp(key, "wino_geom")) c->opt_wino_geom = value;
    else if (!strcmp(key, "wino_tail")) c->opt_wino_tail = value;
    else if (!strcmp(key, "wino_tail_g")) c->opt_wino_tail_g = value;
    else if (!strcmp(key, "wino_unit_g")) c->opt_wino_unit_g = value;
    else if (!strcmp(key, "wino_split")) c->opt_wino_split = value;
    else if (!strcmp(key, "wino_tail_merge")) c->opt_wino_tail_merge = value;
    else if (!strcmp(key, "ksplit")) c->opt_ksplit = value;
    else if (!strcmp(key, "ksplit_plan")) c->opt_ksplit = value > 0 ? -value : 0;     // decimal digits = chunks per slice, e.g. 3221
    else if (!strcmp(key, "conv_min_lds")) conv_set_min_lds(value);
    else if (!strcmp(key, "conv_v5_lds")) conv_set_v5_lds(value);
    else if (!strcmp(key, "pp_generic")) pp_set_generic(value);
    else if (!strcmp(key, "pp_limbs_slices")) c->opt_limbs_slices = value;
    else if (!strcmp(key, "peaks_gpu_branch")) { c->opt_gpu_branch_peaks = value; c->tab_in_h = -1; }
    else if (!strcmp(key, "kp_flip_x")) c->opt_kp_flip_x = value != 0;
    else { pmx_set_error("pmx_set_option: unknown key '%s'", key); return PMX_ERR_INVALID; }
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------ weights
extern "C" int pmx_set_layer(pmx_ctx* c, const char* name, const float* w, const float* bias, int cout, int cin, int ks)
{
    PMX_CHECK(c && name && w, PMX_ERR_INVALID, "pmx_set_layer: null arg");
    PMX_DEV(c);
    auto it = c->index.find(name);
    PMX_CHECK(it != c->index.end(), PMX_ERR_WEIGHTS, "pmx_set_layer: unknown layer '%s'", name);
    const LayerDesc& d = c->table[it->second];
    PMX_CHECK(d.cin == cin && d.cout == cout && d.ks == ks, PMX_ERR_WEIGHTS,
              "pmx_set_layer: '%s' expects (cout %d, cin %d, k %d), got (%d, %d, %d)", name, d.cout, d.cin, d.ks, cout, cin, ks);
    PackedLayer& L = c->layers[it->second];
    std::vector<int> cmap = (c->kind == NET_POSE && cin == 185) ? concat_map()
                            : (c->kind != NET_POSE && cin == c->n_heat + 128) ? concat_map_cpm(c->n_heat) : identity_map(cin);
    std::vector<float> wp, bp;
    const int cpad = cout_pad_of(cout);
    pack_weights(w, bias, cout, cin, ks, cmap, cpad, wp, bp);
    PMX_HIP(hipStreamSynchronize(c->stream));     // layer may be in use by queued work
    if (!L.d_w) PMX_HIP(hipMalloc((void**)&L.d_w, wp.size() * sizeof(float)));
    if (!L.d_b) PMX_HIP(hipMalloc((void**)&L.d_b, bp.size() * sizeof(float)));
    PMX_HIP(hipMemcpy(L.d_w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(L.d_b, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
    // the bf16x3 pack (1.5x the fp32 weights) and the Winograd pack (16/9 x for 3x3, 81/49 x for 7x7) are derived from the packed fp32
    // weights on first use (ensure_*_pack): a context that never runs those kernels neither holds nor computes them
    if (L.d_w3) { (void)hipFree(L.d_w3); L.d_w3 = nullptr; }
    if (L.d_ww) { (void)hipFree(L.d_ww); L.d_ww = nullptr; }
    L.set = true; L.cin = cin; L.cout = cout; L.ks = ks;
    L.cin_pad = (int)cmap.size(); L.cout_pad = cpad; L.nch = L.cin_pad / CK;
    return PMX_OK;
}

extern "C" int pmx_weights_missing(pmx_ctx* c, int* n)
{
    PMX_CHECK(c && n, PMX_ERR_INVALID, "null arg");
    int m = 0;
    for (auto& l : c->layers) m += l.set ? 0 : 1;
    *n = m;
    return PMX_OK;
}

// ------------------------------------------------------------------------------------------ forward
struct ConvIO { const float* in; int lda; float* out; int ldc; };

// derived weight packs, built from the device-resident packed fp32 weights when a kernel first needs them
static int fetch_packed(const PackedLayer& L, std::vector<float>& wp)
{
    wp.resize((size_t)L.ks * L.ks * L.nch * L.cout_pad * CK);
    PMX_HIP(hipMemcpy(wp.data(), L.d_w, wp.size() * sizeof(float), hipMemcpyDeviceToHost));
    return PMX_OK;
}
static int ensure_wino_pack(PackedLayer& L)
{
    if (L.d_ww) return PMX_OK;
    std::vector<float> wp, ww;
    if (int rc = fetch_packed(L, wp)) return rc;
    pack_wino(wp, L.ks, L.nch, L.cout_pad, ww);
    // (the pack becomes visible only once it is complete: a failed copy must not leave a non-null pointer to garbage behind)
    float* d = nullptr;
    PMX_HIP(hipMalloc((void**)&d, ww.size() * sizeof(float)));
    if (hipMemcpy(d, ww.data(), ww.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        pmx_set_error("winograd weight pack: host-to-device copy failed: %s", hipGetErrorString(hipGetLastError()));
        return PMX_ERR_HIP;
    }
    L.d_ww = d;
    return PMX_OK;
}
// conv1_1's weights in the order conv1_wino_kernel's lanes want them: [channel half 2][k-pair 14][k of the pair 2][channel 32] (the 28th k is
// zero) -- one contiguous 256-byte run per wave and k-pair.  Read out of the direct pack [tap][1 chunk][64][16] each lane gathered 14 dwords
// 64 bytes apart: 1.7 us of a 25 us block (profiles/r05_conv1_wino_ablation.json).  Lives in the layer's otherwise unused Winograd slot.
static int ensure_conv1_pack(PackedLayer& L)
{
    if (L.d_ww) return PMX_OK;
    std::vector<float> wp;
    if (int rc = fetch_packed(L, wp)) return rc;
    std::vector<float> w1(2 * 14 * 2 * 32, 0.f);
    for (int hf = 0; hf < 2; ++hf)
        for (int sp = 0; sp < 14; ++sp)
            for (int kk = 0; kk < 2; ++kk)
                for (int n = 0; n < 32; ++n) {
                    const int k = 2 * sp + kk;
                    if (k < 27) w1[((hf * 14 + sp) * 2 + kk) * 32 + n] = wp[((size_t)(k / 3) * L.cout_pad + hf * 32 + n) * CK + k % 3];
                }
    float* d = nullptr;
    PMX_HIP(hipMalloc((void**)&d, w1.size() * sizeof(float)));
    if (hipMemcpy(d, w1.data(), w1.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        pmx_set_error("conv1_1 weight pack: host-to-device copy failed: %s", hipGetErrorString(hipGetLastError()));
        return PMX_ERR_HIP;
    }
    L.d_ww = d;
    return PMX_OK;
}
static int ensure_bf16x3_pack(PackedLayer& L)
{
    if (L.d_w3) return PMX_OK;
    std::vector<float> wp;
    std::vector<uint16_t> w3;
    if (int rc = fetch_packed(L, wp)) return rc;
    pack_bf16x3(wp, L.ks * L.ks, L.nch, L.cout_pad, w3);
    void* d = nullptr;
    PMX_HIP(hipMalloc(&d, w3.size() * sizeof(uint16_t)));
    if (hipMemcpy(d, w3.data(), w3.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        pmx_set_error("bf16x3 weight pack: host-to-device copy failed: %s", hipGetErrorString(hipGetLastError()));
        return PMX_ERR_HIP;
    }
    L.d_w3 = d;
    return PMX_OK;
}

// Launches one convolution (1 or 2 groups) whose ConvArgs describe the FINAL result (real bias, ReLU, pool, output slices).
// With S > 1 K slices the slice blocks write raw partial sums into the context's slab scratch and conv_splitk_reduce
// produces the final result (slabs added in slice order, then bias, ReLU, pool).
static const int SK_ZERO_BIAS = PMX_SK_ZERO_BIAS;
static int launch_conv(pmx_ctx* c, const ConvArgs& a0, int groups, int v, const SplitPlan& plan)
{
    const int S = plan.S;
    if (S <= 1) {
        ConvArgs a = a0;
        a.ksplit = 1; a.slab_stride = 0;
        return conv_launch(v, a, groups, c->stream);
    }
    PMX_CHECK(a0.cout_pad <= SK_ZERO_BIAS, PMX_ERR_INVALID, "split-K: cout_pad %d too large", a0.cout_pad);
    const size_t slab = (size_t)a0.B * a0.H * a0.W * a0.cout_pad;
    const size_t need = slab * S * groups;
    if (need > c->sk_floats) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->sk_scratch) (void)hipFree(c->sk_scratch);
        c->sk_scratch = nullptr; c->sk_floats = 0;
        PMX_HIP(hipMalloc((void**)&c->sk_scratch, need * sizeof(float)));
        c->sk_floats = need;
    }
    if (!c->sk_zero_bias) {
        PMX_HIP(hipMalloc((void**)&c->sk_zero_bias, SK_ZERO_BIAS * sizeof(float)));
        // stream-ordered: a null-stream hipMemset is not ordered against this (non-blocking) stream and may still be in flight
        // when the first slice kernel reads the vector
        PMX_HIP(hipMemsetAsync(c->sk_zero_bias, 0, SK_ZERO_BIAS * sizeof(float), c->stream));
    }
    ConvArgs a = a0;
    SplitKReduceArgs r;
    memset(&r, 0, sizeof r);
    for (int g = 0; g < groups; ++g) {
        float* base = c->sk_scratch + (size_t)g * S * slab;
        r.slabs[g] = base; r.bias[g] = a0.g[g].bias; r.out[g] = a0.g[g].out; r.cout[g] = a0.g[g].cout;
        a.g[g].out = base; a.g[g].bias = c->sk_zero_bias; a.g[g].cout = a0.cout_pad;
    }
    a.ldc = a0.cout_pad; a.relu = 0; a.pool = 0; a.ksplit = S; a.slab_stride = (long long)slab; a.kbounds = plan.bounds;
    r.slab_stride = (long long)slab; r.ksplit = S; r.B = a0.B; r.H = a0.H; r.W = a0.W; r.ld_slab = a0.cout_pad; r.ldc = a0.ldc;
    r.relu = a0.relu; r.pool = a0.pool;
    int rc = conv_launch(v, a, groups, c->stream);
    if (rc) return rc;
    return conv_splitk_reduce(r, groups, c->stream);
}

// Unit mode of the Winograd kernel (single images): a (tile, group) is cut into S units -- pass 1 over g chunks each (ceil(nch / g) units)
// and, for 7x7, row 6, column 6 and tap (6, 6) -- that run as separate blocks writing slabs, combined in unit order by the split-K combine kernel
// (bias, ReLU, pool there)
static int launch_wino_units(pmx_ctx* c, const ConvArgs& a0, int ks, int groups, int g)
{
    const int S = (a0.nch + g - 1) / g + (ks == 7 ? 3 : 0);
    PMX_CHECK(S >= 2 && S <= 8, PMX_ERR_INVALID, "winograd units: %d slabs", S);
    PMX_CHECK(a0.cout_pad <= SK_ZERO_BIAS, PMX_ERR_INVALID, "split-K: cout_pad %d too large", a0.cout_pad);
    const size_t slab = (size_t)a0.B * a0.H * a0.W * a0.cout_pad;
    const size_t need = slab * S * groups;
    if (need > c->sk_floats) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->sk_scratch) (void)hipFree(c->sk_scratch);
        c->sk_scratch = nullptr; c->sk_floats = 0;
        PMX_HIP(hipMalloc((void**)&c->sk_scratch, need * sizeof(float)));
        c->sk_floats = need;
    }
    if (!c->sk_zero_bias) {
        PMX_HIP(hipMalloc((void**)&c->sk_zero_bias, SK_ZERO_BIAS * sizeof(float)));
        PMX_HIP(hipMemsetAsync(c->sk_zero_bias, 0, SK_ZERO_BIAS * sizeof(float), c->stream));
    }
    ConvArgs a = a0;
    SplitKReduceArgs r;
    memset(&r, 0, sizeof r);
    for (int gi = 0; gi < groups; ++gi) {
        float* base = c->sk_scratch + (size_t)gi * S * slab;
        r.slabs[gi] = base; r.bias[gi] = a0.g[gi].bias; r.out[gi] = a0.g[gi].out; r.cout[gi] = a0.g[gi].cout;
        a.g[gi].out = base; a.g[gi].bias = c->sk_zero_bias; a.g[gi].cout = a0.cout_pad;
    }
    a.ldc = a0.cout_pad; a.relu = 0; a.pool = 0; a.ksplit = S; a.slab_stride = (long long)slab; a.kbounds = (unsigned long long)g;
    r.slab_stride = (long long)slab; r.ksplit = S; r.B = a0.B; r.H = a0.H; r.W = a0.W; r.ld_slab = a0.cout_pad; r.ldc = a0.ldc;
    r.relu = a0.relu; r.pool = a0.pool;
    int rc = conv_wino_launch(a, ks, groups, c->stream);
    if (rc) return rc;
    return conv_splitk_reduce(r, groups, c->stream);
}

// Run geometry of the Winograd kernel (46-pixel-wide maps): the full blocks [0, nfull) of every image as one plain launch; the part-filled
// last block of every image in unit mode (S units of g pass-1 chunks (+ row 6, column 6, tap (6, 6)) writing compact slabs) + the combine
// kernel.  B = 32 at 46 x 46: 16 x 32 x 2 = 1024 full blocks = exactly 4 rounds of the 256 CUs, then 64 x 7 short unit blocks, instead of
// 5 rounds of 18 x 32 x 2 rectangles.  tail_g = 0: every block (also the part-filled one) in the plain launch.
struct WinoProf { std::string name; double flops, bytes, issued; };      // profile entry of the layer (null: not profiled)
// the tails of all images of the launch as one stream of tiles, 32 per block (conv_wino_kernel<KS, 0, 1, 3>)?
static bool wino_tail_merged(const pmx_ctx* c, const ConvArgs& a)
{
    return c->opt_wino_tail_merge != 0 && wino_tail_mergeable(a.B, a.H, a.W, a.lda);
}
static int launch_wino_run(pmx_ctx* c, const ConvArgs& a0, int ks, int groups, int tail_g, const WinoProf* pf = nullptr)
{
    const int ntiles = PMX_WINO_RUN_TX * ((a0.H + 1) / 2), nblk = (ntiles + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES, nfull = ntiles / PMX_WINO_RUN_TILES;
    ConvArgs a = a0;
    a.ksplit = 1; a.slab_stride = 0; a.run_j0 = 0; a.run_nb = tail_g ? nfull : nblk;
    // profile: the three launches of a layer with a unit-mode tail are separate entries -- "<layer>|<kernel>" (the full blocks; carries
    // their share of the FLOP), "...:units", "...:combine" -- so that each can be held against its own rocprofv3 kernel row
    const double share = tail_g ? (double)nfull * PMX_WINO_RUN_TILES / ntiles : 1.0;
    int rc;
    if (pf && c->prof_on == 2) {
        // inside timed regions: the dispatch stamps the two events itself (no hipEventRecord barrier packets around the launch)
        if ((rc = prof_begin(c, pf->name, pf->flops * share, pf->bytes, pf->issued * share, /*record=*/false))) return rc;
        conv_set_launch_events(c->pending.back().e0, c->pending.back().e1);
        c->prof_open = -1;
        rc = conv_wino_run_launch(a, ks, groups, c->stream);
        conv_set_launch_events(nullptr, nullptr);
        if (rc) {       // nothing was dispatched, so nothing will ever stamp the pair: take it back (prof_collect would fail on it for good)
            c->ev_pool.push_back(c->pending.back().e0);
            c->ev_pool.push_back(c->pending.back().e1);
            c->pending.pop_back();
        }
    } else {
        if (pf && (rc = prof_begin(c, pf->name, pf->flops * share, pf->bytes, pf->issued * share))) return rc;
        rc = conv_wino_run_launch(a, ks, groups, c->stream);
        if (pf && !rc) rc = prof_end(c);
    }
    if (rc || !tail_g) return rc;
    PMX_CHECK(nfull >= 1 && nfull < nblk, PMX_ERR_INVALID, "winograd tail: no part-filled block (%d tiles)", ntiles);
    const int S = (a0.nch + tail_g - 1) / tail_g + (ks == 7 ? 3 : 0);
    PMX_CHECK(S >= 2 && S <= 8, PMX_ERR_INVALID, "winograd tail: %d slabs", S);
    PMX_CHECK(a0.cout_pad <= SK_ZERO_BIAS, PMX_ERR_INVALID, "split-K: cout_pad %d too large", a0.cout_pad);
    const int nslab = a0.W / (2 * PMX_WINO_RUN_TX);
    // merged: the tails of all images as one stream of tiles, 32 per block (46 x 46: 17 tiles per image -- every MFMA row a real tile)
    const bool merged = wino_tail_merged(c, a0);
    const size_t slab = merged ? (size_t)wino_tail_merged_blocks(a0.B, a0.H) * PMX_WINO_RUN_TILES * 4 * a0.cout_pad      // [block of the stream][tile][pixel][cout_pad]
                               : (size_t)a0.B * nslab * PMX_WINO_RUN_TILES * 4 * a0.cout_pad;      // one block per (image, slab): [image][slab][tile][pixel][cout_pad]
    const size_t need = slab * S * groups;
    if (need > c->sk_floats) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->sk_scratch) (void)hipFree(c->sk_scratch);
        c->sk_scratch = nullptr; c->sk_floats = 0;
        PMX_HIP(hipMalloc((void**)&c->sk_scratch, need * sizeof(float)));
        c->sk_floats = need;
    }
    if (!c->sk_zero_bias) {
        PMX_HIP(hipMalloc((void**)&c->sk_zero_bias, SK_ZERO_BIAS * sizeof(float)));
        PMX_HIP(hipMemsetAsync(c->sk_zero_bias, 0, SK_ZERO_BIAS * sizeof(float), c->stream));
    }
    WinoTailReduceArgs r;
    memset(&r, 0, sizeof r);
    for (int gi = 0; gi < groups; ++gi) {
        float* base = c->sk_scratch + (size_t)gi * S * slab;
        r.slabs[gi] = base; r.bias[gi] = a0.g[gi].bias; r.out[gi] = a0.g[gi].out; r.cout[gi] = a0.g[gi].cout;
        a.g[gi].out = base; a.g[gi].bias = c->sk_zero_bias; a.g[gi].cout = a0.cout_pad;
    }
    a.ldc = a0.cout_pad; a.relu = 0; a.pool = 0; a.ksplit = S; a.slab_stride = (long long)slab; a.kbounds = (unsigned long long)tail_g;
    a.run_j0 = nfull; a.run_nb = 1;
    r.slab_stride = (long long)slab; r.S = S; r.B = a0.B; r.H = a0.H; r.W = a0.W; r.ld_slab = a0.cout_pad; r.ldc = a0.ldc;
    r.relu = a0.relu; r.run_j0 = nfull; r.run_nb = 1; r.nslab = nslab; r.pool = a0.pool; r.merged = merged;
    // (profile mode 2 -- the dominant kernel only, inside timed regions -- leaves these two short launches without events: an event pair
    //  costs ~5 us of idle stream)
    const bool pf_all = pf && c->prof_on == 1;
    if (pf_all && (rc = prof_begin(c, pf->name + ":units", pf->flops * (1.0 - share), 0.0, pf->issued * (1.0 - share)))) return rc;
    if ((rc = merged ? conv_wino_merged_tail_launch(a, ks, groups, c->stream) : conv_wino_run_launch(a, ks, groups, c->stream))) return rc;
    if (pf_all && (rc = prof_end(c))) return rc;
    if (pf_all && (rc = prof_begin(c, pf->name + ":combine", 0.0, 0.0, 0.0))) return rc;
    if ((rc = conv_wino_tail_reduce(r, groups, c->stream))) return rc;
    return pf_all ? prof_end(c) : PMX_OK;
}

// Which form a 3x3 / 7x7 layer takes (conv_select.hip): 0 = direct kernels (+ split-K), 1 = the Winograd kernel (*run: in the run geometry,
// *tail_g > 0: its part-filled last blocks in unit mode), 2 = the Winograd kernel in unit mode (*unit_g = chunks per pass-1 unit)
static WinoSelectOpts wino_opts(const pmx_ctx* c, int ks, int groups, int lda)
{
    WinoSelectOpts o;
    o.conv_algo = c->opt_conv_algo; o.precision = c->opt_precision; o.forced_variant = c->opt_force[ks]; o.ksplit = c->opt_ksplit;
    o.wino_unit_eff = c->opt_wino_unit_eff; o.wino_min_fill = c->opt_wino_min_fill; o.wino_geom = c->opt_wino_geom; o.wino_tail = c->opt_wino_tail;
    o.wino_tail_g = c->opt_wino_tail_g; o.wino_tail_merge = c->opt_wino_tail_merge; o.groups = groups; o.lda = lda; o.wino_unit_g = c->opt_wino_unit_g;
    o.wino_split = c->opt_wino_split;
    return o;
}
static int wino_mode(const pmx_ctx* c, int ks, int cin_pad, int cout_pad, int cout, int ldc, int images, int H, int W, int pool, int* unit_g,
                     int* run, int* tail_g, int groups = 1, int lda = 0)
{
    return wino_select(wino_opts(c, ks, groups, lda), ks, cin_pad, cout_pad, cout, ldc, images, H, W, pool, unit_g, run, tail_g);
}

// one launch of 1 or 2 groups (same geometry); in/out pointers are already offset to the group's channels
// `level` (heterogeneous forward only, c->segs non-empty): the resolution level of the layer's input, 0 = network input .. 3 = 1/8; B, H, W
// then carry the image count and the LARGEST map of the level (launch checks), the geometry comes from the segment table of the level
static int run_conv(pmx_ctx* c, const char* label, int li0, int li1, const float* in0, const float* in1, int lda,
                    float* out0, float* out1, int ldc, int B, int H, int W, int relu, int pool, int level = -1)
{
    const PackedLayer& L0 = c->layers[li0];
    const int groups = li1 >= 0 ? 2 : 1;
    const bool seg = !c->segs.empty() && level >= 0;
    // a batch whose plain launch would end in a part-filled round of the CUs: the images of the whole rounds first, then the rest through
    // the selection of THEIR count (conv_select.hip::wino_split_images); each half is an ordinary run_conv on its images
    if (!seg && c->split_suffix.empty() && c->opt_wino_split && B >= 2 && L0.ks > 1 && wino_eligible(L0.ks, L0.cin_pad, L0.cout_pad) &&
        (groups == 1 || (wino_eligible(c->layers[li1].ks, c->layers[li1].cin_pad, c->layers[li1].cout_pad) && c->layers[li1].cout == L0.cout))) {
        const int n0 = wino_split_images(wino_opts(c, L0.ks, groups, lda), L0.ks, L0.cin_pad, L0.cout_pad, L0.cout, ldc, B, groups, H, W, pool);
        if (n0 > 0 && n0 < B) {
            const size_t pin = (size_t)n0 * H * W * lda, pout = (size_t)n0 * (pool ? H / 2 : H) * (pool ? W / 2 : W) * ldc;
            c->split_suffix = "@0+" + std::to_string(n0);
            int rc = run_conv(c, label, li0, li1, in0, in1, lda, out0, out1, ldc, n0, H, W, relu, pool, level);
            if (!rc) {
                c->split_suffix = "@" + std::to_string(n0) + "+" + std::to_string(B - n0);
                rc = run_conv(c, label, li0, li1, in0 + pin, in1 ? in1 + pin : nullptr, lda, out0 + pout, out1 ? out1 + pout : nullptr, ldc, B - n0, H, W, relu, pool, level);
            }
            c->split_suffix.clear();
            return rc;
        }
    }
    const double npix = seg ? (double)c->seg_pix[level] : (double)B * H * W;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.g[0].in = in0; a.g[0].w = L0.d_w; a.g[0].bias = L0.d_b; a.g[0].out = out0; a.g[0].cout = L0.cout;
    double flops = 2.0 * npix * (double)L0.cout * L0.cin * L0.ks * L0.ks;
    if (groups == 2) {
        const PackedLayer& L1 = c->layers[li1];
        a.g[1].in = in1; a.g[1].w = L1.d_w; a.g[1].bias = L1.d_b; a.g[1].out = out1; a.g[1].cout = L1.cout;
        flops += 2.0 * npix * (double)L1.cout * L1.cin * L1.ks * L1.ks;
    }
    a.B = B; a.H = H; a.W = W; a.lda = lda; a.ldc = ldc; a.nch = L0.nch; a.cout_pad = L0.cout_pad;
    a.relu = relu; a.pool = pool;
    int rc;
    if (seg) {
        // heterogeneous launch: the plain Winograd kernel on 8 x 16 rectangles over all segments (every 3x3 / 7x7 layer of the pose network
        // qualifies; its per-pixel arithmetic is that of a plain launch of each image alone -- oracle/conv_fma_ref::conv_wino)
        PMX_CHECK(wino_eligible(L0.ks, L0.cin_pad, L0.cout_pad) && c->opt_precision == 0 &&
                  (groups == 1 || (wino_eligible(c->layers[li1].ks, c->layers[li1].cin_pad, c->layers[li1].cout_pad) && c->layers[li1].cout == L0.cout)),
                  PMX_ERR_INVALID, "heterogeneous forward: layer %s has no Winograd form", label);
        if ((rc = ensure_wino_pack(c->layers[li0])) || (groups == 2 && (rc = ensure_wino_pack(c->layers[li1])))) return rc;
        a.nch = L0.cin_pad / 32;
        a.g[0].w = L0.d_ww;
        if (groups == 2) a.g[1].w = c->layers[li1].d_ww;
        const int t = PMX_SEG_RECT(level, pool);
        a.nseg = (int)c->segs.size(); a.segs = c->d_segs + (size_t)t * c->segs.size(); a.seg_tiles = c->seg_tiles[t];
        const bool prof_seg = c->prof_on == 1 || (c->prof_on == 2 && L0.ks == 7);
        if (prof_seg) {
            const double bytes = 4.0 * npix * ((double)L0.cin * groups + (double)L0.cout * groups / (pool ? 4 : 1));
            if ((rc = prof_begin(c, std::string(label) + (L0.ks == 7 ? "|conv_wino_f2x2_7x7/seg" : "|conv_wino_f2x2_3x3/seg"), flops, bytes,
                                 flops * (L0.ks == 7 ? 100.0 / 196.0 : 16.0 / 36.0)))) return rc;
        }
        if ((rc = conv_wino_launch(a, L0.ks, groups, c->stream))) return rc;
        return prof_seg ? prof_end(c) : PMX_OK;
    }
    int v = conv_pick_variant(L0.ks, L0.cout_pad, H, W, B * groups, c->opt_force[L0.ks], c->opt_kernel_gen, pool, groups == 1 ? L0.cin : 9999,
                              c->opt_precision == 1 && L0.ks > 1);
    if (c->opt_precision == 1 && L0.ks > 1 && conv_bf16x3_twin(v) >= 0) {
        if ((rc = ensure_bf16x3_pack(c->layers[li0])) || (groups == 2 && (rc = ensure_bf16x3_pack(c->layers[li1])))) return rc;
        v = conv_bf16x3_twin(v);
        a.g[0].w = (const float*)L0.d_w3;
        if (groups == 2) a.g[1].w = (const float*)c->layers[li1].d_w3;
    }
    const bool prof_this = c->prof_on == 1 || (c->prof_on == 2 && L0.ks == 7);
    const bool wino_ok = wino_eligible(L0.ks, L0.cin_pad, L0.cout_pad) &&
                         (groups == 1 || (wino_eligible(c->layers[li1].ks, c->layers[li1].cin_pad, c->layers[li1].cout_pad) && c->layers[li1].cout == L0.cout));
    int ug = 0, wrun = 0, wtail = 0;
    const int wmode = wino_ok ? wino_mode(c, L0.ks, L0.cin_pad, L0.cout_pad, L0.cout, ldc, B * groups, H, W, pool, &ug, &wrun, &wtail, groups, lda) : 0;
    const bool wino_plain = wmode == 1;
    if (wmode && ((rc = ensure_wino_pack(c->layers[li0])) || (groups == 2 && (rc = ensure_wino_pack(c->layers[li1]))))) return rc;
    if (wmode == 2) {
        a.nch = L0.cin_pad / 32;
        a.g[0].w = L0.d_ww;
        if (groups == 2) a.g[1].w = c->layers[li1].d_ww;
        if (prof_this) {
            const double bytes = 4.0 * B * H * W * ((double)L0.cin * groups + (double)L0.cout * groups);
            if ((rc = prof_begin(c, std::string(label) + (L0.ks == 7 ? "|conv_wino_f2x2_7x7/u" : "|conv_wino_f2x2_3x3/u") + std::to_string(ug), flops, bytes,
                                 flops * (L0.ks == 7 ? 100.0 / 196.0 : 16.0 / 36.0)))) return rc;
        }
        if ((rc = launch_wino_units(c, a, L0.ks, groups, ug))) return rc;
        return prof_this ? prof_end(c) : PMX_OK;
    }
    if (wino_plain) {
        a.nch = L0.cin_pad / 32;
        a.g[0].w = L0.d_ww;
        if (groups == 2) a.g[1].w = c->layers[li1].d_ww;
        if (prof_this) {
            const double bytes = 4.0 * B * H * W * ((double)L0.cin * groups + (double)L0.cout * groups / (pool ? 4 : 1));
            // "r": run geometry; "/t<g>": its part-filled last blocks in unit mode, g chunks per pass-1 unit (part of the arithmetic)
            std::string kn = L0.ks == 7 ? "|conv_wino_f2x2_7x7" : "|conv_wino_f2x2_3x3";
            if (wrun) kn += "r";
            if (wtail) kn += "/t" + std::to_string(wtail) + (wrun && wino_tail_merged(c, a) ? "m" : "");      // "m": merged tails
            // products per 2 x 2 output tile and channel pair: 3x3: 16 of 36; 7x7: 4 x 16 + 4 x 8 + 4 = 100 of 196
            const double issued = flops * (L0.ks == 7 ? 100.0 / 196.0 : 16.0 / 36.0);
            if (wrun) {
                const WinoProf pf{std::string(label) + kn, flops, bytes, issued};
                return launch_wino_run(c, a, L0.ks, groups, wtail, &pf);
            }
            if ((rc = prof_begin(c, std::string(label) + kn, flops, bytes, issued))) return rc;
        }
        if ((rc = wrun ? launch_wino_run(c, a, L0.ks, groups, wtail) : conv_wino_launch(a, L0.ks, groups, c->stream))) return rc;
        return prof_this ? prof_end(c) : PMX_OK;
    }
    SplitPlan plan = conv_pick_ksplit(v, H, W, B, groups, L0.cout_pad, L0.nch, pool, c->opt_ksplit);
    if (L0.cout % 4 != 0 || ldc % 4 != 0 || (groups == 2 && c->layers[li1].cout != L0.cout)) plan.S = 1;
    if (prof_this) {
        const double bytes = 4.0 * B * H * W * ((double)L0.cin * groups + (double)L0.cout * groups / (pool ? 4 : 1));
        std::string kn = conv_variant(v).name;
        if (plan.S > 1) {       // "/k<chunks of slice 0>-<slice 1>-...": the K slices (+ the combine kernel) are part of the launch
            kn += "/k";
            for (int i = 0; i < plan.S; ++i) kn += (i ? "-" : "") + std::to_string(plan.sizes[i]);
        }
        if ((rc = prof_begin(c, std::string(label) + "|" + kn, flops, bytes))) return rc;
    }
    if ((rc = launch_conv(c, a, groups, v, plan))) return rc;
    return prof_this ? prof_end(c) : PMX_OK;
}

// the two 1x1 layers that end a stage (A: 128 -> cmid + ReLU, B: cmid -> cout [+ ReLU]) as ONE launch when the shapes allow
// (else, or with option fuse_pairs = 0, as two run_conv launches through `mid`): bit-identical either way
static int run_pair(pmx_ctx* c, const char* labelA, const char* labelB, int a0, int a1, int b0, int b1, const float* in0,
                    const float* in1, int lda, float* mid0, float* mid1, int ldm, float* out0, float* out1, int ldc, int B, int H,
                    int W, int reluB, int level = -1)
{
    const PackedLayer& LA = c->layers[a0];
    const PackedLayer& LB = c->layers[b0];
    const int groups = a1 >= 0 ? 2 : 1;
    // (heterogeneous forward: a 1x1 layer only sees pixels -- the segments' maps are one run of seg_pix[level] pixels)
    const bool seg = !c->segs.empty() && level >= 0;
    const long long npix = seg ? c->seg_pix[level] : (long long)B * H * W;
    const bool ok = c->opt_fuse_pairs && c->opt_kernel_gen >= 6 && LA.ks == 1 && LB.ks == 1 && LA.cin_pad == 128 && LB.cin == LA.cout &&
                    conv_pair_supported(LA.cin, LA.cout, LB.cout_pad) && (groups == 1 || c->layers[b1].cout_pad == LB.cout_pad);
    int rc;
    PMX_CHECK(ok || !seg, PMX_ERR_INVALID, "heterogeneous forward: the 1x1 pair %s + %s has no fused form", labelA, labelB);
    if (!ok) {
        if ((rc = run_conv(c, labelA, a0, a1, in0, in1, lda, mid0, mid1, ldm, B, H, W, 1, 0))) return rc;
        return run_conv(c, labelB, b0, b1, mid0, mid1, ldm, out0, out1, ldc, B, H, W, reluB, 0);
    }
    PairArgs p;
    memset(&p, 0, sizeof p);
    double flops = 0;
    for (int g = 0; g < groups; ++g) {
        const PackedLayer& A = c->layers[g ? a1 : a0];
        const PackedLayer& Bl = c->layers[g ? b1 : b0];
        p.g[g].in = g ? in1 : in0; p.g[g].w1 = A.d_w; p.g[g].b1 = A.d_b; p.g[g].w2 = Bl.d_w; p.g[g].b2 = Bl.d_b;
        p.g[g].out = g ? out1 : out0; p.g[g].cout = Bl.cout;
        flops += 2.0 * (double)npix * ((double)A.cout * A.cin + (double)Bl.cout * Bl.cin);
    }
    p.npix = npix; p.lda = lda; p.ldc = ldc; p.cmid = LA.cout; p.cout_pad = LB.cout_pad; p.relu2 = reluB;
    if (c->prof_on == 1) {
        const std::string kn = "conv1x1_pair_c" + std::to_string(LA.cout) + "_n" + std::to_string(LB.cout_pad);
        if ((rc = prof_begin(c, std::string(labelA) + "+" + labelB + "|" + kn, flops, 4.0 * (double)npix * groups * (LA.cin + LB.cout)))) return rc;
    }
    if ((rc = conv_pair_launch(p, groups, c->stream))) return rc;
    return prof_end(c);
}

// which form conv1_1 -> conv1_2 (+ pool) takes: *fuse: one launch (direct conv1_2 on 8 x 16 x 64 tiles: large maps); returns true for
// conv1_2 as Winograd F(2x2, 3x3) on 16 x 16 squares (conv1_wino.hip): wherever the Winograd kernels are allowed (conv_algo >= 1) and the
// launch has at least one block per CU (smaller launches: the 8 x 16 direct tiles give twice the blocks); conv1_wino = 2: always
// conv1_1 + conv1_2 have the shapes conv1_wino_kernel is written for (3 -> 64 -> 64, fp32)?
static bool conv1_pairable(const pmx_ctx* c)
{
    const PackedLayer& L1 = c->layers[c->index.at("conv1_1")];
    const PackedLayer& L2 = c->layers[c->index.at("conv1_2")];
    return c->opt_precision == 0 && c->opt_force[3] < 0 && L1.cin == 3 && L1.cout == 64 && L2.cin == 64 && L2.cout == 64;
}
static bool conv1_form(const pmx_ctx* c, int B, int H, int W, bool* fuse_out)
{
    const PackedLayer& L1 = c->layers[c->index.at("conv1_1")];
    const PackedLayer& L2 = c->layers[c->index.at("conv1_2")];
    const int v2 = conv_pick_variant(3, L2.cout_pad, H, W, B, c->opt_force[3], c->opt_kernel_gen, 1, L2.cin, 0);
    const bool fuse = c->opt_fuse_conv1 && c->opt_kernel_gen >= 6 && c->opt_precision == 0 && c->opt_force[3] < 0 && L1.cin == 3 && L1.cout == 64 &&
                      L2.cin == 64 && L2.cout == 64 && !strcmp(conv_variant(v2).name, "conv3x3_v5_t8x16_n64");
    const bool pair_ok = c->opt_precision == 0 && c->opt_force[3] < 0 && L1.cin == 3 && L1.cout == 64 && L2.cin == 64 && L2.cout == 64 && H % 2 == 0 && W % 2 == 0;
    if (fuse_out) *fuse_out = fuse;
    return pair_ok && c->opt_conv1_wino && c->opt_conv_algo >= 1 &&
           (c->opt_conv1_wino == 2 || (fuse && (long long)B * ((H + 15) / 16) * ((W + 15) / 16) >= conv_num_cus()));
}

static int run_conv1(pmx_ctx* c, int B, int H, int W)
{
    const int i1 = c->index.at("conv1_1"), i2 = c->index.at("conv1_2");
    const PackedLayer& L1 = c->layers[i1];
    const PackedLayer& L2 = c->layers[i2];
    bool fuse = false;
    const bool seg = !c->segs.empty();
    const bool wino1 = seg ? conv1_pairable(c) : conv1_form(c, B, H, W, &fuse);
    const uint8_t* in_u8 = c->in_u8;
    c->in_u8 = nullptr;                                   // (valid for this forward only)
    PMX_CHECK(!in_u8 || wino1, PMX_ERR_STATE, "conv1: a uint8 input without the kernel that preprocesses it");
    PMX_CHECK(!seg || (wino1 && in_u8), PMX_ERR_INVALID, "heterogeneous forward: needs conv1 as conv1_wino_kernel on a uint8 input");
    int rc;
    if (!fuse && !wino1) {
        if ((rc = run_conv(c, "conv1_1", i1, -1, c->in16, nullptr, PMX_IN_C, c->act0, nullptr, 64, B, H, W, 1, 0))) return rc;
        return run_conv(c, "conv1_2", i2, -1, c->act0, nullptr, 64, c->act1, nullptr, 64, B, H, W, 1, 1);
    }
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.g[0].in = c->in16; a.g[0].w = L2.d_w; a.g[0].bias = L2.d_b; a.g[0].out = c->act1; a.g[0].cout = L2.cout;
    a.g[1].w = L1.d_w; a.g[1].bias = L1.d_b;
    a.B = B; a.H = H; a.W = W; a.lda = PMX_IN_C; a.ldc = 64; a.nch = L2.nch; a.cout_pad = L2.cout_pad; a.relu = 1; a.pool = 1;
    if (wino1) {
        if ((rc = ensure_wino_pack(c->layers[i2])) || (rc = ensure_conv1_pack(c->layers[i1]))) return rc;
        a.g[0].w = L2.d_ww;
        a.g[1].w = L1.d_ww;
        if (in_u8) {                                      // the kernel preprocesses the uint8 batch itself (pmx_forward_from_u8)
            a.g[1].in = reinterpret_cast<const float*>(in_u8);
            a.kbounds = (unsigned long long)__builtin_bit_cast(unsigned, c->in_div);
        }
        if (seg) { a.nseg = (int)c->segs.size(); a.segs = c->d_segs + (size_t)PMX_SEG_CONV1 * c->segs.size(); a.seg_tiles = c->seg_tiles[PMX_SEG_CONV1]; }
        if (c->prof_on == 1) {
            const double np0 = seg ? (double)c->seg_pix[0] : (double)B * H * W;
            const double f1 = 2.0 * np0 * 9.0 * (double)L1.cout * L1.cin, f2 = 2.0 * np0 * 9.0 * (double)L2.cout * L2.cin;
            if ((rc = prof_begin(c, "conv1_1+conv1_2|conv_wino1_f2x2_t16x16", f1 + f2, (in_u8 ? 3.0 : 4.0 * 3) * np0 + 4.0 * np0 * (64 / 4), f1 + f2 * 16.0 / 36.0))) return rc;
        }
        if ((rc = conv1_wino_launch(a, c->stream))) return rc;
        return prof_end(c);
    }
    if (c->prof_on == 1) {
        const double flops = 2.0 * B * H * W * 9.0 * ((double)L1.cout * L1.cin + (double)L2.cout * L2.cin);
        if ((rc = prof_begin(c, "conv1_1+conv1_2|conv1_fused_t8x16_n64", flops, 4.0 * B * H * W * (3 + 64 / 4)))) return rc;
    }
    if ((rc = conv1_fused_launch(a, c->stream))) return rc;
    return prof_end(c);
}

// FaceNet / HandNet forward (models/FaceNet.py:78-160): one branch, groups = 1 everywhere
static int forward_cpm(pmx_ctx* c, int B, int H, int W)
{
    auto id = [&](const char* n) { return c->index.at(n); };
    int rc;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int CC = c->cat_c;
    float* cat = c->cat;
    float* heat = cat + c->cat_heat;
#define RUN1(name, in, lda, out, ldc, h, w, relu, pool) \
    do { if ((rc = run_conv(c, name, id(name), -1, in, nullptr, lda, out, nullptr, ldc, B, h, w, relu, pool))) return rc; } while (0)
    if ((rc = run_conv1(c, B, H, W))) return rc;
    RUN1("conv2_1", c->act1, 64, c->act0, 128, H2, W2, 1, 0);
    RUN1("conv2_2", c->act0, 128, c->act1, 128, H2, W2, 1, 1);
    RUN1("conv3_1", c->act1, 128, c->act0, 256, H4, W4, 1, 0);
    RUN1("conv3_2", c->act0, 256, c->act1, 256, H4, W4, 1, 0);
    RUN1("conv3_3", c->act1, 256, c->act0, 256, H4, W4, 1, 0);
    RUN1("conv3_4", c->act0, 256, c->act1, 256, H4, W4, 1, 1);
    RUN1("conv4_1", c->act1, 256, c->act0, 512, H8, W8, 1, 0);
    RUN1("conv4_2", c->act0, 512, c->act1, 512, H8, W8, 1, 0);
    RUN1("conv4_3", c->act1, 512, c->act0, 512, H8, W8, 1, 0);
    RUN1("conv4_4", c->act0, 512, c->act1, 512, H8, W8, 1, 0);
    RUN1("conv5_1", c->act1, 512, c->act0, 512, H8, W8, 1, 0);
    RUN1("conv5_2", c->act0, 512, c->act1, 512, H8, W8, 1, 0);
    RUN1("conv5_3_CPM", c->act1, 512, cat, CC, H8, W8, 1, 0);               // feature_map -> cat[:, 0:128]
    // conv6_1_CPM (reads the 128 feature channels) -> conv6_2_CPM (stage-1 heat maps -> cat[:, 128:128+C])
    if ((rc = run_pair(c, "conv6_1_CPM", "conv6_2_CPM", id("conv6_1_CPM"), -1, id("conv6_2_CPM"), -1, cat, nullptr, CC, c->brT, nullptr, 512,
                       heat, nullptr, CC, B, H8, W8, 0))) return rc;
    char nm[48], nm2[48];
    for (int s = 2; s <= 6 && s <= c->opt_stop_stage; ++s) {
        for (int i = 1; i <= 5; ++i) {
            snprintf(nm, sizeof nm, "Mconv%d_stage%d", i, s);
            const float* in; float* out; int lda;
            if (i == 1) { in = cat; lda = CC; }
            else if (i % 2 == 0) { in = c->brA; lda = 128; }
            else { in = c->brB; lda = 128; }
            out = i % 2 == 1 ? c->brA : c->brB;
            RUN1(nm, in, lda, out, 128, H8, W8, 1, 0);
        }
        snprintf(nm, sizeof nm, "Mconv6_stage%d", s);
        snprintf(nm2, sizeof nm2, "Mconv7_stage%d", s);
        if ((rc = run_pair(c, nm, nm2, id(nm), -1, id(nm2), -1, c->brA, nullptr, 128, c->brB, nullptr, 128, heat, nullptr, CC, B, H8, W8, 0)))
            return rc;
    }
#undef RUN1
    c->maps_valid = true; c->maps_external = false;
    c->cur_B = B; c->cur_fh = H8; c->cur_fw = W8;
    c->cur_segs.clear();
    c->pp_valid = false;
    return PMX_OK;
}

int pmx_forward_from_in16(pmx_ctx* c, int B, int H, int W)
{
    PMX_CHECK(c->segs.empty() || c->kind == NET_POSE, PMX_ERR_INVALID, "heterogeneous forward: posenet only");
    if (c->kind != NET_POSE) return forward_cpm(c, B, H, W);
    auto id = [&](const char* n) { return c->index.at(n); };
    int rc;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
#define RUN(...) do { if ((rc = run_conv(c, __VA_ARGS__))) return rc; } while (0)
    // stem (CocoPoseNet.py:136-151)
    // (the last argument = the resolution level, read only by a heterogeneous forward: pmx_multi.hip)
    if ((rc = run_conv1(c, B, H, W))) return rc;
    RUN("conv2_1", id("conv2_1"), -1, c->act1, nullptr, 64, c->act0, nullptr, 128, B, H2, W2, 1, 0, 1);
    RUN("conv2_2", id("conv2_2"), -1, c->act0, nullptr, 128, c->act1, nullptr, 128, B, H2, W2, 1, 1, 1);
    RUN("conv3_1", id("conv3_1"), -1, c->act1, nullptr, 128, c->act0, nullptr, 256, B, H4, W4, 1, 0, 2);
    RUN("conv3_2", id("conv3_2"), -1, c->act0, nullptr, 256, c->act1, nullptr, 256, B, H4, W4, 1, 0, 2);
    RUN("conv3_3", id("conv3_3"), -1, c->act1, nullptr, 256, c->act0, nullptr, 256, B, H4, W4, 1, 0, 2);
    RUN("conv3_4", id("conv3_4"), -1, c->act0, nullptr, 256, c->act1, nullptr, 256, B, H4, W4, 1, 1, 2);
    RUN("conv4_1", id("conv4_1"), -1, c->act1, nullptr, 256, c->act0, nullptr, 512, B, H8, W8, 1, 0, 3);
    RUN("conv4_2", id("conv4_2"), -1, c->act0, nullptr, 512, c->act1, nullptr, 512, B, H8, W8, 1, 0, 3);
    RUN("conv4_3_CPM", id("conv4_3_CPM"), -1, c->act1, nullptr, 512, c->act0, nullptr, 256, B, H8, W8, 1, 0, 3);
    RUN("conv4_4_CPM", id("conv4_4_CPM"), -1, c->act0, nullptr, 256, c->cat + PMX_CAT_FEAT, nullptr, PMX_CAT_C, B, H8, W8, 1, 0, 3);
    // stage 1 (CocoPoseNet.py:154-165); L1 = PAF branch, L2 = heat-map branch
    float* cat = c->cat;
    RUN("conv5_1_CPM", id("conv5_1_CPM_L1"), id("conv5_1_CPM_L2"), cat, cat, PMX_CAT_C, c->brA, c->brA + 128, 256, B, H8, W8, 1, 0, 3);
    RUN("conv5_2_CPM", id("conv5_2_CPM_L1"), id("conv5_2_CPM_L2"), c->brA, c->brA + 128, 256, c->brB, c->brB + 128, 256, B, H8, W8, 1, 0, 3);
    RUN("conv5_3_CPM", id("conv5_3_CPM_L1"), id("conv5_3_CPM_L2"), c->brB, c->brB + 128, 256, c->brA, c->brA + 128, 256, B, H8, W8, 1, 0, 3);
    // conv5_4 (128 -> 512, ReLU) -> conv5_5 (512 -> 38 | 19): one launch
    if ((rc = run_pair(c, "conv5_4_CPM", "conv5_5_CPM", id("conv5_4_CPM_L1"), id("conv5_4_CPM_L2"), id("conv5_5_CPM_L1"), id("conv5_5_CPM_L2"),
                       c->brA, c->brA + 128, 256, c->brT, c->brT + 512, 1024, cat + PMX_CAT_PAF, cat + PMX_CAT_HEAT, PMX_CAT_C, B, H8, W8, 0, 3)))
        return rc;
    // stages 2-6 (CocoPoseNet.py:168-260)
    char n1[48], n2[48], m1[48], m2[48], lab[48], lab2[48];
    for (int s = 2; s <= 6 && s <= c->opt_stop_stage; ++s) {
        for (int i = 1; i <= 5; ++i) {
            snprintf(n1, sizeof n1, "Mconv%d_stage%d_L1", i, s);
            snprintf(n2, sizeof n2, "Mconv%d_stage%d_L2", i, s);
            snprintf(lab, sizeof lab, "Mconv%d_stage%d", i, s);
            const float *in0, *in1; float *o0, *o1; int lda;
            if (i == 1) { in0 = cat; in1 = cat; lda = PMX_CAT_C; }
            else if (i % 2 == 0) { in0 = c->brA; in1 = c->brA + 128; lda = 256; }
            else { in0 = c->brB; in1 = c->brB + 128; lda = 256; }
            if (i % 2 == 1) { o0 = c->brA; o1 = c->brA + 128; }
            else { o0 = c->brB; o1 = c->brB + 128; }
            RUN(lab, id(n1), id(n2), in0, in1, lda, o0, o1, 256, B, H8, W8, 1, 0, 3);
        }
        // Mconv6 (1x1 128 -> 128, ReLU; reads Mconv5's output in brA) -> Mconv7 (1x1 128 -> 38 | 19, into the cat slices): one launch
        snprintf(n1, sizeof n1, "Mconv6_stage%d_L1", s); snprintf(n2, sizeof n2, "Mconv6_stage%d_L2", s);
        snprintf(m1, sizeof m1, "Mconv7_stage%d_L1", s); snprintf(m2, sizeof m2, "Mconv7_stage%d_L2", s);
        snprintf(lab, sizeof lab, "Mconv6_stage%d", s); snprintf(lab2, sizeof lab2, "Mconv7_stage%d", s);
        if ((rc = run_pair(c, lab, lab2, id(n1), id(n2), id(m1), id(m2), c->brA, c->brA + 128, 256, c->brB, c->brB + 128, 256,
                           cat + PMX_CAT_PAF, cat + PMX_CAT_HEAT, PMX_CAT_C, B, H8, W8, 0, 3))) return rc;
    }
#undef RUN
    c->maps_valid = true; c->maps_external = false;
    c->cur_B = B; c->cur_fh = H8; c->cur_fw = W8;
    c->cur_segs = c->segs;           // (empty for a uniform batch)
    c->pp_valid = false;
    return PMX_OK;
}

int pmx_check_weights(pmx_ctx* c)
{
    int missing = 0;
    for (auto& l : c->layers) missing += l.set ? 0 : 1;
    PMX_CHECK(missing == 0, PMX_ERR_WEIGHTS, "forward: %d of %d layers have no weights", missing, (int)c->layers.size());
    return PMX_OK;
}
static int check_forward_args(pmx_ctx* c, const void* p, int B, int H, int W)
{
    PMX_CHECK(c && p, PMX_ERR_INVALID, "forward: null arg");
    PMX_CHECK(B >= 1 && B <= c->max_batch, PMX_ERR_CAPACITY, "forward: batch %d outside 1..%d", B, c->max_batch);
    PMX_CHECK(H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, PMX_ERR_INVALID, "forward: H, W must be multiples of 8 (got %d x %d)", H, W);
    PMX_CHECK((size_t)H * W <= (size_t)c->max_h * c->max_w, PMX_ERR_CAPACITY, "forward: %d x %d exceeds the context capacity %d x %d",
              H, W, c->max_h, c->max_w);
    int missing = 0;
    for (auto& l : c->layers) missing += l.set ? 0 : 1;
    PMX_CHECK(missing == 0, PMX_ERR_WEIGHTS, "forward: %d of %d layers have no weights", missing, (int)c->layers.size());
    return PMX_OK;
}

extern "C" int pmx_forward_u8(pmx_ctx* c, const uint8_t* img, int B, int H, int W, int on_device)
{
    int rc = check_forward_args(c, img, B, H, W);
    if (rc) return rc;
    PMX_DEV(c);
    const uint8_t* d = img;
    if (!on_device) {
        PMX_HIP(hipMemcpyAsync(c->u8_tmp, img, (size_t)B * H * W * 3, hipMemcpyHostToDevice, c->stream));
        d = c->u8_tmp;
    }
    return pmx_forward_from_u8(c, d, B, H, W, c->kind == NET_POSE ? 255.0f : 256.0f);
}

int pmx_forward_from_u8(pmx_ctx* c, const uint8_t* d, int B, int H, int W, float divisor)
{
    int rc;
    if (!c->segs.empty() || conv1_form(c, B, H, W, nullptr)) {                // conv1_wino_kernel reads the uint8 pixels and preprocesses them in its patch load
        c->in_u8 = d; c->in_div = divisor;
        rc = pmx_forward_from_in16(c, B, H, W);
        c->in_u8 = nullptr;
        return rc;
    }
    if (c->prof_on == 1 && (rc = prof_begin(c, "prep_u8|prep_u8", 0, (double)B * H * W * (3 + 64)))) return rc;
    if ((rc = launch_prep_u8(d, c->in16, B, H, W, divisor, c->stream))) return rc;
    if ((rc = prof_end(c))) return rc;
    return pmx_forward_from_in16(c, B, H, W);
}

extern "C" int pmx_forward_f32(pmx_ctx* c, const float* x, int B, int H, int W, int on_device)
{
    int rc = check_forward_args(c, x, B, H, W);
    if (rc) return rc;
    PMX_DEV(c);
    const float* d = x;
    if (!on_device) {
        PMX_HIP(hipMemcpyAsync(c->nchw_tmp, x, (size_t)B * H * W * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d = c->nchw_tmp;
    }
    if ((rc = launch_prep_f32(d, c->in16, B, H, W, c->stream))) return rc;
    return pmx_forward_from_in16(c, B, H, W);
}

// OpenCV INTER_LINEAR uint8 tables for one axis: [idx0 | idx1 | coef0 | coef1], each `dst` ints.
// fx = float((d + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double; s = floor(fx); fx -= s;
// s < 0 -> (0, fx = 0); s >= src - 1 -> (src - 1, fx = 0); coefficients cvRound((1 - fx) * 2048), cvRound(fx * 2048).
void pmx_make_resize_table(int dst, int src, int* tab);
static void make_resize_table(int dst, int src, int* tab) { pmx_make_resize_table(dst, src, tab); }
void pmx_make_resize_table(int dst, int src, int* tab)
{
    const double scale = 1.0 / ((double)dst / (double)src);
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f = f - (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        tab[d] = s;
        tab[dst + d] = s + 1 < src - 1 ? s + 1 : src - 1;
        tab[2 * dst + d] = (int)lrintf((1.0f - f) * 2048.0f);
        tab[3 * dst + d] = (int)lrintf(f * 2048.0f);
    }
}

// uint8 images of one original size -> cv2.resize(..., (w, h)) on the device -> network forward
extern "C" int pmx_forward_u8_resized(pmx_ctx* c, const uint8_t* img, int B, int src_h, int src_w, int h, int w, int on_device)
{
    if (src_h == h && src_w == w) return pmx_forward_u8(c, img, B, h, w, on_device);     // identity (pose_detector.py:493)
    int rc = check_forward_args(c, img, B, h, w);
    if (rc && rc != PMX_ERR_WEIGHTS) return rc;      // without weights the resize still runs (pmx_get_resized), then
                                                     // pmx_forward_u8 below reports PMX_ERR_WEIGHTS
    PMX_CHECK(src_h >= 1 && src_w >= 1, PMX_ERR_INVALID, "forward_u8_resized: bad source size");
    PMX_DEV(c);
    const size_t nsrc = (size_t)B * src_h * src_w * 3;
    const uint8_t* d = img;
    if (!on_device) {
        if (nsrc > c->u8_src_cap) {
            PMX_HIP(hipStreamSynchronize(c->stream));
            if (c->u8_src) (void)hipFree(c->u8_src);
            c->u8_src = nullptr;
            PMX_HIP(hipMalloc((void**)&c->u8_src, nsrc));
            c->u8_src_cap = nsrc;
        }
        PMX_HIP(hipMemcpyAsync(c->u8_src, img, nsrc, hipMemcpyHostToDevice, c->stream));
        d = c->u8_src;
    }
    const size_t ntab = (size_t)4 * (w + h);
    if (ntab > c->rs_tab_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->rs_tab) (void)hipFree(c->rs_tab);
        c->rs_tab = nullptr;
        PMX_HIP(hipMalloc((void**)&c->rs_tab, ntab * sizeof(int)));
        c->rs_tab_cap = ntab;
    }
    std::vector<int> tab(ntab);
    make_resize_table(w, src_w, tab.data());
    make_resize_table(h, src_h, tab.data() + 4 * w);
    PMX_HIP(hipStreamSynchronize(c->stream));     // the table buffer may still be in use by a queued resize
    PMX_HIP(hipMemcpy(c->rs_tab, tab.data(), ntab * sizeof(int), hipMemcpyHostToDevice));
    if (c->prof_on == 1 && (rc = prof_begin(c, "resize_u8|resize_linear_u8", 0, (double)nsrc + (double)B * h * w * 3))) return rc;
    if ((rc = launch_resize_linear_u8(d, c->u8_tmp, c->rs_tab, c->rs_tab + 4 * w, B, src_h, src_w, h, w, c->stream))) return rc;
    if ((rc = prof_end(c))) return rc;
    return pmx_forward_u8(c, c->u8_tmp, B, h, w, 1);
}

// test / parity accessor: the resized uint8 batch of the last pmx_forward_u8_resized (B x h x w x 3)
extern "C" int pmx_get_resized(pmx_ctx* c, uint8_t* out, int B, int h, int w)
{
    PMX_CHECK(c && out, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(B >= 1 && B <= c->max_batch && (size_t)h * w <= (size_t)c->max_h * c->max_w, PMX_ERR_CAPACITY, "pmx_get_resized: size");
    PMX_DEV(c);
    PMX_HIP(hipMemcpyAsync(out, c->u8_tmp, (size_t)B * h * w * 3, hipMemcpyDeviceToHost, c->stream));
    PMX_HIP(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_get_maps(pmx_ctx* c, float* paf, float* heat)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(c->maps_valid, PMX_ERR_STATE, "pmx_get_maps: no forward / set_maps yet");
    PMX_CHECK(c->cur_segs.empty(), PMX_ERR_STATE, "pmx_get_maps: the current maps are those of a mixed-size batch (use pmx_get_image_maps)");
    PMX_DEV(c);
    const int B = c->cur_B, fh = c->cur_fh, fw = c->cur_fw;
    PMX_CHECK(c->kind == NET_POSE || !paf, PMX_ERR_INVALID, "pmx_get_maps: facenet / handnet have no PAF output (pass NULL)");
    const size_t np = c->kind == NET_POSE ? (size_t)B * PMX_N_PAF * fh * fw : 0, nh = (size_t)B * c->n_heat * fh * fw;
    if (c->maps_external) {
        if (paf) PMX_HIP(hipMemcpyAsync(paf, c->ext_paf, np * 4, hipMemcpyDeviceToHost, c->stream));
        if (heat) PMX_HIP(hipMemcpyAsync(heat, c->ext_heat, nh * 4, hipMemcpyDeviceToHost, c->stream));
    } else {
        int rc;
        if (paf) {
            if ((rc = launch_nhwc_to_nchw(c->cat, c->nchw_tmp, B, PMX_N_PAF, fh, fw, PMX_CAT_C, PMX_CAT_PAF, c->stream))) return rc;
            PMX_HIP(hipMemcpyAsync(paf, c->nchw_tmp, np * 4, hipMemcpyDeviceToHost, c->stream));
        }
        if (heat) {
            float* tmp = c->nchw_tmp + np;
            if ((rc = launch_nhwc_to_nchw(c->cat, tmp, B, c->n_heat, fh, fw, c->cat_c, c->cat_heat, c->stream))) return rc;
            PMX_HIP(hipMemcpyAsync(heat, tmp, nh * 4, hipMemcpyDeviceToHost, c->stream));
        }
    }
    PMX_HIP(hipStreamSynchronize(c->stream));
    return PMX_OK;
}

extern "C" int pmx_set_maps(pmx_ctx* c, const float* paf, const float* heat, int B, int fh, int fw)
{
    PMX_CHECK(c && heat && (paf || c->kind != NET_POSE), PMX_ERR_INVALID, "pmx_set_maps: null arg");
    PMX_CHECK(B >= 1 && B <= c->max_batch, PMX_ERR_CAPACITY, "pmx_set_maps: batch %d outside 1..%d", B, c->max_batch);
    PMX_CHECK(fh >= 1 && fw >= 1, PMX_ERR_INVALID, "pmx_set_maps: bad size");
    PMX_DEV(c);
    const size_t need = (size_t)B * fh * fw;
    if (need > c->ext_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (c->ext_paf) (void)hipFree(c->ext_paf);
        if (c->ext_heat) (void)hipFree(c->ext_heat);
        c->ext_paf = c->ext_heat = nullptr;
        PMX_HIP(hipMalloc((void**)&c->ext_paf, need * PMX_N_PAF * 4));
        PMX_HIP(hipMalloc((void**)&c->ext_heat, need * c->n_heat * 4));
        c->ext_cap = need;
    }
    if (paf) PMX_HIP(hipMemcpyAsync(c->ext_paf, paf, need * PMX_N_PAF * 4, hipMemcpyHostToDevice, c->stream));
    PMX_HIP(hipMemcpyAsync(c->ext_heat, heat, need * c->n_heat * 4, hipMemcpyHostToDevice, c->stream));
    PMX_HIP(hipStreamSynchronize(c->stream));    // host buffers may be released by the caller
    c->maps_valid = true; c->maps_external = true;
    c->cur_B = B; c->cur_fh = fh; c->cur_fw = fw;
    c->cur_segs.clear();
    c->pp_valid = false;
    return PMX_OK;
}

// ------------------------------------------------------------------------------------- post-process
extern "C" int pmx_set_gaussian(pmx_ctx* c, const double* taps, int radius)
{
    PMX_CHECK(c && taps, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(radius >= 0 && radius <= PMX_GAUSS_MAX_RADIUS, PMX_ERR_INVALID, "pmx_set_gaussian: radius %d > %d", radius, PMX_GAUSS_MAX_RADIUS);
    c->gauss.assign(taps, taps + 2 * radius + 1);
    c->tab_in_h = -1;   // force table rebuild / upload
    if (!c->tab_cache.empty()) {      // the per-size table sets of mixed batches carry the old taps (a setup call: synchronising is fine)
        PMX_DEV(c);
        PMX_HIP(hipDeviceSynchronize());
        for (auto& kv : c->tab_cache) (void)hipFree(kv.second.xi0);
        c->tab_cache.clear();
    }
    return PMX_OK;
}

// np.linspace(0, in-1, num=out) grid + the corner indices / weights of Chainer's ResizeImages (see oracle)
static void make_grid(int in, int out, std::vector<int>& i0, std::vector<int>& i1, std::vector<double>& lo, std::vector<double>& hi)
{
    i0.resize(out); i1.resize(out); lo.resize(out); hi.resize(out);
    const double start = 0.0, stop = (double)(in - 1);
    const int div = out - 1;
    const double delta = stop - start;
    const double step = div > 0 ? delta / (double)div : 0.0;
    for (int k = 0; k < out; ++k) {
        double u;
        if (div > 0) {
            if (step == 0.0) u = ((double)k / (double)div) * delta + start;
            else u = (double)k * step + start;
            if (k == out - 1 && out > 1) u = stop;
        } else {
            u = start;     // num == 1 -> [start]
        }
        const int f = (int)floor(u);
        const double wl = (double)(f + 1) - u, wh = u - (double)f;
        i0[k] = f < 0 ? 0 : (f > in - 1 ? in - 1 : f);
        i1[k] = f + 1 > in - 1 ? in - 1 : (f + 1 < 0 ? 0 : f + 1);
        lo[k] = wl; hi[k] = wh;
    }
}

int pmx_ensure_tables(pmx_ctx* c, int in_h, int in_w, int out_h, int out_w, int flip_x)
{
    if (c->tab_in_h == in_h && c->tab_in_w == in_w && c->tab_out_h == out_h && c->tab_out_w == out_w && c->tab_flip == flip_x) return PMX_OK;
    const int cap = out_h > out_w ? out_h : out_w;
    PPTables& t = c->tab;
    if (cap > c->tab_cap) {
        PMX_HIP(hipStreamSynchronize(c->stream));
        void* olds[] = {t.xi0, t.xi1, t.xlo, t.xhi, t.yi0, t.yi1, t.ylo, t.yhi};
        for (void* p : olds) if (p) (void)hipFree(p);
        PMX_HIP(hipMalloc((void**)&t.xi0, cap * sizeof(int)));  PMX_HIP(hipMalloc((void**)&t.xi1, cap * sizeof(int)));
        PMX_HIP(hipMalloc((void**)&t.xlo, cap * sizeof(double))); PMX_HIP(hipMalloc((void**)&t.xhi, cap * sizeof(double)));
        PMX_HIP(hipMalloc((void**)&t.yi0, cap * sizeof(int)));  PMX_HIP(hipMalloc((void**)&t.yi1, cap * sizeof(int)));
        PMX_HIP(hipMalloc((void**)&t.ylo, cap * sizeof(double))); PMX_HIP(hipMalloc((void**)&t.yhi, cap * sizeof(double)));
        c->tab_cap = cap;
    }
    if (!t.gauss) PMX_HIP(hipMalloc((void**)&t.gauss, (2 * PMX_GAUSS_MAX_RADIUS + 1) * sizeof(double)));
    std::vector<int> i0, i1; std::vector<double> lo, hi;
    PMX_HIP(hipStreamSynchronize(c->stream));
    make_grid(in_w, out_w, i0, i1, lo, hi);
    if (flip_x) {       // column x of the mirrored map = column out_w - 1 - x of the resized one: same samples, same arithmetic
        std::reverse(i0.begin(), i0.end()); std::reverse(i1.begin(), i1.end());
        std::reverse(lo.begin(), lo.end()); std::reverse(hi.begin(), hi.end());
    }
    PMX_HIP(hipMemcpy(t.xi0, i0.data(), out_w * sizeof(int), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.xi1, i1.data(), out_w * sizeof(int), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.xlo, lo.data(), out_w * sizeof(double), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.xhi, hi.data(), out_w * sizeof(double), hipMemcpyHostToDevice));
    make_grid(in_h, out_h, i0, i1, lo, hi);
    PMX_HIP(hipMemcpy(t.yi0, i0.data(), out_h * sizeof(int), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.yi1, i1.data(), out_h * sizeof(int), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.ylo, lo.data(), out_h * sizeof(double), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(t.yhi, hi.data(), out_h * sizeof(double), hipMemcpyHostToDevice));
    if (c->opt_gpu_branch_peaks) {
        // create_gaussian_kernel(sigma, ksize = 17) (pose_detector.py:38-44): 1/(2 pi sigma^2) exp(-d^2 / 2 sigma^2), NOT
        // normalised to sum 1, applied as a 17x17 zero-padded convolution (:112-113); separable factor per axis
        const int r = 8;
        std::vector<double> g(2 * r + 1);
        const double s2 = PMX_GAUSS_SIGMA * PMX_GAUSS_SIGMA;
        for (int i = -r; i <= r; ++i) g[i + r] = sqrt(1.0 / (s2 * 2.0 * M_PI)) * exp(-0.5 * (double)(i * i) / s2);
        PMX_HIP(hipMemcpy(t.gauss, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
        t.radius = r; t.border_zero = 1; t.nms_ge = 1;
    } else {
        PMX_HIP(hipMemcpy(t.gauss, c->gauss.data(), c->gauss.size() * sizeof(double), hipMemcpyHostToDevice));
        t.radius = ((int)c->gauss.size() - 1) / 2; t.border_zero = 0; t.nms_ge = 0;
    }
    c->tab_in_h = in_h; c->tab_in_w = in_w; c->tab_out_h = out_h; c->tab_out_w = out_w; c->tab_flip = flip_x;
    return PMX_OK;
}

extern "C" int pmx_postprocess(pmx_ctx* c, int B, int map_h, int map_w, double img_len, const double* scale_xy)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(c->kind == NET_POSE, PMX_ERR_STATE, "pmx_postprocess: posenet only (use pmx_keypoints for facenet / handnet)");
    PMX_CHECK(c->maps_valid, PMX_ERR_STATE, "pmx_postprocess: no network output (call forward or set_maps first)");
    PMX_CHECK(B == c->cur_B, PMX_ERR_INVALID, "pmx_postprocess: batch %d != batch of the current maps %d", B, c->cur_B);
    PMX_CHECK(c->cur_segs.empty(), PMX_ERR_STATE, "pmx_postprocess: the current maps are those of a mixed-size batch (use pmx_postprocess_images)");
    PMX_CHECK(map_h >= 1 && map_w >= 1 && (long long)map_h * map_w < (1ll << 31), PMX_ERR_INVALID, "pmx_postprocess: bad map size");
    PMX_DEV(c);
    int rc;
    if ((rc = pmx_ensure_tables(c, c->cur_fh, c->cur_fw, map_h, map_w))) return rc;
    const long long fhw = (long long)c->cur_fh * c->cur_fw;
    PPMaps m;
    if (c->maps_external) {          // NCHW copies installed by pmx_set_maps
        m.heat = c->ext_heat; m.paf = c->ext_paf;
        m.sx = 1; m.sy = c->cur_fw; m.sc = fhw;
        m.sbh = PMX_N_HEAT * fhw; m.sbp = PMX_N_PAF * fhw;
    } else {                         // channel slices of the NHWC cat buffer written by the last stage
        m.heat = c->cat + PMX_CAT_HEAT; m.paf = c->cat + PMX_CAT_PAF;
        m.sc = 1; m.sx = PMX_CAT_C; m.sy = (long long)c->cur_fw * PMX_CAT_C;
        m.sbh = m.sbp = fhw * PMX_CAT_C;
    }
    m.fh = c->cur_fh; m.fw = c->cur_fw;
    if (c->opt_keep_smoothed) {
        const size_t need = (size_t)B * PMX_N_JOINTS * map_h * map_w;
        if (need > c->smoothed_cap) {
            PMX_HIP(hipStreamSynchronize(c->stream));
            if (c->pp.smoothed) (void)hipFree(c->pp.smoothed);
            c->pp.smoothed = nullptr;
            PMX_HIP(hipMalloc((void**)&c->pp.smoothed, need * sizeof(float)));
            c->smoothed_cap = need;
        }
    }
    const double* dscale = nullptr;
    if (scale_xy) {
        PMX_HIP(hipMemcpyAsync(c->d_scale, scale_xy, sizeof(double) * 2 * B, hipMemcpyHostToDevice, c->stream));
        dscale = c->d_scale;
    }
    // the candidate scan of a limb over several blocks where the maps come at full resolution (detect_precise, pmx_set_maps: crowds of
    // peaks, 19 blocks per image otherwise); the batch path's low-resolution maps keep the one-block form
    c->pp_limbs_slices = c->opt_limbs_slices >= 0 ? c->opt_limbs_slices : (c->maps_external ? 8 : 0);
    rc = pp_launch(m, c->tab, c->pp, B, map_h, map_w, img_len, dscale, c->opt_keep_smoothed && c->pp.smoothed, c->stream,
                   c->prof_on == 1 ? pp_prof_cb : nullptr, c, c->pp_limbs_slices);
    if (rc) return rc;
    c->pp_valid = true; c->pp_final = false; c->pp_B = B; c->pp_h = map_h; c->pp_w = map_w;
    c->pp_maps = m; c->pp_img_len = img_len; c->pp_has_scale = scale_xy != nullptr;
    c->pp_calls.clear();
    return PMX_OK;
}

// the post-process buffers as the images [base, ...) see them: every per-image array moved on by `base` images (strides: pp_alloc)
PPBuffers pmx_pp_view(const PPBuffers& p, int base)
{
    PPBuffers v = p;
    const size_t b = (size_t)base, npk = (size_t)PMX_N_JOINTS * p.cap_pk, nl = (size_t)PMX_N_LIMBS;
    v.pk_raw_key += b * npk; v.pk_raw_score += b * npk; v.pk_count += b * PMX_N_JOINTS;
    v.pk_x += b * npk; v.pk_y += b * npk; v.pk_score += b * npk; v.pk_start += b * (PMX_N_JOINTS + 1);
    v.cn_a += b * nl * p.cap_pk; v.cn_b += b * nl * p.cap_pk; v.cn_score += b * nl * p.cap_pk;
    v.cn_count += b * nl; v.cn_need += b * nl;
    v.scan_score += b * nl * p.scan_cap; v.scan_idx += b * nl * p.scan_cap; v.scan_cnt += b * nl;
    if (p.cand_score) { v.cand_score += b * nl * p.cap_cand; v.cand_idx += b * nl * p.cap_cand; v.cand_used += b * nl * 2 * p.cap_pk; }
    if (p.sub_work) v.sub_work += b * p.cap_sub * 20;
    v.subsets += b * p.cap_sub * 20; v.status += b; v.results += b * p.rec_bytes;
    v.smoothed = nullptr;            // (sized for ONE map size: not available per segment)
    return v;
}

// ---- capacities: the reference has none (np.vstack / lists); ours grow on demand ------------------------------------------
static const int PMX_IMG_CAPACITY_BITS = PMX_IMG_PEAK_OVERFLOW | PMX_IMG_CAND_OVERFLOW | PMX_IMG_SUBSET_OVERFLOW | PMX_IMG_PEOPLE_OVERFLOW;
static int next_pow2(long long v)
{
    long long p = 1;
    while (p < v) p <<= 1;
    return (int)(p > (1ll << 30) ? (1ll << 30) : p);
}

// Synchronise and look at the per-image status words of the last post-process.  If an image overflowed a capacity, grow it
// (peaks: to the largest per-joint count seen; candidates: device-memory store sized to the largest accepted count; subsets:
// doubled), reallocate the post-process buffers and run the post-process of the batch again on the same network output --
// until every image fits.  After this the records on the device are final.
// Re-size the post-process buffers: the new set is allocated FIRST and swapped in only when every allocation succeeded -- on failure (out
// of memory on a crowd image, an absurd user capacity) the context keeps its old buffers and capacities and stays usable
static int pp_realloc(pmx_ctx* c, int cap_pk, int cap_sub, int cap_cand, int cap_ppl)
{
    const PPBuffers old = c->pp;
    PPBuffers fresh{};
    fresh.cap_pk = cap_pk; fresh.cap_sub = cap_sub; fresh.cap_cand = cap_cand; fresh.cap_ppl = cap_ppl;
    fresh.smoothed = old.smoothed;               // (owned by the context, sized separately)
    c->pp = fresh;
    const int rc = pp_alloc(c);
    if (rc) {
        pp_free(c);                              // whatever part of the new set exists
        c->pp = old;
        return rc;
    }
    const PPBuffers neu = c->pp;
    c->pp = old;
    pp_free(c);
    c->pp = neu;
    return PMX_OK;
}

static int pp_finalize(pmx_ctx* c)
{
    if (!c->pp_valid || c->pp_final) return PMX_OK;
    for (int round = 0; round < 64; ++round) {
        const int B = c->pp_B;
        std::vector<int> status(B);
        PMX_HIP(hipMemcpyAsync(status.data(), c->pp.status, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
        PMX_HIP(hipStreamSynchronize(c->stream));
        int bits = 0;
        for (int v : status) bits |= v;
        if (!(bits & PMX_IMG_CAPACITY_BITS)) { c->pp_final = true; return PMX_OK; }
        int cap_pk = c->pp.cap_pk, cap_sub = c->pp.cap_sub, cap_cand = c->pp.cap_cand, cap_ppl = c->pp.cap_ppl;
        if (bits & PMX_IMG_PEAK_OVERFLOW) {
            std::vector<int> cnt((size_t)B * PMX_N_JOINTS);
            PMX_HIP(hipMemcpy(cnt.data(), c->pp.pk_count, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
            int need = 0;
            for (int v : cnt) need = v > need ? v : need;
            cap_pk = next_pow2(need);
        }
        if (bits & PMX_IMG_CAND_OVERFLOW) {
            std::vector<int> need_v((size_t)B * PMX_N_LIMBS);
            PMX_HIP(hipMemcpy(need_v.data(), c->pp.cn_need, need_v.size() * sizeof(int), hipMemcpyDeviceToHost));
            int need = PMX_LDS_CANDIDATES;
            for (int v : need_v) need = v > need ? v : need;
            cap_cand = next_pow2(need);
        }
        if (bits & PMX_IMG_SUBSET_OVERFLOW) cap_sub *= 2;
        if (bits & PMX_IMG_PEOPLE_OVERFLOW) {
            int need = 0;
            for (int b = 0; b < B; ++b) {
                pmx_image_info info;
                PMX_HIP(hipMemcpy(&info, c->pp.results + (size_t)b * c->pp.rec_bytes, sizeof info, hipMemcpyDeviceToHost));
                need = info.n_people > need ? info.n_people : need;
            }
            cap_ppl = next_pow2(need);
        }
        // invariant pmx_set_capacities enforces (people <= subsets): keep it through growth, so that a grown state can be replayed
        // through pmx_set_capacities (Engine.state() / load_state() when a PoseDetector re-creates its context)
        if (cap_sub < cap_ppl) cap_sub = cap_ppl;
        PMX_CHECK(cap_pk != c->pp.cap_pk || cap_sub != c->pp.cap_sub || cap_cand != c->pp.cap_cand || cap_ppl != c->pp.cap_ppl,
                  PMX_ERR_STATE, "post-process reports a capacity overflow (0x%x) that growing does not resolve", bits);
        int rc = pp_realloc(c, cap_pk, cap_sub, cap_cand, cap_ppl);
        if (rc) { c->pp_valid = false; return rc; }      // (old buffers and capacities stay in place; this batch has no results)
        c->pp_regrown += 1;
        if (c->pp_calls.empty()) {
            rc = pp_launch(c->pp_maps, c->tab, c->pp, B, c->pp_h, c->pp_w, c->pp_img_len, c->pp_has_scale ? c->d_scale : nullptr,
                           c->opt_keep_smoothed && c->pp.smoothed, c->stream, nullptr, nullptr, c->pp_limbs_slices);
        } else {                     // a mixed batch: every segment again, on its own slice of the (new) buffers
            for (const PPCall& q : c->pp_calls)
                if ((rc = pp_launch(q.maps, q.tab, pmx_pp_view(c->pp, q.base), q.B, q.map_h, q.map_w, q.img_len, q.has_scale ? c->d_scale + 2 * q.base : nullptr,
                                    0, c->stream, nullptr, nullptr, q.limbs_slices))) break;
        }
        if (rc) { c->pp_valid = false; return rc; }
    }
    pmx_set_error("post-process capacities did not converge");
    return PMX_ERR_STATE;
}

extern "C" int pmx_set_capacities(pmx_ctx* c, int peaks_per_joint, int subsets, int people, int candidates)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(peaks_per_joint >= 0 && subsets >= 0 && people >= 0 && candidates >= 0, PMX_ERR_INVALID, "pmx_set_capacities: negative capacity");
    const int lim = 1 << 20;
    PMX_CHECK(peaks_per_joint <= lim && subsets <= lim && people <= lim && candidates <= (1 << 24), PMX_ERR_INVALID, "pmx_set_capacities: capacity too large");
    const int cap_pk = peaks_per_joint ? peaks_per_joint : c->pp.cap_pk, cap_sub = subsets ? subsets : c->pp.cap_sub;
    const int cap_ppl = people ? people : c->pp.cap_ppl;
    PMX_CHECK(cap_ppl <= cap_sub, PMX_ERR_INVALID, "pmx_set_capacities: people (%d) must not exceed subsets (%d)", cap_ppl, cap_sub);
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    c->pp_valid = false;
    return pp_realloc(c, cap_pk, cap_sub, candidates, cap_ppl);
}

extern "C" int pmx_get_capacities(pmx_ctx* c, int* peaks_per_joint, int* subsets, int* people, int* candidates)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    if (peaks_per_joint) *peaks_per_joint = c->pp.cap_pk;
    if (subsets) *subsets = c->pp.cap_sub;
    if (people) *people = c->pp.cap_ppl;
    if (candidates) *candidates = c->pp.cap_cand;
    return PMX_OK;
}

extern "C" int pmx_detect_batch(pmx_ctx* c, const uint8_t* img, int B, int H, int W, int on_device, int map_h, int map_w,
                                double img_len, const double* scale_xy)
{
    int rc = pmx_forward_u8(c, img, B, H, W, on_device);
    if (rc) return rc;
    return pmx_postprocess(c, B, map_h, map_w, img_len, scale_xy);
}

extern "C" int pmx_results_layout(pmx_ctx* c, int* people_cap, size_t* bytes_per_record)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(c->pp_valid, PMX_ERR_STATE, "pmx_results_layout: no post-process results yet");
    PMX_DEV(c);
    int rc = pp_finalize(c);
    if (rc) return rc;
    if (people_cap) *people_cap = c->pp.cap_ppl;
    if (bytes_per_record) *bytes_per_record = c->pp.rec_bytes;
    return PMX_OK;
}

extern "C" int pmx_get_results(pmx_ctx* c, int B, void* out, size_t out_bytes)
{
    PMX_CHECK(c && out, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(c->pp_valid && B >= 1 && B <= c->pp_B, PMX_ERR_STATE, "pmx_get_results: no post-process results for batch %d", B);
    PMX_DEV(c);
    for (;;) {
        const size_t rec = c->pp.rec_bytes, need = rec * (size_t)B;
        PMX_CHECK(out_bytes >= need, PMX_ERR_CAPACITY, "pmx_get_results: %zu bytes for %d records of %zu bytes (see pmx_results_layout)",
                  out_bytes, B, rec);
        if (need > c->h_results_bytes) {
            if (c->h_results) (void)hipHostFree(c->h_results);
            c->h_results = nullptr; c->h_results_bytes = 0;
            const size_t want = rec * (size_t)c->max_batch;
            PMX_HIP(hipHostMalloc((void**)&c->h_results, want, hipHostMallocDefault));
            c->h_results_bytes = want;
        }
        PMX_HIP(hipMemcpyAsync(c->h_results, c->pp.results, need, hipMemcpyDeviceToHost, c->stream));
        PMX_HIP(hipStreamSynchronize(c->stream));
        if (!c->pp_final) {
            // common case: the records just copied carry the status words -- no extra round trip
            int bits = 0;
            for (int b = 0; b < B; ++b) bits |= reinterpret_cast<const pmx_image_info*>(c->h_results + (size_t)b * rec)->status;
            if (B == c->pp_B && !(bits & PMX_IMG_CAPACITY_BITS)) c->pp_final = true;
            else {
                int rc = pp_finalize(c);        // grows + re-runs if needed; the record layout may have changed
                if (rc) return rc;
                continue;
            }
        }
        memcpy(out, c->h_results, need);
        return PMX_OK;
    }
}

extern "C" int pmx_results_device_ptr(pmx_ctx* c, void** p, size_t* bytes)
{
    PMX_CHECK(c && p && bytes, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(c->pp_valid, PMX_ERR_STATE, "no post-process results yet");
    if (hipSetDevice(c->device) != hipSuccess) { pmx_set_error("hipSetDevice failed"); return PMX_ERR_HIP; }
    if (int rc = pp_finalize(c)) return rc;      // never hand out records that still carry capacity-overflow bits / a layout about to change
    *p = c->pp.results;
    *bytes = c->pp.rec_bytes;
    return PMX_OK;
}

extern "C" int pmx_results_snapshot(pmx_ctx* c, int slot, void* dst_device, size_t dst_bytes)
{
    PMX_CHECK(c && dst_device, PMX_ERR_INVALID, "null arg");
    PMX_CHECK(slot >= 0 && slot < PMX_SNAPSHOT_SLOTS, PMX_ERR_INVALID, "pmx_results_snapshot: slot %d outside 0..%d", slot, PMX_SNAPSHOT_SLOTS - 1);
    PMX_CHECK(c->pp_valid, PMX_ERR_STATE, "pmx_results_snapshot: no post-process results yet");
    PMX_DEV(c);
    const size_t need = c->pp.rec_bytes * (size_t)c->pp_B;
    PMX_CHECK(dst_bytes >= need, PMX_ERR_CAPACITY, "pmx_results_snapshot: %zu bytes for %d records of %zu bytes", dst_bytes, c->pp_B, c->pp.rec_bytes);
    if (!c->snap_ev[slot]) PMX_HIP(hipEventCreateWithFlags(&c->snap_ev[slot], hipEventDisableTiming));
    if (!c->snap_status[slot]) PMX_HIP(hipHostMalloc((void**)&c->snap_status[slot], sizeof(int) * (size_t)c->max_batch, hipHostMallocDefault));
    PMX_HIP(hipMemcpyAsync(dst_device, c->pp.results, need, hipMemcpyDeviceToDevice, c->stream));
    PMX_HIP(hipMemcpyAsync(c->snap_status[slot], c->pp.status, sizeof(int) * (size_t)c->pp_B, hipMemcpyDeviceToHost, c->stream));
    PMX_HIP(hipEventRecord(c->snap_ev[slot], c->stream));
    c->snap_B[slot] = c->pp_B; c->snap_cap_ppl[slot] = c->pp.cap_ppl; c->snap_rec[slot] = c->pp.rec_bytes;
    return PMX_OK;
}

extern "C" int pmx_snapshot_wait(pmx_ctx* c, int slot, int* batch, int* people_cap, size_t* bytes_per_record, int* status_or)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(slot >= 0 && slot < PMX_SNAPSHOT_SLOTS && c->snap_ev[slot] && c->snap_B[slot] > 0, PMX_ERR_STATE, "pmx_snapshot_wait: slot %d holds no snapshot", slot);
    PMX_DEV(c);
    PMX_HIP(hipEventSynchronize(c->snap_ev[slot]));
    int bits = 0;
    for (int b = 0; b < c->snap_B[slot]; ++b) bits |= c->snap_status[slot][b];
    if (batch) *batch = c->snap_B[slot];
    if (people_cap) *people_cap = c->snap_cap_ppl[slot];
    if (bytes_per_record) *bytes_per_record = c->snap_rec[slot];
    if (status_or) *status_or = bits;
    return PMX_OK;
}

static int check_pp_image(pmx_ctx* c, int image)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_CHECK(c->pp_valid, PMX_ERR_STATE, "no post-process results yet");
    PMX_CHECK(image >= 0 && image < c->pp_B, PMX_ERR_INVALID, "image %d outside 0..%d", image, c->pp_B - 1);
    if (hipSetDevice(c->device) != hipSuccess) { pmx_set_error("hipSetDevice failed"); return PMX_ERR_HIP; }
    return pp_finalize(c);
}

extern "C" int pmx_get_peaks(pmx_ctx* c, int image, double* peaks5, int cap, int* n_rows)
{
    int rc = check_pp_image(c, image);
    if (rc) return rc;
    PMX_CHECK(peaks5 && n_rows, PMX_ERR_INVALID, "null arg");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    int start[PMX_N_JOINTS + 1];
    PMX_HIP(hipMemcpy(start, c->pp.pk_start + image * (PMX_N_JOINTS + 1), sizeof start, hipMemcpyDeviceToHost));
    const int n = start[PMX_N_JOINTS];
    *n_rows = n;
    PMX_CHECK(n <= cap, PMX_ERR_CAPACITY, "pmx_get_peaks: %d rows > capacity %d", n, cap);
    std::vector<int> x(n), y(n);
    std::vector<float> s(n);
    if (n) {
        const size_t pbase = (size_t)image * PMX_N_JOINTS * c->pp.cap_pk;
        PMX_HIP(hipMemcpy(x.data(), c->pp.pk_x + pbase, n * sizeof(int), hipMemcpyDeviceToHost));
        PMX_HIP(hipMemcpy(y.data(), c->pp.pk_y + pbase, n * sizeof(int), hipMemcpyDeviceToHost));
        PMX_HIP(hipMemcpy(s.data(), c->pp.pk_score + pbase, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    int j = 0;
    for (int i = 0; i < n; ++i) {
        while (j < PMX_N_JOINTS && i >= start[j + 1]) ++j;
        peaks5[i * 5 + 0] = j; peaks5[i * 5 + 1] = x[i]; peaks5[i * 5 + 2] = y[i]; peaks5[i * 5 + 3] = (double)s[i]; peaks5[i * 5 + 4] = i;
    }
    return PMX_OK;
}

extern "C" int pmx_get_connections(pmx_ctx* c, int image, double* conns4, int cap, int* n_rows)
{
    int rc = check_pp_image(c, image);
    if (rc) return rc;
    PMX_CHECK(conns4 && n_rows, PMX_ERR_INVALID, "null arg");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    int cnt[PMX_N_LIMBS];
    PMX_HIP(hipMemcpy(cnt, c->pp.cn_count + image * PMX_N_LIMBS, sizeof cnt, hipMemcpyDeviceToHost));
    int total = 0;
    for (int l = 0; l < PMX_N_LIMBS; ++l) total += cnt[l];
    *n_rows = total;
    PMX_CHECK(total <= cap, PMX_ERR_CAPACITY, "pmx_get_connections: %d rows > capacity %d", total, cap);
    int o = 0;
    const int cap_pk = c->pp.cap_pk;
    std::vector<int> a(cap_pk), b(cap_pk);
    std::vector<double> s(cap_pk);
    for (int l = 0; l < PMX_N_LIMBS; ++l) {
        const int n = cnt[l];
        if (!n) continue;
        const size_t base = ((size_t)image * PMX_N_LIMBS + l) * cap_pk;
        PMX_HIP(hipMemcpy(a.data(), c->pp.cn_a + base, n * sizeof(int), hipMemcpyDeviceToHost));
        PMX_HIP(hipMemcpy(b.data(), c->pp.cn_b + base, n * sizeof(int), hipMemcpyDeviceToHost));
        PMX_HIP(hipMemcpy(s.data(), c->pp.cn_score + base, n * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i, ++o) {
            conns4[o * 4 + 0] = l; conns4[o * 4 + 1] = a[i]; conns4[o * 4 + 2] = b[i]; conns4[o * 4 + 3] = s[i];
        }
    }
    return PMX_OK;
}

extern "C" int pmx_get_subsets(pmx_ctx* c, int image, double* subsets20, int cap, int* n_rows)
{
    int rc = check_pp_image(c, image);
    if (rc) return rc;
    PMX_CHECK(subsets20 && n_rows, PMX_ERR_INVALID, "null arg");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    pmx_image_info info;
    PMX_HIP(hipMemcpy(&info, c->pp.results + (size_t)image * c->pp.rec_bytes, sizeof info, hipMemcpyDeviceToHost));
    int n = info.n_people;      // rows kept by the final filter
    *n_rows = n;
    PMX_CHECK(n <= cap, PMX_ERR_CAPACITY, "pmx_get_subsets: %d rows > capacity %d", n, cap);
    if (n) PMX_HIP(hipMemcpy(subsets20, c->pp.subsets + (size_t)image * c->pp.cap_sub * 20, (size_t)n * 20 * sizeof(double),
                             hipMemcpyDeviceToHost));
    return PMX_OK;
}

extern "C" int pmx_get_smoothed(pmx_ctx* c, int image, int joint, float* out, int map_h, int map_w)
{
    int rc = check_pp_image(c, image);
    if (rc) return rc;
    const int n_ch = c->kind == NET_POSE ? PMX_N_JOINTS : c->n_heat - 1;
    PMX_CHECK(out && joint >= 0 && joint < n_ch, PMX_ERR_INVALID, "bad arg");
    PMX_CHECK(c->pp.smoothed && (c->opt_keep_smoothed || c->kind != NET_POSE), PMX_ERR_STATE, "pmx_get_smoothed: option keep_smoothed was not set");
    PMX_CHECK(map_h == c->pp_h && map_w == c->pp_w, PMX_ERR_INVALID, "pmx_get_smoothed: map size mismatch");
    PMX_DEV(c);
    PMX_HIP(hipStreamSynchronize(c->stream));
    PMX_HIP(hipMemcpy(out, c->pp.smoothed + ((size_t)image * n_ch + joint) * map_h * map_w, (size_t)map_h * map_w * sizeof(float),
                      hipMemcpyDeviceToHost));
    return PMX_OK;
}

// ------------------------------------------------------------------------------------- measurement
extern "C" int pmx_timer_start(pmx_ctx* c)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_DEV(c);
    PMX_HIP(hipEventRecord(c->t0, c->stream));
    return PMX_OK;
}
extern "C" int pmx_timer_stop(pmx_ctx* c, double* ms)
{
    PMX_CHECK(c && ms, PMX_ERR_INVALID, "null arg");
    PMX_DEV(c);
    PMX_HIP(hipEventRecord(c->t1, c->stream));
    PMX_HIP(hipEventSynchronize(c->t1));
    float f = 0.f;
    PMX_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    *ms = f;
    return PMX_OK;
}
extern "C" int pmx_profile_enable(pmx_ctx* c, int on)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_DEV(c);
    int rc = prof_collect(c);
    c->prof_on = on < 0 ? 0 : (on > 2 ? 1 : on);
    return rc;
}
extern "C" int pmx_profile_reset(pmx_ctx* c)
{
    PMX_CHECK(c, PMX_ERR_INVALID, "null ctx");
    PMX_DEV(c);
    int rc = prof_collect(c);
    c->prof.clear();
    c->prof_index.clear();
    return rc;
}
extern "C" int pmx_profile_count(pmx_ctx* c, int* n)
{
    PMX_CHECK(c && n, PMX_ERR_INVALID, "null arg");
    PMX_DEV(c);
    int rc = prof_collect(c);
    *n = (int)c->prof.size();
    return rc;
}
extern "C" int pmx_profile_issued(pmx_ctx* c, int i, double* issued_flop)
{
    PMX_CHECK(c && issued_flop && i >= 0 && i < (int)c->prof.size(), PMX_ERR_INVALID, "bad profile index");
    *issued_flop = c->prof[i].issued;
    return PMX_OK;
}

extern "C" int pmx_profile_entry(pmx_ctx* c, int i, char* name, int cap, double* total_ms, int64_t* launches, double* flops, double* bytes)
{
    PMX_CHECK(c && i >= 0 && i < (int)c->prof.size(), PMX_ERR_INVALID, "bad profile index");
    const ProfEntry& e = c->prof[i];
    if (name && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (total_ms) *total_ms = e.total_ms;
    if (launches) *launches = e.launches;
    if (flops) *flops = e.flops;
    if (bytes) *bytes = e.bytes;
    return PMX_OK;
}

// ---------------------------------------------------------------------------- single-layer test entry
extern "C" int pmx_conv2d(pmx_ctx* c, const float* x, const float* w, const float* bias, int B, int cin, int H, int W, int cout,
                          int ks, int relu, int pool, float* y, int iters, double* avg_ms)
{
    PMX_CHECK(c && x && w && y, PMX_ERR_INVALID, "pmx_conv2d: null arg");
    PMX_CHECK(ks == 1 || ks == 3 || ks == 7, PMX_ERR_INVALID, "pmx_conv2d: ksize must be 1, 3 or 7");
    PMX_CHECK(B >= 1 && cin >= 1 && cout >= 1 && H >= 1 && W >= 1, PMX_ERR_INVALID, "pmx_conv2d: bad shape");
    PMX_CHECK(!pool || (H % 2 == 0 && W % 2 == 0), PMX_ERR_INVALID, "pmx_conv2d: pool needs even H, W");
    PMX_DEV(c);
    std::vector<int> cmap = identity_map(cin);
    const int cin_pad = (int)cmap.size(), cpad = cout_pad_of(cout);
    std::vector<float> wp, bp;
    pack_weights(w, bias, cout, cin, ks, cmap, cpad, wp, bp);
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    float *d_x = nullptr, *d_xn = nullptr, *d_w = nullptr, *d_b = nullptr, *d_y = nullptr, *d_yn = nullptr;
    const size_t nx = (size_t)B * cin * H * W, nxn = (size_t)B * H * W * cin_pad, ny = (size_t)B * cout * Ho * Wo;
    PMX_HIP(hipMalloc((void**)&d_x, nx * 4));
    PMX_HIP(hipMalloc((void**)&d_xn, nxn * 4));
    PMX_HIP(hipMalloc((void**)&d_w, wp.size() * 4));
    PMX_HIP(hipMalloc((void**)&d_b, bp.size() * 4));
    PMX_HIP(hipMalloc((void**)&d_y, ny * 4));
    PMX_HIP(hipMalloc((void**)&d_yn, ny * 4));
    PMX_HIP(hipMemcpy(d_x, x, nx * 4, hipMemcpyHostToDevice));
    PMX_HIP(hipMemsetAsync(d_xn, 0, nxn * 4, c->stream));
    PMX_HIP(hipMemcpy(d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    // poison the output so that unwritten elements are caught by the test
    PMX_HIP(hipMemsetAsync(d_yn, 0xFF, ny * 4, c->stream));
    int rc = launch_nchw_to_nhwc(d_x, d_xn, B, cin, H, W, cin_pad, 0, c->stream);
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.g[0].in = d_xn; a.g[0].w = d_w; a.g[0].bias = d_b; a.g[0].out = d_yn; a.g[0].cout = cout;
    a.B = B; a.H = H; a.W = W; a.lda = cin_pad; a.ldc = cout; a.nch = cin_pad / CK; a.cout_pad = cpad; a.relu = relu; a.pool = pool;
    const int v = conv_pick_variant(ks, cpad, H, W, B, c->opt_force[ks], c->opt_kernel_gen, pool, cin, c->opt_precision == 1 && ks > 1);
    void* d_w3 = nullptr;
    int v_run = v;
    if (c->opt_precision == 1 && conv_bf16x3_twin(v) >= 0 && ks > 1) {
        std::vector<uint16_t> w3;
        pack_bf16x3(wp, ks * ks, cin_pad / CK, cpad, w3);
        PMX_HIP(hipMalloc(&d_w3, w3.size() * sizeof(uint16_t)));
        PMX_HIP(hipMemcpy(d_w3, w3.data(), w3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        a.g[0].w = (const float*)d_w3;
        v_run = conv_bf16x3_twin(v);
    }
    SplitPlan plan = conv_pick_ksplit(v_run, H, W, B, 1, cpad, cin_pad / CK, pool, c->opt_ksplit);
    if (cout % 4 != 0) plan.S = 1;
    float* d_ww = nullptr;
    int ug = 0, wrun = 0, wtail = 0;
    const int wmode = wino_mode(c, ks, cin_pad, cpad, cout, cout, B, H, W, pool, &ug, &wrun, &wtail, 1, a.lda);
    const bool wino = wmode == 1;
    if (wino) {
        std::vector<float> ww;
        pack_wino(wp, ks, cin_pad / CK, cpad, ww);
        PMX_HIP(hipMalloc((void**)&d_ww, ww.size() * sizeof(float)));
        PMX_HIP(hipMemcpy(d_ww, ww.data(), ww.size() * sizeof(float), hipMemcpyHostToDevice));
        a.g[0].w = d_ww; a.nch = cin_pad / 32;
    }
    if (ug) {
        std::vector<float> ww;
        pack_wino(wp, ks, cin_pad / CK, cpad, ww);
        PMX_HIP(hipMalloc((void**)&d_ww, ww.size() * sizeof(float)));
        PMX_HIP(hipMemcpy(d_ww, ww.data(), ww.size() * sizeof(float), hipMemcpyHostToDevice));
        a.g[0].w = d_ww; a.nch = cin_pad / 32;
    }
    auto launch_conv = [&](pmx_ctx* cc, const ConvArgs& aa, int gg, int vv, const SplitPlan& pp) {
        if (ug) return launch_wino_units(cc, aa, ks, gg, ug);
        if (wino && wrun) return launch_wino_run(cc, aa, ks, gg, wtail);
        return wino ? conv_wino_launch(aa, ks, gg, cc->stream) : ::launch_conv(cc, aa, gg, vv, pp);
    };
    if (!rc) rc = launch_conv(c, a, 1, v_run, plan);
    if (!rc && iters > 0) {
        hipEvent_t e0, e1;
        PMX_HIP(hipEventCreate(&e0)); PMX_HIP(hipEventCreate(&e1));
        PMX_HIP(hipEventRecord(e0, c->stream));
        for (int i = 0; i < iters && !rc; ++i) rc = launch_conv(c, a, 1, v_run, plan);
        PMX_HIP(hipEventRecord(e1, c->stream));
        PMX_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        PMX_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (avg_ms) *avg_ms = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (!rc) rc = launch_nhwc_to_nchw(d_yn, d_y, B, cout, Ho, Wo, cout, 0, c->stream);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { pmx_set_error("pmx_conv2d: %s", hipGetErrorString(e)); rc = PMX_ERR_HIP; }
    }
    if (!rc) {
        hipError_t e = hipMemcpy(y, d_y, ny * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { pmx_set_error("pmx_conv2d: %s", hipGetErrorString(e)); rc = PMX_ERR_HIP; }
    }
    (void)hipFree(d_x); (void)hipFree(d_xn); (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_y); (void)hipFree(d_yn);
    if (d_w3) (void)hipFree(d_w3);
    if (d_ww) (void)hipFree(d_ww);
    return rc;
}

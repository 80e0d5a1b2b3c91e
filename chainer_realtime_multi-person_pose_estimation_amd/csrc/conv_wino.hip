// conv_wino.hip -- the fp32 Winograd F(2x2, 3x3) kernel for the 3x3 / 7x7 layers (option "conv_algo"): conv_wino_kernel<KS, POOL,
// UNIT, GEOM> -- rectangles (GEOM 0), runs of 32 consecutive tiles of 46-column slabs (GEOM 1 / 2), merged tails (GEOM 3), unit mode for
// single images and tails -- its launchers and the combine kernel of the unit-mode tails.  Replaces `L.Convolution2D` (+ `F.relu`,
// + `F.max_pooling_2d(2, 2)`) on those layers (models/CocoPoseNet.py:26-129) with a DEFINED fp32 arithmetic that oracle/conv_fma_ref.c::
// conv_wino_ref restates bit for bit.  The direct kernels, the variant table and the split-K machinery live in conv_mfma.hip.
#include <hip/hip_ext.h>
#include <type_traits>
#include "pmx_common.h"

#include "wino_util.h"

// ---- Winograd F(2x2, 3x3) (fp32, option "conv_algo") ------------------------------------------------------------------
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A: a 2 x 2 output tile from a 4 x 4 input window costs 16 multiplies per channel pair instead of
// 36 -> 2.25x less matrix work; the transforms are additions only (B^T, A^T) or done once on the host (G g G^T, in double, rounded
// to fp32).  Block = 32 Winograd tiles (4 rows x 8 columns of 2 x 2 = an 8 x 16 pixel output tile) x 128 output channels; wave w owns
// 32 channels and all 16 "frequencies": 16 accumulator tiles of 32 (Winograd tiles) x 32 (channels) = 256 AGPRs, one block per CU.
// Everything runs as PHASES of 8 planes x 4 k8-steps x 4 MFMAs per wave: while a phase multiplies the 8 planes in one half of the
// U[plane][tile][channel] LDS buffer, every thread transforms its (tile, 4 channels) item of the raw halo for the next phase into the
// other half, one LDS / VALU instruction per slot between two MFMAs; the transformed weights stream from L2 in a register ring 32 MFMAs
// ahead (pinned with sched_barrier: left alone the compiler sinks the loads to one step ahead and the single wave per SIMD stalls on
// L2); one s_barrier per phase (eight MFMAs before its end: see p1_step).  Epilogue: A^T M A per lane (the 16 frequencies of a (tile, channel) sit in one lane's registers),
// bias, ReLU, 2x2 max-pool = max over the tile's four outputs.
// KS = 7 (the 7x7 layers of stages 2-6), 100 instead of 196 products per tile and channel pair: pass 1 -- the taps (0..5, 0..5) are four
// 3x3 sub-kernels, each a Winograd product on its own shifted window, all four accumulated in the SAME frequency-domain accumulators
// (the output transform is linear: 4 x 16); pass 2a -- row 6 as two 1x3 sub-kernels, 1-D F(2,3) along x (2 x 8), and tap (6, 6) direct
// (4); pass 2b -- column 6 as two 3x1 sub-kernels along y (2 x 8).  The raw halo of a 32-channel chunk is staged once per pass, round-robin.
// UNIT = 1 (single images): a block runs one unit (pass 1 over a chunk range / row 6 / column 6 / tap (6, 6)) and writes its share of y to
// a slab; conv_splitk_reduce_kernel adds the slabs in unit order.
// The arithmetic is DEFINED -- transform additions in a fixed order, one sequential FMA chain per plane over (chunk, sub-kernel, k8-step,
// k), output transforms in a fixed order, units added in order -- and oracle/conv_fma_ref.c::conv_wino_ref restates it bit for bit; it
// is not the direct kernels' chain (results agree to fp32 rounding, ~1e-6 of the map scale).
// float4 add / subtract as two packed-fp32 instructions (v_pk_add_f32, the subtrahend negated by the source modifier: same rounding as
// v_sub_f32); the scheduler-pinned one-op-per-slot transform code otherwise compiles to four scalar VALU instructions per float4
// Diagnostic build only (tools/block_timing.py compiles this file with -DPMX_BLOCK_TIMING into its own library; the product library
// never defines it): thread 0 of the first 8192 blocks of a Winograd launch stamps the 100 MHz wall clock at entry / pipeline primed /
// before the stores / exit, and the CU it runs on.  (The stamps perturb the register allocation of the loops -- a 7x7 block runs 1.4x
// slower in that build -- so only the prologue, the epilogue and the hand-over gap between two blocks on a CU are read off it.)
#ifdef PMX_BLOCK_TIMING
__device__ unsigned long long g_blk_t[8192 * 8];
extern "C" int pmx_debug_block_times(unsigned long long* out, size_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_t), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#define PMX_T(k) do { if (threadIdx.x == 0) { const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
                                              if (lin_ < 8192) g_blk_t[lin_ * 8 + (k)] = (k) == 7 ? (unsigned long long)__smid() : wall_clock64(); } } while (0)
#else
#define PMX_T(k)
#endif

// GEOM 0: a block = 4 x 8 Winograd tiles = an 8 x 16 pixel output rectangle (any map size).
// GEOM 1 ("runs", 46-pixel-wide maps = the 46 x 46 maps of a 368 x 368 input): a block = 32 CONSECUTIVE Winograd tiles in row-major order
// of the 23-tile-wide grid of one image (at most 3 tile rows): a 46 x 46 map is 529 tiles = 16 full blocks + 17 tiles instead of 18
// rectangles (8.9 % padding), and the 16 full blocks of 32 images x 2 branch groups are exactly 4 rounds of 256 CUs; the part-filled last
// block of every image runs in unit mode (pmx_api.hip::run_conv).  Raw halo = the 6 + KS - 1 input rows the three tile rows touch x all
// 46 + KS - 1 columns (7x7: 12 x 52 pixels x 32 channels = 90 KB next to the 74 KB of U: 256 bytes short of the 160 KB LDS).
// GEOM 3 ("merged tails", unit mode only): the part-filled last blocks of ALL images of the launch as one stream -- image b's tail tiles
// [32 nfull, ntiles) (they lie in one tile row; nt of them, 16 <= nt <= 23) are the stream positions [b nt, (b + 1) nt), and block j owns the
// positions [32 j, 32 j + 32): up to three images' segments, every MFMA row a real tile (46 x 46: 17 tiles per image -- one image per block
// filled 17 of the 32 rows).  Raw halo = one tile row, the segments side by side, each with its own KS - 1 columns of overlap.
// how far ahead of their MFMAs the transformed weights are requested: pass 1 in steps of 4 MFMAs (ring of 16), pass 2 in steps of 8 (ring
// of 8).  8 / 4 = ~2000 cycles; 10 / 5, 12 / 6 and 15 / 7 measured 0.5 - 3 % slower on the 7x7 layers: the weight stream is not what the
// matrix pipe waits for
constexpr int WLEAD1 = 8, WLEAD2 = 4;
static_assert(WLEAD1 >= 4 && WLEAD1 <= 15 && WLEAD2 >= 4 && WLEAD2 <= 7, "weight ring lead");
// Diagnostic builds only (tools/kernel_variants.py; the product library is built with 0): leave out parts of the phases' side work to see
// what the matrix pipe waits for -- 1: the transform slots (LDS reads of the raw halo, B^T d B, U stores, halo staging), 2: the weight
// loads, 4: the A-fragment LDS reads, 8: the barriers inside the phases; pass 1 only: 16: the raw-halo LDS reads of the transform, 32: its
// VALU work, 64: its U stores, 128 / 256: the halo staging's LDS stores / global loads.  The results are wrong; only the launch time is read.
#ifndef PMX_ABLATE
#define PMX_ABLATE 0
#endif
// Schedule decisions that were A/B-measured and are now fixed (the rejected alternatives are gone from the source; EXPERIMENTS.md has the
// numbers): the output transform two registers at a time through v_pk_add_f32 (-0.5 % on 7x7, profiles/r05_pkout_ab.json); the
// transform's packed adds clustered in ONE gap per group of 16, first cluster in slot 20 (a gap that holds VALU work costs ~3.2 ns of
// matrix-pipe time once plus ~2.2 ns per instruction, tools/mfma_gap_probe.hip: -1.6 % on 7x7, profiles/r05_vcluster_ab.json), on every
// geometry (profiles/r05_vcluster_all_geoms.json); the chunk offset of a halo load in the scalar offset; halo offsets kept in registers
// on the 7x7 forms only (3x3: +6 % on conv3_3); one barrier per pass-2 phase at its end (eight MFMAs earlier: +0.6 %,
// profiles/r05_p2bar_ab.json); blocks of a unit-mode launch in their natural order on the XCDs (placed by weight set: no gain,
// profiles/r05_unit_xcd_ab.json).
template <int KS, int GEOM>
struct WinoCfg {
    static constexpr int TH = 8, TW = 16, PADK = KS / 2, CKW = 32, LDU = CKW + 4;
    // raw-halo pixel pitch (floats).  36 (7x7: all the LDS allows): a transform read of 16 lanes covers two tiles 2 pixels = 72 floats
    // apart -> their 128-byte rows overlap in 24 of 64 banks (PMC: 25 % of the LDS cycles are bank conflicts).  3x3: the halo is small
    // enough for a pitch of 48 -> 2 pixels = 96 floats = 32 banks apart, no overlap
    // (GEOM 3, 7x7: 8 x 82 pixels only fit with a pitch of 32 -- two tiles of a transform read then share their banks: 2-way conflicts,
    //  on a launch that is 3 % of a layer)
    static constexpr int LDR = KS == 3 ? 48 : GEOM == 3 ? CKW : CKW + 4;
    static constexpr int RUN_TX = PMX_WINO_RUN_TX, RUN_W = 2 * RUN_TX;
    // GEOM 3 ("merged tails"): one tile row of up to three images side by side: 32 tiles + three times the KS - 1 columns of overlap
    static constexpr int HH = GEOM == 3 ? 2 + KS - 1 : GEOM ? 6 + KS - 1 : TH + KS - 1;
    static constexpr int HW = GEOM == 3 ? 2 * PMX_WINO_RUN_TILES + 3 * (KS - 1) : GEOM ? RUN_W + KS - 1 : TW + KS - 1, NPX = HH * HW;
    static constexpr int NSUB = KS == 3 ? 1 : 4;                       // 3x3 sub-kernels done as Winograd products
    static constexpr int NDIR = KS == 3 ? 0 : 13;                      // taps outside the 3x3 sub-kernels (KS = 7: row 6, column 6 -> pass 2)
    static constexpr int RAW_ELEMS = NPX * LDR, U_ELEMS = 16 * 32 * LDU;
    static constexpr int LDS_BYTES = (RAW_ELEMS + U_ELEMS) * 4;
    static constexpr int NHF = (NPX * (CKW / 4) + 255) / 256;
    static_assert(KS == 3 || KS == 7, "Winograd kernel: 3x3 or 7x7");
    static_assert(LDS_BYTES <= 160 * 1024, "Winograd kernel: LDS");
};

template <int KS, int POOL, int UNIT, int GEOM>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a)
{
    using C = WinoCfg<KS, GEOM>;
    static_assert(!(POOL && KS != 3), "pooling only with the 3x3 variant");
    static_assert(!(UNIT && POOL), "unit mode: the combine kernel pools");
    static_assert(GEOM != 3 || UNIT, "merged tails run in unit mode");
    constexpr bool MERGE = GEOM == 3;
    // (the transform's packed adds sit clustered in one gap per group of 16.  The rectangle and multi-slab forms first answered the clusters
    //  with 3.4 KB of scratch per lane: not register pressure but a DECLINED UNROLL -- with the clusters in it the body of the 32 x 4 slot
    //  loop crossed the unroller's size limit for `#pragma unroll`, the loop stayed a loop and every register array indexed by the slot
    //  number moved to scratch.  This file is therefore compiled with -mllvm -pragma-unroll-threshold=200000 (native.py::SOURCES);
    //  tools/isa_stats.py shows what is left, tests/test_host.py holds the scratch ceiling)
    // slot of the first cluster (the second follows eight slots later, the stores nine later still): all twelve raw-halo reads long landed
    constexpr int VT = 20;
    static_assert(VT >= 16 && VT <= 22, "clustered transform: first cluster in slots 16 .. 22 (its stores end before the halo slots at 40)");
    // UNIT (single images: 36 blocks of a 46x46 7x7 layer cannot fill 256 CUs): blockIdx.z = unit * groups + group, and a block runs
    // ONE unit of the work -- unit u < nu1: pass 1 over the chunks [u g, u g + g) (g = a.kbounds); 7x7: unit nu1: row 6 (pass 2a without
    // tap (6, 6)); unit nu1 + 1: column 6 (pass 2b); unit nu1 + 2: tap (6, 6) -- and writes its untransformed share of y (no bias / ReLU) to slab `unit`; conv_splitk_reduce_kernel adds the slabs in unit order
    extern __shared__ float4 smem4[];
    PMX_T(0); PMX_T(7);
    float* const s_raw = reinterpret_cast<float*>(smem4);
    float* const s_u = s_raw + C::RAW_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    // Which piece of the launch this workgroup computes.  Workgroups go to the 8 XCDs round-robin in their linear order (x fastest), each
    // XCD has its own L2.  (Unit mode: the gridDim.x blocks that read the same weights land on all 8 XCDs, so every L2 fetches all the
    // weights of the layer -- 125 - 138 MB per merged-tail launch against ~38 MB unique; handing every XCD a contiguous range of pieces
    // instead measured no gain: the redundant fetch is not what the unit launches wait for.)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int unit = UNIT ? bz / a.ngroups : 0;
    const bool g1 = (UNIT ? bz % a.ngroups : bz) != 0;
    ConvGroupArgs G;
    G.in = g1 ? a.g[1].in : a.g[0].in;
    G.w = g1 ? a.g[1].w : a.g[0].w;              // transformed weights [plane][chunk32][k8-step][cout_pad][8] (pmx_api.hip::pack_wino)
    G.bias = g1 ? a.g[1].bias : a.g[0].bias;
    G.out = g1 ? a.g[1].out : a.g[0].out;
    G.cout = g1 ? a.g[1].cout : a.g[0].cout;
    if (UNIT) G.out += (size_t)unit * (size_t)a.slab_stride;
    int tile;
    {                                             // (plain launches: gridDim.x is a multiple of 8 wherever it matters, XCD = blockIdx.x & 7)
        const int nwg = gridDim.x, bid = bx;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // Heterogeneous launches (rectangles, plain mode; a.nseg > 0): the launch is a stream of SEGMENTS -- n images of one map size each
    // (the scales of detect_precise, the size classes of a mixed batch) -- laid end to end in the activation buffers; segment s owns the
    // tiles [segs[s].tile0, segs[s + 1].tile0) of the launch.  All of this is scalar work on block-uniform values (one s_load per field):
    // the block then behaves exactly like a block of a launch of that segment alone -- same tiles, same halo, same arithmetic, same bits
    // (a plain block's result does not depend on which launch it is part of).
    constexpr bool SEGS = GEOM == 0 && !UNIT;
    int H = a.H, W = a.W, s_tiles_x = a.tiles_x, s_tiles_img = a.tiles_x * a.tiles_y;
    size_t s_pix_in = 0, s_pix_out = 0;
    if (SEGS && a.nseg > 0) {
        int sg = 0;
        for (int k = 1; k < a.nseg; ++k) sg = tile >= a.segs[k].tile0 ? k : sg;
        const ConvSeg S = a.segs[sg];
        H = S.H; W = S.W; s_tiles_x = S.tiles_x; s_tiles_img = S.tiles_img;
        s_pix_in = (size_t)(unsigned)S.pix0; s_pix_out = (size_t)(unsigned)S.pixo;
        tile -= S.tile0;
    }
    // GEOM 1: the map is cut into vertical slabs of 46 columns (23 tile columns; 46 / 92 / 184-wide maps = 1 / 2 / 4 slabs); block trem of
    // this launch in (image, slab) bslab = the 32 consecutive Winograd tiles [t0, t0 + 32) of that slab (row-major), tile rows r0 .. r0 + 2;
    // raw halo row 0 / column 0 = image row 2 r0 - PADK / column 46 slab - PADK (halo columns inside the map come from the neighbour slab)
    const int tiles_per_img = MERGE ? 1 : GEOM ? a.run_nb : s_tiles_img;
    const int bslab = tile / tiles_per_img;
    const int trem = tile - bslab * tiles_per_img;
    // (GEOM 1 = a single slab, the 46-wide maps of the 7x7 layers: the slab arithmetic is compiled out -- its extra scalar registers
    //  pushed the 7x7 kernel's transition code into 30 more spill reloads per block, +3 %; GEOM 2 = any number of slabs)
    constexpr bool SLABS = GEOM == 2;
    const int bimg = SLABS ? bslab / a.run_nslab : bslab;
    const int sx0 = SLABS ? (bslab - bimg * a.run_nslab) * C::RUN_W : 0;
    const int t0 = MERGE ? a.run_j0 * PMX_WINO_RUN_TILES : GEOM ? (a.run_j0 + trem) * PMX_WINO_RUN_TILES : 0;      // (MERGE: first tail tile of an image)
    const int r0 = GEOM ? t0 / C::RUN_TX : 0;
    const int ntiles = C::RUN_TX * ((H + 1) >> 1);
    const int y0 = GEOM ? 2 * r0 : (trem / s_tiles_x) * C::TH, x0 = GEOM ? sx0 : (trem % s_tiles_x) * C::TW;
    // MERGE: block `tile` = stream positions [32 tile, 32 tile + 32) = segment s (s = 0, 1, 2) of image mg_img0 + s: mg_n0 / mg_n1 / the
    // rest tiles from tail tile mg_tt0 (s = 0) / 0 on, halo columns from 0 / mg_cb1 / mg_cb2 on (2 n + KS - 1 of them)
    const int mg_nt = ntiles - t0, mg_tx0 = t0 - r0 * C::RUN_TX;
    const int mg_p0 = tile * PMX_WINO_RUN_TILES, mg_img0 = MERGE ? mg_p0 / mg_nt : 0, mg_tt0 = mg_p0 - mg_img0 * mg_nt;
    const int mg_n0 = min(mg_nt - mg_tt0, PMX_WINO_RUN_TILES), mg_n1 = min(mg_nt, PMX_WINO_RUN_TILES - mg_n0);
    const int mg_cb1 = 2 * mg_n0 + KS - 1, mg_cb2 = mg_cb1 + 2 * mg_n1 + KS - 1;
    const int n0 = by * 128;
    const int n = n0 + wave * 32 + li;
    const float* in_b = G.in + (MERGE ? (size_t)0 : (s_pix_in + (size_t)bimg * H * W) * a.lda);
    float bias = G.bias[n];                       // (pinned to a register further down, once the first halo loads are on their way:
                                                  //  pinned here the block waited a full memory round trip before issuing anything else)
    const int nch = a.nch;                        // chunks of 32 input channels
    const int ug = UNIT ? (int)a.kbounds : nch;   // chunks per pass-1 unit
    const int nu1 = UNIT ? (nch + ug - 1) / ug : 1;
    const int c0 = UNIT ? min(unit, nu1 - 1) * ug : 0;                     // pass-1 chunk range of this block
    const int c1 = UNIT ? min(nch, c0 + ug) : nch;
    const bool do_p1 = !UNIT || unit < nu1, do_p2a = !UNIT || unit == nu1, do_p2b = !UNIT || unit == nu1 + 1;
    const bool do_pd = UNIT && unit == nu1 + 2;   // unit mode: tap (6, 6) is a unit of its own (in pass 2a otherwise)

    // raw halo staging: slot r of a thread = pixel (tid >> 3) + 32 r of the halo, channels 4 (tid & 7) .. + 3 of the chunk.  Nothing per slot
    // but one byte offset: the loads go through a buffer resource that spans exactly this image, so rows above / below the map fall out
    // of its range and return 0 = the zero padding; columns left / right of it get an out-of-range offset.  (Round 6: also the rectangles
    // -- until then GEOM 0 kept a clamped global offset and an in-image mask per slot, added the chunk offset with 64-bit vector
    // arithmetic in front of every load and selected zeros at the LDS store.)
    // (MERGE: the resource spans the whole batch -- rows outside an image would land in its neighbour, so they are masked like the columns)
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0,
                                                                           MERGE ? (unsigned)(a.B * H * W * a.lda) * 4u : (unsigned)(H * W * a.lda) * 4u, 0x00020000);
    const int hbase_b = (((y0 - C::PADK) * W + x0 - C::PADK) * a.lda + (tid & 7) * 4) * 4;     // byte offset of halo pixel (0, 0), may be negative
    const int hrow_skip = (SLABS || GEOM == 0) ? W - C::HW : -(KS - 1);                      // image pixels between the end of a halo row and the next
    const int lda_b = a.lda * 4;
    // GEOM 1 / 2: byte offset of slot r's pixel in the image for chunk 0, computed ONCE with the first halo load (0x80000000 = the column is
    // outside the map: stays out of the buffer's range whatever chunk offset is added) and kept in registers -- recomputed per use, the
    // compiler hoisted a second copy of this arithmetic (20 slots x (mul_hi, mul_lo, mad, cmp)) to right in front of the first MFMA
    // (7x7 only: on the 3x3 instantiations the kept offsets measured slower -- conv3_3 +6 % -- than the compiler's own placement)
    constexpr bool HOFF = GEOM == 0 || KS == 7 || GEOM == 3;     // (merged tails: the per-slot segment arithmetic is never repeated)
    int h_off[HOFF ? C::NHF : 1];
    auto halo_off_calc = [&](int r) -> int {
        const unsigned hp = (unsigned)(tid >> 3) + 32u * r;
        const unsigned hy = hp / (unsigned)C::HW, hx = hp - hy * (unsigned)C::HW;
        if constexpr (MERGE) {
            const int sg = (int)hx >= mg_cb2 ? 2 : (int)hx >= mg_cb1 ? 1 : 0;
            const int lx = (int)hx - (sg == 2 ? mg_cb2 : sg == 1 ? mg_cb1 : 0);
            const int ns = sg == 2 ? PMX_WINO_RUN_TILES - mg_n0 - mg_n1 : sg == 1 ? mg_n1 : mg_n0;
            const int img = mg_img0 + sg;
            const int gy = y0 - C::PADK + (int)hy, gx = 2 * (mg_tx0 + (sg == 0 ? mg_tt0 : 0)) - C::PADK + lx;
            const bool ok = lx < 2 * ns + KS - 1 && img < a.B && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && hy < (unsigned)C::HH;
            return ok ? (((img * H + gy) * W + gx) * a.lda + (tid & 7) * 4) * 4 : (int)0x80000000;
        }
        // halo pixel (hy, hx) = image pixel (y0 - PADK + hy, x0 - PADK + hx): hp + (W - HW) hy pixels after halo pixel (0, 0) in the image
        int off = hbase_b + ((int)hp + (int)hy * hrow_skip) * lda_b;
        if ((SLABS || GEOM == 0) ? (unsigned)(x0 - C::PADK) + hx >= (unsigned)W          // (left of the map the sum wraps around: also out)
                                 : hx - (unsigned)C::PADK >= (unsigned)C::RUN_W) off = (int)0x80000000;
        return off;
    };
    auto halo_off_init = [&](int r) { if constexpr (HOFF) h_off[r] = halo_off_calc(r); };
    auto halo_load_slot = [&](float4 (&hv)[C::NHF], int chunk, int r) {       // r is a compile-time constant at every call
        const int off0 = HOFF ? h_off[HOFF ? r : 0] : halo_off_calc(r);
        // (the chunk's byte offset goes into the scalar offset, which the range check ignores: a pixel outside the image stays out of
        //  range, a pixel inside it stays inside its own channel row -- one VALU add less per load)
        hv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off0, chunk * (C::CKW * 4), 0));
    };
    float* const s_raw_t = s_raw + (tid >> 3) * C::LDR + (tid & 7) * 4;       // slot r of this thread: + r * 32 * LDR floats (an immediate offset)
    auto halo_store_slot = [&](const float4 (&hv)[C::NHF], int r) {
        const int f = tid + r * 256;
        const float4 v = hv[r];
        // (only the last slot can fall behind the halo; spelled out because the compiler does not bound tid by the block size)
        if (r * 256 + 255 < C::NPX * (C::CKW / 4) || f < C::NPX * (C::CKW / 4)) *reinterpret_cast<float4*>(s_raw_t + r * (32 * C::LDR)) = v;
    };
    auto halo_load = [&](float4 (&hv)[C::NHF], int chunk) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) halo_load_slot(hv, chunk, r);
    };
    auto halo_store = [&](const float4 (&hv)[C::NHF]) {
#pragma unroll
        for (int r = 0; r < C::NHF; ++r) halo_store_slot(hv, r);
    };
    // position of Winograd tile m of the block in the raw halo (top-left pixel of sub-kernel 0's 4 x 4 window), in pixels.
    // GEOM 0: 4 x 8 grid; GEOM 1: tile t0 + m of the row-major run (tiles past the end of the map repeat the last one; never stored)
    auto tile_px = [&](int m) -> int {
        if (MERGE) {         // (positions past the end of the stream repeat the last tile of the last image; never read back)
            const int mc = min(m, a.B * mg_nt - 1 - mg_p0);
            const int sg = mc >= mg_n0 + mg_n1 ? 2 : mc >= mg_n0 ? 1 : 0;
            return (sg == 2 ? mg_cb2 - 2 * (mg_n0 + mg_n1) : sg == 1 ? mg_cb1 - 2 * mg_n0 : 0) + 2 * mc;
        }
        if (GEOM) {
            const int t = min(t0 + m, ntiles - 1);
            const int ty = t / C::RUN_TX;
            return (2 * (ty - r0)) * C::HW + 2 * (t - ty * C::RUN_TX);
        }
        return (2 * (m >> 3)) * C::HW + 2 * (m & 7);
    };
    // transform item of this thread: Winograd tile tt, channels 4 * tc .. + 3 of the chunk
    const int tt = tid >> 3, tc = tid & 7;
    const int t_raw = tile_px(tt) * C::LDR + tc * 4;
    const int t_u = tt * C::LDU + tc * 4;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G.w), 0, 0x7fffffff, 0x00020000);
    // weight panels are [plane][chunk32][k8-step 4][cout_pad][8]: the 64 lanes of one fragment load (32 channels x 2 halves x 16 B)
    // read 1 KB of contiguous, fully used cache lines (with the channel-major [cout_pad][32] layout each load touched 32 lines and used a
    // quarter of each, relying on the 32 KB L1 to keep them for the next three k8-steps -- it did not: weight loads cost 7.5 %)
    // ([plane][chunk32][cout_pad / 32][k8-step 4][32][8]: the k8-steps of this wave's 32 channels are 1 KB apart -- a constant on the
    //  vector offset = the load's immediate offset, no scalar add per load)
    const unsigned b_off = (unsigned)(((n >> 5) * 1024 + (n & 31) * 8 + kh * 4) * 4);
    constexpr unsigned st_v = 1024u;                                       // bytes between the k8-steps of a panel (in the vector offset)
    const unsigned panel_b = (unsigned)a.cout_pad * C::CKW * 4u;          // bytes of one (plane, chunk) panel
    const unsigned freq_b = panel_b * (unsigned)nch;                       // bytes between planes (sub-kernel * 16 + frequency)

    f32x16 acc[16];
    // the 256 accumulator registers are zeroed in the shadow of the first halo / weight loads (left to the compiler the 256
    // v_accvgpr_write sat right in front of the first MFMA, after both prologue barriers: ~0.5 us per block, fully exposed)
    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < 16; ++f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
            asm volatile("" : "+a"(acc[f]));
        }
    };

    // ---- pass 1, software pipeline.  A (chunk, sub-kernel) step is two PHASES of 8 frequencies x 4 k8-steps x 4 MFMAs: phase 0 runs
    // the frequencies 0..7 (rows 0, 1 of V, U half 0) while the threads transform rows 2, 3 of the same window into U half 1; phase 1
    // runs the frequencies 8..15 while they transform rows 0, 1 of the NEXT step's window into U half 0 (when that step starts a new
    // chunk, the raw halo is replaced first: registers -> LDS, one extra barrier).  Every LDS / VALU instruction of the transform
    // sits in a fixed slot between two MFMAs (36 of the 64 slots of a phase), so the matrix pipe never waits for it; one barrier per
    // phase.  Weight fragments: ring of 16 steps, loaded 8 steps (32 MFMAs, ~2000 cycles) ahead across phase, sub-kernel and chunk
    // boundaries (left alone, the compiler sinks the loads to one step ahead and the single wave per SIMD stalls on L2).
    float4 hreg[C::NHF];
    const int a_off = li * C::LDU + kh * 4;
    f32x4 bw[16];
    if (do_p1) {
#pragma unroll
    for (int r = 0; r < C::NHF; ++r) {               // first halo: offsets computed and loads issued slot by slot
        halo_off_init(r);
        halo_load_slot(hreg, c0, r);
    }
#pragma unroll
    for (int st8 = 0; st8 < WLEAD1; ++st8)   // the first steps of the first phase: frequencies 0, 1, .. (x 4 k8-steps) of plane 0
        bw[st8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)(st8 & 3) * st_v, (unsigned)c0 * panel_b + (unsigned)(st8 >> 2) * freq_b, 0));
    __builtin_amdgcn_sched_barrier(0);
    zero_acc();
    __builtin_amdgcn_sched_barrier(0);
    halo_store(hreg);
    asm volatile("" : "+v"(bias));
    __syncthreads();
    if (c1 - c0 > 1) {
        halo_load(hreg, c0 + 1);
    }
    {   // rows 0, 1 of the first window (not overlapped)
        f32x4 wv4[2][4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (0 * C::HW + jx) * C::LDR]);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (1 * C::HW + jx) * C::LDR]);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(&s_raw[t_raw + (2 * C::HW + jx) * C::LDR]);
            wv4[0][jx] = d0 - d2;
            wv4[1][jx] = d1 + d2;
        }
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 0) * 32 * C::LDU + t_u]) = wv4[il][0] - wv4[il][2];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 1) * 32 * C::LDU + t_u]) = wv4[il][1] + wv4[il][2];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 2) * 32 * C::LDU + t_u]) = wv4[il][2] - wv4[il][1];
            *reinterpret_cast<f32x4*>(&s_u[(4 * il + 3) * 32 * C::LDU + t_u]) = wv4[il][1] - wv4[il][3];
        }
    }
    __syncthreads();
    // A fragments of the first two steps of the first phase; every phase requests those of the phase after it (steps 30, 31)
    f32x4 av[4];
    av[0] = *reinterpret_cast<const f32x4*>(&s_u[a_off]);
    av[1] = *reinterpret_cast<const f32x4*>(&s_u[a_off + 8]);

    // one (chunk, sub-kernel) step = the two phases.  LAST (compile time): the last sub-kernel of a chunk, whose second phase transforms the
    // first window of the NEXT chunk -- the raw halo is replaced in between, spread over the free side slots so that the matrix pipe never
    // waits for it: phase 0 reads the old halo in slots 2..13, then one barrier (slot 14: every wave is done with the old halo) and one
    // ds_write_b128 of the new halo per slot from slot 40 on (the phase's barrier in step 30 publishes it); phase 1 issues one global
    // load of the chunk after next per slot from slot 40 on.  (Before: 10 / 20 stores + a barrier + the loads in one clump in slot 0 of
    // phase 1, exposed: +4 % per block with the 12 x 52 halo of the run geometry.)
    auto p1_step = [&](auto sub_c, int ch, bool more, unsigned chunk_b, unsigned next_b) {
        constexpr int sub = decltype(sub_c)::value;
        constexpr bool LAST = sub == C::NSUB - 1;
        constexpr int sub_n = LAST ? 0 : sub + 1;
        const bool repl = LAST && ((C::NDIR > 0 && !UNIT) ? true : more);      // chunks are staged round-robin over the passes (0 .. nch-1, then 0 ..
        int cn = ch + 2;                                                        // again for pass 2a, 2b): the registers hold the chunk after next
        if (UNIT) cn = cn < c1 ? cn : c1 - 1;
        if (cn >= nch) cn -= nch;
        if (cn >= nch) cn -= nch;
        const unsigned plane_b = chunk_b + (unsigned)(sub * 16) * freq_b;                       // plane sub * 16 + 0 of this chunk
        const unsigned nplane_b = (LAST ? next_b : chunk_b) + (unsigned)(sub_n * 16) * freq_b;     // plane 0 of the next step
        const int src_cur = t_raw + ((3 * (sub >> 1)) * C::HW + 3 * (sub & 1)) * C::LDR;
        const int src_nxt = t_raw + ((3 * (sub_n >> 1)) * C::HW + 3 * (sub_n & 1)) * C::LDR;
        // (the two phases as two instances of a generic lambda rather than a loop: `#pragma unroll` on a body of this size is a request
        //  the unroller may decline -- it did for the rectangle / multi-slab forms once the clustered transform was in the body, and the
        //  register arrays indexed by r, s turned into 3 KB of scratch per lane)
        auto phase_r = [&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            // side work of this phase: rows (2, 3) of the current window (r = 0) / rows (0, 1) of the next one (r = 1)
            constexpr int q = r ^ 1;                                    // V row pair produced
            const int src = (r == 0 ? src_cur : src_nxt) + q * C::HW * C::LDR;      // d rows q .. q + 2
            float* const udst = s_u + (q * 8) * 32 * C::LDU + t_u;
            f32x4 dd[3][4], wv[2][4], vvs[8];
#pragma unroll
            for (int s = 0; s < 32; ++s) {                              // step = (frequency r * 8 + s / 4, k8-step s % 4)
                const int f = r * 8 + (s >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 3][e], bw[s & 15][e], acc[f], 0, 0, 0);
                    if (e == 0) {                                       // weights of step s + lead
                        const int sn = s + WLEAD1;
                        unsigned so;
                        if (sn < 32) so = plane_b + (unsigned)(r * 8 + (sn >> 2)) * freq_b;
                        else if (r == 0) so = plane_b + (unsigned)(8 + ((sn - 32) >> 2)) * freq_b;
                        else so = nplane_b + (unsigned)((sn - 32) >> 2) * freq_b;
                        if (!(PMX_ABLATE & 2))
                        bw[sn & 15] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)(sn & 3) * st_v, so, 0));
                        // THE barrier of the phase sits here, eight MFMAs before its end: U half q is complete (its last store is in slot
                        // 39; LAST, r = 0: and the new raw halo, slot 40 + NHF - 1 <= 60), every wave has read all it needs of U half r (the
                        // fragments of steps 30, 31 were requested at steps 28, 29).  The MFMAs that follow have their operands in
                        // registers, and the next phase's first two fragments are requested behind it -- at the phase boundary itself
                        // nothing waits (with the barrier there, the first MFMA of every phase waited for the barrier AND an LDS read)
                        if (s == 30 && !(PMX_ABLATE & 8)) __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (e == 1) {                                // A fragment of step s + 2 (steps 30, 31: of the next phase's steps 0, 1)
                        const int fn = s + 2 < 32 ? r * 8 + ((s + 2) >> 2) : q * 8, sn = (s + 2) & 3;
                        if (!(PMX_ABLATE & 4))
                        av[(s + 2) & 3] = *reinterpret_cast<const f32x4*>(&s_u[fn * 32 * C::LDU + a_off + sn * 8]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (!(PMX_ABLATE & 1)) {                     // transform slot t
                        const int t = 2 * s + (e - 2);
                        if (t >= 2 && t < 14) {                         // 12 reads: d rows q .. q + 2, column by column
                            const int jx = (t - 2) / 3, ri = (t - 2) % 3;
                            if (PMX_ABLATE & 16) asm volatile("" : "=v"(dd[ri][jx]));
                            else
                            dd[ri][jx] = *reinterpret_cast<const f32x4*>(&s_raw[src + (ri * C::HW + jx) * C::LDR]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (LAST && t == 14 && r == 0) {         // every wave has read what it needs of the old halo
                            if (repl) __syncthreads();
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (t == VT || t == VT + 8) {
                            // the 16 packed adds of B^T d (slot 16) / of (.) B (slot 24) in one gap each; the eight U stores follow one per slot
#pragma unroll
                            for (int k8 = 0; k8 < 8; ++k8) {
                                if (t == VT) {
                                    const int jx = k8 >> 1, wi = k8 & 1;
                                    if (PMX_ABLATE & 32) asm volatile("" : "=v"(wv[wi][jx]));
                                    else if (q == 0) wv[wi][jx] = wi == 0 ? pk_sub4(dd[0][jx], dd[2][jx]) : pk_add4(dd[1][jx], dd[2][jx]);
                                    else wv[wi][jx] = wi == 0 ? pk_sub4(dd[1][jx], dd[0][jx]) : pk_sub4(dd[0][jx], dd[2][jx]);
                                } else {
                                    const int il = k8 >> 2, jv = k8 & 3;
                                    if (PMX_ABLATE & 32) asm volatile("" : "=v"(vvs[k8]));
                                    else vvs[k8] = jv == 0 ? pk_sub4(wv[il][0], wv[il][2]) : jv == 1 ? pk_add4(wv[il][1], wv[il][2]) : jv == 2 ? pk_sub4(wv[il][2], wv[il][1]) : pk_sub4(wv[il][1], wv[il][3]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (t >= VT + 9 && t < VT + 17) {
                            const int k8 = t - (VT + 9), il = k8 >> 2, jv = k8 & 3;
                            if (PMX_ABLATE & 64) asm volatile("" :: "v"(vvs[k8]));
                            else *reinterpret_cast<f32x4*>(&udst[(4 * il + jv) * 32 * C::LDU]) = vvs[k8];
                            __builtin_amdgcn_sched_barrier(0);
                        } else if (LAST && ((t >= 40 && t < 40 + (C::NHF < 20 ? C::NHF : 20)) || (C::NHF > 20 && t == 15))) {
                            // the halo of the next chunk -> LDS (r = 0) / of the one after it -> registers: slots 40 .. 59 (before the
                            // phase's barrier in step 30), a 21st staging slot (merged tails, 7x7) in slot 15 right behind the barrier
                            const int hs = t == 15 ? 20 : t - 40;       // (a constant once the loops are unrolled)
                            if (repl) {
                                if (r == 0) { if (!(PMX_ABLATE & 128)) halo_store_slot(hreg, hs); }
                                else if (!(PMX_ABLATE & 256)) halo_load_slot(hreg, cn, hs);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        };
        phase_r(std::integral_constant<int, 0>{});
        phase_r(std::integral_constant<int, 1>{});
    };
    static_assert(C::NHF <= 21, "halo slots");
    PMX_T(2);
    for (int ch = c0; ch < c1; ++ch) {
        const bool more = ch + 1 < c1;
        const unsigned chunk_b = (unsigned)ch * panel_b;
        const unsigned next_b = (unsigned)(more ? ch + 1 : ch) * panel_b;
        // (the sub-kernels are unrolled: a run-time loop over three of them + a peeled last one made the register allocator shuttle
        //  accumulator tiles between the two copies with v_accvgpr_mov + s_nop 15)
        p1_step(std::integral_constant<int, 0>{}, ch, more, chunk_b, next_b);
        if constexpr (C::NSUB > 1) {
            p1_step(std::integral_constant<int, 1>{}, ch, more, chunk_b, next_b);
            p1_step(std::integral_constant<int, 2>{}, ch, more, chunk_b, next_b);
            p1_step(std::integral_constant<int, 3>{}, ch, more, chunk_b, next_b);
        }
    }

    }   // do_p1

    // ---- output transform Y = A^T M A per (tile, channel): y[2 * i + j] = pixel (i, j) of the tile
    // (two registers at a time through the packed-fp32 adds -- the same roundings as scalar adds, half the VALU instructions)
    f32x16 y[4];
    if (!UNIT || do_p1) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            f32x2 t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 m0 = {acc[0 + j][2 * rp], acc[0 + j][2 * rp + 1]}, m1 = {acc[4 + j][2 * rp], acc[4 + j][2 * rp + 1]};
                const f32x2 m2 = {acc[8 + j][2 * rp], acc[8 + j][2 * rp + 1]}, m3 = {acc[12 + j][2 * rp], acc[12 + j][2 * rp + 1]};
                t0[j] = pk_add2(pk_add2(m0, m1), m2);
                t1[j] = pk_sub2(pk_sub2(m1, m2), m3);
            }
            const f32x2 o0 = pk_add2(pk_add2(t0[0], t0[1]), t0[2]), o1 = pk_sub2(pk_sub2(t0[1], t0[2]), t0[3]);
            const f32x2 o2 = pk_add2(pk_add2(t1[0], t1[1]), t1[2]), o3 = pk_sub2(pk_sub2(t1[1], t1[2]), t1[3]);
            y[0][2 * rp] = o0[0]; y[0][2 * rp + 1] = o0[1]; y[1][2 * rp] = o1[0]; y[1][2 * rp + 1] = o1[1];
            y[2][2 * rp] = o2[0]; y[2][2 * rp + 1] = o2[1]; y[3][2 * rp] = o3[0]; y[3][2 * rp + 1] = o3[1];
        }
    } else {
        // a unit block without pass 1 (row 6 / column 6 / tap (6, 6)): the transform of all-zero accumulators is +0 -- no accumulator
        // is zeroed or read for it
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) y[pp][reg] = 0.f;
    }

    if (C::NDIR > 0) {
        // ---- pass 2 (7x7): the 13 taps of row 6 and column 6.  Row 6 = two 1x3 sub-kernels (kx 0..2, 3..5) as 1-D F(2,3) along x:
        // per output row i of the tile 4 horizontal frequencies -> 8 planes (i * 4 + f), both sub-kernels summed in the same planes;
        // column 6 = two 3x1 sub-kernels (ky 0..2, 3..5) as 1-D F(2,3) along y: 8 planes (j * 4 + f); tap (6, 6) direct into the four
        // pixel planes.  2 * 8 + 2 * 8 + 4 = 36 products per tile and channel pair instead of 13 * 4 = 52.  Weight planes (same
        // [plane][chunk32][cout_pad][32] array as pass 1): 64 + sub * 4 + f (row 6), 72 + sub * 4 + f (column 6), 80 (tap (6, 6)).
        // Pass 2a per chunk: D phase (16 steps x 4 MFMAs from the raw halo; the threads transform row-6 sub-kernel 0 meanwhile), H0 and
        // H1 phases (16 steps x 8 MFMAs; during H0 sub-kernel 1 is transformed, during H1 the raw halo of the next chunk replaces this
        // one); then y += A^T-transform of the row planes.  Pass 2b per chunk: V0, V1 phases; then y += transform of the column planes.
        constexpr int PH = 64, PV = 72, PD = 80;
        constexpr int HBAR = C::NHF > 20 ? C::NHF : 20;      // H1 / V1: the slot of the barrier behind the halo stores (slots 0 .. NHF - 1)
        f32x16 e8[8];
        f32x4 bwr[8], bd[4], av[4];
        auto zero8 = [&]() {
#pragma unroll
            for (int pl = 0; pl < 8; ++pl)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) e8[pl][r16] = 0.f;
        };
        auto wload = [&](int plane, unsigned chb, int st) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off + (unsigned)st * st_v, chb + (unsigned)plane * freq_b, 0));
        };
        // 1-D transform of this thread's (tile, 4 channels): two lines (output rows i for the row class, output columns j for the
        // column class) of 4 samples each -> (d0 - d2, d1 + d2, d2 - d1, d1 - d3), slot by slot
        f32x4 dd[2][4], vv[2][4];
        auto side1d = [&](int t, int base, int line_stride, int samp_stride, float* udst) {     // t compile-time
            if (t >= 2 && t < 10) {
                const int l = (t - 2) >> 2, c = (t - 2) & 3;
                dd[l][c] = *reinterpret_cast<const f32x4*>(&s_raw[base + l * line_stride + c * samp_stride]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (t == 12) {                        // the 16 packed adds in one gap
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int l = k8 >> 2, f = k8 & 3;
                    vv[l][f] = f == 0 ? pk_sub4(dd[l][0], dd[l][2]) : f == 1 ? pk_add4(dd[l][1], dd[l][2]) : f == 2 ? pk_sub4(dd[l][2], dd[l][1]) : pk_sub4(dd[l][1], dd[l][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (t >= 20 && t < 28) {
                const int l = (t - 20) >> 2, f = (t - 20) & 3;
                *reinterpret_cast<f32x4*>(&udst[(l * 4 + f) * 32 * C::LDU]) = vv[l][f];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // one 8-plane phase: 16 steps s = f * 4 + st, per step the two lines l = 0, 1 x 4 MFMAs; A fragments one step ahead, weights
        // four steps ahead (ring of 8; `wnext(s)` loads step s of whatever phase follows), side slots m = 2, 3, 6, 7 of every step
        auto phase8 = [&](const float* ub, int wplane, unsigned chb, auto&& wnext, auto&& side) {
            av[0] = *reinterpret_cast<const f32x4*>(&ub[a_off]);
            av[1] = *reinterpret_cast<const f32x4*>(&ub[4 * 32 * C::LDU + a_off]);
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int f = s2 >> 2, st = s2 & 3;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int l = m >> 2, e = m & 3;
                    e8[l * 4 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(s2 & 1) * 2 + l][e], bwr[s2 & 7][e], e8[l * 4 + f], 0, 0, 0);
                    if (m == 0) {
                        constexpr int L2 = WLEAD2;
                        if (PMX_ABLATE & 2) {}
                        else if (s2 + L2 < 16) bwr[(s2 + L2) & 7] = wload(wplane + ((s2 + L2) >> 2), chb, (s2 + L2) & 3);
                        else wnext(s2 + L2 - 16);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if ((m == 1 || m == 5) && s2 + 1 < 16) {
                        const int ln = m == 1 ? 0 : 1, fn = (s2 + 1) >> 2, sn = (s2 + 1) & 3;
                        if (!(PMX_ABLATE & 4))
                        av[((s2 + 1) & 1) * 2 + ln] = *reinterpret_cast<const f32x4*>(&ub[(ln * 4 + fn) * 32 * C::LDU + a_off + sn * 8]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if ((m == 2 || m == 3 || m == 6 || m == 7) && !(PMX_ABLATE & 1)) {
                        side(4 * s2 + (m < 4 ? m - 2 : m - 4));
                    }
                }
            }
        };
        float* const u0 = s_u + t_u;
        float* const u1 = s_u + 8 * 32 * C::LDU + t_u;
        const int a2_off = tile_px(li) * C::LDR + kh * 4;

        // ================= pass 2a: tap (6, 6) + row 6 =================
        if (do_p2a) {
        zero8();                                    // (pass 1's last phase already staged the raw halo of chunk 0 again)
#pragma unroll
        for (int st = 0; st < 4; ++st) bd[st] = wload(PD, 0u, st);
        if (UNIT) {                                 // standalone: stage chunk 0, keep chunk 1 in the registers
            {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
            halo_store(hreg);
            __syncthreads();
            if (nch > 1) {
                halo_load(hreg, 1);
            }
        }
        for (int ch = 0; ch < nch; ++ch) {
            const bool more = ch + 1 < nch;
            const unsigned chb = (unsigned)ch * panel_b, nxb = (unsigned)(more ? ch + 1 : ch) * panel_b;
            // ---- D phase: step q = st * 4 + p; side: row-6 sub-kernel 0 -> U half 0; weights of H0's first four steps
            {
                f32x4 ad[4];
                ad[0] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + 0) * C::HW + 6 + 0) * C::LDR]);
                ad[1] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + 0) * C::HW + 6 + 1) * C::LDR]);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int st = q >> 2, pp = q & 3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!UNIT) y[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[q & 3][e], bd[st][e], y[pp], 0, 0, 0);
                        if (e == 0) {
                            if (q < WLEAD2) { bwr[q] = wload(PH + (q >> 2), chb, q & 3); __builtin_amdgcn_sched_barrier(0); }
                        } else if (e == 1) {
                            if (q + 2 < 16) {
                                const int qn = q + 2, pn = qn & 3, sn = qn >> 2;
                                ad[qn & 3] = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + (pn >> 1)) * C::HW + 6 + (pn & 1)) * C::LDR + sn * 8]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                            side1d(2 * q + (e - 2), t_raw + (6 * C::HW + 0) * C::LDR, C::HW * C::LDR, C::LDR, u0);
                        }
                    }
                }
            }
            __syncthreads();                        // U half 0 = row-6 sub-kernel 0
            // The raw halo is replaced without ever stopping the matrix pipe and without the staging registers meeting the transform's:
            // ---- H0: side = sub-kernel 1 (kx 3..5) -> U half 1 (the last reads of this chunk's raw halo, slots 2..27); then the halo of the
            // next chunk (chunk 0 again after the last one: pass 2b starts from it) global -> registers, one load per slot from slot 28
            // on; afterwards H1's first weights
            const int cnx = more ? ch + 1 : 0;
            phase8(s_u, PH + 0, chb,
                   [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PH + 4 + (s2n >> 2), chb, s2n & 3); },
                   [&](int t) {
                       if (t >= 28 && t < 28 + C::NHF) { halo_load_slot(hreg, cnx, t - 28); __builtin_amdgcn_sched_barrier(0); }
                       else side1d(t, t_raw + (6 * C::HW + 3) * C::LDR, C::HW * C::LDR, C::LDR, u1);
                   });
            __syncthreads();                        // U half 1 = row-6 sub-kernel 1; nobody reads the old raw halo any more
            // ---- H1: side = registers -> LDS, one ds_write_b128 per slot from slot 0 on, a barrier (slot 20), then column-6 sub-kernel 0
            // of the new chunk -> U half 0 (free: H1 reads half 1; needed after the last chunk, otherwise unused and overwritten by the
            // next D phase -- unconditional, because a branch per slot would cut the schedule into pieces); afterwards the next chunk's
            // tap-(6,6) weights
            phase8(s_u + 8 * 32 * C::LDU, PH + 4, chb,
                   [&](int s2n) {
                       if (s2n < 4) bd[s2n] = wload(PD, nxb, s2n);
                       bwr[(s2n + 16) & 7] = wload(PV + (s2n >> 2), 0u, s2n & 3);  // pass 2b's first weights (used after the last chunk)
                   },
                   [&](int t) {
                       if (t < C::NHF) { halo_store_slot(hreg, t); __builtin_amdgcn_sched_barrier(0); }
                       else if (t == HBAR) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
                       else if (t >= HBAR + 2) side1d(t - HBAR, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
                   });
        }
        __syncthreads();                            // U half 0 = column-6 sub-kernel 0 of chunk 0
        // y += A^T-transform of the row planes: e8[i * 4 + f]
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                y[i2 * 2 + 0][reg] = y[i2 * 2 + 0][reg] + ((e8[i2 * 4 + 0][reg] + e8[i2 * 4 + 1][reg]) + e8[i2 * 4 + 2][reg]);
                y[i2 * 2 + 1][reg] = y[i2 * 2 + 1][reg] + ((e8[i2 * 4 + 1][reg] - e8[i2 * 4 + 2][reg]) - e8[i2 * 4 + 3][reg]);
            }

        }   // do_p2a

        // ================= pass 2b: column 6 =================
        if (do_p2b) {
        zero8();
        if (UNIT) {                                 // standalone: stage chunk 0, first weights, sub-kernel 0 of chunk 0 (not overlapped)
            {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
#pragma unroll
            for (int s2n = 0; s2n < WLEAD2; ++s2n) bwr[s2n] = wload(PV + (s2n >> 2), 0u, s2n & 3);
            halo_store(hreg);
            __syncthreads();
            if (nch > 1) {
                halo_load(hreg, 1);
            }
#pragma unroll
            for (int t = 0; t < 28; ++t) side1d(t, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
            __syncthreads();
        }
        // one chunk; MORE (compile time): another chunk follows -- its raw halo replaces this one inside V0 (barrier in slot 10 after
        // the last reads of the old halo, one ds_write_b128 per slot from slot 28 on), the halo after it is requested inside V1
        auto p2b_chunk = [&](auto more_c, int ch) {
            constexpr bool MORE = decltype(more_c)::value;
            // y rests during pass 2b: pin it to the accumulator file (64 of its registers are free here) -- left in VGPRs next to the 20-slot
            // halo it pushed the halo addresses to scratch, each reload with an s_waitcnt vmcnt(0) that also drains the weight ring
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) asm volatile("" : "+a"(y[pp]));
            const unsigned chb = (unsigned)ch * panel_b, nxb = (unsigned)(MORE ? ch + 1 : ch) * panel_b;
            // ---- V0: side = sub-kernel 1 (ky 3..5) -> U half 1, then the next chunk's halo global -> registers (slots 28 ..)
            phase8(s_u, PV + 0, chb,
                   [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 4 + (s2n >> 2), chb, s2n & 3); },
                   [&](int t) {
                       if (MORE && t >= 28 && t < 28 + C::NHF) { halo_load_slot(hreg, ch + 1, t - 28); __builtin_amdgcn_sched_barrier(0); }
                       else side1d(t, t_raw + (3 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u1);
                   });
            __syncthreads();
            // ---- V1: side = registers -> LDS (slots 0 ..), barrier (slot 20), the next chunk's sub-kernel 0 -> U half 0
            if constexpr (MORE)
                phase8(s_u + 8 * 32 * C::LDU, PV + 4, chb,
                       [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 0 + (s2n >> 2), nxb, s2n & 3); },
                       [&](int t) {
                           if (t < C::NHF) { halo_store_slot(hreg, t); __builtin_amdgcn_sched_barrier(0); }
                           else if (t == HBAR) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
                           else if (t >= HBAR + 2) side1d(t - HBAR, t_raw + (0 * C::HW + 6) * C::LDR, C::LDR, C::HW * C::LDR, u0);
                       });
            else
                phase8(s_u + 8 * 32 * C::LDU, PV + 4, chb,
                       [&](int s2n) { bwr[(s2n + 16) & 7] = wload(PV + 0 + (s2n >> 2), nxb, s2n & 3); },
                       [&](int) {});
            __syncthreads();
        };
        static_assert(HBAR + 28 <= 60 && 28 + C::NHF <= 60, "halo slots");
        for (int ch = 0; ch < nch - 1; ++ch) p2b_chunk(std::true_type{}, ch);
        p2b_chunk(std::false_type{}, nch - 1);
        // y += A^T-transform of the column planes: e8[j * 4 + f]
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                y[0 * 2 + j2][reg] = y[0 * 2 + j2][reg] + ((e8[j2 * 4 + 0][reg] + e8[j2 * 4 + 1][reg]) + e8[j2 * 4 + 2][reg]);
                y[1 * 2 + j2][reg] = y[1 * 2 + j2][reg] + ((e8[j2 * 4 + 1][reg] - e8[j2 * 4 + 2][reg]) - e8[j2 * 4 + 3][reg]);
            }
        }   // do_p2b

        if (do_pd) {
            // ================= unit mode: tap (6, 6) over all chunks, straight from the raw halo =================
            f32x4 bdn[4];
            {
#pragma unroll
                for (int r = 0; r < C::NHF; ++r) halo_off_init(r);
            }
            halo_load(hreg, 0);
#pragma unroll
            for (int st = 0; st < 4; ++st) bd[st] = wload(PD, 0u, st);
            for (int ch = 0; ch < nch; ++ch) {
                if (ch) __syncthreads();
                halo_store(hreg);
                __syncthreads();
                const int cn = ch + 1 < nch ? ch + 1 : ch;
                halo_load(hreg, cn);
#pragma unroll
                for (int st = 0; st < 4; ++st) bdn[st] = wload(PD, (unsigned)cn * panel_b, st);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int st = q >> 2, pp = q & 3;
                    const f32x4 ad = *reinterpret_cast<const f32x4*>(&s_raw[a2_off + ((6 + (pp >> 1)) * C::HW + 6 + (pp & 1)) * C::LDR + st * 8]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[e], bd[st][e], y[pp], 0, 0, 0);
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) bd[st] = bdn[st];
            }
        }
    }

    // ---- bias, ReLU, (pool), store.  The stores go through a buffer resource that spans exactly this image's output (32-bit byte offsets,
    // an out-of-range offset = the store is dropped): no 64-bit address arithmetic and no branch per store.  (Written with pointers and
    // `if (inside) out[...] = v` this epilogue compiled to ~1100 instructions -- 270 quarter-rate integer multiplies / 64-bit mads, 80
    // exec-mask branches -- and took 4-5 us of a block that lasts 23 us (conv2_1) to 200 us (7x7): tools/block_timing.py.)
    PMX_T(5);
    const bool nok = n < G.cout;
    const int Hp = H >> 1, Wp = W >> 1;
    const int opix = POOL ? Hp * Wp : H * W;                         // output pixels per image
    const int ldc_b = a.ldc * 4;
    if (UNIT && GEOM) {
        // unit mode of a run (the part-filled last block of an image): compact slab [image][block of the launch][tile][pixel][cout_pad];
        // conv_wino_tail_reduce_kernel adds the units in order and drops the tiles past the end of the map
        // (MERGE: bslab * 1 + 0 = the block of the stream)
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(G.out + (size_t)(bslab * (MERGE ? 1 : a.run_nb) + trem) * (32 * 4) * a.ldc, 0,
                                                                                  (unsigned)(32 * 4 * ldc_b), 0x00020000);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mr = (reg & 3) + 8 * (reg >> 2) + 4 * kh;      // Winograd tile of this register row
            const int o = (int)__umul24(mr * 4, ldc_b) + n * 4;
            // (through float temporaries: __builtin_bit_cast applied directly to the vector element y[k][reg] compiled to element 0 for every reg)
            const float y00 = y[0][reg], y01 = y[1][reg], y10 = y[2][reg], y11 = y[3][reg];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), srsrc, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), srsrc, o + ldc_b, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), srsrc, o + 2 * ldc_b, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), srsrc, o + 3 * ldc_b, 0, 0);
        }
    } else {
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(G.out + (s_pix_out + (size_t)bimg * opix) * a.ldc, 0, (unsigned)(opix * ldc_b), 0x00020000);
        const int n_b = nok ? n * 4 : -1;                            // (a lane without a real output channel: every offset out of range)
        // run geometry: (tile row, tile column) of the lane's first tile by one division, then stepped from register row to register row
        // (the rows of a lane are the tiles tb + 0, 1, 2, 3, 8, 9, ...: steps of 1 or 5 < 23, at most one wrap)
        const unsigned tb = (unsigned)(t0 + 4 * kh);
        int rty = (int)(tb / (unsigned)C::RUN_TX), rtx = (int)(tb - (unsigned)rty * (unsigned)C::RUN_TX);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mrc = (reg & 3) + 8 * (reg >> 2);              // (+ 4 kh): Winograd tile of this register row
            float y00 = y[0][reg], y01 = y[1][reg], y10 = y[2][reg], y11 = y[3][reg];
            int gy, gx;
            if (GEOM) {
                if (reg) {
                    rtx += (reg & 3) ? 1 : 5;
                    const bool wrap = rtx >= C::RUN_TX;
                    rtx = wrap ? rtx - C::RUN_TX : rtx;
                    rty += wrap ? 1 : 0;
                }
                gy = 2 * rty; gx = x0 + 2 * rtx;                     // tiles past the end of the map land on rows >= H
            } else {
                gy = y0 + 2 * ((mrc >> 3)) ; gx = x0 + 2 * ((mrc & 7) + 4 * kh);
            }
            if (POOL) {
                float v = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias;
                if (a.relu) v = fmaxf(v, 0.f);
                const int py = gy >> 1, px = gx >> 1;
                const int o = (py < Hp && px < Wp && nok) ? (int)__umul24(__umul24(py, Wp) + px, ldc_b) + n_b : -1;      // (24-bit operands: full-rate multiplies)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, o, 0, 0);
            } else {
                y00 += bias; y01 += bias; y10 += bias; y11 += bias;
                if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                const int o00 = (int)__umul24(__umul24(gy, W) + gx, ldc_b) + n_b;      // (24-bit operands: full-rate multiplies; < 2^31 by the launcher's check)
                // (run geometry: the map is a whole number of 46-column slabs and a tile column is < 23, so both pixel columns are inside)
                const bool r0 = gy < H && nok, r1 = gy + 1 < H && nok, c0v = GEOM ? true : gx < W, c1v = GEOM ? true : gx + 1 < W;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), orsrc, (r0 && c0v) ? o00 : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), orsrc, (r0 && c1v) ? o00 + ldc_b : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), orsrc, (r1 && c0v) ? o00 + W * ldc_b : -1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), orsrc, (r1 && c1v) ? o00 + (W + 1) * ldc_b : -1, 0, 0);
            }
        }
    }
    PMX_T(6);
}

template <int KS, int POOL, int UNIT = 0>
static int launch_wino(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, 0>;
    ConvArgs a = a0;
    PMX_CHECK(!!a.pool == !!POOL, PMX_ERR_INVALID, "conv wino: pool mismatch");
    PMX_CHECK(!POOL || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)(a.H + C::TH + KS) * (a.W + C::TW + KS) * a.lda * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit byte offsets");
    PMX_CHECK((long long)(a.H + 2) * (a.W + 2) * a.ldc * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: output image too large for 32-bit byte offsets");
    a.tiles_x = (a.W + C::TW - 1) / C::TW;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.run_j0 = a.run_nb = 0;
    auto kern = conv_wino_kernel<KS, POOL, UNIT, 0>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    if (UNIT) { a.ngroups = groups; PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan"); }
    // heterogeneous launch: the caller built the segment table (conv_build_segs: sizes checked there) and says how many tiles it holds
    PMX_CHECK(a.nseg == 0 || (!UNIT && a.segs && a.seg_tiles > 0), PMX_ERR_INVALID, "conv wino: segments only in plain mode, with a table");
    const unsigned gx = a.nseg ? (unsigned)a.seg_tiles : (unsigned)(a.tiles_x * a.tiles_y * a.B);
    dim3 grid(gx, (unsigned)(a.cout_pad / 128), (unsigned)(groups * (UNIT ? a.ksplit : 1)));
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// a.nch = input channels / 32 (chunks of the Winograd kernel), a.g[].w = transformed weights (pmx_api.hip::pack_wino)
// a.ksplit > 1: unit mode -- a.ksplit = ceil(nch / g) (+ 3 for 7x7: row 6, column 6, tap (6, 6)) slabs at a.g[].out + unit * a.slab_stride, g = a.kbounds
int conv_wino_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    if (a.ksplit > 1) return ks == 7 ? launch_wino<7, 0, 1>(a, groups, stream) : launch_wino<3, 0, 1>(a, groups, stream);
    if (ks == 7) return launch_wino<7, 0>(a, groups, stream);
    return a.pool ? launch_wino<3, 1>(a, groups, stream) : launch_wino<3, 0>(a, groups, stream);
}

// run geometry (46-pixel-wide maps): the blocks [a.run_j0, a.run_j0 + a.run_nb) of every image, 32 consecutive Winograd tiles each
// Events for the NEXT run-geometry launch of this thread (profile mode 2): handed to hipExtLaunchKernelGGL, which stamps them from the
// dispatch's own completion signal -- the kernel's execution time without the two barrier packets of hipEventRecord (~6 us of idle stream
// per pair, 25 pairs per step inside bench.py's timed region)
static thread_local hipEvent_t g_launch_ev0 = nullptr, g_launch_ev1 = nullptr;
void conv_set_launch_events(hipEvent_t e0, hipEvent_t e1) { g_launch_ev0 = e0; g_launch_ev1 = e1; }

template <int KS, int POOL, int UNIT, int GEOM>
static int launch_wino_run_g(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, GEOM>;
    ConvArgs a = a0;
    PMX_CHECK(!!a.pool == !!POOL && a.W % C::RUN_W == 0 && a.W > 0, PMX_ERR_INVALID, "conv wino runs: map width must be a multiple of %d (W %d)", C::RUN_W, a.W);
    PMX_CHECK(!POOL || (a.H % 2 == 0 && a.W % 2 == 0), PMX_ERR_INVALID, "conv: pooled layer needs even H, W");
    a.run_nslab = a.W / C::RUN_W;
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.H * a.W * a.lda * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: image too large for 32-bit offsets");
    PMX_CHECK((long long)(a.H + 2) * (a.W + 2) * a.ldc * 4 < (1ll << 31), PMX_ERR_INVALID, "conv: output image too large for 32-bit byte offsets");
    const int nblk = (C::RUN_TX * ((a.H + 1) / 2) + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES;
    PMX_CHECK(a.run_j0 >= 0 && a.run_nb >= 1 && a.run_j0 + a.run_nb <= nblk, PMX_ERR_INVALID, "conv wino runs: blocks [%d, %d) of %d", a.run_j0, a.run_j0 + a.run_nb, nblk);
    a.tiles_x = a.tiles_y = 0;
    PMX_CHECK(GEOM == 2 || a.run_nslab == 1, PMX_ERR_INVALID, "conv wino runs: single-slab kernel on a %d-wide map", a.W);
    auto kern = conv_wino_kernel<KS, POOL, UNIT, GEOM>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    if (UNIT) { a.ngroups = groups; PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan"); }
    dim3 grid((unsigned)(a.run_nb * a.B * a.run_nslab), (unsigned)(a.cout_pad / 128), (unsigned)(groups * (UNIT ? a.ksplit : 1)));
    if (g_launch_ev0) {
        hipExtLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, g_launch_ev0, g_launch_ev1, 0, a);
        g_launch_ev0 = g_launch_ev1 = nullptr;
    } else
        hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int KS, int POOL, int UNIT>
static int launch_wino_run(const ConvArgs& a, int groups, hipStream_t stream)
{
    return a.W == 2 * PMX_WINO_RUN_TX ? launch_wino_run_g<KS, POOL, UNIT, 1>(a, groups, stream) : launch_wino_run_g<KS, POOL, UNIT, 2>(a, groups, stream);
}

template <int KS>
static int launch_wino_merged(const ConvArgs& a0, int groups, hipStream_t stream)
{
    using C = WinoCfg<KS, 3>;
    ConvArgs a = a0;
    PMX_CHECK(wino_tail_mergeable(a.B, a.H, a.W, a.lda) && a.run_j0 == C::RUN_TX * ((a.H + 1) / 2) / PMX_WINO_RUN_TILES, PMX_ERR_INVALID,
              "conv wino merged tails: not a mergeable tail (B %d, %d x %d, full blocks %d)", a.B, a.H, a.W, a.run_j0);
    PMX_CHECK(a.cout_pad % 128 == 0, PMX_ERR_INVALID, "conv wino: cout_pad %d not a multiple of 128", a.cout_pad);
    PMX_CHECK((long long)a.B * a.H * a.W * a.lda * 4 < (1ll << 31), PMX_ERR_INVALID, "conv wino merged tails: batch too large for 32-bit offsets");
    PMX_CHECK(a.ksplit >= 2 && a.ksplit <= 8 && a.kbounds >= 1, PMX_ERR_INVALID, "conv wino: bad unit plan");
    a.run_nslab = 1; a.run_nb = wino_tail_merged_blocks(a.B, a.H); a.tiles_x = a.tiles_y = 0; a.ngroups = groups;
    auto kern = conv_wino_kernel<KS, 0, 1, 3>;
    static bool attr_set[PMX_MAX_DEVICES] = {};
    if (int rc = conv_allow_big_lds(reinterpret_cast<const void*>(kern), attr_set)) return rc;
    dim3 grid((unsigned)a.run_nb, (unsigned)(a.cout_pad / 128), (unsigned)(groups * a.ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int conv_wino_merged_tail_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    return ks == 7 ? launch_wino_merged<7>(a, groups, stream) : launch_wino_merged<3>(a, groups, stream);
}

int conv_wino_run_launch(const ConvArgs& a, int ks, int groups, hipStream_t stream)
{
    if (a.ksplit > 1) return ks == 7 ? launch_wino_run<7, 0, 1>(a, groups, stream) : launch_wino_run<3, 0, 1>(a, groups, stream);      // (the combine pools)
    if (ks == 7) return launch_wino_run<7, 0, 0>(a, groups, stream);
    return a.pool ? launch_wino_run<3, 1, 0>(a, groups, stream) : launch_wino_run<3, 0, 0>(a, groups, stream);
}

// ---- combine of the unit-mode slabs of a block range of the run geometry (see WinoTailReduceArgs) -------------------------------------------
// One thread per (image, block, tile, pixel of the tile, 4 output channels): slabs added in unit order (left to right), then bias, ReLU --
// the arithmetic of conv_splitk_reduce_kernel on the compact slab layout of conv_wino_kernel<KS, 0, 1, 1>.
__global__ __launch_bounds__(256) void conv_wino_tail_reduce_kernel(const WinoTailReduceArgs r)
{
    const int g = blockIdx.z;
    const float* slabs = g ? r.slabs[1] : r.slabs[0];
    const float* bias = g ? r.bias[1] : r.bias[0];
    float* out = g ? r.out[1] : r.out[0];
    const int cout = g ? r.cout[1] : r.cout[0];
    const int c4n = cout >> 2;
    // pooled layers: one thread per TILE (its four pixels are the pooling window), else one per pixel
    const int ppt = r.pool ? 1 : 4;
    const int nt = PMX_WINO_RUN_TX * ((r.H + 1) >> 1) - r.run_j0 * PMX_WINO_RUN_TILES;     // merged: tail tiles per image
    const long long total = r.merged ? (long long)r.B * nt * ppt * c4n : (long long)r.B * r.nslab * r.run_nb * (PMX_WINO_RUN_TILES * ppt) * c4n;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4n) * 4;
    const long long q = i / c4n;                        // ((image-slab * run_nb + block) * 32 + tile) [* 4 + pixel]
    const int k = r.pool ? 0 : (int)(q & 3);
    const long long qt = r.pool ? q : q >> 2;           // (image-slab * run_nb + block) * 32 + tile
    int b, sx0, t;
    if (r.merged) {                                     // qt = stream position: image * nt + tail tile (= block of the stream * 32 + row)
        b = (int)(qt / nt); sx0 = 0;
        t = r.run_j0 * PMX_WINO_RUN_TILES + (int)(qt - (long long)b * nt);
    } else {
        const int m = (int)(qt & (PMX_WINO_RUN_TILES - 1));
        const long long bj = qt >> 5;
        const int jl = (int)(bj % r.run_nb);
        const long long bs = bj / r.run_nb;
        b = (int)(bs / r.nslab); sx0 = (int)(bs % r.nslab) * (2 * PMX_WINO_RUN_TX);
        t = (r.run_j0 + jl) * PMX_WINO_RUN_TILES + m;
    }
    const int ty = t / PMX_WINO_RUN_TX, tx = t - ty * PMX_WINO_RUN_TX;
    float4 best;
    for (int kk = 0; kk < (r.pool ? 4 : 1); ++kk) {
        const int kq = r.pool ? kk : k;
        const float* src = slabs + (qt * 4 + kq) * r.ld_slab + c;
        float4 acc = *reinterpret_cast<const float4*>(src);
        for (int s = 1; s < r.S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long long)s * r.slab_stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (kk == 0) best = acc;
        else { best.x = fmaxf(best.x, acc.x); best.y = fmaxf(best.y, acc.y); best.z = fmaxf(best.z, acc.z); best.w = fmaxf(best.w, acc.w); }
    }
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    best.x += bv.x; best.y += bv.y; best.z += bv.z; best.w += bv.w;
    if (r.relu) { best.x = fmaxf(best.x, 0.f); best.y = fmaxf(best.y, 0.f); best.z = fmaxf(best.z, 0.f); best.w = fmaxf(best.w, 0.f); }
    if (r.pool) {
        const int Hp = r.H >> 1, Wp = r.W >> 1, py = ty, px = (sx0 >> 1) + tx;
        if (py >= Hp || px >= Wp) return;
        *reinterpret_cast<float4*>(out + (((long long)b * Hp + py) * Wp + px) * r.ldc + c) = best;
    } else {
        const int gy = 2 * ty + (k >> 1), gx = sx0 + 2 * tx + (k & 1);
        if (gy >= r.H || gx >= r.W) return;                 // tiles past the end of the map, the odd last row
        *reinterpret_cast<float4*>(out + (((long long)b * r.H + gy) * r.W + gx) * r.ldc + c) = best;
    }
}

int conv_wino_tail_reduce(const WinoTailReduceArgs& r, int groups, hipStream_t stream)
{
    PMX_CHECK(r.cout[0] % 4 == 0 && (groups < 2 || r.cout[1] == r.cout[0]) && r.ldc % 4 == 0 && r.ld_slab % 4 == 0, PMX_ERR_INVALID,
              "winograd tail reduce: channel counts / strides must be multiples of 4");
    static_assert(PMX_WINO_RUN_TILES == 32, "tile index bits");
    const int nt = PMX_WINO_RUN_TX * ((r.H + 1) / 2) - r.run_j0 * PMX_WINO_RUN_TILES;
    const long long total = (r.merged ? (long long)r.B * nt * (r.pool ? 1 : 4) : (long long)r.B * r.nslab * r.run_nb * (PMX_WINO_RUN_TILES * (r.pool ? 1 : 4))) * (r.cout[0] / 4);
    hipLaunchKernelGGL(conv_wino_tail_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, (unsigned)groups), dim3(256), 0, stream, r);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// Kernel selection for the 3x3 / 7x7 layers: which form of the Winograd kernel (plain, run geometry with unit-mode tails, unit mode) or the
// direct kernels a launch takes -- a pure function of the launch shape, the device's CU count and the context options, kept apart from the
// C ABI (pmx_api.hip) so that it can be read and tested on its own (tests/test_gpu_selection.py times the alternatives it chooses between).
// The direct kernels' variant / split-K choice lives next to those kernels (conv_mfma.hip: conv_pick_variant, conv_pick_ksplit).
#include "pmx_common.h"

#include <math.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

bool wino_eligible(int ks, int cin_pad, int cout_pad) { return (ks == 3 || ks == 7) && cin_pad % 32 == 0 && cout_pad % 128 == 0; }

// Which form a 3x3 / 7x7 layer takes: 0 = direct kernels (+ split-K), 1 = the Winograd kernel, 2 = the Winograd kernel in unit mode
// (*unit_g = chunks per pass-1 unit).  Blocks are equal and run one per CU, so the plain kernel costs ceil(blocks / CUs) rounds however
// full the last one is; the unit mode costs the same work at finer grain (no round quantisation, ~0.8 of the plain kernel's in-round
// efficiency) plus the slab traffic of the combine kernel; the direct kernels win when neither fills the chip
// (tools/wino_batch_sweep.py).  A forced split-K option (never, n slices, an explicit plan) is a statement about the direct kernels:
// no unit mode then.
// Measured block times of the Winograd kernel (MI355X, one block per CU), microseconds per 32-channel chunk of a plain block / per phase of
// 128 MFMAs per wave, and the fixed cost of a unit block (prologue: first halo + weights exposed; epilogue: output transform, slab store)
static const double WINO_T7_CHUNK_US = 50.0, WINO_T3_CHUNK_US = 9.3, WINO_PHASE_US = 4.0, WINO_UNIT_FIXED_US = 8.0;

// Makespan (microseconds) of the unit-mode launch of `nblk` part-filled blocks (images x 128-channel blocks) cut into pass-1 units of g chunks
// (+ row 6, column 6, tap (6, 6) for 7x7): blocks are dispatched unit by unit (blockIdx.z = unit * groups + group) to the CU that frees first
static double wino_tail_makespan(int ks, int nch, int g, long long nblk, int ncu)
{
    std::vector<double> unit_us;
    for (int c = 0; c < nch; c += g) unit_us.push_back(std::min(g, nch - c) * (ks == 7 ? 8.0 : 2.0) * WINO_PHASE_US + WINO_UNIT_FIXED_US);
    if (ks == 7) {
        unit_us.push_back(nch * 2.0 * WINO_PHASE_US + WINO_UNIT_FIXED_US);      // row 6
        unit_us.push_back(nch * 2.0 * WINO_PHASE_US + WINO_UNIT_FIXED_US);      // column 6
        unit_us.push_back(nch * 0.5 * WINO_PHASE_US + WINO_UNIT_FIXED_US);      // tap (6, 6)
    }
    std::vector<double> cu((size_t)ncu, 0.0);        // min-heap of the CUs' free times
    auto cmp = [](double a, double b) { return a > b; };
    double end = 0.0;
    for (double t : unit_us)
        for (long long b = 0; b < nblk; ++b) {
            std::pop_heap(cu.begin(), cu.end(), cmp);
            cu.back() += t;
            end = std::max(end, cu.back());
            std::push_heap(cu.begin(), cu.end(), cmp);
        }
    return end;
}

// The unit plan of `nblk` blocks (images x 128-channel blocks x groups) that finishes first: chunks per pass-1 unit g -> S = ceil(nch / g)
// (+ 3 for 7x7) units, 2 <= S <= 8, by the dispatch simulation above + the combine's slab reads; forced_g > 0 pins g (if it is a valid plan).
// A pure function of its arguments: memoised -- run_conv asks for every layer of every forward, and the simulation is a heap walk over
// units x blocks per candidate.  Returns the makespan in microseconds, *g_out = 0 when no plan exists (nch too small).
static double wino_best_unit_g(int ks, int nch, long long nblk, int ncu, int forced_g, int* g_out)
{
    const int extra = ks == 7 ? 3 : 0;
    struct Key { int ks, nch, forced; long long nblk, ncu; bool operator<(const Key& k) const { return std::tie(ks, nch, forced, nblk, ncu) < std::tie(k.ks, k.nch, k.forced, k.nblk, k.ncu); } };
    static std::mutex mu;
    static std::map<Key, std::pair<double, int>> memo;
    const Key key{ks, nch, forced_g, nblk, ncu};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(key);
        if (it != memo.end()) { *g_out = it->second.second; return it->second.first; }
    }
    double best = 1e30;
    int best_g = 0;
    for (int gg = 1; gg <= nch; ++gg) {
        const int SS = (nch + gg - 1) / gg + extra;
        if (SS < 2 || SS > 8 || (gg > 1 && (nch + gg - 2) / (gg - 1) + extra == SS)) continue;      // (same unit count as a smaller g: skip)
        const double t = wino_tail_makespan(ks, nch, gg, nblk, ncu) + 2.0 * SS;   // + the combine's slab reads
        if (t < best) { best = t; best_g = gg; }
    }
    if (forced_g > 0 && forced_g <= nch) {
        const int SS = (nch + forced_g - 1) / forced_g + extra;
        if (SS >= 2 && SS <= 8) { best_g = forced_g; best = wino_tail_makespan(ks, nch, best_g, nblk, ncu) + 2.0 * SS; }
    }
    std::lock_guard<std::mutex> lk(mu);
    memo[key] = std::make_pair(best, best_g);
    *g_out = best_g;
    return best;
}

// *run = 1: mode 1 in the run geometry (46-pixel-wide maps, no pool); *tail_g > 0: its part-filled last blocks in unit mode, g chunks per
// pass-1 unit (launch_wino_run)
int wino_select(const WinoSelectOpts& o, int ks, int cin_pad, int cout_pad, int cout, int ldc, int images, int H, int W, int pool, int* unit_g,
                int* run, int* tail_g)
{
    *unit_g = 0; *run = 0; *tail_g = 0;
    if (o.conv_algo < 1 || o.precision != 0 || o.forced_variant >= 0 || !wino_eligible(ks, cin_pad, cout_pad)) return 0;
    const int nch = cin_pad / 32, extra = ks == 7 ? 3 : 0;     // 7x7: + row 6, column 6, tap (6, 6)
    const long long ncu = conv_num_cus(), nb = cout_pad / 128;
    int g = 0, S = 0;        // the plan with as many units as 8 slabs allow (tails under conv_algo 2; whole launches with wino_unit_g = -1)
    int gu = 0;              // the plan of a whole launch in unit mode (single images, small batches)
    if (o.ksplit == 0 && cout % 4 == 0 && ldc % 4 == 0 && nch >= 2) {
        const int nu1_max = 8 - extra;
        g = (nch + nu1_max - 1) / nu1_max;
        const int nu1 = (nch + g - 1) / g;
        if (nu1 >= 2) S = nu1 + extra; else g = 0;
        // Until round 6 whole launches used that plan too -- which for the wide 3x3 layers of one 368 x 368 image is 576 unit blocks = 2.25
        // rounds of the 256 CUs paid as 3 (conv4_2: 72 tile blocks x 8 units of 2 chunks).  Now the plan the dispatch simulation finishes
        // first (the one the tails use): conv4_2 as 3 units of 6 / 6 / 4 chunks = 216 blocks in ONE round, conv4_1 / conv4_3 as 3 / 6 units;
        // the 46 x 46 7x7 layers keep their 7 units (252 blocks), the 46 x 62 ones take 5 (240 blocks instead of 336).
        gu = g;
        if (g && o.wino_unit_g >= 0) {
            const long long ub = (long long)((H + 7) / 8) * ((W + 15) / 16) * images * nb;         // unit mode keeps the rectangles
            int bg = 0;
            (void)wino_best_unit_g(ks, nch, ub, (int)ncu, o.wino_unit_g, &bg);
            if (bg) gu = bg;
        }
    }
    // run geometry: blocks of 32 consecutive tiles; the part-filled last block of an image (if any) can run as S unit blocks
    // (3x3 layers with fewer than four chunks keep the rectangles: their blocks are so short -- conv2_1: 24 us -- that the larger first
    //  halo of a run costs more than the padding it saves: measured +3 %)
    const bool geom_run = W > 0 && W % (PMX_WINO_RUN_TX * 2) == 0 && (!pool || H % 2 == 0) && o.wino_geom != 0 &&
                          (ks == 7 || nch >= 4 || W == PMX_WINO_RUN_TX * 2 || o.conv_algo == 2);
    const long long nslab = geom_run ? W / (PMX_WINO_RUN_TX * 2) : 1;
    const int ntiles = PMX_WINO_RUN_TX * ((H + 1) / 2), nblk = (ntiles + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES, nfull = ntiles / PMX_WINO_RUN_TILES;
    const bool tail_ok = geom_run && g > 0 && nfull >= 1 && nfull < nblk && o.wino_tail != 0;
    if (o.conv_algo == 2) {                  // tests: the plain kernel on every eligible layer (the tail in units only when asked for)
        *run = geom_run;
        if (tail_ok && o.wino_tail == 1) *tail_g = g;
        return 1;
    }
    if (o.conv_algo == 3) { *unit_g = gu; return gu ? 2 : 0; }             // tests: unit mode wherever it applies
    // cost of the plain kernel in rounds of one block per CU (equal blocks: a round costs the same however full it is)
    const long long blocks = geom_run ? (long long)nblk * nslab * images * nb : (long long)((H + 7) / 8) * ((W + 15) / 16) * images * nb;
    const long long rounds = (blocks + ncu - 1) / ncu;
    double plain_cost = (double)rounds;
    int tg = 0;
    if (tail_ok) {
        // the full blocks as whole rounds + the tail as unit blocks (best g by a dispatch simulation) + two more launches and the combine
        const double t_block = nch * (ks == 7 ? WINO_T7_CHUNK_US : WINO_T3_CHUNK_US);
        const long long main_rounds = ((long long)nfull * nslab * images * nb + ncu - 1) / ncu;
        // (merged tails: the tail tiles of a group's images as one stream, 32 per block)
        const int grp = std::max(1, o.groups);
        const bool merge = o.wino_tail_merge != 0 && wino_tail_mergeable(images / grp, H, W, o.lda);
        const long long tail_blocks = merge ? (long long)grp * wino_tail_merged_blocks(images / grp, H) * nb : (long long)images * nslab * nb;
        int best_g = 0;
        const double best = wino_best_unit_g(ks, nch, tail_blocks, (int)ncu, o.wino_tail_g, &best_g);
        const double cost = (double)main_rounds + (best + 10.0) / t_block;
        if (best_g && (o.wino_tail == 1 || cost < plain_cost)) { plain_cost = cost; tg = best_g; }
    }
    if (g) {
        const double t_block = nch * (ks == 7 ? 52e-6 : 18.5e-6);                 // one plain block (measured), seconds
        // (3x3 units are short -- 256 MFMAs per chunk against ~10 us of block prologue / epilogue: 3/4 of the 7x7 figure)
        const double eff = o.wino_unit_eff / 100.0 * (ks == 7 ? 1.0 : 0.75);
        const long long ublocks = (long long)((H + 7) / 8) * ((W + 15) / 16) * images * nb;       // unit mode keeps the rectangles
        const double est_unit = (double)ublocks / (ncu * eff) +
                                (double)ublocks * (S + 1) * 65536.0 / 3.0e12 / t_block + 0.03;      // in rounds of the plain kernel
        // (a handful of unit blocks cannot beat the direct kernels' split-K, which cuts the same work into more and smaller blocks:
        //  184 x 248 input, one image: 12 tiles x 7 units = 84 blocks took 1.9 ms per forward against 1.4 ms)
        // (the choice of the MODE is made with the many-unit plan, as before round 6; the plan itself is gu)
        if (est_unit < plain_cost && ublocks * S * 2 >= ncu) { *unit_g = gu; return 2; }
    }
    if (blocks * 100 >= (long long)o.wino_min_fill * rounds * ncu) { *run = geom_run; *tail_g = tg; return 1; }
    return 0;
}

// A launch of the plain kernel whose last round of the CUs is part-filled pays a whole round for it (equal one-per-CU blocks).  Where that
// costs more than running the images of that round in the finer-grained forms, run_conv cuts the batch in two: the first *n0 images --
// whole rounds -- through the plain kernel, the rest through whatever the selection gives THEIR count (unit mode with the plan of
// wino_best_unit_g; a landscape batch of 8: 5 images = 240 blocks in one round + 3 images as 720 unit blocks instead of 384 blocks = 2
// rounds).  Returns n0 (0: no split).  Part of the arithmetic like every launch form: the results of image i then depend on which side of
// n0 it lies (profile label "...@<first image>+<count>").  Option "wino_split" = 0 turns it off.
int wino_split_images(const WinoSelectOpts& o, int ks, int cin_pad, int cout_pad, int cout, int ldc, int B, int groups, int H, int W, int pool)
{
    if (!o.wino_split || B < 2 || o.conv_algo != 1) return 0;
    int ug = 0, run = 0, tg = 0;
    if (wino_select(o, ks, cin_pad, cout_pad, cout, ldc, B * groups, H, W, pool, &ug, &run, &tg) != 1) return 0;
    const int nch = cin_pad / 32;
    const long long ncu = conv_num_cus(), nb = cout_pad / 128;
    const long long rects = (long long)((H + 7) / 8) * ((W + 15) / 16);
    long long bpi;                   // blocks of the main launch per image (all groups)
    if (run) {
        const int ntiles = PMX_WINO_RUN_TX * ((H + 1) / 2), nblk = (ntiles + PMX_WINO_RUN_TILES - 1) / PMX_WINO_RUN_TILES, nfull = ntiles / PMX_WINO_RUN_TILES;
        bpi = (long long)(tg ? nfull : nblk) * (W / (PMX_WINO_RUN_TX * 2)) * nb * groups;
    } else
        bpi = rects * nb * groups;
    const long long total = bpi * B, full = total / ncu;
    if (full < 1 || total % ncu == 0) return 0;
    const int n0 = (int)(full * ncu / bpi);
    if (n0 < 1 || n0 >= B) return 0;
    const int rem = B - n0;
    int ug2 = 0, run2 = 0, tg2 = 0;
    const int m2 = wino_select(o, ks, cin_pad, cout_pad, cout, ldc, rem * groups, H, W, pool, &ug2, &run2, &tg2);
    const double t_block = nch * (ks == 7 ? WINO_T7_CHUNK_US : WINO_T3_CHUNK_US);
    double rem_us;
    if (m2 == 2) {
        int g = 0;
        rem_us = wino_best_unit_g(ks, nch, rects * rem * groups * nb, (int)ncu, o.wino_unit_g, &g) + 10.0;      // + the combine launch
        if (!g) return 0;
    } else if (m2 == 1)
        rem_us = (double)((bpi * rem + ncu - 1) / ncu) * t_block;
    else
        return 0;
    const double whole_us = (double)(full + 1) * t_block;
    const double split_us = (double)((bpi * n0 + ncu - 1) / ncu) * t_block + rem_us + 10.0;                       // + one more launch
    const double thr = o.wino_split >= 50 ? o.wino_split / 100.0 : 0.97;          // (option value >= 50: the threshold in percent, for measurements)
    return split_us < thr * whole_us ? n0 : 0;
}

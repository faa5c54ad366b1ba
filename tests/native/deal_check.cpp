// Host check of csrc/layout.h's deal (tests/test_host_logic.py compiles and runs it with g++): for equal and unequal shares,
// full and partial grids, every strip goes to exactly one wave, a wave's strips come in increasing order, no wave gets more
// strips than deal_rounds reserves, and equal shares are the plain boustrophedon deal.
#define __host__
#define __device__
#include "layout.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace sucre;

int main() {
    int bad = 0, cases = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (uint32_t blocks : {1u, 7u, 300u, 256u * (uint32_t)(mode ? kClosedWaves : kFitWaves)})
            for (uint32_t n_strips : {1u, 4u, 5119u, 5120u, 32400u, 131072u}) {
                const DealShares sh = deal_shares(mode, blocks);
                const uint32_t W = blocks * 4;
                std::vector<uint32_t> lev(n_strips);
                for (uint32_t s = 0; s < n_strips; ++s) lev[s] = (uint32_t)(65.0 * std::pow(1.0 - (double)s / n_strips, 0.7));
                std::vector<int> seen(n_strips, 0);
                const uint32_t R = deal_rounds(W, n_strips, sh);
                bool equal = true;
                for (uint32_t g = 0; g < sh.G; ++g) equal = equal && sh.p[g] == kDealDen;
                for (uint32_t wid = 0; wid < W; ++wid) {
                    uint32_t last = 0, cnt = 0;
                    const uint32_t K = deal_walk(wid, W, n_strips, sh, [&](uint32_t s) { if (s >= n_strips) ++bad; return s < n_strips ? lev[s] : 0u; },
                                                 [&](uint32_t k, uint32_t s) {
                                                     if (k != cnt || (cnt && s <= last) || s >= n_strips) ++bad;
                                                     else ++seen[s];
                                                     if (equal && s != k * W + ((k & 1u) ? W - 1u - wid : wid)) ++bad;
                                                     last = s; ++cnt;
                                                 });
                    if (K != cnt || K > R) ++bad;
                }
                for (int v : seen) if (v != 1) ++bad;
                ++cases;
            }
    // the light kernels' grids (csrc/light.hip: as many workgroups as are resident, 4 or 5 per CU; deal_shares_resident -- ADVICE round 5):
    // the same properties, and no wave gets more strips than deal_rounds reserves in light_deal_kernel's table
    for (uint32_t blocks : {768u, 1024u, 1280u, 300u})
        for (uint32_t n_strips : {1u, 4800u, 32400u, 131072u}) {
            const DealShares sh = deal_shares_resident(blocks);
            const uint32_t W = blocks * 4, R = deal_rounds(W, n_strips, sh);
            std::vector<uint32_t> lev(n_strips);
            for (uint32_t s = 0; s < n_strips; ++s) lev[s] = (uint32_t)(65.0 * std::pow(1.0 - (double)s / n_strips, 0.7));
            std::vector<int> seen(n_strips, 0);
            for (uint32_t wid = 0; wid < W; ++wid) {
                uint32_t cnt = 0;
                const uint32_t K = deal_walk(wid, W, n_strips, sh, [&](uint32_t s) { return s < n_strips ? lev[s] : 0u; },
                                             [&](uint32_t k, uint32_t s) { if (k != cnt || s >= n_strips) ++bad; else ++seen[s]; ++cnt; });
                if (K != cnt || K > R) ++bad;
            }
            for (int v : seen) if (v != 1) ++bad;
            ++cases;
        }
    // StripEntry.counts: every strip height up to kMaxViews levels with every split into unmasked / masked full chunks that the
    // plan kernel can write (8-bit fields wrapped at 1024 levels)
    for (uint32_t levels = 0; levels <= (uint32_t)kMaxViews; ++levels) {
        const uint32_t nfull = levels >> 2, r = levels & 3u;
        for (uint32_t nu : {0u, nfull / 2u, nfull}) {
            const uint32_t c = strip_counts(nu, nfull - nu, r);
            if (counts_unmasked(c) != nu || counts_masked(c) != nfull - nu || counts_tail(c) != r) ++bad;
        }
        ++cases;
    }
    // equal shares on the full grid reserve ceil(n_strips / W) strips per wave, not G times that
    for (int mode = 0; mode < 2; ++mode) {
        const uint32_t blocks = 256u * (uint32_t)(mode ? kClosedWaves : kFitWaves), W = blocks * 4u;
        const DealShares sh = deal_shares(mode, blocks);
        bool equal = true;
        for (uint32_t g = 0; g < (uint32_t)(mode ? kClosedWaves : kFitWaves); ++g) equal = equal && sh.p[g] == kDealDen;
        if (equal && (sh.G != 1u || deal_rounds(W, 131072u, sh) != (131072u + W - 1u) / W)) ++bad;
        ++cases;
    }
    std::printf("%d cases, %d violations\n", cases, bad);
    return bad != 0;
}

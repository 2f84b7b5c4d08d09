// In-wave search (round 6; VERDICT r05 #1, where it pays): the preamble search of a tile by the K1 wave that has just produced
// it, for rows short enough that a lane's whole row of decisions is still on the chip when its block ends (rows of 16 words =
// BlockSize 512: chip length 8 with one of rtlamr's preambles whose taps all fit the row-as-ring scheme of k2_row.h).
//
// Why here and not at chip 72: at BlockSize 512 a K1 launch is 16 384 short waves in eight rounds, there is no
// lock-step to lose, and the search kernel -- 16 385 waves of ~700 instructions for 4 KiB of bitstream each -- is bound by
// instruction issue wherever it runs: next to K1 (the early search) it costs K1 what it takes alone.  In the K1 wave the row is
// already in LDS (the parked output chunks), nothing is loaded, nothing is waited for, and the sweep is 200 instructions.
//
// What it writes is what k2_search_row writes for the tile (staging slot in stream order, count, group sum, overflow bits);
// lane 63's words whose windows reach past its row -- the next tile belongs to another wave -- are left to k2_row_cleanup.
#pragma once
#include "k1_common.h"
#include "k2_row.h"

namespace amr {

// first word of a row whose taps reach into the row behind it: words below it can be searched without a look-ahead
template <int SL, int KIND, int WPB>
constexpr int k1s_wcut()
{
    constexpr int D = k2r_taps<SL, KIND, WPB>();
    return WPB - 1 - (((D - 1) * SL) >> 5);
}
template <int SL, int KIND, int WPB>
constexpr bool k1s_ok()
{
    return k2r_taps<SL, KIND, WPB>() == (int)kK2WKnownLen[KIND] && k1s_wcut<SL, KIND, WPB>() >= 1;   // the whole preamble in the sweep
}

// Tile T (qt numbering: K1's wave-tile + 1) by one wave; R = this lane's row, chunk c = words 4c .. 4c + 3; lds: 896 words of
// scratch nobody else uses any more.
template <int SL, int KIND, int WPB>
__device__ __forceinline__ void k1s_search_tile(const K1Search &s, uint32_t T, uint32_t lane, K2WRing<WPB / 4> &R, uint32_t *lds, uint32_t lg_bs)
{
    constexpr int D = k2r_taps<SL, KIND, WPB>();
    static_assert(k1s_ok<SL, KIND, WPB>(), "the sweep must apply the whole preamble");
    uint32_t *mylist = lds;                                          // [kK2WList][2]
    uint32_t *cnts = mylist + kK2WList * 2;                          // [64] hits per lane-row
    uint32_t *bases = cnts + 4 * 64;                                 // [64]
    cnts[lane] = 0;
    k2w_v4u X = {0u, 0u, 0u, 0u};                                    // what follows lane 63: not here (clean-up launch)

    // valid word range of this lane's row: n_lo <= R*BS + 32w < n_hi (k2_search_row)
    const int64_t rowbase = ((int64_t)T * 64 + lane - 64) << lg_bs;
    const int64_t lo64 = (s.n_lo - rowbase) >> 5, hi64 = (s.n_hi - rowbase) >> 5;
    const uint32_t w_lo = (uint32_t)(lo64 < 0 ? 0 : lo64 > (int64_t)WPB ? WPB : lo64);
    uint32_t w_hi = (uint32_t)(hi64 < 0 ? 0 : hi64 > (int64_t)WPB ? WPB : hi64);
    constexpr uint32_t w_cut = (uint32_t)k1s_wcut<SL, KIND, WPB>();
    if (lane == 63 && w_hi > w_cut) w_hi = w_cut;

    uint32_t Bc = 0, list_n = 0;
    k2r_groups<SL, WPB, KIND, D, 0>(R, X, 0u, w_lo, w_hi, lane, mylist, list_n, Bc);

    // every candidate is a hit (the sweep applied the whole preamble): hits per lane-row
    const uint32_t n_keep = list_n < (uint32_t)kK2WList ? list_n : (uint32_t)kK2WList;
    for (uint32_t e0 = 0; e0 < n_keep; e0 += 64) {
        const uint32_t e = e0 + lane;
        if (e < n_keep) atomicAdd(&cnts[(mylist[e * 2] >> 8) & 63], (uint32_t)__popc(mylist[e * 2 + 1]));
    }
    const uint32_t val = cnts[lane];
    const uint32_t inc = k2w_wave_scan(val);
    const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
    bases[lane] = inc - val;

    // emit (k2_search_row): the list is word-major across the lanes, ascending inside a lane
    uint32_t run = 0;
    for (uint32_t e0 = 0; e0 < n_keep; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t key = 0, m = 0;
        if (e < n_keep) { key = mylist[e * 2]; m = mylist[e * 2 + 1]; }
        const uint32_t l = (key >> 8) & 63, c = (uint32_t)__popc(m);
        uint32_t slot = bases[l] + (uint32_t)__builtin_amdgcn_ds_bpermute((int)(l << 2), (int)run);
        const uint32_t n = n_keep - e0 < 64u ? n_keep - e0 : 64u;    // wave-uniform
        uint32_t add = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)j), cj = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)j);
            slot += (lane > j && l == lj) ? cj : 0u;
            add += (lane == lj) ? cj : 0u;
        }
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j), mj = (uint32_t)__builtin_amdgcn_readlane((int)m, (int)j);
            const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)j);
            if (lane < 32 && ((mj >> (31 - lane)) & 1)) {
                const uint32_t rank = sj + (lane ? __popc(mj >> (32 - lane)) : 0);
                if (rank < s.cap) s.staging[(size_t)T * s.cap + rank] = (((kj >> 8) & 63) << lg_bs) + ((kj & 0xff) << 5) + lane;
            }
        }
        run += add;
    }
    if (lane == 0) {
        const uint32_t c = total < s.cap ? total : s.cap;
        s.counts[T] = c;
        if (c) atomicAdd(&s.gcnt[(T >> 6) * kGroupStride], c);
        if (total > s.cap) atomicOr(s.overflow, 1u);
        if (list_n > (uint32_t)kK2WList) atomicOr(s.overflow, 2u);
    }
}

}  // namespace amr

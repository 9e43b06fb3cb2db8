// mth_tile_dev.h -- device helpers shared by the PDR + LPMD kernels (mth_pdr_lpmd.hip: dense tile kernel, mth_pdr_wide.hip: hashed-site form).
#pragma once
#include "mth_common.h"

namespace mth {

// ---------------------------------------------------------------------------------------------
// DPP controls (gfx9 family): no LDS traffic, one VALU per step
#define MTH_DPP(v, ctrl, rmask, bctl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, (bctl)))
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {   // wave-uniform result
    v += MTH_DPP(v, 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, true);
    v += MTH_DPP(v, 0x4e /*quad_perm [2,3,0,1]*/, 0xf, true);
    v += MTH_DPP(v, 0x141 /*row_half_mirror*/, 0xf, true);
    v += MTH_DPP(v, 0x140 /*row_mirror*/, 0xf, true);          // every lane: sum of its row of 16
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += MTH_DPP(v, 0x111 /*row_shr:1*/, 0xf, true);
    v += MTH_DPP(v, 0x112 /*row_shr:2*/, 0xf, true);
    v += MTH_DPP(v, 0x114 /*row_shr:4*/, 0xf, true);
    v += MTH_DPP(v, 0x118 /*row_shr:8*/, 0xf, true);
    v += MTH_DPP(v, 0x142 /*row_bcast:15*/, 0xa, false);
    v += MTH_DPP(v, 0x143 /*row_bcast:31*/, 0xc, false);
    return v;
}

constexpr int TILE_BUCKET_SHIFT = 8;   // 256 tiles per bucket

// arguments of the PDR + LPMD tile kernels (mth_pdr_lpmd.hip: dense counters, 4096-bp tiles; mth_pdr_wide.hip: hashed sites, wide tiles)
struct TileArgs {
    const int32_t  *read_start;
    const uint8_t  *read_mapq;
    const uint32_t *cpg_off;
    const uint32_t *cpg_pos;
    const void     *cpg_rel;
    const uint32_t *idx;
    const uint32_t *idx2;         // dense kernel only: second family of its tile-granular index (nullptr: idx is the fine index)
    const DevState *st;
    uint32_t *tile_cnt;
    unsigned long long *bucket;   // per 256-tile bucket: [nbk] rows, then [nbk][4] LPMD partial sums
    uint32_t nbk;
    SiteRec  *scratch;     // TILE_W rows per tile
    int32_t region_beg, region_end, idx_base, max_span;
    uint32_t n_reads, n_cpgs;
    uint32_t min_cov;      // max(pdr_min_depth, 1)
    uint32_t min_cpgs;
    int32_t  min_dist, max_dist;
    uint8_t  pdr_min_qual, lpmd_min_qual, want_pdr, want_lpmd;
    uint32_t tile_w_rt;    // wide form only: positions per tile when it is not the slice width 1 << SHIFT (0: it is) -- see launch_pdr_lpmd
#ifdef MTH_TILE_TRACE
    unsigned long long *trace;     // experiment build: 8 ticks per tile (tools/tile_trace.py)
#endif
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));

// ---- per-slot liveness without compare + select (profiles/r02_ubench_valu.md: v_cmp, v_cndmask, v_min/max are half rate) ----
// The live call slots of a read are a PREFIX (k < n), so everything that depends on "slot k is dead" is a function of n
// alone and comes from two 9-row LDS tables (n = 0..8; built once per pass, 576 bytes):
//   mtab[n][k]   0xffffffff if k < n else 0                       -> masks for the span check / dead-word insertion
//   dtab[n][c]   relpos offsets of the packed fields (below) that push every dead slot 0x400 * (k + 1) past the live
//                ones, so that any distance involving a dead slot is >= 0x400 - 255 and all distances stay >= 0
// ---- windowed pair counts, two pairs per instruction (8-bit relpos) ----
// Slots are packed two per register as 16-bit fields: Q_e = (slot 2e, slot 2e+1), O_e = (slot 2e+1, slot 2e+2), slot 8
// being a dummy that is always dead.  The pairs at call-index gap g are then  later - earlier  with earlier = Q_m and
// later = O_{(g-1)/2+m} (g odd) or Q_{g/2+m} (g even): one 32-bit subtraction gives two distances (no borrows: relpos
// ascends with the slot, dead offsets ascend faster).  With A = D + (0x8000 - min) and B = (0x8000 + max) - D per
// field, bit 15 of A & B says "min <= distance <= max"; the call states sit in bit 15 of a second set of packed words,
// so one xor + and gives "in the window and discordant".  Only full-rate VALU ops (add / sub / and / xor / or / shift).
// hashed-site form for sparse batches (mth_pdr_wide.hip): shift = log2 of the tile width (14 or 15)
void launch_tile_wide(const TileArgs &a, uint32_t ntiles, int shift, bool rel8, hipStream_t s);

struct SlotTabs {
    uint32_t mtab[9][8];
    uint32_t dtab[9][8];
};
__device__ __forceinline__ void slot_tabs_init(SlotTabs &T, const int tid) {
    if (tid < 72) {
        const uint32_t nn = (uint32_t)tid >> 3, c = (uint32_t)tid & 7u;
        T.mtab[nn][c] = c < nn ? 0xffffffffu : 0u;
        auto dead = [&](uint32_t k) { return k >= 8u ? 0x2400u : (k >= nn ? 0x400u * (k + 1u) : 0u); };
        const uint32_t lo = c < 4 ? 2u * c : 2u * (c - 4u) + 1u;       // first slot of the packed register Q_c / O_(c-4)
        T.dtab[nn][c] = dead(lo) | (dead(lo + 1u) << 16);
    }
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   // v_bfi_b32


}  // namespace mth

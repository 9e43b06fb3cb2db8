// mth_wave_tile.h -- device helpers of the one-wave-per-tile kernels (mth_fdrp_wtile.hip, mth_mhl_wtile.hip): wave-wide maxima, ballots as
// scalar masks, a lane write, the wave-level LDS fence (a workgroup is ONE wave there: no s_barrier), and the gather's bucket scan.
#pragma once
#include "mth_tile_dev.h"

namespace mth {

__device__ __forceinline__ uint32_t fw_wave_max(uint32_t v) {   // wave-uniform result
    v = max(v, MTH_DPP(v, 0xb1 /*quad_perm [1,0,3,2]*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x4e /*quad_perm [2,3,0,1]*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x141 /*row_half_mirror*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x140 /*row_mirror*/, 0xf, true));
    // (readlane returns int: the maxima are taken as unsigned)
    return max(max((uint32_t)__builtin_amdgcn_readlane(v, 0), (uint32_t)__builtin_amdgcn_readlane(v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane(v, 32), (uint32_t)__builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ uint32_t fw_wave_scan_max_incl(uint32_t v) {   // values >= 0; lanes outside a row read 0
    v = max(v, MTH_DPP(v, 0x111 /*row_shr:1*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x112 /*row_shr:2*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x114 /*row_shr:4*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x118 /*row_shr:8*/, 0xf, true));
    v = max(v, MTH_DPP(v, 0x142 /*row_bcast:15*/, 0xa, false));
    v = max(v, MTH_DPP(v, 0x143 /*row_bcast:31*/, 0xc, false));
    return v;
}
// lane `idx` of vec <- val (both wave-uniform, SALU-made: no VALU-written SGPR feeds the lane select)
__device__ __forceinline__ uint32_t fw_writelane(uint32_t vec, const uint32_t val, const uint32_t idx) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(idx) : "m0");   // (one SGPR operand per VALU instruction on gfx9: the lane select goes through m0)
    return vec;
}
// the lanes where p holds, as a mask (hip's __ballot goes through a select and a compare: two vector instructions more per call)
__device__ __forceinline__ unsigned long long fw_ballot(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
// the wave's LDS writes are visible to its later reads (one wave per workgroup: no s_barrier)
#define FW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// exclusive scan of the buckets' row counts (one workgroup; a few thousand buckets) -- mth_fdrp_wtile.hip
__global__ __launch_bounds__(1024) void k_fw_bucket_scan(const unsigned long long *__restrict__ bucket, unsigned long long *__restrict__ bucket_pre, const uint32_t nbk);

}  // namespace mth

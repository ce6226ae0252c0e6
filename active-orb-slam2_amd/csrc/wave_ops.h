// Wave64 reductions / scans on DPP for gfx950 (no LDS round trips).  HIP's __shfl_xor / __shfl_up compile to
// ds_bpermute_b32, i.e. one dependent LDS access per step; these helpers use row-level DPP steps and v_readlane.
// All 64 lanes must be active at the call.
#pragma once
#include <hip/hip_runtime.h>

namespace aos2 {

__device__ __forceinline__ int wave_row_sum_i32(int v)   // every lane: the sum of its row of 16 lanes
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return v;
}

__device__ __forceinline__ int wave_sum_i32(int v)       // wave-uniform sum of the 64 lanes
{
    v = wave_row_sum_i32(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ int wave_incl_scan_i32(int v)  // inclusive prefix sum over the lanes
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive scan inside each row
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

}  // namespace aos2

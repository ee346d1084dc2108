// gs_wave_match.h -- which lanes of the wave hold the same radix digit as this one: the stable-rank primitive of the radix scatter
// (gs_radix.hip) and of the emit pass that carries the tile sort's first pass (gs_tiles_fast.hip).  Not installed.
#pragma once

#include "gs_internal.h"

namespace gs
{

// add + the number of set bits of `mask` that belong to lanes BELOW this one: v_mbcnt_lo / v_mbcnt_hi (two instructions, the addend rides
// along; popcount(mask & lowerLanes) is two ANDs and two popcounts and needs the lane's own mask in registers).  `mask` may differ
// from lane to lane.
__device__ __forceinline__ int lanesBelow(unsigned long long mask, int add = 0)
{
    return (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, (unsigned) add));
}

// Inclusive prefix sum over the 64 lanes of the wave in six DPP adds: shifts by 1, 2, 4, 8 inside each row of 16 lanes (a lane with no
// source keeps the 0 it was given), then lane 15 of every row broadcast to the row above it (rows 1 and 3), then lane 31 to the upper
// half.  No LDS: __shfl_up compiles to ds_bpermute_b32 plus five instructions of index arithmetic per step.
__device__ __forceinline__ int waveInclusiveSum(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// The same for 64 values below 2^32 each whose sum needs more than 32 bits (pair totals of the digits: a demand beyond 2^31 must show as
// such): the low 20 and the upper 12 bits are summed apart -- 64 x 2^20 and 64 x 2^12 fit 32 bits -- and put together again.
__device__ __forceinline__ unsigned long long waveInclusiveSumWide(unsigned x)
{
    const unsigned lo = (unsigned) waveInclusiveSum((int) (x & 0xFFFFFu)), hi = (unsigned) waveInclusiveSum((int) (x >> 20));
    return ((unsigned long long) hi << 20) + lo;
}

// Lanes of the wave whose (valid) digit equals this lane's digit.  Four VALU instructions per digit bit on the two 32-bit halves of
// the mask: the bit as 0 / -1 (v_bfe_i32), its ballot (v_cmp_ne), and per half ONE three-input boolean, mask & ~(ballot ^ mine)
// (v_bitop3_b32, truth table 0x90 in the a = 0xF0, b = 0xCC, c = 0xAA convention: 0xF0 & ~(0xCC ^ 0xAA)).  Round 3's form (`lo &=
// ~(bal ^ mine)` with the bit taken by shift-and-subtract) compiled to seven per bit: the compiler re-made every ballot from a 0/1
// select and spent two xors before its own v_bitop3 (GS_MATCH_BITOP3 = 0 keeps it for A/B builds).
#ifndef GS_MATCH_BITOP3
#define GS_MATCH_BITOP3 1
#endif
template <int BITS>
__device__ __forceinline__ unsigned long long matchDigit(int digit, bool valid)
{
    const unsigned long long all = __ballot(valid);
    unsigned lo = (unsigned) all, hi = (unsigned) (all >> 32);
#pragma unroll
    for (int bit = 0; bit != BITS; ++bit)
    {
        // lanes outside `valid` may vote anything: they were removed from the mask above and stay removed
#if GS_MATCH_BITOP3
        // all ones when my bit is set: v_bfe_i32 (written as such: the builtin is expanded into a shift pair and not matched back)
        unsigned mine;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mine) : "v"(digit), "n"(bit));
        const unsigned long long bal = __ballot(mine != 0u);
        lo = __builtin_amdgcn_bitop3_b32(lo, (unsigned) bal, mine, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (unsigned) (bal >> 32), mine, 0x90);
#else
        const unsigned long long bal = __ballot(((unsigned) digit & (1u << bit)) != 0u);
        const unsigned mine = 0u - (((unsigned) digit >> bit) & 1u); // all ones when my bit is set
        lo &= ~((unsigned) bal ^ mine);
        hi &= ~((unsigned) (bal >> 32) ^ mine);
#endif
    }
    const unsigned long long m = ((unsigned long long) hi << 32) | lo;
    return valid ? m : 0ull;
}

} // namespace gs

// gs_wave_match.h -- which lanes of the wave hold the same radix digit as this one: the stable-rank primitive of the radix scatter
// (gs_radix.hip) and of the emit pass that carries the tile sort's first pass (gs_tiles_fast.hip).  Not installed.
#pragma once

#include "gs_internal.h"

namespace gs
{

// Lanes of the wave whose (valid) digit equals this lane's digit.  Written on the two 32-bit halves of the mask with
// xnor/and against the ballot of each digit bit: seven VALU instructions per bit and no scalar temporaries beyond the
// ballot itself (the obvious `m &= b ? bal : ~bal` makes the compiler keep every ballot alive and spill SGPRs).
template <int BITS>
__device__ __forceinline__ unsigned long long matchDigit(int digit, bool valid)
{
    const unsigned long long all = __ballot(valid);
    unsigned lo = (unsigned) all, hi = (unsigned) (all >> 32);
#pragma unroll
    for (int bit = 0; bit != BITS; ++bit)
    {
        // lanes outside `valid` may vote anything: they were removed from the mask above and stay removed
        const unsigned long long bal = __ballot(((unsigned) digit & (1u << bit)) != 0u);
        const unsigned mine = 0u - (((unsigned) digit >> bit) & 1u); // all ones when my bit is set
        lo &= ~((unsigned) bal ^ mine);
        hi &= ~((unsigned) (bal >> 32) ^ mine);
    }
    const unsigned long long m = ((unsigned long long) hi << 32) | lo;
    return valid ? m : 0ull;
}

} // namespace gs

// lds_atomic_order.hip -- does a returning LDS atomic add (ds_add_rtn_u32) of one wave instruction hand out its old values in LANE
// order among the lanes that hit the same address?  It would make a stable radix rank ONE LDS instruction per key instead of the
// ~63 VALU instructions of the ballot match (gs_radix.hip: matchDigit<9>).  Answer on MI355X (round 3): NO -- 69 % of the adds come
// back out of lane order with 2 bins, 2 % with 512 (whenever lanes collide), so the scatter kernels keep their ballot ranks.
// Every wave draws digits from a hash, adds 1 to its own counter of
// that digit and compares the returned value with the number of lower lanes holding the same digit (from ballots); 4 waves per
// workgroup hammer the LDS side by side, several workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_order.hip -o tools/micro/lds_atomic_order && tools/micro/lds_atomic_order
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(int iterations, int bins, unsigned long long* mismatches, unsigned long long* checked)
{
    __shared__ unsigned cnt[4][512];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long bad = 0, done = 0;
    const unsigned long long lower = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int it = 0; it != iterations; ++it)
    {
        for (int d = lane; d < 512; d += 64) cnt[wave][d] = 0;
        __builtin_amdgcn_wave_barrier();
        unsigned expectBase[4] = {0, 0, 0, 0};
        (void) expectBase;
        // four rows back to back, like the scatter kernel's ITEMS rows: a later row must see the earlier rows' adds
        unsigned seenBefore = 0;
        unsigned digitOf[4];
        for (int r = 0; r != 4; ++r)
        {
            unsigned h = (unsigned) (blockIdx.x * 9781u + it * 6271u + r * 101u + t * 2654435761u);
            h ^= h >> 15;
            h *= 2246822519u;
            h ^= h >> 13;
            // skewed digits now and then: many lanes on one address
            const unsigned digit = ((it + r) % 5 == 0) ? (h % 3u) : (h % (unsigned) bins);
            digitOf[r] = digit;
            const bool active = (h >> 20) % 7u != 0u; // some lanes sit a row out
            unsigned old = 0;
            if (active) old = atomicAdd(&cnt[wave][digit], 1u);
            // lanes with my digit, this row: ballot match over 9 bits
            unsigned long long m = __ballot(active);
            for (int b = 0; b != 9; ++b)
            {
                const unsigned long long bal = __ballot((digit >> b) & 1u);
                m &= ((digit >> b) & 1u) ? bal : ~bal;
            }
            // adds of earlier rows to this digit by ANY lane of the wave: count them through the counter itself before this row
            // (kept in registers: the sum over earlier rows of the popcount of their match masks for THIS lane's current digit is
            // not available cheaply, so earlier rows are checked through a second counter array below)
            (void) seenBefore;
            if (active)
            {
                // old must be (adds of earlier rows to this digit) + (lower lanes of this row with this digit)
                const unsigned inRow = (unsigned) __popcll(m & lower);
                // earlier rows' adds to this digit: recompute by replaying their digits through shuffles
                unsigned earlier = 0;
                for (int q = 0; q != r; ++q)
                {
                    // how many lanes had digitOf[q] == digit in row q (and were active there)?
                    unsigned hq = (unsigned) (blockIdx.x * 9781u + it * 6271u + q * 101u + t * 2654435761u);
                    hq ^= hq >> 15;
                    hq *= 2246822519u;
                    hq ^= hq >> 13;
                    const bool activeQ = (hq >> 20) % 7u != 0u;
                    for (int src = 0; src != 64; ++src)
                    {
                        const unsigned dq = (unsigned) __shfl((int) digitOf[q], src);
                        const int aq = __shfl(activeQ ? 1 : 0, src);
                        earlier += (aq && dq == digit) ? 1u : 0u;
                    }
                }
                if (old != earlier + inRow) ++bad;
                ++done;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(checked, done);
}

int main()
{
    unsigned long long *bad = nullptr, *checked = nullptr;
    (void) hipMalloc((void**) &bad, 8);
    (void) hipMalloc((void**) &checked, 8);
    int status = 0;
    for (int bins : {2, 16, 128, 512})
    {
        (void) hipMemset(bad, 0, 8);
        (void) hipMemset(checked, 0, 8);
        hipLaunchKernelGGL(probe, dim3(256 * 8), dim3(256), 0, 0, 64, bins, bad, checked);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        unsigned long long b = 0, c = 0;
        (void) hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
        (void) hipMemcpy(&c, checked, 8, hipMemcpyDeviceToHost);
        std::printf("bins %3d: %llu returning LDS atomic adds checked, %llu out of lane order\n", bins, c, b);
        if (b != 0) status = 1;
    }
    return status;
}

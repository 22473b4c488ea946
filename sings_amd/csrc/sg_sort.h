// Sorting of one tile's (depth_bits << 32 | gaussian_id) keys in LDS, shared by the forward composite kernel (which
// sorts the short lists of its own tile) and the long-list kernels of sg_binning.hip.  Keys are unique (the Gaussian
// id is in the low word), so the ascending order is exactly the stable order of upstream's (tile, depth) radix sort.
#pragma once
#include "sg_common.h"

#define SG_WSORT_MAX 1024      // longest list a composite workgroup sorts itself (8 KiB of keys in the staging buffer)
static_assert(SG_WSORT_MAX == SG_TILE_KEY_CAP, "a row of SgBin::tile_keys holds what a compositing workgroup sorts");
#define SG_RANKSORT_MAX 128    // up to here: rank sort (every thread counts the smaller keys); beyond: one-wave bitonic

// One WAVE sorts s[0, n2) (n2 a power of two <= SG_WSORT_MAX, padded with ~0) -- no workgroup barriers.
__device__ __forceinline__ void sg_bitonic_wave(uint64_t *__restrict__ s, int n2, int lane)
{
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                const bool up = (i & k) == 0;
                const uint64_t a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
}

// A 256-thread workgroup sorts the n <= SG_WSORT_MAX keys at `src` into s[0, n) (LDS, >= SG_WSORT_MAX entries;
// `tmp`: another SG_RANKSORT_MAX entries).  Contains workgroup barriers: every thread must call it; n is uniform.
//  * n <= 128: rank sort.  Thread e < n counts the keys smaller than its own by walking the (broadcast-read) list: n
//    independent LDS reads per thread instead of the ~28 dependent compare-exchange rounds of a bitonic network -- a
//    third of the latency at the mean list length of cfg3 (96) for the same number of vector instructions, and two
//    waves share the work;
//  * 129 .. 256: wave 0 runs the bitonic network (n^2 comparisons would cost more vector issue than they save latency);
//  * 257 .. 1024: the same network on the whole workgroup.
__device__ __forceinline__ void sg_sort_short_list(const uint64_t *__restrict__ src, int n, uint64_t *__restrict__ s,
                                                   uint64_t *__restrict__ tmp, int tid)
{
    if (n <= SG_RANKSORT_MAX) {
        uint64_t key = 0;
        if (tid < n) { key = src[tid]; tmp[tid] = key; }
        __syncthreads();
        if (tid < n) {
            int rank = 0;
            int j = 0;
            for (; j + 2 <= n; j += 2) {
                const ulonglong2 kk = *(const ulonglong2 *)(tmp + j);          // one 16-B broadcast read, two keys
                rank += (kk.x < key) + (kk.y < key);
            }
            if (j < n) rank += tmp[j] < key;
            s[rank] = key;
        }
        __syncthreads();
        return;
    }
    int n2 = 1; while (n2 < n) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) s[i] = i < n ? src[i] : ~0ull;
    __syncthreads();
    if (n2 <= 256) {
        if (tid < 64) sg_bitonic_wave(s, n2, tid);
        __syncthreads();
        return;
    }
    // 257 .. 1024 entries (an avatar's typical tile): the bitonic network on all four waves, n2 / 512 comparators per thread and
    // stage.  In round 1 these lists went to sg_tile_sort_kernel -- one 1024-thread workgroup per tile, ~1200 of them on an avatar
    // frame, 60-90 us between the scan and the composite; here a tile's sort overlaps the compositing of the other tiles on its CU.
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                const bool up = (i & k) == 0;
                const uint64_t a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            __syncthreads();
        }
}

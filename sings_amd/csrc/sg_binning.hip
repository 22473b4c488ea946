// Tile binning: produces, for every 16x16 tile, the depth-ordered list of Gaussian ids.
//
// Replaces InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs + identifyTileRanges
// of the upstream rasterizer.  The result is bit-identical to a stable
// sort of the (tile << 32 | depth_bits) keys generated in Gaussian order: entries of one
// tile with equal depth bits come from different Gaussians and a stable sort keeps them in
// ascending Gaussian id, so the order is exactly the lexicographic order of
// (tile, depth_bits, gaussian_id).  MI355X-first formulation (integer, HBM-light):
//   1. count  : one lane per Gaussian adds 1 to tile_count[t] for every tile of its rectangle
//   2. scan   : one 1024-thread workgroup turns counts into [start,end) ranges, R = total
//   3. scatter: one lane per Gaussian writes (depth_bits<<32 | id) into its tiles' segments
//   4. sort   : one workgroup per tile sorts its segment in LDS (bitonic, u64 keys) and
//               writes point_list (and, on request, the upstream-format keys)
// Traffic: 8 B written + 8 B read + 4 B written per pair, instead of six 24-B/pair radix passes.
#include "sg_common.h"

__global__ void __launch_bounds__(256)
sg_tile_count_kernel(int P, const int32_t *__restrict__ radii, const float4 *__restrict__ recC, int gx,
                     uint32_t *__restrict__ tile_count)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P || !(radii[idx] > 0)) return;
    float4 rc = recC[idx];
    uint32_t mn = __float_as_uint(rc.z), wh = __float_as_uint(rc.w);
    int x0 = mn & 0xffff, y0 = mn >> 16, w = wh & 0xffff, h = wh >> 16;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) atomicAdd(&tile_count[(y0 + y) * gx + x0 + x], 1u);
}

// single workgroup, 1024 threads: exclusive scan over T tile counts
__global__ void __launch_bounds__(1024)
sg_tile_scan_kernel(int T, const uint32_t *__restrict__ tile_count, uint2 *__restrict__ ranges,
                    uint32_t *__restrict__ cursor, uint32_t *__restrict__ header, uint32_t cap)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        int i = base + tid;
        uint32_t v = i < T ? tile_count[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        uint32_t carry = carry_s;
        uint32_t start = carry + woff + incl - v;
        if (i < T) {
            uint32_t s = start < cap ? start : cap, e = start + v < cap ? start + v : cap;
            ranges[i] = v ? make_uint2(s, e) : make_uint2(0u, 0u);
            cursor[i] = start;
        }
        __syncthreads();
        if (tid == 1023) carry_s = start + v;
        __syncthreads();
    }
    if (tid == 0) {
        header[0] = carry_s;
        header[1] = carry_s > cap ? 1u : 0u;
        header[3] = (uint32_t)T;
    }
}

__global__ void __launch_bounds__(256)
sg_tile_scatter_kernel(int P, const int32_t *__restrict__ radii, const float4 *__restrict__ recC,
                       const float *__restrict__ depth, int gx, uint32_t *__restrict__ cursor,
                       uint64_t *__restrict__ pair_keys, uint32_t cap)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P || !(radii[idx] > 0)) return;
    float4 rc = recC[idx];
    uint32_t mn = __float_as_uint(rc.z), wh = __float_as_uint(rc.w);
    int x0 = mn & 0xffff, y0 = mn >> 16, w = wh & 0xffff, h = wh >> 16;
    uint64_t key = ((uint64_t)__float_as_uint(depth[idx]) << 32) | (uint32_t)idx;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t slot = atomicAdd(&cursor[(y0 + y) * gx + x0 + x], 1u);
            if (slot < cap) pair_keys[slot] = key;
        }
}

// ---- per-tile sort ---------------------------------------------------------------------
#define SG_SORT_THREADS 256
#define SG_SORT_LDS 4096   // u64 entries sorted in LDS (32 KiB)

__device__ __forceinline__ void sg_bitonic_lds(uint64_t *s, int n2, int tid, int nthreads)
{
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += nthreads) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
                int ixj = i | j;
                bool up = (i & k) == 0;
                uint64_t a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(SG_SORT_THREADS)
sg_tile_sort_kernel(const uint2 *__restrict__ ranges, uint64_t *__restrict__ pair_keys,
                    uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys)
{
    __shared__ uint64_t s[SG_SORT_LDS];
    const int tile = blockIdx.x, tid = threadIdx.x;
    uint2 r = ranges[tile];
    uint32_t n = r.y - r.x;
    if (n == 0) return;
    uint64_t *seg = pair_keys + r.x;
    if (n <= SG_SORT_LDS) {
        int n2 = 1; while (n2 < (int)n) n2 <<= 1;
        for (int i = tid; i < n2; i += SG_SORT_THREADS) s[i] = i < (int)n ? seg[i] : ~0ull;
        __syncthreads();
        if (n > 1) sg_bitonic_lds(s, n2, tid, SG_SORT_THREADS);
        for (int i = tid; i < (int)n; i += SG_SORT_THREADS) {
            uint64_t k = s[i];
            point_list[r.x + i] = (uint32_t)k;
            if (point_keys) point_keys[r.x + i] = ((uint64_t)tile << 32) | (k >> 32);
        }
    } else {
        // Rare long tile: sort LDS-sized chunks in place, then place every element by rank.
        // Keys are unique (Gaussian id in the low word), so the final position of a key is
        // the number of smaller keys summed over all sorted chunks (binary searches).
        const uint32_t nchunks = (n + SG_SORT_LDS - 1) / SG_SORT_LDS;
        for (uint32_t c = 0; c < nchunks; c++) {
            uint32_t off = c * SG_SORT_LDS, m = n - off < SG_SORT_LDS ? n - off : SG_SORT_LDS;
            int n2 = 1; while (n2 < (int)m) n2 <<= 1;
            for (int i = tid; i < n2; i += SG_SORT_THREADS) s[i] = i < (int)m ? seg[off + i] : ~0ull;
            __syncthreads();
            sg_bitonic_lds(s, n2, tid, SG_SORT_THREADS);
            for (int i = tid; i < (int)m; i += SG_SORT_THREADS) seg[off + i] = s[i];
            __syncthreads();
        }
        __threadfence();   // chunk stores must be visible to every wave of this workgroup
        __syncthreads();
        for (uint32_t i = tid; i < n; i += SG_SORT_THREADS) {
            uint64_t k = seg[i];
            uint32_t pos = 0;
            for (uint32_t c = 0; c < nchunks; c++) {
                uint32_t off = c * SG_SORT_LDS, m = n - off < SG_SORT_LDS ? n - off : SG_SORT_LDS;
                uint32_t lo = 0, hi = m;
                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (seg[off + mid] < k) lo = mid + 1; else hi = mid; }
                pos += lo;
            }
            point_list[r.x + pos] = (uint32_t)k;
            if (point_keys) point_keys[r.x + pos] = ((uint64_t)tile << 32) | (k >> 32);
        }
    }
}

void sg_launch_binning(const SgCam &c, int P, const int32_t *radii, SgGeom g, SgBin b, size_t cap,
                       int write_keys, hipStream_t st)
{
    const int T = c.gx * c.gy;
    uint32_t cap32 = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
    sg_prof_begin(SG_K_TILE_COUNT, st);
    if (P > 0)
        hipLaunchKernelGGL(sg_tile_count_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, radii, g.recC, c.gx, b.tile_count);
    sg_prof_end(SG_K_TILE_COUNT, st);
    sg_prof_begin(SG_K_TILE_SCAN, st);
    hipLaunchKernelGGL(sg_tile_scan_kernel, dim3(1), dim3(1024), 0, st, T, b.tile_count, b.ranges, b.cursor, b.header, cap32);
    sg_prof_end(SG_K_TILE_SCAN, st);
    sg_prof_begin(SG_K_TILE_SCATTER, st);
    if (P > 0)
        hipLaunchKernelGGL(sg_tile_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, radii, g.recC, g.depth,
                           c.gx, b.cursor, b.pair_keys, cap32);
    sg_prof_end(SG_K_TILE_SCATTER, st);
    sg_prof_begin(SG_K_TILE_SORT, st);
    hipLaunchKernelGGL(sg_tile_sort_kernel, dim3(T), dim3(SG_SORT_THREADS), 0, st, b.ranges, b.pair_keys,
                       b.point_list, write_keys ? b.point_keys : (uint64_t *)nullptr);
    sg_prof_end(SG_K_TILE_SORT, st);
}

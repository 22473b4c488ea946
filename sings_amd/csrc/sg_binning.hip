// Tile binning: produces, for every 16x16 tile, the depth-ordered list of Gaussian ids.
//
// Replaces InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs + identifyTileRanges
// of the upstream rasterizer.  The result is bit-identical to a stable
// sort of the (tile << 32 | depth_bits) keys generated in Gaussian order: entries of one
// tile with equal depth bits come from different Gaussians and a stable sort keeps them in
// ascending Gaussian id, so the order is exactly the lexicographic order of
// (tile, depth_bits, gaussian_id).  MI355X-first formulation (integer, HBM-light):
//   1. count  : fused into the preprocess kernel (sg_project.h::sg_store_proj): every
//               (tile,Gaussian) pair takes a RETURNING atomic on its tile counter and records
//               (Gaussian, tile, arrival rank) in Gaussian-major order, load-balanced per wave
//   2. scan   : one 1024-thread workgroup turns counts into [start,end) ranges, R = total
//   3. scatter: one lane per PAIR, no atomics: pair_keys[start[tile] + rank] = depth_bits<<32 | id
//   4. sort   : one wave per tile sorts its segment in LDS (bitonic, u64 keys; one workgroup per
//               tile for lists longer than 256) and writes point_list (+ upstream-format keys)
// Traffic: 12 B + 8 B written, 20 B read, 4 B written per pair -- instead of six 24-B/pair radix passes.
#include "sg_common.h"

// single workgroup, 1024 threads x 8 consecutive tiles each: exclusive scan over T tile counts
__global__ void __launch_bounds__(1024)
sg_tile_scan_kernel(int T, const uint32_t *__restrict__ tile_count, uint32_t tc_stride, uint2 *__restrict__ ranges,
                    uint32_t *__restrict__ cursor, uint32_t *__restrict__ header, uint32_t cap,
                    uint32_t *__restrict__ long_tiles, uint32_t long_threshold, uint32_t *__restrict__ items,
                    uint32_t *__restrict__ ck_start, uint32_t items_cap)
{
    __shared__ uint32_t wsum[16], wsum_i[16], wsum_c[16];
    __shared__ uint32_t carry_s, carry_i, carry_c;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { carry_s = 0; carry_i = 0; carry_c = 0; }
    __syncthreads();
    for (int base = 0; base < T; base += 8192) {
        const int i0 = base + tid * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = i0 + k < T ? tile_count[(size_t)(i0 + k) * tc_stride] : 0u; sum += v[k]; }
        // backward work items (1 per tile, one per SG_SEG entries for longer lists) and checkpoint slots
        uint32_t isum = 0, csum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t seg = sg_nseg(v[k]);
            if (i0 + k < T) { isum += seg ? seg : 1u; csum += seg; }
        }
        uint32_t incl = sum, incl_i = isum, incl_c = csum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t u = __shfl_up(incl, o, 64), ui = __shfl_up(incl_i, o, 64), uc = __shfl_up(incl_c, o, 64);
            if (lane >= o) { incl += u; incl_i += ui; incl_c += uc; }
        }
        if (lane == 63) { wsum[wid] = incl; wsum_i[wid] = incl_i; wsum_c[wid] = incl_c; }
        __syncthreads();
        uint32_t woff = 0, woff_i = 0, woff_c = 0;
        for (int w = 0; w < wid; w++) { woff += wsum[w]; woff_i += wsum_i[w]; woff_c += wsum_c[w]; }
        uint32_t start = carry_s + woff + incl - sum;
        uint32_t istart = carry_i + woff_i + incl_i - isum, cstart = carry_c + woff_c + incl_c - csum;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k < T) {
                uint32_t s = start < cap ? start : cap, e = start + v[k] < cap ? start + v[k] : cap;
                ranges[i0 + k] = v[k] ? make_uint2(s, e) : make_uint2(0u, 0u);
                cursor[i0 + k] = start;
                if (e - s > long_threshold) long_tiles[atomicAdd(&header[4], 1u)] = (uint32_t)(i0 + k);
                const uint32_t nseg = sg_nseg(v[k]), nit = nseg ? nseg : 1u;
                for (uint32_t sg = 0; sg < nit; sg++)
                    if (istart + sg < items_cap) items[istart + sg] = (uint32_t)(i0 + k) | (sg << 20);
                ck_start[i0 + k] = nseg ? cstart : 0xffffffffu;
                istart += nit; cstart += nseg;
            }
            start += v[k];
        }
        __syncthreads();
        if (tid == 1023) { carry_s = start; carry_i = istart; carry_c = cstart; }
        __syncthreads();
    }
    if (tid == 0) {
        header[5] = carry_i < items_cap ? carry_i : items_cap;
        header[0] = carry_s;
        header[1] = carry_s > cap ? 1u : 0u;
        header[3] = (uint32_t)T;
    }
}

// one lane per pair (Gaussian-major list written by the preprocess); no atomics
__global__ void __launch_bounds__(256)
sg_pair_scatter_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ pair_gid,
                       const uint32_t *__restrict__ pair_tile, const uint32_t *__restrict__ pair_local,
                       const float *__restrict__ depth, const uint32_t *__restrict__ start,
                       uint64_t *__restrict__ pair_keys, uint32_t cap)
{
    const uint32_t R = header[0] < cap ? header[0] : cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) {
        uint32_t gid = pair_gid[i];
        uint32_t slot = start[pair_tile[i]] + pair_local[i];
        if (slot < cap) pair_keys[slot] = ((uint64_t)__float_as_uint(depth[gid]) << 32) | gid;
    }
}

// ---- per-tile sort ---------------------------------------------------------------------
#define SG_SORT_THREADS 1024   // long lists are a tail of a few workgroups: give each the whole CU
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

#define SG_WSORT_MAX 256   // longest list sorted by a single wave

// one wave per tile, 4 tiles per workgroup, no workgroup barriers
__global__ void __launch_bounds__(256)
sg_tile_sort_wave_kernel(int T, const uint2 *__restrict__ ranges, const uint64_t *__restrict__ pair_keys,
                         uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys)
{
    __shared__ uint64_t sall[4][SG_WSORT_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n == 0 || n > SG_WSORT_MAX) return;
    uint64_t *s = sall[wave];
    int n2 = 1; while (n2 < n) n2 <<= 1;
    for (int i = lane; i < n2; i += 64) s[i] = i < n ? pair_keys[r.x + i] : ~0ull;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                bool up = (i & k) == 0;
                uint64_t a = s[i], b = s[ixj];
                if ((a > b) == up) { s[i] = b; s[ixj] = a; }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    for (int i = lane; i < n; i += 64) {
        uint64_t kx = s[i];
        point_list[r.x + i] = (uint32_t)kx;
        if (point_keys) point_keys[r.x + i] = ((uint64_t)tile << 32) | (kx >> 32);
    }
}

// long lists: one workgroup per tile
__global__ void __launch_bounds__(SG_SORT_THREADS)
sg_tile_sort_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ long_tiles,
                    const uint2 *__restrict__ ranges, uint64_t *__restrict__ pair_keys,
                    uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys, uint32_t *__restrict__ rank)
{
    __shared__ uint64_t s[SG_SORT_LDS];
    const int tid = threadIdx.x;
    const uint32_t nlong = header[4];
    for (uint32_t li = blockIdx.x; li < nlong; li += gridDim.x) {      // work list written by the scan kernel
    const int tile = (int)long_tiles[li];
    uint2 r = ranges[tile];
    uint32_t n = r.y - r.x;
    __syncthreads();
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
        // Long tile (> SG_SORT_LDS entries): sort LDS-sized chunks in place, then place every element by rank.
        // Keys are unique (Gaussian id in the low word), so the final position of a key is the number of
        // smaller keys summed over all sorted chunks.  Each sorted chunk is brought back into LDS once and all
        // threads binary-search it there; the running ranks live in `rank` (the pair_local array, dead after
        // the scatter), indexed like the segment.
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
        uint32_t *rk = rank + r.x;
        for (uint32_t i = tid; i < n; i += SG_SORT_THREADS) rk[i] = 0;
        __threadfence();   // chunk stores must be visible to every wave of this workgroup
        __syncthreads();
        for (uint32_t c = 0; c < nchunks; c++) {
            uint32_t off = c * SG_SORT_LDS, m = n - off < SG_SORT_LDS ? n - off : SG_SORT_LDS;
            for (uint32_t i = tid; i < m; i += SG_SORT_THREADS) s[i] = seg[off + i];
            __syncthreads();
            for (uint32_t i = tid; i < n; i += SG_SORT_THREADS) {
                uint64_t k = seg[i];
                uint32_t lo = 0, hi = m;
                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (s[mid] < k) lo = mid + 1; else hi = mid; }
                rk[i] += lo;
            }
            __syncthreads();
        }
        for (uint32_t i = tid; i < n; i += SG_SORT_THREADS) {
            uint64_t k = seg[i];
            uint32_t pos = rk[i];
            point_list[r.x + pos] = (uint32_t)k;
            if (point_keys) point_keys[r.x + pos] = ((uint64_t)tile << 32) | (k >> 32);
        }
    }
    }
}

void sg_launch_binning(const SgCam &c, int P, const int32_t *radii, SgGeom g, SgBin b, size_t cap,
                       int write_keys, hipStream_t st)
{
    (void)radii;
    const int T = c.gx * c.gy;
    const uint32_t cap32 = sg_cap32(cap);
    uint64_t *pk = write_keys ? b.point_keys : (uint64_t *)nullptr;
    sg_prof_begin(SG_K_TILE_SCAN, st);
    hipLaunchKernelGGL(sg_tile_scan_kernel, dim3(1), dim3(1024), 0, st, T, b.tile_count, b.tc_stride, b.ranges, b.cursor, b.header, cap32,
                       b.long_tiles, (uint32_t)SG_WSORT_MAX, b.items, b.ck_start, sg_items_cap(T, cap));
    sg_prof_end(SG_K_TILE_SCAN, st);
    sg_prof_begin(SG_K_TILE_SCATTER, st);
    if (P > 0) {
        size_t want = (cap + 255) / 256;
        int grid = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(sg_pair_scatter_kernel, dim3(grid), dim3(256), 0, st, b.header, b.pair_gid, b.pair_tile,
                           b.pair_local, g.depth, b.cursor, b.pair_keys, cap32);
    }
    sg_prof_end(SG_K_TILE_SCATTER, st);
    sg_prof_begin(SG_K_TILE_SORT, st);
    hipLaunchKernelGGL(sg_tile_sort_wave_kernel, dim3((T + 3) / 4), dim3(256), 0, st, T, b.ranges, b.pair_keys,
                       b.point_list, pk);
    hipLaunchKernelGGL(sg_tile_sort_kernel, dim3(T < 2048 ? T : 2048), dim3(SG_SORT_THREADS), 0, st, b.header, b.long_tiles,
                       b.ranges, b.pair_keys, b.point_list, pk, b.pair_local);
    sg_prof_end(SG_K_TILE_SORT, st);
}

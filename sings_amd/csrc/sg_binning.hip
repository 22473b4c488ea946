// Tile binning: produces, for every 16x16 tile, the depth-ordered list of Gaussian ids.
//
// Replaces InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs + identifyTileRanges
// of the upstream rasterizer.  The result is bit-identical to a stable
// sort of the (tile << 32 | depth_bits) keys generated in Gaussian order: entries of one
// tile with equal depth bits come from different Gaussians and a stable sort keeps them in
// ascending Gaussian id, so the order is exactly the lexicographic order of
// (tile, depth_bits, gaussian_id).  MI355X-first formulation (integer, HBM-light):
//   1. count  : fused into the preprocess kernels (sg_project.h::sg_store_proj): every (tile,Gaussian) pair gets its
//               arrival rank in the tile -- a RETURNING atomic on the tile counter, or (few tiles) an LDS histogram per
//               workgroup + one global atomic per touched tile -- and is recorded as (Gaussian, tile, rank) in
//               Gaussian-major order, load-balanced per wave
//   2. scan   : a few workgroups turn counts into [start,end) ranges, R, and the per-tile plans of the work lists
//   3. scatter: one lane per PAIR, no atomics: pair_keys[start[tile] + rank] = depth_bits<<32 | id; + work lists
//   4. sort   : one wave per tile sorts its segment in LDS (bitonic, u64 keys); lists longer than 256: one workgroup
//               per 4096-entry chunk, longer still: chunks merged by rank; writes point_list (+ upstream-format keys)
// Traffic: 12 B + 8 B written, 20 B read, 4 B written per pair -- instead of six 24-B/pair radix passes.
#include "sg_common.h"

#define SG_WSORT_MAX 256       // longest list sorted by a single wave
#define SG_SORT_THREADS 1024   // longer lists: one 1024-thread workgroup per chunk of
#define SG_SORT_LDS 4096       // u64 entries sorted in LDS (32 KiB)

// Exclusive scans over the T tile counts of
//   q0 pairs (-> ranges, cursors), q1 backward work items, q2 checkpoint slots, q3 sort items, q4 rank items.
// Workgroup b owns tiles [b * SG_SCAN_BS * tpt, (b + 1) * SG_SCAN_BS * tpt), one tile per thread and round.  Instead of a second
// kernel (or a look-back chain) every workgroup first REDUCES the counts of all tiles in front of its range itself:
// at most T words per workgroup, coalesced and L2-resident.  The kernel is a handful of waves that start with a cold
// instruction cache, so its loops are deliberately NOT unrolled: the fully unrolled version (1200 instructions of
// straight-line code) took 22 us at 8160 tiles, this one 12 us.
// The work lists themselves are written by the (chip-wide) scatter kernel from the per-tile
// `plan` = (first backward item, first sort item, first rank item, pair count).
#define SG_SCAN_NQ 5
// BS = threads (= tiles per round) of a scan workgroup: 256 for images of few tiles (several CUs even at 1000 tiles),
// 1024 for many tiles (fewer workgroups re-reducing the counts in front of them)
__device__ __forceinline__ void sg_scan_derive(uint32_t v, uint32_t q[SG_SCAN_NQ])
{
    const uint32_t seg = sg_nseg(v);
    const uint32_t nch = v > SG_WSORT_MAX ? (v + SG_SORT_LDS - 1) / SG_SORT_LDS : 0u;
    q[0] = v; q[1] = seg ? seg : 1u; q[2] = seg; q[3] = nch; q[4] = nch > 1 ? nch : 0u;
}

template <int SG_SCAN_BS>
__global__ void __launch_bounds__(SG_SCAN_BS)
sg_tile_scan_kernel(int T, int tpt, const uint32_t *__restrict__ tile_count,
                    uint2 *__restrict__ ranges, uint32_t *__restrict__ cursor, uint32_t *__restrict__ header,
                    uint32_t cap, uint32_t sort_cap, uint32_t rank_cap, uint4 *__restrict__ plan,
                    uint32_t *__restrict__ ck_start, uint32_t items_cap)
{
    constexpr int NQ = SG_SCAN_NQ;
    __shared__ uint32_t wsum[NQ][SG_SCAN_BS / 64];
    __shared__ uint32_t carry[NQ];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int first = blockIdx.x * SG_SCAN_BS * tpt;
    // ---- 1. totals of everything in front of this workgroup's range
    uint32_t acc[NQ] = { 0, 0, 0, 0, 0 };
    for (int t = tid; t < first; t += SG_SCAN_BS) {
        uint32_t q[NQ];
        const uint32_t v = tile_count[t];
        sg_scan_derive(v, q);
#pragma unroll
        for (int a = 0; a < NQ; a++) acc[a] += q[a];
    }
#pragma unroll 1
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < NQ; a++) acc[a] += __shfl_xor(acc[a], o, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NQ; a++) wsum[a][wid] = acc[a];
    }
    __syncthreads();
    if (tid < NQ) {
        uint32_t t = 0;
#pragma unroll 1
        for (int w = 0; w < SG_SCAN_BS / 64; w++) t += wsum[tid][w];
        carry[tid] = t;
    }
    __syncthreads();
    // ---- 2. scan of the own range, SG_SCAN_BS tiles per round
    for (int r = 0; r < tpt; r++) {
        const int tile = first + r * SG_SCAN_BS + tid;
        const bool ok = tile < T;
        uint32_t q[NQ] = { 0, 0, 0, 0, 0 };
        const uint32_t v = ok ? tile_count[tile] : 0u;
        if (ok) sg_scan_derive(v, q);
        uint32_t incl[NQ];
#pragma unroll
        for (int a = 0; a < NQ; a++) incl[a] = q[a];
#pragma unroll 1
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int a = 0; a < NQ; a++) {
                const uint32_t u = __shfl_up(incl[a], o, 64);
                if (lane >= o) incl[a] += u;
            }
        }
        if (lane == 63) {
#pragma unroll
            for (int a = 0; a < NQ; a++) wsum[a][wid] = incl[a];
        }
        __syncthreads();
        uint32_t st[NQ], tot[NQ];
#pragma unroll
        for (int a = 0; a < NQ; a++) {
            uint32_t woff = 0, all = 0;
#pragma unroll 1
            for (int w = 0; w < SG_SCAN_BS / 64; w++) { const uint32_t x = wsum[a][w]; woff += w < wid ? x : 0u; all += x; }
            st[a] = carry[a] + woff + incl[a] - q[a];
            tot[a] = carry[a] + all;
        }
        if (ok) {
            const uint32_t s = st[0] < cap ? st[0] : cap, e = st[0] + v < cap ? st[0] + v : cap;
            ranges[tile] = v ? make_uint2(s, e) : make_uint2(0u, 0u);
            cursor[tile] = st[0];
            ck_start[tile] = q[2] ? st[2] : 0xffffffffu;
            plan[tile] = make_uint4(st[1], st[3], st[4], v);
        }
        __syncthreads();
        if (tid < NQ) {
            // (every thread computed the same tot[]; thread a publishes quantity a)
            uint32_t t = 0;
#pragma unroll
            for (int a = 0; a < NQ; a++) t = tid == a ? tot[a] : t;
            carry[tid] = t;
        }
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        header[0] = carry[0];
        header[1] = carry[0] > cap ? 1u : 0u;
        header[3] = (uint32_t)T;
        header[4] = carry[3] < sort_cap ? carry[3] : sort_cap;
        header[5] = carry[1] < items_cap ? carry[1] : items_cap;
        header[6] = carry[4] < rank_cap ? carry[4] : rank_cap;
    }
}

// One lane per pair (Gaussian-major list written by the preprocess): key -> its slot, no atomics.
// The first T threads also expand their tile's plan into the work lists of the sort / merge / backward kernels.
__global__ void __launch_bounds__(256)
sg_pair_scatter_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ pair_gid,
                       const uint32_t *__restrict__ pair_tile, const uint32_t *__restrict__ pair_local,
                       const float *__restrict__ depth, const uint32_t *__restrict__ start,
                       uint64_t *__restrict__ pair_keys, uint32_t cap, int T, const uint4 *__restrict__ plan,
                       uint2 *__restrict__ sort_items, uint2 *__restrict__ rank_items, uint32_t sort_cap,
                       uint32_t rank_cap, uint32_t *__restrict__ items, uint32_t items_cap)
{
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (uint32_t tile = gtid; tile < (uint32_t)T; tile += nthreads) {
        const uint4 pl = plan[tile];
        const uint32_t nseg = sg_nseg(pl.w), nit = nseg ? nseg : 1u;
        for (uint32_t sg = 0; sg < nit; sg++)
            if (pl.x + sg < items_cap) items[pl.x + sg] = tile | (sg << 20);
        const uint32_t nch = pl.w > SG_WSORT_MAX ? (pl.w + SG_SORT_LDS - 1) / SG_SORT_LDS : 0u;
        for (uint32_t c = 0; c < nch; c++) {
            if (pl.y + c < sort_cap) sort_items[pl.y + c] = make_uint2(tile, c);
            if (nch > 1 && pl.z + c < rank_cap) rank_items[pl.z + c] = make_uint2(tile, c);
        }
    }
    const uint32_t R = header[0] < cap ? header[0] : cap;
    for (uint32_t i = gtid; i < R; i += nthreads) {
        uint32_t gid = pair_gid[i];
        uint32_t slot = start[pair_tile[i]] + pair_local[i];
        if (slot < cap) pair_keys[slot] = ((uint64_t)__float_as_uint(depth[gid]) << 32) | gid;
    }
}

// ---- per-tile sort ---------------------------------------------------------------------

// Bitonic sort of s[0, n2) (n2 a power of two <= SG_SORT_LDS) by the whole SG_SORT_THREADS workgroup.
// Wave w owns the comparators of a contiguous block of B = 128 * cpt elements: every stage with 2j <= B touches only
// the wave's own block and needs no workgroup barrier (LDS operations of one wave complete in order), which leaves
// 14 workgroup barriers instead of 78 at n2 = 4096.
__device__ __forceinline__ void sg_bitonic_lds(uint64_t *s, int n2, int tid)
{
    const int cpt = n2 > 2048 ? n2 / 2048 : 1;          // comparators per thread
    const int wave = tid >> 6, lane = tid & 63;
    const int B = 128 * cpt;
    const bool active = wave * B < n2;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (active) {
                for (int c = 0; c < cpt; c++) {
                    const int t = (wave * cpt + c) * 64 + lane;
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                    const bool up = (i & k) == 0;
                    const uint64_t a = s[i], b = s[ixj];
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            if (2 * j <= B) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); }
            else __syncthreads();
        }
        if (2 * k > B) __syncthreads();                  // the next k starts with a cross-wave stage (or we are done)
    }
    __syncthreads();
}


// lists of at most SG_WSORT_MAX entries: one wave per tile, bitonic in the wave's slice of LDS, no workgroup barriers
__device__ __forceinline__ void sg_tile_sort_wave(int tile, uint64_t *__restrict__ s, int lane, const uint2 *__restrict__ ranges,
                                                  const uint64_t *__restrict__ pair_keys, uint32_t *__restrict__ point_list,
                                                  uint64_t *__restrict__ point_keys)
{
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n == 0 || n > SG_WSORT_MAX) return;
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

// One launch for both kinds of list.  Workgroups [0, wave_blocks): 16 short lists each, one per wave (above).  The
// others: one workgroup per work item (tile, chunk of SG_SORT_LDS entries) of the lists longer than SG_WSORT_MAX,
// items written by the scatter kernel; a list of one chunk is sorted and written out, the chunks of a longer list are
// sorted in place and merged by sg_tile_rank_kernel.
__global__ void __launch_bounds__(SG_SORT_THREADS)
sg_tile_sort_kernel(int T, int wave_blocks, const uint32_t *__restrict__ header, const uint2 *__restrict__ sort_items,
                    const uint2 *__restrict__ ranges, uint64_t *__restrict__ pair_keys,
                    uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys)
{
    __shared__ uint64_t s[SG_SORT_LDS];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < wave_blocks) {
        const int tile = blockIdx.x * (SG_SORT_THREADS / 64) + (tid >> 6);
        if (tile < T) sg_tile_sort_wave(tile, s + (tid >> 6) * SG_WSORT_MAX, tid & 63, ranges, pair_keys, point_list, point_keys);
        return;
    }
    const uint32_t nitems = header[4];
    for (uint32_t li = blockIdx.x - wave_blocks; li < nitems; li += gridDim.x - wave_blocks) {
        const uint2 it = sort_items[li];
        const int tile = (int)it.x;
        const uint2 r = ranges[tile];
        const uint32_t n = r.y - r.x, off = it.y * SG_SORT_LDS;
        if (off >= n) continue;
        const int m = (int)(n - off < SG_SORT_LDS ? n - off : SG_SORT_LDS);
        uint64_t *seg = pair_keys + r.x + off;
        int n2 = 1; while (n2 < m) n2 <<= 1;
        __syncthreads();
        for (int i = tid; i < n2; i += SG_SORT_THREADS) s[i] = i < m ? seg[i] : ~0ull;
        __syncthreads();
        if (m > 1) sg_bitonic_lds(s, n2, tid);
        if (n <= SG_SORT_LDS) {
            for (int i = tid; i < m; i += SG_SORT_THREADS) {
                const uint64_t k = s[i];
                point_list[r.x + i] = (uint32_t)k;
                if (point_keys) point_keys[r.x + i] = ((uint64_t)tile << 32) | (k >> 32);
            }
        } else {
            for (int i = tid; i < m; i += SG_SORT_THREADS) seg[i] = s[i];
        }
    }
}

// Lists longer than SG_SORT_LDS: one workgroup per (tile, chunk).  Keys are unique (Gaussian id in the low word), so
// the final position of a key is its index in its own sorted chunk plus the number of smaller keys in every other
// chunk; each other chunk is brought into LDS once and binary-searched there.
__global__ void __launch_bounds__(SG_SORT_THREADS)
sg_tile_rank_kernel(const uint32_t *__restrict__ header, const uint2 *__restrict__ rank_items,
                    const uint2 *__restrict__ ranges, const uint64_t *__restrict__ pair_keys,
                    uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys)
{
    __shared__ uint64_t s[SG_SORT_LDS];
    constexpr int KPT = SG_SORT_LDS / SG_SORT_THREADS;
    const int tid = threadIdx.x;
    const uint32_t nitems = header[6];
    for (uint32_t li = blockIdx.x; li < nitems; li += gridDim.x) {
        const uint2 it = rank_items[li];
        const int tile = (int)it.x;
        const uint2 r = ranges[tile];
        const uint32_t n = r.y - r.x, off = it.y * SG_SORT_LDS;
        if (off >= n) continue;
        const uint32_t m = n - off < SG_SORT_LDS ? n - off : SG_SORT_LDS;
        const uint64_t *seg = pair_keys + r.x;
        uint64_t key[KPT];
        uint32_t rk[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i = tid + q * SG_SORT_THREADS;
            key[q] = i < m ? seg[off + i] : ~0ull;
            rk[q] = i;
        }
        const uint32_t nchunks = (n + SG_SORT_LDS - 1) / SG_SORT_LDS;
        for (uint32_t c = 0; c < nchunks; c++) {
            if (c == it.y) continue;
            const uint32_t co = c * SG_SORT_LDS, cm = n - co < SG_SORT_LDS ? n - co : SG_SORT_LDS;
            __syncthreads();
            for (uint32_t i = tid; i < cm; i += SG_SORT_THREADS) s[i] = seg[co + i];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < KPT; q++) {
                uint32_t lo = 0, hi = cm;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s[mid] < key[q]) lo = mid + 1; else hi = mid; }
                rk[q] += lo;
            }
        }
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i = tid + q * SG_SORT_THREADS;
            if (i < m) {
                point_list[r.x + rk[q]] = (uint32_t)key[q];
                if (point_keys) point_keys[r.x + rk[q]] = ((uint64_t)tile << 32) | (key[q] >> 32);
            }
        }
        __syncthreads();
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
    const int tpt = (T + 65535) / 65536 > 0 ? (T + 65535) / 65536 : 1;       // at most 64 (1024-thread) / 256 (256-thread) workgroups
#define SG_SCAN_GRID(BS) ((T + (BS) * tpt - 1) / ((BS) * tpt) > 0 ? (T + (BS) * tpt - 1) / ((BS) * tpt) : 1)
#define SG_SCAN(BS) hipLaunchKernelGGL((sg_tile_scan_kernel<BS>), dim3(SG_SCAN_GRID(BS)), dim3(BS), 0, st, T, tpt, b.tile_count, \
                                       b.ranges, b.cursor, b.header, cap32, sg_sort_items_cap(T, cap),                 \
                                       sg_rank_items_cap(cap), b.plan, b.ck_start, sg_items_cap(T, cap))
    if (T <= SG_HIST_TILES_MAX) SG_SCAN(256); else SG_SCAN(1024);
#undef SG_SCAN
#undef SG_SCAN_GRID
    sg_prof_end(SG_K_TILE_SCAN, st);
    sg_prof_begin(SG_K_TILE_SCATTER, st);
    {
        size_t want = ((cap > (size_t)T ? cap : (size_t)T) + 255) / 256;
        int grid = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(sg_pair_scatter_kernel, dim3(grid), dim3(256), 0, st, b.header, b.pair_gid, b.pair_tile,
                           b.pair_local, g.depth, b.cursor, b.pair_keys, cap32, T, b.plan, b.sort_items, b.rank_items,
                           sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.items, sg_items_cap(T, cap));
    }
    sg_prof_end(SG_K_TILE_SCATTER, st);
    sg_prof_begin(SG_K_TILE_SORT, st);
    static_assert(SG_WSORT_MAX * (SG_SORT_THREADS / 64) <= SG_SORT_LDS, "16 wave slices fit the sort buffer");
    const int wave_blocks = (T + SG_SORT_THREADS / 64 - 1) / (SG_SORT_THREADS / 64);
    const uint32_t sgrid = sg_sort_items_cap(T, cap) < 512 ? sg_sort_items_cap(T, cap) : 512;     // 2 x 256 CUs
    hipLaunchKernelGGL(sg_tile_sort_kernel, dim3(wave_blocks + sgrid), dim3(SG_SORT_THREADS), 0, st, T, wave_blocks, b.header,
                       b.sort_items, b.ranges, b.pair_keys, b.point_list, pk);
    const uint32_t rgrid = sg_rank_items_cap(cap) < 128 ? sg_rank_items_cap(cap) : 128;
    hipLaunchKernelGGL(sg_tile_rank_kernel, dim3(rgrid), dim3(SG_SORT_THREADS), 0, st, b.header, b.rank_items,
                       b.ranges, b.pair_keys, b.point_list, pk);
    sg_prof_end(SG_K_TILE_SORT, st);
}

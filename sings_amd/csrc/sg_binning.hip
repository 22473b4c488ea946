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
//   2. scan + scatter, ONE launch (sg_scan_scatter_kernel): every workgroup turns the T tile counts into the exclusive
//               prefix it needs -- in its own LDS, T words, L2-resident reads -- instead of waiting for a scan kernel;
//               then one lane per PAIR, no atomics: pair_keys[start[tile] + rank] = depth_bits<<32 | id.  The per-tile
//               outputs (ranges, plans, work lists, R) are written by the workgroups between them.
//   3. sort   : lists of <= 1024 entries are sorted by the forward composite kernel itself, in the LDS of the tile's
//               workgroup, just before it gathers the records (sg_sort.h; no launch, no extra pass over the keys);
//               longer lists: one workgroup per 4096-entry chunk (bitonic), longer still: chunks merged by rank.
// Traffic: 12 B + 8 B written, 20 B read, 4 B written per pair -- instead of six 24-B/pair radix passes.
// (Round 1 ran scan / scatter / sort / rank as four launches: 10 + 12 + 18 + 4 us at cfg3, all latency.)
#include "sg_sort.h"

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

// Early pair count: one 64-bit system-scope store to a mapped, coherent host word (valid bit | flags << 32 | R) -- visible to a
// polling host thread while this kernel and the composite behind it are still running.
__device__ __forceinline__ void sg_publish_count(unsigned long long *signal, uint32_t R, uint32_t flags)
{
    if (signal) __hip_atomic_store(signal, (1ull << 63) | ((unsigned long long)flags << 32) | R, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int SG_SCAN_BS>
__global__ void __launch_bounds__(SG_SCAN_BS)
sg_tile_scan_kernel(int T, int tpt, const uint32_t *__restrict__ tile_count,
                    uint2 *__restrict__ ranges, uint32_t *__restrict__ cursor, uint32_t *__restrict__ header,
                    uint32_t cap, uint32_t sort_cap, uint32_t rank_cap, uint4 *__restrict__ plan,
                    uint32_t *__restrict__ ck_start, uint32_t items_cap, int short_lists, unsigned long long *signal)
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
        const uint32_t hflags = (carry[0] > cap ? 1u : 0u) | (short_lists && carry[3] ? 2u : 0u);
        header[1] = hflags;
        sg_publish_count(signal, carry[0], hflags);
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

// ---- scan + scatter in one launch ---------------------------------------------------------
// Every workgroup: (1) copies the T tile counts into LDS, (2) thread t sums the derived quantities of its tpt
// consecutive tiles, the workgroup scans the 1024 partial sums, (3) thread t walks its tiles again and leaves the
// exclusive pair prefix (= cursor) in LDS; the threads with t % gridDim == blockIdx ALSO write their tiles' outputs
// (ranges, cursor, checkpoint slot, plan, work-list entries), so the T tiles are written once, by all workgroups
// between them; (4) the workgroup scatters its share of the pairs with the cursors in LDS.  No workgroup waits for
// another one.  LDS: 4 T bytes (T <= SG_SS_MAX_TILES; larger images take the two-kernel path below).
#define SG_SS_THREADS 1024
#define SG_SS_MAX_TILES 32768
__global__ void __launch_bounds__(SG_SS_THREADS)
sg_scan_scatter_kernel(int T, const uint32_t *__restrict__ tile_count, uint2 *__restrict__ ranges,
                       uint32_t *__restrict__ cursor, uint32_t *__restrict__ header, uint32_t cap, uint32_t sort_cap,
                       uint32_t rank_cap, uint4 *__restrict__ plan, uint32_t *__restrict__ ck_start, uint32_t items_cap,
                       const uint32_t *__restrict__ pair_gid, const uint32_t *__restrict__ pair_tile,
                       const uint32_t *__restrict__ pair_local, const float *__restrict__ depth,
                       uint64_t *__restrict__ pair_keys, uint2 *__restrict__ sort_items, uint2 *__restrict__ rank_items,
                       uint32_t *__restrict__ items, int short_lists, unsigned long long *signal)
{
    constexpr int NQ = SG_SCAN_NQ;
    extern __shared__ uint32_t sStart[];                     // [T] counts, then exclusive pair prefix
    __shared__ uint32_t wsum[NQ][SG_SS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int t = tid; t < T; t += SG_SS_THREADS) sStart[t] = tile_count[t];
    __syncthreads();
    const int tpt = (T + SG_SS_THREADS - 1) / SG_SS_THREADS;
    const int t0 = tid * tpt < T ? tid * tpt : T, t1 = t0 + tpt < T ? t0 + tpt : T;
    uint32_t own[NQ] = { 0, 0, 0, 0, 0 };
#pragma unroll 8
    for (int t = t0; t < t1; t++) {
        uint32_t q[NQ];
        sg_scan_derive(sStart[t], q);
#pragma unroll
        for (int a = 0; a < NQ; a++) own[a] += q[a];
    }
    uint32_t incl[NQ];
#pragma unroll
    for (int a = 0; a < NQ; a++) incl[a] = own[a];
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
    // the 16 wave totals: lane w < 16 holds wave w's total, a 16-lane scan gives every wave its offset -- five pipelined
    // LDS reads and four shuffle steps instead of 80 dependent LDS round trips (this kernel is pure latency)
    uint32_t run[NQ], tot[NQ];
    {
        uint32_t wt[NQ], wi[NQ];
#pragma unroll
        for (int a = 0; a < NQ; a++) { wt[a] = lane < SG_SS_THREADS / 64 ? wsum[a][lane] : 0u; wi[a] = wt[a]; }
#pragma unroll 1
        for (int o = 1; o < SG_SS_THREADS / 64; o <<= 1) {
#pragma unroll
            for (int a = 0; a < NQ; a++) {
                const uint32_t u = __shfl_up(wi[a], o, 64);
                if (lane >= o) wi[a] += u;
            }
        }
#pragma unroll
        for (int a = 0; a < NQ; a++) {
            const uint32_t woff = __shfl(wi[a] - wt[a], wid, 64);
            tot[a] = __shfl(wi[a], SG_SS_THREADS / 64 - 1, 64);
            run[a] = woff + incl[a] - own[a];
        }
    }
    const bool writer = (uint32_t)tid % gridDim.x == blockIdx.x;
    for (int t = t0; t < t1; t++) {
        const uint32_t v = sStart[t];
        uint32_t q[NQ];
        sg_scan_derive(v, q);
        if (writer) {
            const uint32_t s = run[0] < cap ? run[0] : cap, e = run[0] + v < cap ? run[0] + v : cap;
            ranges[t] = v ? make_uint2(s, e) : make_uint2(0u, 0u);
            cursor[t] = run[0];
            ck_start[t] = q[2] ? run[2] : 0xffffffffu;
            plan[t] = make_uint4(run[1], run[3], run[4], v);
            for (uint32_t sg = 0; sg < q[1]; sg++)
                if (run[1] + sg < items_cap) items[run[1] + sg] = (uint32_t)t | (sg << 20);
            for (uint32_t c = 0; c < q[3]; c++) {
                if (run[3] + c < sort_cap) sort_items[run[3] + c] = make_uint2((uint32_t)t, c);
                if (q[4] && run[4] + c < rank_cap) rank_items[run[4] + c] = make_uint2((uint32_t)t, c);
            }
        }
        sStart[t] = run[0];
#pragma unroll
        for (int a = 0; a < NQ; a++) run[a] += q[a];
    }
    if (blockIdx.x == 0 && tid == 0) {
        header[0] = tot[0];
        // bit 0: more pairs than the workspace holds; bit 1: a list needs the long-list kernels the caller told us to skip.
        // Either way the lists are incomplete / unsorted and the composite kernels touch nothing.
        const uint32_t hflags = (tot[0] > cap ? 1u : 0u) | (short_lists && tot[3] ? 2u : 0u);
        header[1] = hflags;
        sg_publish_count(signal, tot[0], hflags);         // the host may be waiting for exactly this (SgRasterSettings.count_signal)
        header[3] = (uint32_t)T;
        header[4] = tot[3] < sort_cap ? tot[3] : sort_cap;
        header[5] = tot[1] < items_cap ? tot[1] : items_cap;
        header[6] = tot[4] < rank_cap ? tot[4] : rank_cap;
    }
    __syncthreads();
    const uint32_t R = tot[0] < cap ? tot[0] : cap;
    for (uint32_t i = blockIdx.x * SG_SS_THREADS + tid; i < R; i += gridDim.x * SG_SS_THREADS) {
        const uint32_t gid = pair_gid[i];
        const uint32_t slot = sStart[pair_tile[i]] + pair_local[i];
        if (slot < cap) pair_keys[slot] = ((uint64_t)__float_as_uint(depth[gid]) << 32) | gid;
    }
}

// ---- long lists ---------------------------------------------------------------------------

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


// Lists longer than SG_WSORT_MAX (the forward composite sorts the others itself): one workgroup per work item (tile,
// chunk of SG_SORT_LDS entries), items written by the scan; a list of one chunk is sorted and written out, the chunks of
// a longer list are sorted in place and merged by sg_tile_rank_kernel.  Exits at once when there is no such list.
__global__ void __launch_bounds__(SG_SORT_THREADS)
sg_tile_sort_kernel(const uint32_t *__restrict__ header, const uint2 *__restrict__ sort_items,
                    const uint2 *__restrict__ ranges, uint64_t *__restrict__ pair_keys,
                    uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys)
{
    __shared__ uint64_t s[SG_SORT_LDS];
    const int tid = threadIdx.x;
    const uint32_t nitems = header[1] ? 0u : header[4];
    for (uint32_t li = blockIdx.x; li < nitems; li += gridDim.x) {
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
    const uint32_t nitems = header[1] ? 0u : header[6];
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
    (void)radii; (void)P;
    const int T = c.gx * c.gy;
    const uint32_t cap32 = sg_cap32(cap);
    uint64_t *pk = write_keys ? b.point_keys : (uint64_t *)nullptr;
    const int short_lists = (c.flags & SG_FLAG_SHORT_LISTS) ? 1 : 0;
    // 4 T bytes of dynamic LDS (up to 128 KiB).  Above the default 64-KiB limit (images of >= ~16 k tiles: 2048 x 2048) the
    // kernel needs its limit raised -- per DEVICE and not remembered in a process-wide flag (a second GPU used from the same
    // process, or a failed call, must not leave the launch below without it): asked for whenever it is needed (a host-side
    // table update, no stream operation), and if the runtime refuses, the two-kernel path below does the same job.
    bool fused = T <= SG_SS_MAX_TILES;
    if (fused && (size_t)T * 4 + 1024 > 64 * 1024)
        fused = hipFuncSetAttribute((const void *)sg_scan_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SG_SS_MAX_TILES * 4) == hipSuccess;
    if (fused) {
        sg_prof_begin(SG_K_TILE_SCAN, st);
        size_t want = (cap + 4 * SG_SS_THREADS - 1) / (4 * SG_SS_THREADS);     // ~4 pairs per thread
        const int grid = (int)(want < 8 ? 8 : (want > 256 ? 256 : want));
        hipLaunchKernelGGL(sg_scan_scatter_kernel, dim3(grid), dim3(SG_SS_THREADS), (size_t)T * 4, st, T, b.tile_count, b.ranges,
                           b.cursor, b.header, cap32, sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.plan, b.ck_start,
                           sg_items_cap(T, cap), b.pair_gid, b.pair_tile, b.pair_local, g.depth, b.pair_keys, b.sort_items,
                           b.rank_items, b.items, short_lists, c.count_signal);
        sg_prof_end(SG_K_TILE_SCAN, st);
    } else {
        sg_prof_begin(SG_K_TILE_SCAN, st);
        const int tpt = (T + 65535) / 65536 > 0 ? (T + 65535) / 65536 : 1;
        const int sgrid = (T + 1024 * tpt - 1) / (1024 * tpt) > 0 ? (T + 1024 * tpt - 1) / (1024 * tpt) : 1;
        hipLaunchKernelGGL((sg_tile_scan_kernel<1024>), dim3(sgrid), dim3(1024), 0, st, T, tpt, b.tile_count, b.ranges, b.cursor,
                           b.header, cap32, sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.plan, b.ck_start,
                           sg_items_cap(T, cap), short_lists, c.count_signal);
        sg_prof_end(SG_K_TILE_SCAN, st);
        sg_prof_begin(SG_K_TILE_SCATTER, st);
        size_t want = ((cap > (size_t)T ? cap : (size_t)T) + 255) / 256;
        int grid = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(sg_pair_scatter_kernel, dim3(grid), dim3(256), 0, st, b.header, b.pair_gid, b.pair_tile,
                           b.pair_local, g.depth, b.cursor, b.pair_keys, cap32, T, b.plan, b.sort_items, b.rank_items,
                           sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.items, sg_items_cap(T, cap));
        sg_prof_end(SG_K_TILE_SCATTER, st);
    }
    // lists longer than 1024 entries (the composite kernel sorts the others): both kernels exit at once when there are none
    if (short_lists) return;     // the caller vouches for short lists (checked on the device: header[1] bit 1)
    sg_prof_begin(SG_K_TILE_SORT, st);
    const uint32_t sgrid = sg_sort_items_cap(T, cap) < 512 ? sg_sort_items_cap(T, cap) : 512;     // 2 x 256 CUs
    hipLaunchKernelGGL(sg_tile_sort_kernel, dim3(sgrid), dim3(SG_SORT_THREADS), 0, st, b.header, b.sort_items, b.ranges,
                       b.pair_keys, b.point_list, pk);
    const uint32_t rgrid = sg_rank_items_cap(cap) < 128 ? sg_rank_items_cap(cap) : 128;
    hipLaunchKernelGGL(sg_tile_rank_kernel, dim3(rgrid), dim3(SG_SORT_THREADS), 0, st, b.header, b.rank_items,
                       b.ranges, b.pair_keys, b.point_list, pk);
    sg_prof_end(SG_K_TILE_SORT, st);
}

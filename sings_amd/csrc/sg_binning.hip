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
//               longer lists (round 3): a BUCKET sort -- sg_tile_partition_kernel splits a list by key value into groups of
//               <= 1024 entries (one O(n) pass pair per tile), sg_group_sort_kernel sorts every group in LDS, one workgroup
//               per group, chip-wide.  (Rounds 1-2: bitonic sort of 4096-entry chunks + a rank merge of the chunks:
//               O(n log^2 n) comparators per tile on ONE workgroup per chunk, 44 + 21 us on an avatar frame.)
// Traffic: 12 B + 8 B written, 20 B read, 4 B written per pair -- instead of six 24-B/pair radix passes.
// (Round 1 ran scan / scatter / sort / rank as four launches: 10 + 12 + 18 + 4 us at cfg3, all latency.)
#include "sg_sort.h"
#include <atomic>

// Exclusive scans over the T tile counts of
//   q0 pairs (-> ranges, cursors), q1 backward work items, q2 checkpoint slots, q3 sort items, q4 rank items, q5 lists > 512 (total only).
// Workgroup b owns tiles [b * SG_SCAN_BS * tpt, (b + 1) * SG_SCAN_BS * tpt), one tile per thread and round.  Instead of a second
// kernel (or a look-back chain) every workgroup first REDUCES the counts of all tiles in front of its range itself:
// at most T words per workgroup, coalesced and L2-resident.  The kernel is a handful of waves that start with a cold
// instruction cache, so its loops are deliberately NOT unrolled: the fully unrolled version (1200 instructions of
// straight-line code) took 22 us at 8160 tiles, this one 12 us.
// The work lists themselves are written by the (chip-wide) scatter kernel from the per-tile
// `plan` = (first backward item, first sort item, first rank item, pair count).
#define SG_SCAN_NQ 6
// BS = threads (= tiles per round) of a scan workgroup: 256 for images of few tiles (several CUs even at 1000 tiles),
// 1024 for many tiles (fewer workgroups re-reducing the counts in front of them)
__device__ __forceinline__ void sg_scan_derive(uint32_t v, uint32_t q[SG_SCAN_NQ])
{
    const uint32_t seg = sg_nseg(v);
    // q3: partition work items (one per list the compositing workgroup does not sort itself); q4: group slots reserved for it
    // q5: lists of more than half a row of SgBin::tile_keys (512; bits 20+: 8192, half a long row; T < 2^20) -- none of them: the
    // published count says so (SG_COUNT_FLAG_HALF_ROWS, SG_COUNT_FLAG_HALF_LONG_ROWS)
    q[0] = v; q[1] = seg ? seg : 1u; q[2] = seg; q[3] = v > SG_WSORT_MAX ? 1u : 0u; q[4] = sg_group_slots(v);
    q[5] = (v > SG_WSORT_MAX / 2 ? 1u : 0u) | (v > SG_TILE_ROW_LONG / 2 ? 1u << 20 : 0u);
}

// Early pair count: one 64-bit system-scope store to a mapped, coherent host word (valid bit | flags << 32 | R) -- visible to a
// polling host thread while this kernel and the composite behind it are still running.
__device__ __forceinline__ void sg_publish_count(unsigned long long *signal, uint32_t R, uint32_t flags)
{
    if (signal) __hip_atomic_store(signal, (1ull << 63) | ((unsigned long long)flags << 32) | R, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int SG_SCAN_BS>
__global__ void __launch_bounds__(SG_SCAN_BS)
sg_tile_scan_kernel(int T, int gx, int tpt, const uint32_t *__restrict__ tile_count,
                    uint2 *__restrict__ ranges, uint32_t *__restrict__ cursor, uint32_t *__restrict__ header,
                    uint32_t cap, uint32_t sort_cap, uint32_t rank_cap, uint4 *__restrict__ plan,
                    uint32_t *__restrict__ ck_start, uint32_t items_cap, int short_lists, unsigned long long *signal, size_t bin_stride,
                    int direct)
{
    constexpr int NQ = SG_SCAN_NQ;
    __shared__ uint32_t wsum[NQ][SG_SCAN_BS / 64];
    __shared__ uint32_t carry[NQ];
    {   // frame blockIdx.y of the launch: its binning workspace (K = 1: offset 0)
        const size_t off = (size_t)blockIdx.y * bin_stride;
        tile_count = sg_at(tile_count, off); ranges = sg_at(ranges, off); cursor = sg_at(cursor, off); header = sg_at(header, off);
        plan = sg_at(plan, off); ck_start = sg_at(ck_start, off);
        if (signal) signal += blockIdx.y;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int first = blockIdx.x * SG_SCAN_BS * tpt;
    // ---- 1. totals of everything in front of this workgroup's range
    uint32_t acc[NQ] = { 0, 0, 0, 0, 0, 0 };
    for (int t = tid; t < first; t += SG_SCAN_BS) {
        uint32_t q[NQ];
        const uint32_t v = tile_count[sg_ctr_of_tile((uint32_t)t, (uint32_t)gx)];
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
        uint32_t q[NQ] = { 0, 0, 0, 0, 0, 0 };
        const uint32_t v = ok ? tile_count[sg_ctr_of_tile((uint32_t)tile, (uint32_t)gx)] : 0u;
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
        const uint32_t hflags = (carry[0] > cap ? 1u : 0u) | ((short_lists && carry[3]) || (direct && header[8]) ? 2u : 0u);
        header[8] = 0u;
        header[1] = hflags;
        sg_publish_count(signal, carry[0], hflags | ((carry[5] & 0xfffffu) ? 0u : SG_COUNT_FLAG_HALF_ROWS) | ((carry[5] >> 20) ? 0u : SG_COUNT_FLAG_HALF_LONG_ROWS));
        header[3] = (uint32_t)T;
        header[4] = carry[3] < sort_cap ? carry[3] : sort_cap;
        header[5] = carry[1] < items_cap ? carry[1] : items_cap;
        header[6] = carry[4] < rank_cap ? carry[4] : rank_cap;
        header[7] = 0u;
    }
}

// One lane per pair (Gaussian-major list written by the preprocess): key -> its slot, no atomics.
// The first T threads also expand their tile's plan into the work lists of the sort / merge / backward kernels.
__global__ void __launch_bounds__(256)
sg_pair_scatter_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ pair_gid,
                       const uint32_t *__restrict__ pair_tile, const uint32_t *__restrict__ pair_local,
                       const float *__restrict__ depth, const uint32_t *__restrict__ start,
                       uint64_t *__restrict__ pair_keys, uint32_t cap, int T, const uint4 *__restrict__ plan,
                       uint4 *__restrict__ sort_items, uint2 *__restrict__ rank_items, uint32_t sort_cap,
                       uint32_t rank_cap, uint32_t *__restrict__ items, uint32_t items_cap, uint32_t *__restrict__ item_w,
                       size_t bin_stride, size_t geom_stride, uint8_t *__restrict__ rec_valid, int direct)
{
    rec_valid = sg_at(rec_valid, (size_t)blockIdx.y * bin_stride);
    {   // frame blockIdx.y
        const size_t off = (size_t)blockIdx.y * bin_stride;
        header = sg_at(header, off); pair_gid = sg_at(pair_gid, off); pair_tile = sg_at(pair_tile, off); pair_local = sg_at(pair_local, off);
        start = sg_at(start, off); pair_keys = sg_at(pair_keys, off); plan = sg_at(plan, off); sort_items = sg_at(sort_items, off);
        rank_items = sg_at(rank_items, off); items = sg_at(items, off); item_w = sg_at(item_w, off);
        depth = sg_at(depth, (size_t)blockIdx.y * geom_stride);
    }
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (uint32_t tile = gtid; tile < (uint32_t)T; tile += nthreads) {
        const uint4 pl = plan[tile];
        const uint32_t nseg = sg_nseg(pl.w), nit = nseg ? nseg : 1u;
        for (uint32_t sg = 0; sg < nit; sg++)
            if (pl.x + sg < items_cap) { items[pl.x + sg] = tile | (sg << 20); item_w[pl.x + sg] = 0u; }
        if (pl.w > SG_WSORT_MAX && pl.y < sort_cap) {                                              // (tile, first group slot, first entry, entries)
            const uint32_t s0 = start[tile] < cap ? start[tile] : cap, e0 = start[tile] + pl.w < cap ? start[tile] + pl.w : cap;
            sort_items[pl.y] = make_uint4(tile, pl.z, s0, e0 - s0);
        }
    }
    const uint32_t R = direct ? 0u : (header[0] < cap ? header[0] : cap);     // (direct binning: the preprocess placed the keys itself)
    for (uint32_t i = gtid; i < R; i += nthreads) {
        uint32_t gid = pair_gid[i];
        uint32_t slot = start[pair_tile[i]] + pair_local[i];
        if (slot < cap) pair_keys[slot] = ((uint64_t)__float_as_uint(depth[gid]) << 32) | gid;
        if (rec_valid) rec_valid[i] = 0;                    // pair i IS gradient-record slot i (Gaussian-major): nothing written yet
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
#ifndef SG_DIRECT_SCAN_WGS
#define SG_DIRECT_SCAN_WGS 8          // direct binning: workgroups per frame that scan (each all T counts) and write a share of the outputs
#endif
#define SG_SS_MAX_TILES 32768
__global__ void __launch_bounds__(SG_SS_THREADS)
sg_scan_scatter_kernel(int T, int gx, const uint32_t *__restrict__ tile_count, uint2 *__restrict__ ranges,
                       uint32_t *__restrict__ cursor, uint32_t *__restrict__ header, uint32_t cap, uint32_t sort_cap,
                       uint32_t rank_cap, uint4 *__restrict__ plan, uint32_t *__restrict__ ck_start, uint32_t items_cap,
                       const uint32_t *__restrict__ pair_gid, const uint32_t *__restrict__ pair_tile,
                       const uint32_t *__restrict__ pair_local, const float *__restrict__ depth,
                       uint64_t *__restrict__ pair_keys, uint4 *__restrict__ sort_items, uint2 *__restrict__ rank_items,
                       uint32_t *__restrict__ items, uint32_t *__restrict__ item_w, int short_lists, unsigned long long *signal,
                       size_t bin_stride, size_t geom_stride, uint8_t *__restrict__ rec_valid, int direct)
{
    rec_valid = sg_at(rec_valid, (size_t)blockIdx.y * bin_stride);
    constexpr int NQ = SG_SCAN_NQ;
    extern __shared__ uint32_t sStart[];                     // [T] counts, then exclusive pair prefix
    {   // frame blockIdx.y of the launch: its binning workspace and depth array (K = 1: offsets 0)
        const size_t off = (size_t)blockIdx.y * bin_stride;
        tile_count = sg_at(tile_count, off); ranges = sg_at(ranges, off); cursor = sg_at(cursor, off); header = sg_at(header, off);
        plan = sg_at(plan, off); ck_start = sg_at(ck_start, off); pair_gid = sg_at(pair_gid, off); pair_tile = sg_at(pair_tile, off);
        pair_local = sg_at(pair_local, off); pair_keys = sg_at(pair_keys, off); sort_items = sg_at(sort_items, off);
        rank_items = sg_at(rank_items, off); items = sg_at(items, off); item_w = sg_at(item_w, off);
        depth = sg_at(depth, (size_t)blockIdx.y * geom_stride);
        if (signal) signal += blockIdx.y;
    }
    __shared__ uint32_t wsum[NQ][SG_SS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int t = tid; t < T; t += SG_SS_THREADS) sStart[t] = tile_count[sg_ctr_of_tile((uint32_t)t, (uint32_t)gx)];
    __syncthreads();
    const int tpt = (T + SG_SS_THREADS - 1) / SG_SS_THREADS;
    const int t0 = tid * tpt < T ? tid * tpt : T, t1 = t0 + tpt < T ? t0 + tpt : T;
    uint32_t own[NQ] = { 0, 0, 0, 0, 0, 0 };
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
    // (direct binning: the tiles are dealt out one by one -- a thread then writes 8 / gridDim of its 8 tiles at cfg3, not all or none)
    const bool wthread = (uint32_t)tid % gridDim.x == blockIdx.x;
    for (int t = t0; t < t1; t++) {
        const uint32_t v = sStart[t];
        uint32_t q[NQ];
        sg_scan_derive(v, q);
        const bool writer = direct ? (uint32_t)t % gridDim.x == blockIdx.x : wthread;
        if (writer) {
            const uint32_t s = run[0] < cap ? run[0] : cap, e = run[0] + v < cap ? run[0] + v : cap;
            ranges[t] = v ? make_uint2(s, e) : make_uint2(0u, 0u);
            cursor[t] = run[0];
            ck_start[t] = q[2] ? run[2] : 0xffffffffu;
            plan[t] = make_uint4(run[1], run[3], run[4], v);
            for (uint32_t sg = 0; sg < q[1]; sg++)
                if (run[1] + sg < items_cap) { items[run[1] + sg] = (uint32_t)t | (sg << 20); item_w[run[1] + sg] = 0u; }
            if (q[3] && run[3] < sort_cap) sort_items[run[3]] = make_uint4((uint32_t)t, run[4], s, e - s);     // (tile, first group slot, first entry, entries)
        }
        sStart[t] = run[0];
#pragma unroll
        for (int a = 0; a < NQ; a++) run[a] += q[a];
    }
    if (blockIdx.x == 0 && tid == 0) {
        header[0] = tot[0];
        // bit 0: more pairs than the workspace holds; bit 1: a list needs the long-list kernels the caller told us to skip.
        // Either way the lists are incomplete / unsorted and the composite kernels touch nothing.
        // (header[8]: a lane of the direct preprocess met a rank beyond its tile's row)
        const uint32_t hflags = (tot[0] > cap ? 1u : 0u) | ((short_lists && tot[3]) || (direct && header[8]) ? 2u : 0u);
        header[8] = 0u;
        header[1] = hflags;
        sg_publish_count(signal, tot[0], hflags | ((tot[5] & 0xfffffu) ? 0u : SG_COUNT_FLAG_HALF_ROWS) | ((tot[5] >> 20) ? 0u : SG_COUNT_FLAG_HALF_LONG_ROWS));     // the host may be waiting for exactly this (SgRasterSettings.count_signal)
        header[3] = (uint32_t)T;
        header[4] = tot[3] < sort_cap ? tot[3] : sort_cap;
        header[5] = tot[1] < items_cap ? tot[1] : items_cap;
        header[6] = tot[4] < rank_cap ? tot[4] : rank_cap;
        header[7] = 0u;                                   // groups the partition kernel hands to the group kernel
    }
    if (direct) return;                                   // direct binning: the preprocess placed the keys itself (sg_store_proj)
    __syncthreads();
    const uint32_t R = tot[0] < cap ? tot[0] : cap;
    for (uint32_t i = blockIdx.x * SG_SS_THREADS + tid; i < R; i += gridDim.x * SG_SS_THREADS) {
        const uint32_t gid = pair_gid[i];
        const uint32_t slot = sStart[pair_tile[i]] + pair_local[i];
        if (slot < cap) pair_keys[slot] = ((uint64_t)__float_as_uint(depth[gid]) << 32) | gid;
        // pair i IS gradient-record slot i (the preprocess reserves both Gaussian-major): few-tile frames mark the records the
        // sparse backward composite writes; here, one lane per pair anyway, every mark is cleared -- instead of a kernel that
        // streams 36 B of zeros per pair in front of every backward (round 3: 27 MB, 10 us per avatar frame)
        if (rec_valid) rec_valid[i] = 0;
    }
}

// ---- long lists: bucket sort ---------------------------------------------------------------
// A list of n > 1024 keys (depth_bits << 32 | gid, unique) is split BY VALUE: bucket = (key - min) >> shift, 1024 buckets over
// the list's own key range; whole buckets are packed greedily into groups of <= 1024 keys, and every group is sorted on its
// own by sg_group_sort_kernel.  Groups are ranges of the bucket-ordered array, so the concatenation of the sorted groups IS the
// sorted list -- no merge.  One workgroup per tile does min / max, histogram, scan, scatter and the packing: three coalesced
// passes over the tile's keys (L2-resident after the first) instead of O(n log^2 n) comparator stages.
// A bucket that alone holds more than 1024 keys (a dense depth cluster next to an outlier) is split again over ITS key range
// (level 1: ping-pong back into the first buffer); what is still too large after that -- > 1024 keys within 2^-20 of the
// list's key range -- is ordered by counting (every key counts the smaller ones; O(m^2), correct for any input, never seen in a
// rendered scene; tests/test_gpu_raster.py crafts one).
// Group slots: the scan reserves sg_group_slots(n) consecutive slots per long tile (plan.z); unused ones are written with
// length 0, so the group kernel needs no counter and no atomics.  A group item is (first index | buffer flag, length | tile << 11).
#define SG_PT_THREADS 1024
#define SG_PT_NB 1024
#define SG_PT_BIGS 256
#define SG_GROUP_IN_A 0x80000000u      // group item flag: the group's keys are in pair_keys (level-1 output), else in the scratch

__device__ __forceinline__ uint64_t sg_shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}

struct SgPartLds {
    uint32_t off[SG_PT_NB + 1];        // exclusive bucket offsets of the current pass
    uint32_t cur[SG_PT_NB];            // histogram, then scatter cursors
    uint32_t nx[SG_PT_NB];             // greedy packing: first bucket of the group after the one that starts at this bucket
    uint32_t wtot[SG_PT_THREADS / 64];
    uint64_t wmin[SG_PT_THREADS / 64], wmax[SG_PT_THREADS / 64];
    uint2 bigs[SG_PT_BIGS];            // (start, length) of the buckets of the current pass that hold more than 1024 keys
    uint2 bigs0[SG_PT_BIGS];           // the level-0 list, kept while the level-1 passes reuse the tables above
    uint32_t nbig;                     // entries of bigs[] (counts on past SG_PT_BIGS)
    uint32_t gslot;                    // next group slot of this tile
};

// Keys of in[0, m) -> bucket order in out[0, m); groups of whole buckets (<= 1024 keys) are appended to `groups` (slots below gend).
// `abs0`: index of in[0] / out[0] in the per-pair arrays; `flag`: SG_GROUP_IN_A iff `out` is pair_keys.  Buckets with more than 1024
// keys are listed in L.bigs / L.nbig (relative to out).  Workgroup-uniform; contains barriers; ends with one.
__device__ void sg_partition_pass(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t m, uint32_t abs0,
                                  uint32_t flag, uint32_t tile, SgPartLds &L, uint2 *__restrict__ groups, uint32_t gend, int tid)
{
    // A thread's keys: U per sweep, every load of a sweep issued before the first use (as a plain strided loop hipcc emits load ->
    // wait -> use: one request in flight per wave, a memory latency per key); lists of up to U * 1024 keys -- every list of an avatar
    // frame -- are read ONCE and stay in registers through range, histogram and scatter.
    constexpr int U = 8;
    constexpr uint64_t NONE = ~0ull;                                   // (a real key never has all depth bits set: depth > 0.2)
    const int lane = tid & 63, wid = tid >> 6;
    const bool resident = m <= (uint32_t)U * SG_PT_THREADS;
    uint64_t kr[U];
    auto sweep = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < U; u++) { const uint32_t i = i0 + (uint32_t)u * SG_PT_THREADS + tid; kr[u] = i < m ? in[i] : NONE; }
    };
    // 1. key range
    uint64_t kmin = NONE, kmax = 0ull;
    for (uint32_t i0 = 0; i0 < m; i0 += U * SG_PT_THREADS) {
        sweep(i0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (kr[u] != NONE) { kmin = kr[u] < kmin ? kr[u] : kmin; kmax = kr[u] > kmax ? kr[u] : kmax; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = sg_shfl_xor_u64(kmin, o), b = sg_shfl_xor_u64(kmax, o);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { L.wmin[wid] = kmin; L.wmax[wid] = kmax; }
    L.cur[tid] = 0u;
    if (tid == 0) L.nbig = 0u;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < SG_PT_THREADS / 64; w++) { const uint64_t a = L.wmin[w], b = L.wmax[w]; kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
    const uint64_t span = kmax - kmin;
    const int bits = span ? 64 - __builtin_clzll(span) : 0;
    const int shift = bits > 10 ? bits - 10 : 0;                       // (span >> shift) < 1024
    // 2. histogram
    for (uint32_t i0 = 0; i0 < m; i0 += U * SG_PT_THREADS) {
        if (!resident) sweep(i0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (kr[u] != NONE) atomicAdd(&L.cur[(uint32_t)((kr[u] - kmin) >> shift)], 1u);
    }
    __syncthreads();
    // 3. exclusive scan over the 1024 buckets (thread = bucket)
    const uint32_t cnt = L.cur[tid];
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) L.wtot[wid] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll 1
    for (int w = 0; w < wid; w++) woff += L.wtot[w];
    const uint32_t excl = woff + incl - cnt;
    L.off[tid] = excl;
    if (tid == SG_PT_THREADS - 1) L.off[SG_PT_NB] = excl + cnt;
    if (cnt > SG_WSORT_MAX) { const uint32_t k = atomicAdd(&L.nbig, 1u); if (k < SG_PT_BIGS) L.bigs[k] = make_uint2(excl, cnt); }
    __syncthreads();                                                   // (everybody has read its cur[] entry and the wave totals)
    L.cur[tid] = excl;
    __syncthreads();
    // 4. scatter into bucket order (the order inside a bucket is arbitrary: every group is sorted afterwards)
    for (uint32_t i0 = 0; i0 < m; i0 += U * SG_PT_THREADS) {
        if (!resident) sweep(i0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (kr[u] != NONE) out[atomicAdd(&L.cur[(uint32_t)((kr[u] - kmin) >> shift)], 1u)] = kr[u];
    }
    // 5. greedy packing.  nx[b]: the largest j > b with off[j] - off[b] <= 1024 (binary search; b + 1 if bucket b alone is larger)
    {
        int lo = tid + 1, hi = SG_PT_NB;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L.off[mid] - excl <= SG_WSORT_MAX) lo = mid; else hi = mid - 1; }
        L.nx[tid] = (uint32_t)lo;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t g = L.gslot;
        for (uint32_t b = 0; b < SG_PT_NB;) {
            const uint32_t j = L.nx[b], s0 = L.off[b], len = L.off[j] - s0;
            if (len > SG_WSORT_MAX) { b++; continue; }                 // a big bucket (listed in bigs[])
            if (len && g < gend) groups[g++] = make_uint2((abs0 + s0) | flag, len | (tile << 11));
            b = j;
        }
        L.gslot = g;
    }
    __syncthreads();
}

// A whole list in LDS: keys in[0, n) (n <= SG_PT_U * 1024) are split into SG_PT_FINE buckets by value -- eight times finer than
// the multi-level path above: the keys of a tile cluster in depth (the front and the back surface of a limb: 5 000 keys within
// 7 % of the tile's depth range), and at 1024 buckets every key of such a band shared its bucket with ~75 others -- scattered
// into sKeys, and every key then counts the smaller keys of its own bucket: final position = bucket start + that count (keys are
// unique).  Cost: sum over the buckets of (keys in it)^2 LDS reads, ~10 per key.  A bucket of more than SG_PT_DENSE keys (equal
// depth bits, or a cluster next to a far outlier) makes the caller take the multi-level path instead (returns false, nothing
// written).  LDS: sKeys [SG_PT_U * 1024] followed by the bucket table [SG_PT_FINE].  Workgroup-uniform; contains barriers.
#define SG_PT_U 12
#define SG_PT_FINE 8192
#define SG_PT_DENSE 1024
__device__ bool sg_sort_resident(const uint64_t *__restrict__ in, uint32_t n, uint32_t abs0, uint32_t tile, SgPartLds &L,
                                 uint64_t *__restrict__ sKeys, uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys,
                                 int tid)
{
    constexpr int U = SG_PT_U, BPT = SG_PT_FINE / SG_PT_THREADS;        // keys / buckets per thread
    constexpr uint64_t NONE = ~0ull;
    uint32_t *tab = (uint32_t *)(sKeys + (size_t)U * SG_PT_THREADS);    // counts -> cursors -> bucket ends
    const int lane = tid & 63, wid = tid >> 6;
    uint64_t kr[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t i = (uint32_t)u * SG_PT_THREADS + tid; kr[u] = i < n ? in[i] : NONE; }
#pragma unroll
    for (int q = 0; q < BPT; q++) tab[q * SG_PT_THREADS + tid] = 0u;
    uint64_t kmin = NONE, kmax = 0ull;
#pragma unroll
    for (int u = 0; u < U; u++)
        if (kr[u] != NONE) { kmin = kr[u] < kmin ? kr[u] : kmin; kmax = kr[u] > kmax ? kr[u] : kmax; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = sg_shfl_xor_u64(kmin, o), b = sg_shfl_xor_u64(kmax, o);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { L.wmin[wid] = kmin; L.wmax[wid] = kmax; }
    if (tid == 0) L.nbig = 0u;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < SG_PT_THREADS / 64; w++) { const uint64_t a = L.wmin[w], b = L.wmax[w]; kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
    const uint64_t span = kmax - kmin;
    const int bits = span ? 64 - __builtin_clzll(span) : 0;
    const int shift = bits > 13 ? bits - 13 : 0;                        // (span >> shift) < 8192
    uint32_t bk[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        bk[u] = (uint32_t)((kr[u] - kmin) >> shift);
        if (kr[u] != NONE) atomicAdd(&tab[bk[u]], 1u);
    }
    __syncthreads();
    // exclusive scan: thread t owns buckets [BPT t, BPT t + BPT)
    uint32_t c[BPT], own = 0;
    bool dense = false;
#pragma unroll
    for (int q = 0; q < BPT; q++) { c[q] = tab[BPT * tid + q]; own += c[q]; dense |= c[q] > SG_PT_DENSE; }
    uint32_t incl = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) L.wtot[wid] = incl;
    if (dense) atomicAdd(&L.nbig, 1u);
    __syncthreads();
    if (L.nbig) { __syncthreads(); return false; }                     // (uniform; the barrier keeps nbig intact until everybody has read it)
    uint32_t run = incl - own;
#pragma unroll 1
    for (int w = 0; w < wid; w++) run += L.wtot[w];
#pragma unroll
    for (int q = 0; q < BPT; q++) { tab[BPT * tid + q] = run; run += c[q]; }
    __syncthreads();
    // scatter; afterwards tab[b] is the END of bucket b (= start of bucket b + 1)
#pragma unroll
    for (int u = 0; u < U; u++)
        if (kr[u] != NONE) sKeys[atomicAdd(&tab[bk[u]], 1u)] = kr[u];
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < U; u++) {
        const uint64_t key = kr[u];
        if (key == NONE) continue;
        const uint32_t s0 = bk[u] ? tab[bk[u] - 1] : 0u, e0 = tab[bk[u]];
        uint32_t rank = 0;
        uint32_t j = s0;
        // (phase clocks on an avatar frame: the mean tile spends 2 us here, the one with the densest buckets 13 -- the loop is LDS
        //  latency, so sixteen independent reads are in flight per round: aligned 16-byte reads of two keys each)
        if ((j & 1u) && j < e0) { rank += sKeys[j] < key; j++; }
        for (; j + 16 <= e0; j += 16) {
            ulonglong2 q[8];
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = *(const ulonglong2 *)(sKeys + j + 2 * i);
#pragma unroll
            for (int i = 0; i < 8; i++) rank += (uint32_t)(q[i].x < key) + (uint32_t)(q[i].y < key);
        }
        for (; j + 4 <= e0; j += 4) {
            const ulonglong2 q0 = *(const ulonglong2 *)(sKeys + j), q1 = *(const ulonglong2 *)(sKeys + j + 2);
            rank += (uint32_t)(q0.x < key) + (uint32_t)(q0.y < key) + (uint32_t)(q1.x < key) + (uint32_t)(q1.y < key);
        }
        for (; j < e0; j++) rank += sKeys[j] < key;
        point_list[abs0 + s0 + rank] = (uint32_t)key;
        if (point_keys) point_keys[abs0 + s0 + rank] = ((uint64_t)tile << 32) | (key >> 32);
    }
    __syncthreads();
    return true;
}

// keys buf[0, m) (unsorted) -> point_list / point_keys [abs0, abs0 + m) in ascending order, by counting: O(m^2 / 1024) per thread
__device__ void sg_count_sort(const uint64_t *__restrict__ buf, uint32_t m, uint32_t abs0, uint32_t tile, uint64_t *__restrict__ slab,
                              uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys, int tid)
{
    for (uint32_t i0 = 0; i0 < m; i0 += SG_PT_THREADS) {
        const uint32_t i = i0 + tid;
        const uint64_t key = i < m ? buf[i] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t t0 = 0; t0 < m; t0 += SG_PT_NB) {
            __syncthreads();
            slab[tid] = t0 + tid < m ? buf[t0 + tid] : ~0ull;
            __syncthreads();
            const uint32_t tn = m - t0 < SG_PT_NB ? m - t0 : SG_PT_NB;
            for (uint32_t j = 0; j < tn; j++) rank += slab[j] < key;
        }
        if (i < m) {
            point_list[abs0 + rank] = (uint32_t)key;
            if (point_keys) point_keys[abs0 + rank] = ((uint64_t)tile << 32) | (key >> 32);
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(SG_PT_THREADS)
sg_tile_partition_kernel(uint32_t *header, const uint4 *__restrict__ part_items,
                         uint64_t *__restrict__ pair_keys, uint64_t *__restrict__ scratch, uint2 *__restrict__ groups,
                         uint32_t group_cap, uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys,
                         uint32_t resident_max, size_t bin_stride, const uint64_t *__restrict__ row_keys, uint32_t key_pitch)
{
    __shared__ SgPartLds L;
    {   // frame blockIdx.y
        const size_t off = (size_t)blockIdx.y * bin_stride;
        header = sg_at(header, off); part_items = sg_at(part_items, off); pair_keys = sg_at(pair_keys, off); scratch = sg_at(scratch, off);
        groups = sg_at(groups, off); point_list = sg_at(point_list, off); point_keys = sg_at(point_keys, off);
        row_keys = sg_at(row_keys, off);
    }
    extern __shared__ __attribute__((aligned(16))) uint64_t sKeys[];                               // resident_max keys (>= 1024: the counting fallback's slab)
    const int tid = threadIdx.x;
    // (a latency chain: the work item carries the list's range itself, and the first item is requested together with the header)
    uint4 it = part_items[blockIdx.x];
    const uint32_t nitems = header[1] ? 0u : header[4];
    for (uint32_t li = blockIdx.x; li < nitems; li += gridDim.x) {
        if (li != blockIdx.x) it = part_items[li];                    // (tile, first reserved group slot, first entry, entries)
        const uint32_t tile = it.x;
        const uint2 r = make_uint2(it.z, it.z + it.w);
        const uint32_t n = it.w;
        const uint32_t g0 = it.y < group_cap ? it.y : group_cap;
        const uint32_t gend = g0 + sg_group_slots(n) < group_cap ? g0 + sg_group_slots(n) : group_cap;
        __syncthreads();
        if (tid == 0) L.gslot = g0;
        __syncthreads();
        // Lists that fit the workgroup's LDS (every list of an avatar frame): sorted right here -- partition into LDS, then every
        // key counts the smaller keys of ITS bucket (ten on average): no second kernel, no trip through memory.
        // (direct binning: the list's unsorted keys are row `tile` of tile_keys)
        const uint64_t *in0 = key_pitch ? row_keys + (size_t)tile * key_pitch : pair_keys + r.x;
        if (n <= resident_max && sg_sort_resident(in0, n, r.x, tile, L, sKeys, point_list, point_keys, tid)) {
            for (uint32_t g = g0 + tid; g < gend; g += SG_PT_THREADS) groups[g] = make_uint2(0u, 0u);
            continue;
        }
        // level 0: pair_keys -> scratch
        sg_partition_pass(in0, scratch + r.x, n, r.x, 0u, tile, L, groups, gend, tid);
        const uint32_t nb0 = L.nbig;
        if (nb0) {
            // level 1 for every bucket of more than 1024 keys: scratch -> pair_keys over the bucket's own key range.  (The level-1
            // passes reuse the LDS tables: the level-0 list moves to bigs0 first.)
            if (tid < SG_PT_BIGS) L.bigs0[tid] = L.bigs[tid];
            __syncthreads();
            for (uint32_t k = 0; k < nb0 && k < SG_PT_BIGS; k++) {
                const uint32_t s0 = L.bigs0[k].x, m = L.bigs0[k].y;
                __syncthreads();
                sg_partition_pass(scratch + r.x + s0, pair_keys + r.x + s0, m, r.x + s0, SG_GROUP_IN_A, tile, L, groups, gend, tid);
                const uint32_t nb1 = L.nbig;
                if (nb1 > SG_PT_BIGS) {
                    // more big sub-buckets than the list holds: order the whole bucket by counting (its groups, emitted above, then
                    // re-sort parts of it into the same places: harmless)
                    sg_count_sort(pair_keys + r.x + s0, m, r.x + s0, tile, sKeys, point_list, point_keys, tid);
                } else {
                    for (uint32_t q = 0; q < nb1; q++) {
                        const uint32_t s1 = L.bigs[q].x, m1 = L.bigs[q].y;
                        __syncthreads();
                        sg_count_sort(pair_keys + r.x + s0 + s1, m1, r.x + s0 + s1, tile, sKeys, point_list, point_keys, tid);
                    }
                }
            }
            if (nb0 > SG_PT_BIGS) {
                // (a list with more than 256 separate dense clusters of > 1024 keys each: > 262 144 entries in ONE tile.)  The
                // level-0 list is incomplete: order the whole list by counting, from a copy in pair_keys (the scratch array may be
                // the point_keys output).
                __syncthreads();
                for (uint32_t i = tid; i < n; i += SG_PT_THREADS) pair_keys[r.x + i] = scratch[r.x + i];
                __syncthreads();
                sg_count_sort(pair_keys + r.x, n, r.x, tile, sKeys, point_list, point_keys, tid);
            }
        }
        __syncthreads();
        // unused reserved slots: length 0
        for (uint32_t g = L.gslot + tid; g < gend; g += SG_PT_THREADS) groups[g] = make_uint2(0u, 0u);
        if (tid == 0 && L.gslot > g0) atomicAdd(&header[7], L.gslot - g0);     // the group kernel has work
    }
}

// One workgroup per group: <= 1024 keys -> sorted, as Gaussian ids into point_list (and upstream-format keys on request).
__global__ void __launch_bounds__(256)
sg_group_sort_kernel(const uint32_t *__restrict__ header, const uint2 *__restrict__ groups, const uint64_t *__restrict__ pair_keys,
                     const uint64_t *__restrict__ scratch, uint32_t *__restrict__ point_list, uint64_t *__restrict__ point_keys,
                     size_t bin_stride)
{
    __shared__ uint64_t s[SG_WSORT_MAX + SG_RANKSORT_MAX];
    {   // frame blockIdx.y
        const size_t off = (size_t)blockIdx.y * bin_stride;
        header = sg_at(header, off); groups = sg_at(groups, off); pair_keys = sg_at(pair_keys, off); scratch = sg_at(scratch, off);
        point_list = sg_at(point_list, off); point_keys = sg_at(point_keys, off);
    }
    const int tid = threadIdx.x;
    const uint32_t nslots = header[1] || header[7] == 0u ? 0u : header[6];    // (header[7]: groups emitted; 0 = every list was sorted in LDS)
    for (uint32_t gi = blockIdx.x; gi < nslots; gi += gridDim.x) {
        const uint2 g = groups[gi];
        const uint32_t len = g.y & 0x7ffu, tile = g.y >> 11;
        if (len == 0u) continue;
        const uint32_t start = g.x & ~SG_GROUP_IN_A;
        const uint64_t *src = ((g.x & SG_GROUP_IN_A) ? pair_keys : scratch) + start;
        __syncthreads();
        sg_sort_short_list(src, (int)len, s, s + SG_WSORT_MAX, tid);     // (all of src is in LDS before anything below is written)
        for (uint32_t i = tid; i < len; i += 256) {
            const uint64_t k = s[i];
            point_list[start + i] = (uint32_t)k;
            if (point_keys) point_keys[start + i] = ((uint64_t)tile << 32) | (k >> 32);
        }
    }
}

// Raising a kernel's dynamic-LDS limit is a host-side call into the runtime (a table update under a lock): asked ONCE per
// (kernel, device) and remembered per device id -- not in a process-wide flag (a second GPU used from the same process must get
// its own call), and a refusal is remembered too (the callers then take their fallback paths).  Round 3 asked on every forward.
static bool sg_dyn_lds_limit(int which, const void *fn, int bytes)
{
    static std::atomic<signed char> state[2][64];                // 0 unknown, 1 granted, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess || ((void)hipGetLastError(), false);
    signed char st = state[which][dev].load(std::memory_order_acquire);
    if (st == 0) {
        st = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : -1;
        if (st < 0) (void)hipGetLastError();
        state[which][dev].store(st, std::memory_order_release);
    }
    return st > 0;
}

void sg_launch_binning(const SgCam &c, const SgBatch &bt, int P, const int32_t *radii, SgGeom g, SgBin b, size_t cap,
                       int write_keys, hipStream_t st)
{
    (void)radii; (void)P;
    const unsigned K = (unsigned)bt.K;                        // frames: blockIdx.y of every kernel below
    const int T = c.gx * c.gy;
    const uint32_t cap32 = sg_cap32(cap);
    uint64_t *pk = write_keys ? b.point_keys : (uint64_t *)nullptr;
    const int short_lists = (c.flags & SG_FLAG_SHORT_LISTS) ? 1 : 0;
    // 4 T bytes of dynamic LDS (up to 128 KiB).  Above the default 64-KiB limit (images of >= ~16 k tiles: 2048 x 2048) the
    // kernel needs its limit raised -- per DEVICE and not remembered in a process-wide flag (a second GPU used from the same
    // process, or a failed call, must not leave the launch below without it): asked for whenever it is needed (a host-side
    // table update, no stream operation), and if the runtime refuses, the two-kernel path below does the same job.
    // K > 1 frames of MANY tiles take the two-kernel path as well: in the fused kernel every workgroup repeats the scan of the T tile
    // counts in front of its share of the scatter -- the price of saving a launch when ONE frame's 211 workgroups are all the GPU
    // has; with K x 211 workgroups queued a 10-us scan kernel (one workgroup row per frame) + a plain scatter are faster (cfg3, 8
    // cameras: 10.4 + 78.0 against 110.9 us; few-tile frames: 75.6 against 77.0, left fused)
    // DIRECT binning (sg_direct_keys): nothing to scatter -- the fused kernel's scan part alone, a few workgroups per frame
    const uint32_t key_pitch = sg_key_pitch(c.gx, c.gy, c.flags);
    const int direct = key_pitch ? 1 : 0;
    bool fused = T <= SG_SS_MAX_TILES && (K == 1 || sg_lds_hist(c.gx, c.gy) || direct);
    if (fused && (size_t)T * 4 + 1024 > 64 * 1024)
        fused = sg_dyn_lds_limit(0, (const void *)sg_scan_scatter_kernel, SG_SS_MAX_TILES * 4);
    if (fused) {
        sg_prof_begin(SG_K_TILE_SCAN, st);
        size_t want = (cap + 4 * SG_SS_THREADS - 1) / (4 * SG_SS_THREADS);     // ~4 pairs per thread
        const int grid = direct ? SG_DIRECT_SCAN_WGS : (int)(want < 8 ? 8 : (want > 256 ? 256 : want));
        hipLaunchKernelGGL(sg_scan_scatter_kernel, dim3(grid, K), dim3(SG_SS_THREADS), (size_t)T * 4, st, T, c.gx, b.tile_count, b.ranges,
                           b.cursor, b.header, cap32, sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.plan, b.ck_start,
                           sg_items_cap(T, cap), b.pair_gid, b.pair_tile, b.pair_local, g.depth, b.pair_keys, b.sort_items,
                           b.rank_items, b.items, b.item_w, short_lists, c.count_signal, bt.bin, bt.geom,
                           sg_lds_hist(c.gx, c.gy) ? b.rec_valid : (uint8_t *)nullptr, direct);
        sg_prof_end(SG_K_TILE_SCAN, st);
    } else {
        sg_prof_begin(SG_K_TILE_SCAN, st);
        const int tpt = (T + 65535) / 65536 > 0 ? (T + 65535) / 65536 : 1;
        const int sgrid = (T + 1024 * tpt - 1) / (1024 * tpt) > 0 ? (T + 1024 * tpt - 1) / (1024 * tpt) : 1;
        hipLaunchKernelGGL((sg_tile_scan_kernel<1024>), dim3(sgrid, K), dim3(1024), 0, st, T, c.gx, tpt, b.tile_count, b.ranges, b.cursor,
                           b.header, cap32, sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.plan, b.ck_start,
                           sg_items_cap(T, cap), short_lists, c.count_signal, bt.bin, direct);
        sg_prof_end(SG_K_TILE_SCAN, st);
        sg_prof_begin(SG_K_TILE_SCATTER, st);
        size_t want = (((direct || cap < (size_t)T) ? (size_t)T : cap) + 255) / 256;
        int grid = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(sg_pair_scatter_kernel, dim3(grid, K), dim3(256), 0, st, b.header, b.pair_gid, b.pair_tile,
                           b.pair_local, g.depth, b.cursor, b.pair_keys, cap32, T, b.plan, b.sort_items, b.rank_items,
                           sg_sort_items_cap(T, cap), sg_rank_items_cap(cap), b.items, sg_items_cap(T, cap), b.item_w, bt.bin, bt.geom,
                           sg_lds_hist(c.gx, c.gy) ? b.rec_valid : (uint8_t *)nullptr, direct);
        sg_prof_end(SG_K_TILE_SCATTER, st);
    }
    // lists longer than 1024 entries (the composite kernel sorts the others): both kernels exit at once when there are none
    if (short_lists) return;     // the caller vouches for short lists (checked on the device: header[1] bit 1)
    sg_prof_begin(SG_K_TILE_SORT, st);
    const uint32_t pgrid = sg_sort_items_cap(T, cap) < 512 ? sg_sort_items_cap(T, cap) : 512;
    // 128 KiB of dynamic LDS (96 KiB of keys + a 32-KiB bucket table): a list of up to 12 288 keys is sorted inside the workgroup.
    // If the runtime refuses the limit (asked for per launch: per device, no process-wide flag) only the counting slab is
    // allocated and every list takes the multi-level path.
    uint32_t resident = SG_PT_U * SG_PT_THREADS;
    size_t dyn = (size_t)resident * 8 + (size_t)SG_PT_FINE * 4;
    if (!sg_dyn_lds_limit(1, (const void *)sg_tile_partition_kernel, (int)dyn)) { resident = SG_PT_NB; dyn = (size_t)SG_PT_NB * 8; }
    hipLaunchKernelGGL(sg_tile_partition_kernel, dim3(pgrid, K), dim3(SG_PT_THREADS), dyn, st, b.header, b.sort_items,
                       b.pair_keys, b.point_keys, b.rank_items, sg_rank_items_cap(cap), b.point_list, pk,
                       resident > SG_PT_NB ? resident : 0u, bt.bin, b.tile_keys, key_pitch);
    const uint32_t ggrid = sg_rank_items_cap(cap) < 256 ? sg_rank_items_cap(cap) : 256;       // (usually nothing to do: header[7])
    hipLaunchKernelGGL(sg_group_sort_kernel, dim3(ggrid, K), dim3(256), 0, st, b.header, b.rank_items, b.pair_keys, b.point_keys,
                       b.point_list, pk, bt.bin);
    sg_prof_end(SG_K_TILE_SORT, st);
}

// Returning integer atomics on 65 k counter words, as the pair expansion of the preprocess issues them (four in flight per lane,
// scattered addresses): device scope on one array / device scope on one array per XCD / WORKGROUP scope on one array per XCD
// (a counter copy only ever touched from one XCD: its L2 is the coherence point).  Checks that no update is lost.  LAB 6.8
//   hipcc -O3 --offload-arch=gfx950 tools/atomic_scope_probe.hip -o /tmp/asp && /tmp/asp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define T 65536
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t *ctr, uint32_t *sink, int per_thread)
{
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xfu;
    uint32_t *c = MODE == 0 ? ctr : ctr + (size_t)xcc * T;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; i += 4) {
        uint32_t r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // neighbouring lanes: neighbouring counters (a splat's tiles), a random base per 4-lane group
            const uint32_t t = (hash((g >> 2) * 977u + (uint32_t)(i + u)) + (g & 3u)) & (T - 1);
            if (MODE == 2) r[u] = __hip_atomic_fetch_add(&c[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else r[u] = atomicAdd(&c[t], 1u);
        }
        acc += r[0] + r[1] + r[2] + r[3];
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}
// GROUP neighbouring lanes share a 64-B line of counters (a splat's tiles: 4x4 blocks of counters); W64: one 64-bit add per lane
// (two or four packed counters) -- half / a quarter of the operations for the same pairs
template <int GROUP, bool W64>
__global__ void __launch_bounds__(256) kg(uint32_t *ctr, uint32_t *sink, int per_thread)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned long long acc = 0;
    for (int i = 0; i < per_thread; i += 4) {
        unsigned long long r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = (((hash((g / GROUP) * 977u + (uint32_t)(i + u)) * 16u) & (T - 1)) + (g % GROUP) * (W64 ? 2u : 1u)) & (T - 2);
            if (W64) r[u] = atomicAdd((unsigned long long *)&ctr[t], 0x0000000100000001ull);
            else r[u] = atomicAdd(&ctr[t], 1u);
        }
        acc += r[0] + r[1] + r[2] + r[3];
    }
    if (acc == ~0ull) sink[0] = (uint32_t)acc;
}
template <int GROUP, bool W64>
static void run(const char *name, uint32_t *ctr, uint32_t *sink, int blocks, int per_thread)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        hipMemset(ctr, 0, 8 * T * 4); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((kg<GROUP, W64>), dim3(blocks), dim3(256), 0, 0, ctr, sink, per_thread);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%s %8.1f us  %.1f operations/ns\n", name, best * 1e3, (double)blocks * 256 * per_thread / (best * 1e6));
}
int main()
{
    uint32_t *ctr, *sink;
    hipMalloc(&ctr, 8 * T * 4); hipMalloc(&sink, 4);
    const int blocks = 782 * 8, per_thread = 4;          // 6.4 M atomics: eight cfg3 views
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = { "device scope, one array        ", "device scope, array per XCD    ", "workgroup scope, array per XCD " };
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipMemset(ctr, 0, 8 * T * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, ctr, sink, per_thread);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, ctr, sink, per_thread);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, ctr, sink, per_thread);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<uint32_t> h(8 * T);
        hipMemcpy(h.data(), ctr, 8 * T * 4, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; int used = 0;
        for (int x = 0; x < 8; x++) { unsigned long long s = 0; for (int t = 0; t < T; t++) s += h[(size_t)x * T + t]; tot += s; used += s != 0; }
        const unsigned long long want = (unsigned long long)blocks * 256 * per_thread;
        printf("%s %8.1f us  %.1f atomics/ns  sum %llu of %llu %s  arrays used %d\n", names[mode], best * 1e3, want / (best * 1e6), tot, want,
               tot == want ? "ok" : "LOST UPDATES", used);
    }
    run<1, false>("32-bit, every lane a line of its own     ", ctr, sink, blocks, per_thread);
    run<4, false>("32-bit, 4 lanes per 64-B line            ", ctr, sink, blocks, per_thread);
    run<16, false>("32-bit, 16 lanes per 64-B line           ", ctr, sink, blocks, per_thread);
    run<1, true>("64-bit, every lane a line of its own     ", ctr, sink, blocks, per_thread);
    run<4, true>("64-bit, 4 lanes per 64-B line            ", ctr, sink, blocks, per_thread);
    run<4, true>("64-bit, 4 lanes per line, HALF the ops   ", ctr, sink, blocks / 2, per_thread);
    return 0;
}

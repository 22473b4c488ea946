// Stand-alone timing probe for sg_tile_scan_kernel (build: see tools/README or the command in DESIGN.md notes):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I. tools/scan_probe.hip -o /tmp/scan_probe && /tmp/scan_probe
#include "../sings_amd/csrc/sg_binning.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
void sg_prof_begin(int, hipStream_t) {}
void sg_prof_end(int, hipStream_t) {}
__global__ void touch_kernel(uint32_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) atomicAdd(&p[i], 0u); }
template <int K> static void run(int T, uint32_t stride, int mean)
{
    size_t words = (size_t)T * K * stride;
    std::vector<uint32_t> h(words, 0);
    for (int t = 0; t < T; t++) for (int k = 0; k < K; k++) h[((size_t)t * K + k) * stride] = (uint32_t)(rand() % (2 * mean / K + 1));
    uint32_t *tc, *cursor, *header, *ck; uint2 *ranges; uint4 *plan;
    hipMalloc(&tc, words * 4); hipMalloc(&cursor, (size_t)T * 8 * 4); hipMalloc(&header, 256); hipMalloc(&ck, T * 4);
    hipMalloc(&ranges, T * 8); hipMalloc(&plan, T * 16);
    hipMemcpy(tc, h.data(), words * 4, hipMemcpyHostToDevice);
    const int tpt = 1, grid = (T + 1023) / 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e9, sum = 0;
        for (int it = 0; it < 20; it++) {
            if (mode) hipLaunchKernelGGL(touch_kernel, dim3((words + 255) / 256), dim3(256), 0, 0, tc, words);   // counters last written by atomics
            hipEventRecord(e0);
            hipLaunchKernelGGL(sg_tile_scan_kernel<K>, dim3(grid), dim3(1024), 0, 0, T, tpt, tc, stride, ranges, cursor, header,
                               1u << 30, 1u << 30, 1u << 30, plan, ck, 1u << 30);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it > 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("T=%d K=%d stride=%u %s: best %.1f us  mean %.1f us\n", T, K, stride, mode ? "after atomics" : "warm", best * 1e3, sum / 17 * 1e3);
    }
}
int main() { run<1>(8160, 1, 100); run<8>(1792, 32, 400); return 0; }
